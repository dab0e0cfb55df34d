"""Device-resident entry points: the expression blocks already live in HBM (fp32, genes x cells,
column-major = a torch tensor of shape (cells, genes)).  PyTorch is only the allocator here; all
compute runs in libsharp_hip.so on its own stream (synchronise torch before handing a tensor over)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import SharpError, check, lib


def _ip(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def synth_fill(dX, seed, cell0, G=12, nmark=1000):
    """Fill a (cells, genes) float32 cuda tensor with the counter-based synthetic counts."""
    _lib.ensure_init()
    ncell, m = dX.shape
    assert dX.is_contiguous() and str(dX.dtype) == "torch.float32"
    check(lib().sharp_synth_fill_dev(C.c_uint(seed), m, C.c_longlong(cell0), ncell, G, nmark, C.c_void_p(dX.data_ptr()),
                                     C.c_longlong(m)))
    check(lib().sharp_synchronize())


def synth_labels(seed, cell0, ncell, G=12):
    out = np.zeros(ncell, np.int32)
    check(lib().sharp_synth_labels(C.c_uint(seed), C.c_longlong(cell0), ncell, G, _ip(out)))
    return out


def csc_to_dev(sp):
    """scipy.sparse (genes x cells; stands in for R's dgCMatrix) -> resident (cells, genes) float32 cuda tensor.  Only the non-zeros
    cross PCIe; the dense block is built on the device (sharp_csc_to_dense_dev)."""
    import torch

    _lib.ensure_init()
    sp = sp.tocsc()
    if not sp.has_canonical_format:
        sp = sp.copy()
        sp.sum_duplicates()
    m, n = sp.shape
    if sp.nnz >= 2**31:
        raise ValueError("sparse input: more than 2^31 - 1 stored entries")
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    cp = np.ascontiguousarray(sp.indptr, np.int32)
    ri = np.ascontiguousarray(sp.indices, np.int32)
    xv = np.ascontiguousarray(sp.data, np.float64)
    check(lib().sharp_csc_to_dense_dev(_ip(cp), _ip(ri), _dp(xv), m, C.c_longlong(n), C.c_void_p(dX.data_ptr()), C.c_longlong(m)))
    return dX


def SHARP_dev(dX, ensize_K=0, reduced_ndim=0, base_ncells=0, partition_ncells=0, hmethod=1, N_cluster=0, enpN_cluster=0,
              indN_cluster=0, minN_cluster=0, maxN_cluster=0, sil_thre=-1.0, height_Ntimes=0.0, flag=True, projector=0,
              rN_seed=0.5, forview=False, view_out=None):
    """SHARP() (R/SHARP.R:44-318) on a resident block (float32, or float64 for TPM / CPM-like values: sharp_SHARP_dev64);
    returns (pred_clusters, info).  forview (the reference's default, R/SHARP.R:46,844): info also carries "viE" (n x p, the
    ensemble-mean projection) and "x0" (n x G, the soft cluster matrix of :717-731,763-783)."""
    _lib.ensure_init()
    n, m = dX.shape
    pred = np.zeros(n, np.int32)
    npred, pu, Ku, path, x0c = C.c_int(), C.c_int(), C.c_int(), C.c_int(), C.c_int()
    viE = x0 = None
    cap = 0
    if forview:
        pmax = reduced_ndim if reduced_ndim > 0 else int(np.ceil(np.log2(n) / 0.04))
        cap = max(maxN_cluster, 40, -(-n // 5000)) + 2
        if view_out is not None and view_out[0].shape == (n, pmax) and view_out[1].size == n * cap:
            viE, x0 = view_out                        # (a caller that keeps its result buffers between calls: no first-touch page faults)
        else:
            viE = np.empty((n, pmax))
            x0 = np.empty(n * cap)
    entry = lib().sharp_SHARP_dev64 if str(dX.dtype) == "torch.float64" else lib().sharp_SHARP_dev
    rc = check(entry(C.c_void_p(dX.data_ptr()), m, C.c_longlong(n), C.c_longlong(dX.stride(0)), ensize_K,
                                     reduced_ndim, base_ncells, partition_ncells, hmethod, N_cluster, enpN_cluster,
                                     indN_cluster, minN_cluster, maxN_cluster, C.c_double(sil_thre), C.c_double(height_Ntimes),
                                     int(bool(flag)), projector, C.c_double(rN_seed), _ip(pred), C.byref(npred), _dp(viE), _dp(x0), cap,
                                     C.byref(x0c), C.byref(pu), C.byref(Ku), C.byref(path)), allow=48)
    info = {"N.pred_cluster": npred.value, "reduced.dim": pu.value, "ensize.K": Ku.value,
            "path": "SHARP_large" if path.value else "SHARP_small", "warn": rc}
    if forview:
        info["view_out"] = (viE, x0)
        # (the library writes n x p_used doubles row-major from the start of the buffer)
        info["viE"] = viE if pu.value == viE.shape[1] else viE.reshape(-1)[: n * pu.value].reshape(n, pu.value)
        info["x0"] = x0[: n * x0c.value].reshape(x0c.value, n).T
    return pred, info


def unlimited_block_dev(dX, p, projector, ensize_K, rN_seed, cap_rows=4096, flag=True, viE=None, next_block=None, view_dim=0, view_seed=None):
    """One block of SHARP_unlimited: labels, per-cluster means of viE (G x p) and cluster sizes.

    flag: the log flag of the block's SHARP() call; viE: optional (nb, p) float64 host array that receives the block's
    ensemble-mean projection (viewflag) -- or, with view_dim > 0, an (nb, view_dim) array that receives those rows reduced on the device
    (sharp_unlimited_block_viewk_dev: what SHARP_unlimited returns as viE above 1e5 cells, R/SHARP_unlimited.R:216-228; view_seed: the integer
    seed of the RUN's z0, the same for all of its blocks -- default 50 + rN_seed + ensize_K + 1; an unseeded run must name one); next_block: the resident
    block of the NEXT call (same genes / projector / parameters): its projection and distance matrices are prepared under this call's tail
    (sharp_unlimited_next_block_dev)."""
    _lib.ensure_init()
    nb, m = dX.shape
    f64 = str(dX.dtype) == "torch.float64"
    if next_block is not None and not f64 and str(next_block.dtype) == "torch.float32":
        check(lib().sharp_unlimited_next_block_dev(C.c_void_p(next_block.data_ptr()), C.c_longlong(next_block.shape[0]),
                                                   C.c_longlong(next_block.stride(0))))
    if viE is not None and view_dim > 0 and view_seed is None:
        if rN_seed == 0.5:
            raise SharpError("unlimited_block_dev: an unseeded run reduced block by block needs view_seed, the one seed of the run's z0")
        view_seed = 50 + rN_seed + (ensize_K if ensize_K > 0 else 5) + 1
    pred = np.zeros(nb, np.int32)
    means = np.empty((cap_rows, p))                 # only the first G rows are written and returned
    counts = np.empty(cap_rows, np.int64)
    G = C.c_int()
    entry = lib().sharp_unlimited_block_viewk_dev64 if f64 else lib().sharp_unlimited_block_viewk_dev      # (a float64 block: TPM / CPM-like values)
    check(entry(C.c_void_p(dX.data_ptr()), m, C.c_longlong(nb), C.c_longlong(dX.stride(0)), p,
                                                projector, ensize_K, C.c_double(rN_seed), int(bool(flag)), _ip(pred), C.byref(G),
                                                _dp(means), cap_rows, counts.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                int(view_dim) if viE is not None else 0, C.c_double(view_seed or 0.0), _dp(viE)))
    return pred, means[: G.value].copy(), counts[: G.value].copy()


def unlimited_blocks_dev(blocks, p, projector, ensize_K, rN_seed, cap_rows=4096):
    """Several resident blocks of one rank in one call (sharp_unlimited_blocks_dev): a list of (labels, cluster means G x p, cluster
    sizes), one per block -- what unlimited_block_dev returns block by block, with the base clustering of all blocks as one pipelined
    batch and their tails on helper threads."""
    _lib.ensure_init()
    import torch

    B = len(blocks)
    m = blocks[0].shape[1]
    ptrs = (C.c_void_p * B)(*[b.data_ptr() for b in blocks])
    f64 = np.array([1 if b.dtype == torch.float64 else 0 for b in blocks], np.int32)
    ncb = np.array([b.shape[0] for b in blocks], np.int64)
    ldb = np.array([b.stride(0) for b in blocks], np.int64)
    pred = np.zeros(int(ncb.sum()), np.int32)
    ncl = np.zeros(B, np.int32)
    means = np.empty((cap_rows * B, p))
    counts = np.empty(cap_rows * B, np.int64)
    check(lib().sharp_unlimited_blocks_dev(ptrs, _ip(f64), ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           B, m, p, projector, ensize_K, C.c_double(rN_seed), _ip(pred), _ip(ncl), _dp(means), cap_rows * B,
                                           counts.ctypes.data_as(C.POINTER(C.c_longlong))))
    out, o, r = [], 0, 0
    for b in range(B):
        g = int(ncl[b])
        out.append((pred[o:o + int(ncb[b])].copy(), means[r:r + g].copy(), counts[r:r + g].copy()))
        o += int(ncb[b]); r += g
    return out


def unlimited_dev(blocks, ensize_K=0, N_cluster=0, minN_cluster=0, maxN_cluster=0, rN_seed=0.5, viewflag=False, view_dim=None, viE_out=None):
    """SHARP_unlimited (R/SHARP_unlimited.R:29-242) on float32 blocks resident on the current GPU -> (pred, n_pred, p, viE or None).
    viewflag: viE as the reference returns it (:214-228): E1 (ncells x p) up to 1e5 cells, above that E1 reduced to 50 columns by one
    more sparse projection -- taken per block on the device (sharp_SHARP_unlimited_viewk_dev).  view_dim forces the reduction (tests)."""
    _lib.ensure_init()
    B = len(blocks)
    ptrs = (C.c_void_p * B)(*[b.data_ptr() for b in blocks])
    ncb = np.array([b.shape[0] for b in blocks], np.int64)
    ldb = np.array([b.stride(0) for b in blocks], np.int64)
    n = int(ncb.sum())
    pred = np.zeros(n, np.int32)
    npred, pu = C.c_int(), C.c_int()
    viE, kdim = None, 0
    if viewflag:
        kdim = int(view_dim) if view_dim is not None else (50 if n > 1e5 else 0)
        cols = kdim if kdim else int(np.ceil(np.log2(n) / 0.04))
        viE = viE_out if viE_out is not None and viE_out.shape == (n, cols) else np.zeros((n, cols))
    check(lib().sharp_SHARP_unlimited_viewk_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)), B,
                                                int(blocks[0].shape[1]), int(ensize_K), int(N_cluster), int(minN_cluster), int(maxN_cluster),
                                                C.c_double(rN_seed), _ip(pred), C.byref(npred), C.byref(pu), kdim, _dp(viE)), allow=48)
    return pred, npred.value, pu.value, viE


def unlimited_multi_dev(blocks, device_of_block, devices, ensize_K=0, N_cluster=0, minN_cluster=0, maxN_cluster=0, rN_seed=0.5,
                        viewflag=False):
    """sharp_SHARP_unlimited_multi_dev: SHARP_unlimited (R/SHARP_unlimited.R:125-183) over blocks that already live on the GPUs of
    `devices`; blocks[b] is a (cells, genes) row-major tensor (fp32 or fp64) on devices[device_of_block[b]].
    -> (pred, n_pred, p, viE or None)"""
    import torch

    B = len(blocks)
    m = int(blocks[0].shape[1])
    ptrs = (C.c_void_p * B)(*[b.data_ptr() for b in blocks])
    f64 = np.array([1 if b.dtype == torch.float64 else 0 for b in blocks], np.int32)
    ncb = np.array([b.shape[0] for b in blocks], np.int64)
    ldb = np.array([b.stride(0) for b in blocks], np.int64)
    dob = np.ascontiguousarray(device_of_block, np.int32)
    dv = np.ascontiguousarray(devices, np.int32)
    n = int(ncb.sum())
    p = int(np.ceil(np.log2(n) / 0.04))
    pred = np.zeros(n, np.int32)
    viE = np.zeros((n, p)) if viewflag else None
    npred, pu = C.c_int(), C.c_int()
    for b in blocks:
        torch.cuda.synchronize(b.device)
    check(lib().sharp_SHARP_unlimited_multi_dev(ptrs, _ip(f64), ncb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                ldb.ctypes.data_as(C.POINTER(C.c_longlong)), _ip(dob), B, m, int(ensize_K), int(N_cluster),
                                                int(minN_cluster), int(maxN_cluster), C.c_double(rN_seed), _ip(dv), len(dv), _ip(pred),
                                                C.byref(npred), C.byref(pu), _dp(viE) if viE is not None else None), allow=48)
    return pred, npred.value, pu.value, viE


def multi_timeline(cap=4096):
    """sharp_multi_timeline: rows of (worker, block, upload start, upload end, clustering start, clustering end) of the last multi-device call"""
    rows = np.zeros((cap, 6))
    n = C.c_int()
    check(lib().sharp_multi_timeline(_dp(rows), cap, C.byref(n)))
    return rows[:min(n.value, cap)]


def unlimited_merge(means, counts, ncells, N_cluster=0, minN_cluster=0, maxN_cluster=0):
    """Cross-block sMetaC on gathered centroids -> final 1-based id per (block, cluster) row."""
    _lib.ensure_init()
    means = np.ascontiguousarray(means, np.float64)
    counts = np.ascontiguousarray(counts, np.int64)
    nC, p = means.shape
    fid = np.zeros(nC, np.int32)
    nf = C.c_int()
    check(lib().sharp_unlimited_merge(_dp(means), counts.ctypes.data_as(C.POINTER(C.c_longlong)), nC, p, C.c_longlong(ncells),
                                      N_cluster, minN_cluster, maxN_cluster, _ip(fid), C.byref(nf)))
    return fid, nf.value


def marker_genes_dev(dX, labels, n_cluster, theta=1e-4, ng=1):
    """Per-gene (auc, icluster, pvalue, sparsity, FC) of get_marker_genes on a resident (cells, genes) float32 block."""
    _lib.ensure_init()
    n, m = dX.shape
    lab = np.ascontiguousarray(labels, np.int32)
    out = np.zeros((m, 5))
    check(lib().sharp_marker_genes_dev(C.c_void_p(dX.data_ptr()), m, C.c_longlong(n), C.c_longlong(dX.stride(0)), _ip(lab), int(n_cluster),
                                       C.c_double(theta), int(ng), _dp(out)))
    return out


def profile(enable=True):
    _lib.ensure_init()
    check(lib().sharp_profile_enable(1 if enable else 0))
    check(lib().sharp_profile_reset())


def profile_table():
    buf = C.create_string_buffer(1 << 16)
    check(lib().sharp_profile_dump(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split()
        out[name] = (float(ms), int(cnt))
    return out
