"""BASELINE.json configs 3, 4 and 5 at their full sizes on one MI355X, and the edges of the path they reach:

  cfg3  500 000 cells x 20 000 genes as 10 blocks, SHARP_unlimited, K = 5, p = 474      -- whole, one call and block by block
  cfg4  1.3 M cells x 27 000 genes, 8 blocks of 162 500 (one per GPU in the 8-GPU run), p = 508 -- all eight shares in turn
        on this GPU through sharp_unlimited_block_dev, then sharp_unlimited_merge at ncells = 1.3 M (the n >= 1e6 branch of
        sMetaC, R/sMetaC.R:110-119), checked against the oracle's merge on the same centroid tables
  cfg5  10 M cells x 20 000 genes, 200 streamed blocks of 50 000 (25 per GPU in the 8-GPU run), p = 582 -- all 200 generated
        on the fly into one buffer, then the merge at ncells = 1e7 (k = 200 .. 2000: 1801 candidate levels) against the oracle
  cfg2  50 000 x 20 000, K = 15 WHOLE, label for label and decision for decision against the oracle
  sharp_sMetaC at n = 1.2 M cells against the oracle; clustering tasks beyond the LDS-resident limits (> 4096, > 7168 rows)

The oracle is the CPU restatement of the reference (oracle/); where it would need hours (whole configs) the checks are the
size-independent properties of the path: determinism, numbering rules, agreement of the one-call and the sharded form,
recovery of the planted clusters."""
import ctypes as C
import os
import time

import numpy as np
import pytest
from sklearn.metrics import adjusted_rand_score

pytestmark = pytest.mark.gpu
SEED, RN = 20261003, 2103


@pytest.fixture(scope="module")
def env():
    import torch

    import sharp_amd
    from sharp_amd import device

    sharp_amd.init(0)
    return sharp_amd, device, torch


def _purity(truth, pred):
    """fraction of cells whose predicted cluster's majority planted label is their own"""
    tab = np.zeros((pred.max() + 1, truth.max() + 1), np.int64)
    np.add.at(tab, (pred, truth), 1)
    return tab.max(1).sum() / truth.size


def _block_steps(sa, dev, torch, m, nb, nblocks, p, K=5, window=1):
    """every block in turn through sharp_unlimited_block_dev (what a rank of the sharded run does with its blocks); the blocks
    are generated into ONE device buffer.  window > 1: `window` blocks at a time through sharp_unlimited_blocks_dev (a rank that
    holds several blocks: one pipelined batch per window).  Returns per-block labels, centroid tables, counts, planted labels, seconds."""
    proj = sa.Projector(m, p, [50 + RN + k for k in range(1, K + 1)])
    preds, means, counts, truth = [], [], [], []
    t_run = 0.0
    if window > 1:
        bufs = [torch.empty((nb, m), dtype=torch.float32, device="cuda") for _ in range(window)]
        try:
            for b0 in range(0, nblocks, window):
                cur = bufs[: min(window, nblocks - b0)]
                for q, x in enumerate(cur):
                    dev.synth_fill(x, SEED, (b0 + q) * nb)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                res = dev.unlimited_blocks_dev(cur, p, proj.handle, K, RN)
                t_run += time.perf_counter() - t0
                for q, (pr, mn, cn) in enumerate(res):
                    assert pr.min() == 1 and pr.max() == mn.shape[0] == cn.size and cn.sum() == nb
                    np.testing.assert_array_equal(np.bincount(pr)[1:], cn)
                    preds.append(pr); means.append(mn); counts.append(cn)
                    truth.append(dev.synth_labels(SEED, (b0 + q) * nb, nb))
        finally:
            proj.close()
            del bufs
            torch.cuda.empty_cache()
        return preds, means, counts, truth, t_run
    dX = torch.empty((nb, m), dtype=torch.float32, device="cuda")
    try:
        for b in range(nblocks):
            dev.synth_fill(dX, SEED, b * nb)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            pr, mn, cn = dev.unlimited_block_dev(dX, p, proj.handle, K, RN)
            t_run += time.perf_counter() - t0
            assert pr.min() == 1 and pr.max() == mn.shape[0] == cn.size and cn.sum() == nb
            first = [int(np.argmax(pr == j)) for j in range(1, pr.max() + 1)]
            assert first == sorted(first)                                 # block labels by first appearance (R/SHARP.R:828-843)
            np.testing.assert_array_equal(np.bincount(pr)[1:], cn)
            preds.append(pr); means.append(mn); counts.append(cn)
            truth.append(dev.synth_labels(SEED, b * nb, nb))
    finally:
        proj.close()
        del dX
        torch.cuda.empty_cache()
    return preds, means, counts, truth, t_run


def _apply_merge(preds, means, fid):
    first = np.concatenate([[0], np.cumsum([mn.shape[0] for mn in means])])
    return np.concatenate([np.asarray(fid)[first[b] + preds[b] - 1] for b in range(len(preds))])


def test_cfg3_whole_and_sharded(env):
    sa, dev, torch = env
    lib = sa.lib()
    nb, m, B = 50000, 20000, 10                       # BASELINE.json configs[2]
    blocks = []
    for b in range(B):
        x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
        dev.synth_fill(x, SEED, b * nb)
        blocks.append(x)
    ptrs = (C.c_void_p * B)(*[x.data_ptr() for x in blocks])
    ncb = np.full(B, nb, np.int64)
    ldb = np.full(B, m, np.int64)

    def run():
        pred = np.zeros(B * nb, np.int32)
        npred, pu = C.c_int(), C.c_int()
        torch.cuda.synchronize()
        rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           B, m, 5, 0, 0, 0, C.c_double(RN), pred.ctypes.data_as(C.POINTER(C.c_int)), C.byref(npred),
                                           C.byref(pu))
        assert rc in (0, 16, 32, 48), lib.sharp_last_error()
        return pred, npred.value, pu.value

    p1, n1, p = run()
    t0 = time.perf_counter()
    p2, n2, _ = run()
    dt = time.perf_counter() - t0
    # soak: the call pipelines blocks on side streams (next block's front under the current block's tail, batched windows, shared chunk
    # buffers of the RP stage -- a race between them once showed in one run of two): ten more calls, identical labels every time
    for rep in range(10):
        pk, nk, _ = run()
        assert nk == n1 and np.array_equal(pk, p1), "call %d of the soak differs" % (rep + 3)
    print("cfg3: %.3f s per call = %.0f cells/s, %d clusters" % (dt, B * nb / dt, n1))
    assert p == 474 == int(np.ceil(np.log2(B * nb) / 0.04))               # R/SHARP_unlimited.R:65-66
    assert np.array_equal(p1, p2) and n1 == n2 == p1.max() and p1.min() == 1
    sizes = np.bincount(p1)[1:]
    assert np.all(np.diff(sizes) <= 0)                                    # ids by decreasing size (:180-183)
    assert (sizes < 10).sum() <= 1                                        # < 10-cell clusters merged (:168-177)
    truth = np.concatenate([dev.synth_labels(SEED, b * nb, nb) for b in range(B)])
    assert adjusted_rand_score(truth, p1) > 0.9
    # the sharded form (block step per block, then the merge on the gathered centroid tables) gives the same labels
    proj = sa.Projector(m, p, [50 + RN + k for k in range(1, 6)])
    preds, means, counts = [], [], []
    for b in range(B):                                                    # (blocks 5.. with the next block prepared under the current one's tail)
        pr, mn, cn = dev.unlimited_block_dev(blocks[b], p, proj.handle, 5, RN, next_block=blocks[b + 1] if 5 <= b < B - 1 else None)
        preds.append(pr); means.append(mn); counts.append(cn)
    proj.close()
    fid, nf = dev.unlimited_merge(np.concatenate(means, 0), np.concatenate(counts, 0), B * nb)
    assert nf == n1 and np.array_equal(_apply_merge(preds, means, fid), p1)
    del blocks
    torch.cuda.empty_cache()


def test_cfg4_shares_and_merge_at_1p3M_cells(env, oracle):
    sa, dev, torch = env
    nb, m, B, p = 162500, 27000, 8, 508                                   # BASELINE.json configs[3]: 1.3 M x 27 000, one block per GPU
    ncells = nb * B
    assert p == int(np.ceil(np.log2(ncells) / 0.04))
    preds, means, counts, truth, t_run = _block_steps(sa, dev, torch, m, nb, B, p)
    print("cfg4: %d shares of %d cells: %.3f s per share = %.0f cells/s per GPU" % (B, nb, t_run / B, nb * B / t_run))
    M, Cn = np.concatenate(means, 0), np.concatenate(counts, 0)
    fid, nf = dev.unlimited_merge(M, Cn, ncells)
    ref = oracle.unlimited_merge(M, Cn, ncells)
    assert ref["rc"] in (0, 16) and nf == ref["n_final"] and np.array_equal(fid, ref["final_id"])
    # n >= 1e6 (R/sMetaC.R:110-119): the candidate numbers of clusters start at floor(n / 50000) = 26 and go to 260 (or nC - 1)
    assert 26 <= nf <= min(260, M.shape[0] - 1)
    pred = _apply_merge(preds, means, fid)
    sizes = np.bincount(pred)[1:]
    assert pred.min() == 1 and pred.max() == nf and np.all(np.diff(sizes) <= 0)
    t = np.concatenate(truth)
    # >= 26 clusters are forced on 12 planted ones: the merge does not mix what the blocks kept apart
    assert _purity(t, pred) > _purity(t, np.concatenate([preds[b] + 1000 * b for b in range(B)])) - 0.01
    # every share on its own recovers the planted clusters
    for b in range(B):
        assert adjusted_rand_score(truth[b], preds[b]) > 0.9


def test_cfg5_two_hundred_streamed_blocks_and_merge_at_1e7_cells(env, oracle):
    sa, dev, torch = env
    nb, m, B, p = 50000, 20000, 200, 582                                  # BASELINE.json configs[4]: 10 M x 20 000, 25 blocks per GPU
    ncells = nb * B
    assert p == int(np.ceil(np.log2(ncells) / 0.04))
    preds, means, counts, truth, t_run = _block_steps(sa, dev, torch, m, nb, B, p, window=10)   # (a rank's 25 blocks: windows of ten resident blocks)
    M, Cn = np.concatenate(means, 0), np.concatenate(counts, 0)
    t0 = time.perf_counter()
    fid, nf = dev.unlimited_merge(M, Cn, ncells)
    t_merge = time.perf_counter() - t0
    print("cfg5: 200 blocks %.2f s (%.0f cells/s on one GPU; a rank of the 8-GPU run does 25: %.2f s), merge of %d rows %.3f s"
          % (t_run, ncells / t_run, t_run / 8, M.shape[0], t_merge))
    # k = floor(1e7 / 50000) .. floor(1e7 / 5000) = 200 .. 2000 (capped by nC - 1): up to 1801 candidate levels
    assert 200 <= nf <= min(2000, M.shape[0] - 1)
    t0 = time.perf_counter()
    ref = oracle.unlimited_merge(M, Cn, ncells)
    print("oracle merge: %.1f s" % (time.perf_counter() - t0))
    assert ref["rc"] in (0, 16) and nf == ref["n_final"] and np.array_equal(fid, ref["final_id"])
    pred = _apply_merge(preds, means, fid)
    sizes = np.bincount(pred)[1:]
    assert pred.max() == nf and np.all(np.diff(sizes) <= 0)
    # >= 200 clusters are forced on 12 planted ones: the merge joins block-level clusters of the same planted cluster, it does not mix
    # what the blocks kept apart
    t = np.concatenate(truth)
    assert _purity(t, pred) > _purity(t, np.concatenate([preds[b] + 100 * b for b in range(B)])) - 0.01
    assert min(adjusted_rand_score(truth[b], preds[b]) for b in range(0, B, 20)) > 0.9


def test_cfg2_full_size_matches_oracle(env, oracle):
    """BASELINE.json configs[1] whole: 50 000 x 20 000, ensize.K = 15, p = 391 (the reduced dimension SHARP() derives from 50 000 cells),
    SHARP_large with 375 base tasks: labels identical to the oracle's, cell for cell (R/SHARP.R:478-851) -- and the two decision logs
     agree decision for decision: 375 base tasks, 25 per-fold wMetaC, the sMetaC across the folds.  (One block of
    cfg3 -- K = 5, p = 474 -- against the oracle: tools/parity_fullsize.py cfg3_block, profiles/r06_cfg3_block_parity.txt.)"""
    from test_decisions_gpu import compare_logs

    sa, dev, torch = env
    n, m, K = 50000, 20000, 15
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, SEED, 0)
    sa.decision_log(True)
    try:
        pred, info = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN)
        got = sa.last_decisions()
    finally:
        sa.decision_log(False)
    assert info["path"] == "SHARP_large" and info["reduced.dim"] == 391
    X = dX.cpu().numpy().T.astype(np.float64)                             # (genes, cells), column-major: 8 GB
    del dX
    torch.cuda.empty_cache()
    cores = min(len(os.sched_getaffinity(0)), 32)                         # 375 tasks; every thread holds a 320 MB fold copy; the oracle is memory-bound: 32 threads beat 64 (bench.py cpu_baseline)
    t0 = time.perf_counter()
    oracle.decision_log(True)
    try:
        ref = oracle.SHARP(X, K=K, rN_seed=RN, nthreads=cores, want_view=False)
        want = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    print("oracle: %.1f s on %d threads (%.0f cells/s)" % (time.perf_counter() - t0, cores, n / (time.perf_counter() - t0)))
    assert ref["rc"] in (0, 16)
    assert info["N.pred_cluster"] == ref["pred_clusters"].max()
    assert np.array_equal(pred, ref["pred_clusters"])
    assert got.shape[0] == 375 + 25 + 1
    compare_logs(sa, got, want, "cfg2 whole")


def test_block_of_1e5_cells_no_reshuffle_branch_matches_oracle(env, oracle):
    """The branch a block of BASELINE.json configs[3] takes (162 500 cells per GPU): n >= 1e5, so SHARP_large keeps the cells in their
    original order (no `scExp[, reind]`, R/SHARP.R:504-507) and skips the un-shuffle of labels, viE and x0 (:777-783); 51 folds whose
    last two are re-balanced (:513-536); a within-block sMetaC over ~ 500 fold clusters with minN = min(max(n / 1e4, 2), 10) = 10
    (R/sMetaC.R:103-109).  One block of 100 200 cells x 1500 genes, K = 3, through SHARP() on a resident block -- labels, viE and x0 --
    and through the per-block entry of SHARP_unlimited (R/SHARP_unlimited.R:135-143): identical to the oracle, cell for cell."""
    sa, dev, torch = env
    n, m, K = 100200, 1500, 3
    p = int(np.ceil(np.log2(n) / 0.04))
    assert n >= 100000 and p == 416
    dX = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(dX, SEED + 5, 0, 8, 150)
    pred, info = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN, forview=True)
    assert info["path"] == "SHARP_large" and info["reduced.dim"] == p
    proj = sa.Projector(m, p, [50 + RN + k for k in range(1, K + 1)])
    viE_b = np.empty((n, p))
    pred_b, means_b, counts_b = dev.unlimited_block_dev(dX, p, proj.handle, K, RN, viE=viE_b)
    proj.close()
    X = dX.cpu().numpy().T.astype(np.float64)                             # (genes, cells)
    del dX
    torch.cuda.empty_cache()
    cores = min(len(os.sched_getaffinity(0)), 32)
    t0 = time.perf_counter()
    ref = oracle.SHARP_large(X, K=K, p=p, rN_seed=RN, nthreads=cores)
    print("oracle: %.1f s on %d threads" % (time.perf_counter() - t0, cores))
    assert ref["rc"] in (0, 16)
    assert len(set(ref["pred_clusters"])) >= 6                               # (the planted clusters were found: a real comparison)
    assert np.array_equal(pred, ref["pred_clusters"])
    assert np.array_equal(pred_b, ref["pred_clusters"])
    tol = 2e-12 * np.abs(ref["viE"]).max()
    np.testing.assert_allclose(info["viE"], ref["viE"], rtol=0, atol=tol)   # rows in the ORIGINAL cell order, never shuffled
    np.testing.assert_allclose(viE_b, ref["viE"], rtol=0, atol=tol)
    assert info["x0"].shape == ref["x0"].shape
    np.testing.assert_allclose(info["x0"], ref["x0"], rtol=0, atol=1e-15)
    # the per-block summary the cross-block merge consumes: centroids = colMeans(viE[cluster, ]) (R/sMetaC.R:58-63)
    for g in (1, int(pred_b.max())):
        np.testing.assert_allclose(means_b[g - 1], ref["viE"][ref["pred_clusters"] == g].mean(0), rtol=0, atol=1e-10)
    assert np.array_equal(counts_b, np.bincount(ref["pred_clusters"])[1:])


def test_cfg1_shaped_call_matches_oracle(env, oracle):
    """BASELINE.json configs[0] is the reference's own example (479 cells of TPM values, default ensize.K = 15, rN.seed = 2103,
    README.md:88-114); its data blob is not in the repository, so the SHAPE is run: 479 cells x 20 000 genes of TPM-like doubles,
    p = ceiling(log2(479) / 0.04) = 223, prep = TRUE (all-zero genes removed, R/SHARP.R:104-106), testlog with fixed cells (:877-924),
    SHARP_small (:339-454) -- labels, viE, x0 and allrpinfo against the oracle on the same prepared matrix."""
    sa, dev, torch = env
    n, m, K = 479, 20000, 15
    X = oracle.synth_fill(SEED, m, 0, n, 5, 1500)
    X = X / np.maximum(X.sum(0, keepdims=True), 1.0) * 1e6               # TPM: non-fp32-exact doubles -> an fp64 block in HBM
    X[[17, 4040, 19999]] = 0.0                                            # all-zero genes
    cells = np.arange(0, n, 5)[:90]
    res = sa.SHARP(X, exp_type="TPM", prep=True, rN_seed=RN, forview=True, testlog_cells=cells)
    assert sa.lib().sharp_x_storage() == 64
    assert res["path"] == "SHARP_small" and res["paras"]["ensize.K"] == K and res["reduced.dim"] == 223 and res["N.cells"] == n
    keep = X.sum(1) != 0
    assert res["N.genes"] == m and int(keep.sum()) == m - 3                 # (N.genes is taken before prep, R/SHARP.R:56,293)
    Xp = X[keep]
    flag_ref, _ms = oracle.testlog(Xp, 223, cells)
    assert bool(res["paras"]["logmark"]) == flag_ref
    ref = oracle.SHARP_small(Xp, K=K, p=223, flag=flag_ref, rN_seed=RN, want_view=True)
    assert ref["rc"] in (0, 16)
    assert np.array_equal(res["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_allclose(res["viE"], ref["viE"], rtol=0, atol=2e-12 * np.abs(ref["viE"]).max())
    np.testing.assert_allclose(res["x0"], ref["x0"], rtol=0, atol=1e-15)
    rp = res["allrpinfo"]
    assert len(rp) == K
    from sharp_amd.api import colorL
    for k in range(K):
        e = oracle.project(Xp, oracle.ranM(Xp.shape[0], 223, 50 + RN + k + 1), flag_ref)
        np.testing.assert_allclose(rp[k]["indE"], e, rtol=0, atol=2e-12 * np.abs(e).max())
        assert rp[k]["rowColor"] == [colorL[j - 1] for j in ref["enrp"][:, k]]
        assert rp[k]["N.cluster"] == len(set(rp[k]["rowColor"]))


def test_smetac_at_1p2M_cells_matches_oracle(env, oracle):
    """sharp_sMetaC with n >= 1e6 (R/sMetaC.R:110-119: maxN = max(maxN, n/5000), minN = max(minN, n/50000))."""
    sa, dev, torch = env
    rng = np.random.default_rng(12)
    n, p, nlab, G = 1200000, 8, 300, 30
    cen = rng.standard_normal((G, p)) * 3
    lab_centre = rng.integers(0, G, nlab)
    labels = rng.integers(0, nlab, n).astype(np.int32)
    labels[:nlab] = rng.permutation(nlab)                                 # every id present
    E = cen[lab_centre[labels]] + 0.3 * rng.standard_normal((n, p)) + 0.2 * rng.standard_normal((nlab, p))[labels]
    ref = oracle.sMetaC(labels, E, maxN=40)
    res = sa.sMetaC(labels, E, maxN_cluster=40)
    assert ref["rc"] in (0, 16) and ref["nC"] == nlab
    assert np.array_equal(res["tf"], ref["tf"])
    assert np.array_equal(res["finalColor"], ref["finalColor"])
    assert 24 <= res["tf"].max() <= 240                                   # k range 24 .. 240


@pytest.mark.parametrize("n", [4500, 8000])
def test_similarity_task_beyond_the_lds_limits(env, oracle, n, monkeypatch):
    """get_opt_hclust on an n x n similarity with n > 4096 (bulk-synchronous kernel) and n > 7168 (sequential kernel): the
    state of the agglomeration lives in global memory there.  Rows = cluster centroids as a cross-block sMetaC sees them."""
    sa, dev, torch = env
    rng = np.random.default_rng(n)
    G, p = 25, 60
    cen = rng.standard_normal((G, p))
    Z = cen[rng.integers(0, G, n)] + 0.35 * rng.standard_normal((n, p))
    S = np.corrcoef(Z)
    np.fill_diagonal(S, 1.0)
    S = (S + S.T) / 2
    ref = oracle.get_opt_hclust(S, minN=2, maxN=40, sil_thre=0.35)
    dev.profile(True)
    res = sa.get_opt_hclust(S, minN_cluster=2, maxN_cluster=40, sil_thre=0.35)
    prof = dev.profile_table()
    assert prof["host:hclust_tasks_bulk_synchronous"][1] == 1             # no exact ties: the bulk-synchronous kernel did it
    assert res["branch"] == ref["branch"] and res["optN_cluster"] == ref["optN"]
    np.testing.assert_allclose(res["height"], ref["height"], rtol=1e-9)
    assert np.array_equal(res["v"], ref["v"])                             # every cutree level
    np.testing.assert_allclose(res["msil"], ref["msil"], atol=1e-10)
    np.testing.assert_allclose(res["CHind"], ref["CHind"], rtol=1e-8)
    assert np.array_equal(res["f"], ref["f"])
    if n > 7168:                                                          # the sequential kernel with its state in global memory
        monkeypatch.setenv("SHARP_HC_SEQ", "1")
        seq = sa.get_opt_hclust(S, minN_cluster=2, maxN_cluster=40, sil_thre=0.35)
        np.testing.assert_allclose(seq["height"], ref["height"], rtol=1e-12)
        assert np.array_equal(seq["v"], ref["v"]) and np.array_equal(seq["f"], ref["f"])


def test_unlimited_merge_beyond_7168_rows(env, oracle):
    """sharp_unlimited_merge on 7400 (block, cluster) rows, beyond the 7168 the agglomeration keeps in LDS -- cfg5's 200 blocks x 40 clusters
    would be 8000 (R/SHARP_unlimited.R:163); test_cfg5_* runs that merge at its true size."""
    sa, dev, torch = env
    rng = np.random.default_rng(5)
    nC, p, G = 7400, 64, 40
    cen = rng.standard_normal((G, p))
    M = cen[rng.integers(0, G, nC)] + 0.3 * rng.standard_normal((nC, p))
    Cn = rng.integers(5, 200, nC).astype(np.int64)          # < 1e6 cells in all: the k range stays 10 .. 40
    ncells = int(Cn.sum())
    fid, nf = dev.unlimited_merge(M, Cn, ncells, maxN_cluster=16)          # k = 10 .. 16 (seven levels: the oracle's cost is per level)
    ref = oracle.unlimited_merge(M, Cn, ncells, maxN=16)
    assert nf == ref["n_final"] and np.array_equal(fid, ref["final_id"])


def test_unlimited_batched_base_clustering_equals_block_by_block(env, monkeypatch):
    """SHARP_unlimited on several large-path blocks runs the base-clustering tasks of ALL blocks as one pipelined batch (three buffer
    sets in rotation, each block's wMetaC / sMetaC tail from the batch's progress callback as a nested batch); block after block
    (SHARP_UNLIMITED_BATCH=0) must give the same labels, cluster count and ensemble projection."""
    sa, dev, torch = env
    lib = sa.lib()
    B, nb, m, K = 12, 28000, 1500, 5                     # 12 blocks x 14 folds x 5 RPs = 840 tasks: five chunks, three buffer sets
    blocks = []
    for b in range(B):
        x = torch.empty((nb + 37 * b, m), dtype=torch.float32, device="cuda")       # ragged block sizes
        dev.synth_fill(x, 20261003, b * 40000, 6, 200)
        blocks.append(x)

    def run():
        ptrs = (C.c_void_p * B)(*[x.data_ptr() for x in blocks])
        ncb = np.array([x.shape[0] for x in blocks], np.int64)
        ldb = np.array([x.stride(0) for x in blocks], np.int64)
        pred = np.zeros(int(ncb.sum()), np.int32)
        npred, pu = C.c_int(), C.c_int()
        rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           B, m, K, 0, 0, 0, C.c_double(2103), pred.ctypes.data_as(C.POINTER(C.c_int)),
                                           C.byref(npred), C.byref(pu))
        assert rc in (0, 16, 32, 48), lib.sharp_last_error()
        return pred, npred.value, pu.value

    p1, n1, pu1 = run()
    p1b, n1b, _ = run()                                   # (and the batched form is reproducible)
    monkeypatch.setenv("SHARP_UNLIMITED_BATCH", "0")
    p0, n0, pu0 = run()
    monkeypatch.delenv("SHARP_UNLIMITED_BATCH")
    assert pu1 == pu0 and n1 == n0 == n1b
    assert np.array_equal(p1, p0) and np.array_equal(p1, p1b)
    # windows (SHARP_UNLIMITED_WINDOW_MB; 0.51 GB of projections per block): 4200 MB = a batched window of eight blocks (560 tasks) and
    # then four blocks one after the other (280 tasks are too few for a batch); 2100 MB = windows of four, i.e. all block after block
    for mb in ("4200", "2100"):
        monkeypatch.setenv("SHARP_UNLIMITED_WINDOW_MB", mb)
        pw, nw, _ = run()
        monkeypatch.delenv("SHARP_UNLIMITED_WINDOW_MB")
        assert nw == n1 and np.array_equal(pw, p1)
    # the blocks' tails of a batched window run on helper threads with their own device slots (four by default; they finish in any
    # order): none (the calling thread, from the batch's progress callback), one and three must give the same call
    for h in ("0", "1", "3", "7"):
        monkeypatch.setenv("SHARP_TAIL_THREADS", h)
        ph, nh, _ = run()
        monkeypatch.delenv("SHARP_TAIL_THREADS")
        assert nh == n1 and np.array_equal(ph, p1), h
    # the first chunk's size (equal chunks / a short first chunk): the same tasks on the same inputs, whatever the order they are enqueued in
    for var, val in (("SHARP_HC_FIRST_CHUNK", "-1"), ("SHARP_HC_FIRST_CHUNK", "150")):
        monkeypatch.setenv(var, val)
        ps, ns_, _ = run()
        monkeypatch.delenv(var)
        assert ns_ == n1 and np.array_equal(ps, p1), (var, val)

    def run_view():                                      # the viewflag form: every block's E1 rows, written by whichever helper ran its tail
        ptrs = (C.c_void_p * B)(*[x.data_ptr() for x in blocks])
        ncb = np.array([x.shape[0] for x in blocks], np.int64)
        ldb = np.array([x.stride(0) for x in blocks], np.int64)
        pred = np.zeros(int(ncb.sum()), np.int32)
        viE = np.zeros((int(ncb.sum()), pu1), np.float64)
        npred, pu = C.c_int(), C.c_int()
        rc = lib.sharp_SHARP_unlimited_view_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                                B, m, K, 0, 0, 0, C.c_double(2103), pred.ctypes.data_as(C.POINTER(C.c_int)),
                                                C.byref(npred), C.byref(pu), viE.ctypes.data_as(C.POINTER(C.c_double)))
        assert rc in (0, 16, 32, 48), lib.sharp_last_error()
        return pred, viE

    # a rank's blocks in one call (sharp_unlimited_blocks_dev: one pipelined batch, tails on helpers) = block after block
    # (sharp_unlimited_block_dev), table for table
    proj = sa.Projector(m, pu1, [50 + 2103 + k for k in range(1, K + 1)])
    try:
        many = dev.unlimited_blocks_dev(blocks, pu1, proj.handle, K, 2103)
        for b in (0, 5, 11):
            pr, mn, cn = dev.unlimited_block_dev(blocks[b], pu1, proj.handle, K, 2103)
            assert np.array_equal(many[b][0], pr) and np.array_equal(many[b][2], cn) and np.array_equal(many[b][1], mn), b
    finally:
        proj.close()
    assert sum(len(t[0]) for t in many) == len(p1)

    pv, v4 = run_view()
    monkeypatch.setenv("SHARP_TAIL_THREADS", "0")
    pv0, v0 = run_view()
    monkeypatch.delenv("SHARP_TAIL_THREADS")
    assert np.array_equal(pv, p1) and np.array_equal(pv0, p1)
    assert np.array_equal(v4, v0) and np.abs(v4).max() > 0
    # the in-process multi-device entry with blocks that are resident already: a worker takes all of its blocks in one go (batch windows
    # where they qualify).  One worker with all twelve; two workers (two slots on this one GPU) with eight (a window) and four
    for dob, devs in (([0] * B, [0]), ([0] * 8 + [1] * 4, [0, 0])):
        pm, nm, _, vm = dev.unlimited_multi_dev(blocks, dob, devs, ensize_K=K, rN_seed=2103, viewflag=True)
        assert nm == n1 and np.array_equal(pm, p1) and np.array_equal(vm, v4), devs
