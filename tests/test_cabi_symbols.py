"""CPU-side check of the drop-in boundary: libsharp_hip.so loads without a GPU and exports
every symbol include/sharp_hip.h declares; without a device every compute entry fails loudly."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "sharp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sharp_[A-Za-z0-9_]+)\s*\(", src)))


@pytest.fixture(scope="module")
def so():
    import __graft_entry__ as g

    path = os.path.join(ROOT, "sharp_amd", "libsharp_hip.so")
    if not os.path.exists(path):
        g.build()
    return C.CDLL(path)


def test_header_declares_the_path():
    names = _declared()
    for must in ["sharp_projector_create", "sharp_project", "sharp_project_dev", "sharp_last_error", "sharp_init"]:
        assert must in names


def test_every_declared_symbol_is_exported(so):
    missing = [n for n in _declared() if not hasattr(so, n)]
    assert not missing, f"declared in include/sharp_hip.h but not exported: {missing}"


def test_no_device_fails_loudly(so):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    so.sharp_last_error.restype = C.c_char_p
    n = C.c_int(-1)
    assert so.sharp_device_count(C.byref(n)) == 0 and n.value == 0
    assert so.sharp_init(0) != 0
    assert b"no HIP device" in so.sharp_last_error()
    h = C.c_int()
    seeds = (C.c_double * 1)(2154.0)
    assert so.sharp_projector_create(100, 10, 1, seeds, C.byref(h)) != 0  # no context -> error, not a CPU path
    assert b"no device context" in so.sharp_last_error()


def test_dotc_convention_without_a_device(so):
    """The .C()-convention entry points (all-pointer arguments, void return, status out) report the missing device through
    *status and sharp_C_last_error(char **, int *) -- what r/sharp_hip.R turns into stop()."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    names = [n for n in _declared() if n.startswith("sharp_C_")]
    assert len(names) >= 20
    st = (C.c_int * 1)(-1)
    dev = (C.c_int * 1)(0)
    so.sharp_C_init.restype = None
    so.sharp_C_init(dev, st)
    assert st[0] == 3                                            # SHARP_ERR_NO_DEVICE
    buf = C.create_string_buffer(b" " * 255)
    msg = (C.c_char_p * 1)(C.addressof(buf))
    ln = (C.c_int * 1)(256)
    so.sharp_C_last_error.restype = None
    so.sharp_C_last_error(msg, ln)
    assert b"no HIP device" in buf.value
    h = (C.c_int * 1)(0)
    so.sharp_C_projector_create.restype = None
    so.sharp_C_projector_create((C.c_int * 1)(100), (C.c_int * 1)(10), (C.c_int * 1)(1), (C.c_double * 1)(2154.0), h, st)
    assert st[0] != 0 and h[0] == 0


def test_python_package_has_no_cpu_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import sharp_amd

    with pytest.raises(sharp_amd.SharpError):
        sharp_amd.ranM2(100, 10, 2154)


def test_product_does_not_import_oracle():
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "sharp_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"(import\s+oracle|from\s+oracle|liboracle|sharp_oracle\.c|oracle/)", txt):
                    bad.append(f)
    assert not bad, f"product files referencing the oracle: {bad}"


def test_list_of_blocks_entry_points_fail_loudly_without_a_device(so):
    """The round-4 entry points (lists of sparse blocks, the in-process multi-device runner with its worker and upload threads, marker
    genes over block lists) report the missing device through their status and sharp_last_error(); nothing computes on the CPU."""
    import numpy as np
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    so.sharp_last_error.restype = C.c_char_p
    m, sizes = 50, [7, 9]
    rng = np.random.default_rng(0)
    dense = [np.asfortranarray(rng.integers(0, 3, size=(m, n)).astype(np.float64)) for n in sizes]
    ncb = np.array(sizes, np.int64)
    pred = np.zeros(sum(sizes), np.int32)
    npred, pu = C.c_int(), C.c_int()
    dv = np.array([0, 0], np.int32)
    ptrs = (C.POINTER(C.c_double) * 2)(*[d.ctypes.data_as(C.POINTER(C.c_double)) for d in dense])
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))     # noqa: E731
    rc = so.sharp_SHARP_unlimited_multi(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), 2, m, 3, 0, 0, 0, C.c_double(7.0), ip(dv), 2, ip(pred),
                                        C.byref(npred), C.byref(pu), None)
    assert rc != 0 and b"no HIP device" in so.sharp_last_error()
    import scipy.sparse as sp

    cs = [sp.csc_matrix(d) for d in dense]
    cps = [c.indptr.astype(np.int32) for c in cs]
    ris = [c.indices.astype(np.int32) for c in cs]
    vxs = [c.data.astype(np.float64) for c in cs]
    cpp = (C.POINTER(C.c_int) * 2)(*[ip(a) for a in cps])
    rip = (C.POINTER(C.c_int) * 2)(*[ip(a) for a in ris])
    vxp = (C.POINTER(C.c_double) * 2)(*[a.ctypes.data_as(C.POINTER(C.c_double)) for a in vxs])
    rc = so.sharp_SHARP_unlimited_csc(cpp, rip, vxp, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), 2, m, 3, 0, 0, 0, C.c_double(7.0), ip(pred),
                                      C.byref(npred), C.byref(pu), None)
    assert rc != 0 and b"no device context" in so.sharp_last_error()
    out = np.zeros((m, 5))
    lab = np.ones(sum(sizes), np.int32)
    lab[::2] = 2
    rc = so.sharp_marker_genes_blocks_csc(cpp, rip, vxp, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), 2, m, ip(lab), 2, C.c_double(1e-5), 1,
                                          out.ctypes.data_as(C.POINTER(C.c_double)))
    assert rc != 0 and b"no device context" in so.sharp_last_error()
    n = C.c_int(-1)
    assert so.sharp_multi_timeline(None, 0, C.byref(n)) == 0 and n.value >= 0
