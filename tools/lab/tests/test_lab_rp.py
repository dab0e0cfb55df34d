"""LAB build only: the two-kernel form of the RP matmul (tools/lab/rp2.hip, SHARP_RP_KERNEL=split: compaction + apply, rounds 2-3) and its
compaction ahead of the projector build, against the product's producer / consumer kernel, bit for bit, and the oracle."""
import numpy as np
import pytest

SEED = 20261003


@pytest.fixture(scope="module")
def sa():
    import sharp_amd

    sharp_amd.init(0)
    return sharp_amd


def both(sa, monkeypatch, pr, X, logflag):
    monkeypatch.setenv("SHARP_RP_KERNEL", "split")
    E2 = pr.project(X, logflag=logflag)
    monkeypatch.setenv("SHARP_RP_KERNEL", "pc")
    E3 = pr.project(X, logflag=logflag)
    monkeypatch.delenv("SHARP_RP_KERNEL")
    return E2, E3


@pytest.mark.parametrize("m,n,K,logflag", [(1500, 96, 3, True), (1500, 96, 3, False), (2003, 130, 1, True), (6000, 700, 15, True),
                                            (4097, 1033, 5, True), (20000, 600, 5, True), (27000, 300, 5, True), (1024, 64, 5, True)])
def test_pc_kernel_equals_two_kernel_form_and_oracle(sa, oracle, monkeypatch, m, n, K, logflag):
    X = oracle.synth_fill(SEED, m, 0, n, 4, max(1, m // 8))
    p = int(np.ceil(np.log2(max(n, 2)) / 0.04)) if m < 20000 else 474
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E2, E3 = both(sa, monkeypatch, pr, X, logflag)
    assert np.array_equal(E2, E3)                     # the same integer sums
    for k in range(min(K, 2)):
        ref = oracle.project(X[:, :64], oracle.ranM(m, p, seeds[k]), logflag)
        np.testing.assert_allclose(E3[:64, k * p:(k + 1) * p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("m,n,p,K", [(900, 40, 600, 15), (64, 24, 150, 15)])   # two launch groups; every gene in overflow segments
def test_pc_kernel_launch_groups_and_overflow_segments(sa, oracle, monkeypatch, m, n, p, K):
    X = oracle.synth_fill(SEED, m, 0, n, 3, max(1, m // 4))
    seeds = [50 + 2103 + k for k in range(1, K + 1)]
    pr = sa.Projector(m, p, seeds)
    E2, E3 = both(sa, monkeypatch, pr, X, True)
    assert np.array_equal(E2, E3)
    ref = oracle.project(X, oracle.ranM(m, p, seeds[0]), True)
    np.testing.assert_allclose(E3[:, :p], ref, rtol=0, atol=2e-12 * np.abs(ref).max())


@pytest.mark.parametrize("rp_chunk", [None, "1500", "300"])
def test_compaction_ahead_of_the_projector_build(sa, oracle, monkeypatch, rp_chunk):
    # SHARP() draws its projectors per call; the block's compaction (needs X only) is started first and runs beside the draw
    # (rp_compact_ahead): every chunk in a buffer of its own (one chunk; four chunks), or -- more chunks than ring buffers -- the first
    # two ahead and the rest in rotation.  Same projections bit for bit as with SHARP_RP_AHEAD=0, same labels as the oracle.
    m, n, G, nm = 3000, 6000, 6, 300
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    ref = oracle.SHARP(X, K=3, base_ncells=300, partition_ncells=2000, rN_seed=2103, nthreads=4)
    kw = dict(ensize_K=3, base_ncells=300, partition_ncells=2000, rN_seed=2103, logflag=False, prep=False)
    if rp_chunk: monkeypatch.setenv("SHARP_RP_CHUNK", rp_chunk)
    monkeypatch.setenv("SHARP_RP_KERNEL", "split")       # (the compaction ahead belongs to the two-kernel form)
    res = sa.SHARP(X, **kw)
    assert sa.lib().sharp_trim() == 0                    # gives back the per-chunk entry buffers and the cached projector blocks ...
    res_again = sa.SHARP(X, **kw)                        # ... which the next call allocates anew
    monkeypatch.setenv("SHARP_RP_AHEAD", "1")            # behind the draw kernel only (the default also runs beside it)
    res_behind = sa.SHARP(X, **kw)
    monkeypatch.setenv("SHARP_RP_AHEAD", "0")
    res_plain = sa.SHARP(X, **kw)
    assert res["path"] == "SHARP_large"
    np.testing.assert_array_equal(res_behind["viE"], res_plain["viE"])
    for r in (res, res_again, res_behind, res_plain):
        assert np.array_equal(r["pred_clusters"], ref["pred_clusters"])
    np.testing.assert_array_equal(res["viE"], res_plain["viE"])
    np.testing.assert_array_equal(res_again["viE"], res_plain["viE"])


def test_compaction_ahead_serves_both_launch_groups(sa, oracle, monkeypatch):
    # K * reduced.ndim = 9000 components: two launch groups (13 + 2 projectors) read the SAME compacted lists -- with the compaction done
    # ahead of the projector build the second group must find them intact (every chunk kept its own buffer) and its cell queue rewound
    m, n, G, nm = 2000, 4500, 5, 200
    X = oracle.synth_fill(SEED, m, 0, n, G, nm)
    kw = dict(ensize_K=15, reduced_ndim=600, base_ncells=300, partition_ncells=2300, rN_seed=2103, logflag=False, prep=False)
    monkeypatch.setenv("SHARP_RP_KERNEL", "split")       # (the compaction ahead belongs to the two-kernel form)
    res = sa.SHARP(X, **kw)
    monkeypatch.setenv("SHARP_RP_AHEAD", "0")
    res_plain = sa.SHARP(X, **kw)
    ref = oracle.SHARP(X, K=15, reduced_ndim=600, base_ncells=300, partition_ncells=2300, rN_seed=2103, nthreads=8)
    assert res["path"] == "SHARP_large" and res["reduced.dim"] == 600
    np.testing.assert_array_equal(res["viE"], res_plain["viE"])
    for r in (res, res_plain):
        assert np.array_equal(r["pred_clusters"], ref["pred_clusters"])
