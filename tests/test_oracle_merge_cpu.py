"""The oracle's merge-from-centroids (oracle_unlimited_merge, R/SHARP_unlimited.R:163-183) against the oracle's own whole
SHARP_unlimited on the same blocks: the GPU tests of the sharded path use the former on centroid tables the device produced."""
import numpy as np

SEED = 20261003


def test_merge_from_centroids_equals_whole_unlimited(oracle):
    m, G, nm = 1200, 5, 120
    sizes = [5400, 5100, 600]                       # two SHARP_large blocks and a SHARP_small one
    offs = np.concatenate([[0], np.cumsum(sizes)])
    blocks = [oracle.synth_fill(SEED, m, int(offs[i]), sizes[i], G, nm) for i in range(3)]
    whole = oracle.SHARP_unlimited(blocks, K=3, rN_seed=2103, nthreads=8, want_view=True)
    p = whole["p"]
    tern = np.concatenate([oracle.ranM(m, p, 50 + 2103 + k).ravel() for k in range(1, 4)])
    means, counts, preds = [], [], []
    for b in blocks:
        r = oracle.SHARP(b, K=3, reduced_ndim=p, tern=tern, rN_seed=2103, nthreads=8)
        pr = r["pred_clusters"]
        preds.append(pr)
        for j in range(1, pr.max() + 1):
            rows = r["viE"][pr == j].astype(np.longdouble)
            acc = np.zeros(p, np.longdouble)
            for row in rows:                        # colMeans accumulates in cell order in long double
                acc += row
            means.append((acc / rows.shape[0]).astype(np.float64))
            counts.append(rows.shape[0])
    mg = oracle.unlimited_merge(np.array(means), np.array(counts), int(offs[-1]))
    first = np.concatenate([[0], np.cumsum([pr.max() for pr in preds])])
    pred = np.concatenate([mg["final_id"][first[i] + preds[i] - 1] for i in range(3)])
    assert mg["n_final"] == whole["pred_clusters"].max()
    assert np.array_equal(pred, whole["pred_clusters"])
