"""The oracle's side of the decision log (oracle_decision_log / oracle_last_decisions): its rows restate what
get_opt_hclust returned -- checked here against the function's own outputs -- and name the call's place in the run."""
import numpy as np

F = {n: i for i, n in enumerate(("level", "block", "k", "fold", "n", "branch", "chosen_k", "ties", "best", "runner_up", "sil_minus_thre",
                                 "height_ratio", "smetac_override_k", "levels"))}


def test_direct_call_rows_restate_the_outputs(oracle):
    rng = np.random.default_rng(11)
    E = rng.standard_normal((240, 30)) + rng.standard_normal((5, 30))[rng.integers(0, 5, 240)] * 2
    oracle.decision_log(True)
    try:
        a = oracle.get_opt_hclust(E)                          # silhouette rule
        b = oracle.get_opt_hclust(E, sil_thre=2.0)            # CH rule (every median below the threshold)
        c = oracle.get_opt_hclust(E, N_cluster=3)
        rows = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    assert rows.shape == (3, oracle.DECISION_COLS) and len(oracle.DECISION_FIELDS) == oracle.DECISION_COLS
    assert oracle.last_decisions().shape[0] == 0
    ra, rb, rc = rows                                          # (equal keys: insertion order)
    assert ra[F["level"]] == -1 and ra[F["n"]] == 240 and ra[F["levels"]] == a["msil"].size == 39
    assert ra[F["branch"]] == a["branch"] == 0 and ra[F["chosen_k"]] == a["optN"]
    assert ra[F["best"]] == a["msil"].max() and ra[F["ties"]] == (a["msil"] == a["msil"].max()).sum()
    below = a["msil"][a["msil"] < a["msil"].max()]
    assert ra[F["runner_up"]] == below.max() and ra[F["sil_minus_thre"]] == a["msil"].max() - 0.35
    assert rb[F["branch"]] == b["branch"] and b["branch"] in (1, 2) and rb[F["chosen_k"]] == b["optN"]
    assert rb[F["best"]] == b["CHind"].max() and rb[F["sil_minus_thre"]] == b["msil"].max() - 2.0
    if b["branch"] == 2:
        assert rb[F["height_ratio"]] > 1
    assert rc[F["branch"]] == 3 and rc[F["chosen_k"]] == 3 and rc[F["levels"]] == 1


def test_rows_name_their_place_in_the_run(oracle):
    X = oracle.synth_fill(20261003, 1500, 0, 530, 5, 250)
    oracle.decision_log(True)
    try:
        oracle.SHARP(X, K=3, base_ncells=100, partition_ncells=200, rN_seed=7)       # folds 200 / 165 / 165
        rows = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    assert rows.shape[0] == 3 * 3 + 3 + 1
    base = rows[rows[:, 0] == 0]
    assert [tuple(r) for r in base[:, [2, 3]].astype(int)] == [(k, t) for k in range(3) for t in range(3)]
    assert list(base[:, F["n"]].astype(int)) == [200, 165, 165] * 3
    assert list(rows[rows[:, 0] == 1][:, F["fold"]].astype(int)) == [0, 1, 2] and (rows[:, 0] == 2).sum() == 1
    blocks = [oracle.synth_fill(20261003, 1500, c0, n, 5, 250) for n, c0 in [(300, 0), (320, 400)]]
    oracle.decision_log(True)
    try:
        oracle.SHARP_unlimited(blocks, K=2, rN_seed=7)
        rows = oracle.last_decisions()
    finally:
        oracle.decision_log(False)
    assert set(rows[:, F["block"]].astype(int)) == {0, 1} and (rows[:, 0] == 3).sum() == 1 and rows[-1, 0] == 3
