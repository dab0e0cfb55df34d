"""N > 1 path on CPU: world_size-2 and world_size-8 gloo runs of the SHARP_unlimited sharding (block b on rank b mod N, all-gather of
the per-block centroid tables, redundant centroid-level sMetaC) must equal the single-process oracle result."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def test_block_ownership_and_global_p():
    from sharp_amd import dist as sdist

    assert [sdist.block_owner(b, 8) for b in range(10)] == [0, 1, 2, 3, 4, 5, 6, 7, 0, 1]
    assert sdist.global_reduced_dim(1_306_127) == 508      # cfg4
    assert sdist.global_reduced_dim(50_000) == 391


import pytest


@pytest.mark.parametrize("world,nblocks", [(2, 4),       # two blocks per rank
                                            (8, 10),      # the eight ranks of an 8-GPU node: ranks 0 and 1 hold two blocks, the others one (block b on rank b mod 8)
                                            (8, 5)])      # fewer blocks than ranks: three ranks hold nothing and still take part in the exchange
def test_unlimited_sharded_matches_single_process_oracle(tmp_path, oracle, world, nblocks):
    port = _free_port()
    env = dict(os.environ, OMP_NUM_THREADS="2")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), str(r), str(world), port,
                               str(tmp_path), str(nblocks)], env=env) for r in range(world)]
    for p in procs:
        assert p.wait(timeout=900) == 0
    seed, m, G, nm, nb, K = 20261003, 1500, 5, 250, 900, 3
    blocks = [oracle.synth_fill(seed, m, b * nb, nb, G, nm) for b in range(nblocks)]
    ref = oracle.SHARP_unlimited(blocks, K=K, rN_seed=2103, nthreads=4)
    got = np.zeros(nb * nblocks, np.int32)
    seen = set()
    for r in range(world):
        z = np.load(os.path.join(tmp_path, f"rank{r}.npz"))
        assert int(z["p"]) == ref["p"]
        for b in z["blocks"]:
            got[b * nb:(b + 1) * nb] = z[f"pred{b}"]
            seen.add(int(b))
    assert seen == set(range(nblocks))
    assert np.array_equal(got, ref["pred_clusters"])
