"""The N > 1 path on the GPU box: (1) the sharded SHARP_unlimited driver with its collectives running through RCCL (a world of one
rank: a 1-GPU box cannot host two RCCL ranks) equals the single-call library result; (2) `bench.py --gpus 2` end to end, two ranks
sharing the GPU over gloo (the bench's test hook), prints one well-formed JSON line."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return str(p)


def test_sharded_unlimited_through_rccl_world_of_one(tmp_path, oracle):
    out = str(tmp_path / "rccl.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_rccl_worker.py"), _free_port(), out], env=env, timeout=900,
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    z = np.load(out)
    assert float(z["allreduce"]) == 1.0
    seed, m, nb, nblocks, K = 20261003, 1500, 5200, 3, 3
    blocks = [oracle.synth_fill(seed, m, b * nb, nb, 5, 250) for b in range(nblocks)]
    ref = oracle.SHARP_unlimited(blocks, K=K, rN_seed=2103, nthreads=8)
    assert int(z["p"]) == ref["p"]
    assert np.array_equal(z["pred"], ref["pred_clusters"])


@pytest.mark.parametrize("cells,genes,nblocks", [(12000, 20000, 2),     # blocks below 5000 cells would leave the SHARP_large path: one block per rank
                                                 (80000, 13000, 8)])     # the eight blocks of configs[3], four per rank (each prepared under the previous one's tail)
def test_bench_two_ranks_sharing_the_gpu(tmp_path, cells, genes, nblocks):
    env = dict(os.environ, SHARP_BENCH_SHARE_GPU="1", SHARP_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--cells", str(cells), "--genes", str(genes)]   # N > 1 runs cfg4 (1.3 M x 27 000 as eight blocks, block b on GPU b mod N), here shrunk
    r = subprocess.run(cmd, env=env, timeout=900, capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                         # rank 0 only, ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "strong" and d["unit"] == "cells/s"
    assert d["value"] == pytest.approx(cells * 2 / (d["ms_per_step"] * 2e-3), rel=1e-3)      # whole-job cells / max-over-ranks time
    assert d["config"]["workload"].startswith("SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d cells, block b on GPU b mod 2"
                                              % (cells, genes, nblocks, cells // nblocks))
    assert d["config"]["baseline_config"] == "configs[3]" and d["config"]["n_RP"] == 5 and d["config"]["cells_per_gpu"] == cells // 2
    assert d["config"]["cells_per_block"] == cells // nblocks
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert d["ari_vs_planted_truth"] > 0.85                        # (what the algorithm finds on this data; parity is tested elsewhere)
    assert len(d["labels_crc32_by_block"]) == nblocks and "cfg4_one_gpu" in d["scaling_curve"]["n1_point"]


def test_bench_default_line_names_cfg3_with_forview_roofline_and_cpu_baseline():
    """`python bench.py` (N = 1): the headline is BASELINE.json configs[2], the largest single-GPU configuration, under the steps / warm-up
    contract, with the forview step, the RP-stage roofline of its block shape (traffic measured in the run), the CPU baseline with its
    per-stage seconds and the GPU-vs-oracle ARI of every configuration's sample.  (--no-extra: the other configurations' timings are the
    driver's bench run; tools/dryrun_8ranks.sh and the two-rank test above cover N > 1.)"""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # the bench is a process of its own on the same GPU: this one gives back what earlier tests left in its worker / helper slots and in
    # torch's cache first
    import sharp_amd
    import torch

    if torch.cuda.is_initialized():
        sharp_amd.init(0)
        assert sharp_amd.lib().sharp_trim() == 0
        torch.cuda.empty_cache()
    # (--no-traffic: the two rocprofv3 child passes that measure roofline.traffic are the driver's bench run's; here the committed figure is named)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-extra", "--no-traffic"], env=env, timeout=1200,
                       capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["scaling"] == "weak" and d["config"]["baseline_config"] == "configs[2]" and d["steps"] == 3
    assert d["config"]["workload"].startswith("SHARP_unlimited on synthetic 500000 cells x 20000 genes as 10 blocks of 50000, ensize.K=5")
    assert d["config"]["reduced_dim"] == 474 and d["ari_vs_planted_truth"] > 0.9
    assert d["value"] == pytest.approx(500000 / (d["ms_per_step"] * 1e-3), rel=1e-3)
    assert d["consistency"]["ms_per_step_x_steps_s"] == pytest.approx(d["consistency"]["timed_region_s"], rel=1e-2)
    assert d["ms_per_step_forview"] > 0                          # (no timing is compared with another timing or a constant: a slower box must not fail parity)
    rf = d["roofline"]
    assert rf["kernel"].startswith("RP matmul stage") and rf["peak"] == 8000.0 and rf["bound"] == "hbm"
    assert rf["frac"] == pytest.approx(50000 * 20000 * 4 / (rf["stage"]["ms"] * 1e-3) / 8e12, rel=2e-3)
    # the stage is ONE launch of the producer / consumer kernel per block: its HIP-event launch time is the stage's time
    assert rf["stage"]["launches_per_stage"] == 1.0 and "rp_pc_kernel" in rf["kernel"]
    assert rf["stage"]["rp_pc_kernel"]["launch_ms"] > 0
    # roofline.traffic: HBM bytes of one launch, measured in the run when rocprofv3 is on the box (else the committed figure, labelled):
    # X once plus E once -- between 1.0 and 2.0 times the algorithmic read
    assert rf["traffic"] > 0 and ("measured in this run" in rf["traffic_source"] or "not measured in this run" in rf["traffic_source"])
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] <= cb["cores_available"] <= cb["cores_present"]
    assert set(cb["by_config"]) == {"cfg2", "cfg2_ch", "cfg3", "cfg4"} and cb["stage_seconds"]["base_clustering_thread_s"] > 0
    for k in ("cfg2", "cfg2_ch", "cfg3", "cfg4"):
        assert d["parity"][k]["ari_gpu_vs_oracle_on_sample"] >= 0.99, (k, d["parity"][k])
    assert d["parity"]["ari_gpu_vs_oracle_on_sample"] == d["parity"]["cfg3"]["ari_gpu_vs_oracle_on_sample"]
