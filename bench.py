#!/usr/bin/env python3
"""Headline benchmark: cells/sec, end-to-end SHARP (fixed genes, n.RP) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL).  W untimed warm-up
steps, then exactly K timed steps bracketed by barrier + device synchronisation on both sides, MAX over
ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
  N = 1 (default, --config cfg2): BASELINE.json configs[1]: one SHARP() call on 50 000 cells x 20 000 genes,
          ensize.K = 15 (SHARP_large: projectors, RP matmul, 375 base-clustering tasks, 25 wMetaC, sMetaC).
  N = 1, --config cfg3: BASELINE.json configs[2]: SHARP_unlimited on 500 000 cells x 20 000 genes as 10 blocks, K = 5.
          (The default run also times cfg3 once, after the timed region, and reports it under "other_configs".)
  N > 1, and N = 1 with --config cfg4: BASELINE.json configs[3]: SHARP_unlimited on 1.3 M cells x 27 000 genes, ensize.K = 5, as
          its EIGHT blocks of 162 500 cells whatever N is, block b on GPU b mod N (one per GPU at N = 8; all eight one after the other
          on the one GPU at N = 1: 140 GB of X), p = 508 from the global count: the total problem and its labels are the same for
          every N ("scaling": "strong"), so `--gpus 1 --config cfg4` is the N = 1 point of the curve the N > 1 runs draw (the default
          run reports it under "other_configs.cfg4_one_gpu").  The only data-path collective is the all-gather of the per-block
          centroid table before the final sMetaC (sharp_amd/dist.py).
The JSON carries `roofline` for the RP matmul stage (rp_pc_kernel, rp3.hip; HBM-bound: X is read once for
all K projectors) from HIP events on the library's streams inside the timed region, the same stage at the
K = 5 shapes of cfg3 and of cfg4's per-GPU share (`roofline.by_config`), and `cpu_baseline`: the fp64 CPU oracle (a
port of the reference's R path, not R itself) timed on the host cores on a bounded sample."""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA_SEED = 20261003
RN_SEED = 2103
G_TRUE, N_MARK = 12, 1000
CFG2 = dict(cells=50000, genes=20000, K=15)
CFG3 = dict(cells=500000, genes=20000, K=5, blocks=10)
CFG4 = dict(cells=1300000, genes=27000, K=5, blocks=8)     # configs[3]: one block per GPU of an 8-GPU node
METRIC = "cells/sec end-to-end SHARP (fixed genes, n.RP); ARI vs reference labels"


def rp_stage_numbers(prof, n, m, K, p, steps_of):
    """Roofline numbers of the RP matmul stage from the library's HIP-event table: per SHARP() call the stage reads
    n*m*4 B of X (once for all K projectors) and writes n*K*p*4 B of E (SURVEY.md 8d's algorithmic bytes)."""
    cms, ccalls = prof.get("rp_compact", (0.0, 0))
    ams, acalls = prof.get("rp_apply", (0.0, 0))
    sms, scalls = prof.get("rp_stage", (0.0, 0))
    pms, pcalls = prof.get("rp_pc", (0.0, 0))            # the stage as ONE producer / consumer kernel (rp3.hip): the default form
    if not scalls:
        return None
    # chunks compacted beside the projector build (rp_compact_ahead, second stream) are the stage's work too: their time is ADDED, as if
    # they had run where they used to, behind the build
    ahead_ms = prof.get("rp_stage_ahead", (0.0, 0))[0]
    main_ms = sms
    sms += ahead_ms
    t_stage = sms / scalls * 1e-3
    read_b, write_b = n * m * 4, n * K * p * 4
    out = {"ms": round(t_stage * 1e3, 4), "cells": n, "genes": m, "n_RP": K, "reduced_dim": p,
           "algorithmic_read_bytes": read_b, "algorithmic_write_bytes": write_b,
           "achieved_read": round(read_b / t_stage / 1e9, 1), "frac_read": round(read_b / t_stage / 8e12, 4),
           "frac_read_write": round((read_b + write_b) / t_stage / 8e12, 4), "launches_per_stage": round((ccalls + acalls + pcalls) / scalls, 2)}
    if pcalls:
        tl = pms / pcalls * 1e-3
        bl = read_b / (pcalls / scalls)
        out["rp_pc_kernel"] = {"launch_ms": round(tl * 1e3, 4), "algorithmic_bytes": int(bl), "achieved": round(bl / tl / 1e9, 1),
                               "frac": round(bl / tl / 8e12, 4)}
    if ccalls:
        tl = cms / ccalls * 1e-3
        bl = read_b / (ccalls / scalls)
        out["rp_compact_kernel"] = {"launch_ms": round(tl * 1e3, 4), "algorithmic_bytes": int(bl), "achieved": round(bl / tl / 1e9, 1),
                                    "frac": round(bl / tl / 8e12, 4)}
    if acalls:
        out["rp_apply_kernel"] = {"launch_ms": round(ams / acalls, 4)}
    if ahead_ms:
        out["ms_main_stream"] = round(main_ms / scalls, 4)           # what the step still sees of the stage
        out["ms_ahead_stream"] = round(ahead_ms / scalls, 4)         # the chunks compacted beside the projector draw (they share the chip with it)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4"], default="cfg2", help="N = 1 workload (N > 1 always runs cfg4)")
    ap.add_argument("--cells", type=int, default=0, help="override the TOTAL number of cells of the workload (tests)")
    ap.add_argument("--genes", type=int, default=0, help="override the number of genes (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg3 / cfg4-share measurements after the timed region")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic under rocprofv3 (two child processes after the timed region)")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
    # test hook: run several ranks on ONE GPU with gloo (a 1-GPU box cannot host an RCCL world of 2)
    share_gpu = os.environ.get("SHARP_BENCH_SHARE_GPU") == "1"
    backend = os.environ.get("SHARP_BENCH_BACKEND", "nccl")
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    xdev = "cuda" if backend == "nccl" else "cpu"      # where the centroid tables are exchanged

    import sharp_amd
    from sharp_amd import device as dev
    from sharp_amd import dist as sdist
    from sharp_amd.api import ARI

    sharp_amd.init(local_rank)
    lib = sharp_amd.lib()

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        lib.sharp_synchronize()

    def synth_block(cell0, n, m):
        x = torch.empty((n, m), dtype=torch.float32, device="cuda")
        dev.synth_fill(x, DATA_SEED, cell0, G_TRUE, N_MARK)
        return x

    def unlimited_call(blocks, K):
        """sharp_SHARP_unlimited_dev on resident blocks -> (pred, n_pred, p)"""
        B = len(blocks)
        ptrs = (C.c_void_p * B)(*[b.data_ptr() for b in blocks])
        ncb = np.array([b.shape[0] for b in blocks], np.int64)
        ldb = np.array([b.stride(0) for b in blocks], np.int64)
        pred = np.zeros(int(ncb.sum()), np.int32)
        npred, pu = C.c_int(), C.c_int()
        rc = lib.sharp_SHARP_unlimited_dev(ptrs, ncb.ctypes.data_as(C.POINTER(C.c_longlong)), ldb.ctypes.data_as(C.POINTER(C.c_longlong)),
                                           B, blocks[0].shape[1], K, 0, 0, 0, C.c_double(RN_SEED), pred.ctypes.data_as(C.POINTER(C.c_int)),
                                           C.byref(npred), C.byref(pu))
        if rc not in (0, 16, 32, 48):
            raise RuntimeError(lib.sharp_last_error().decode())
        return pred, npred.value, pu.value

    # ---- workload of this run: synthetic blocks generated on the device (counter-based: identical on CPU and GPU)
    state = {}
    if world > 1 or args.config == "cfg4":
        cfg, tag = dict(CFG4), "cfg4"
    else:
        cfg, tag = (dict(CFG2), "cfg2") if args.config == "cfg2" else (dict(CFG3), "cfg3")
    sharded = tag == "cfg4"                                  # blocks dealt to the ranks (all of them to the one rank at N = 1)
    if args.cells:
        cfg["cells"] = args.cells
    if args.genes:
        cfg["genes"] = args.genes
    n_total, m, K = cfg["cells"], cfg["genes"], cfg["K"]
    if sharded:
        # The data set is cut into the EIGHT blocks of configs[3] whatever N is (one block per GPU at N = 8; at N = 2 / 4 a rank runs 4 / 2
        # blocks one after the other, block b on rank b mod N), so every N clusters the same blocks and finds the same labels: strong
        # scaling of one fixed problem.  (Tests shrink the data set: blocks below 5000 cells would leave the SHARP_large path, so then one per rank.)
        B = cfg["blocks"] if n_total // cfg["blocks"] >= 5000 and cfg["blocks"] >= world else world
        bounds = [n_total * b // B for b in range(B + 1)]
        ncb = [bounds[b + 1] - bounds[b] for b in range(B)]
        mine = [b for b in range(B) if sdist.block_owner(b, world) == rank]
        blocks = [synth_block(bounds[b], ncb[b], m) for b in mine]
        truth = np.concatenate([dev.synth_labels(DATA_SEED, bounds[b], ncb[b], G_TRUE) for b in mine])
        dX = blocks[0]

        def step():
            p = sdist.global_reduced_dim(n_total)                    # R/SHARP_unlimited.R:65-66: from the GLOBAL cell count
            proj = sharp_amd.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])   # :97-104, regenerated on every rank

            def run_block(blk, p_, nxt):                             # (nxt: this rank's next block, prepared under this one's tail)
                return dev.unlimited_block_dev(blk, p_, proj.handle, K, RN_SEED, next_block=nxt)

            out, nfin, p = sdist.unlimited_sharded(blocks, mine, ncb, run_block, dev.unlimited_merge, device=xdev)
            proj.close()
            state["p"], state["pred"], state["n_clusters"] = p, np.concatenate([out[b] for b in mine]), nfin
        workload = ("SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d cells, block b on GPU b mod %d, ensize.K=%d, rN.seed=%d"
                    % (n_total, m, B, ncb[0], world, K, RN_SEED))
    elif tag == "cfg2":
        dX = synth_block(0, n_total, m)
        truth = dev.synth_labels(DATA_SEED, 0, n_total, G_TRUE)

        def step():
            pred, info = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN_SEED)
            state["p"], state["pred"], state["n_clusters"] = info["reduced.dim"], pred, info["N.pred_cluster"]
        workload = "SHARP() on synthetic %d cells x %d genes, ensize.K=%d, SHARP_large path, rN.seed=%d" % (n_total, m, K, RN_SEED)
    else:
        B = cfg["blocks"]
        nb = n_total // B
        blocks = [synth_block(b * nb, nb, m) for b in range(B)]
        truth = np.concatenate([dev.synth_labels(DATA_SEED, b * nb, nb, G_TRUE) for b in range(B)])

        def step():
            pred, npred, p = unlimited_call(blocks, K)
            state["p"], state["pred"], state["n_clusters"] = p, pred, npred
        workload = ("SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d, ensize.K=%d, rN.seed=%d"
                    % (n_total, m, B, nb, K, RN_SEED))
    torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    dev.profile(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    prof_timed = dev.profile_table()
    # Attribution pass, outside the timed region: in the timed steps the two chunks of base-clustering tasks are in flight together
    # (SHARP_HC_PIPE) and a chunk of 194 tasks or more runs as two ranges on two streams, so the per-kernel event times overlap and
    # kernels sharing the chip run slower than alone; one more step with one chunk and one range at a time gives each kernel's own time.
    os.environ["SHARP_HC_RANGES"] = "1"
    os.environ["SHARP_HC_PIPE"] = "0"
    sharp_amd.reload_options()                           # (the library reads its switches once; this asks it to read them again)
    dev.profile(True)
    step()
    barrier()
    prof = dev.profile_table()
    del os.environ["SHARP_HC_RANGES"]
    del os.environ["SHARP_HC_PIPE"]
    sharp_amd.reload_options()
    dev.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    ms_per_step = dt / args.steps * 1e3
    cells_per_s = n_total * args.steps / dt

    if rank == 0:
        p = state["p"]
        n_local = int(dX.shape[0]) if tag != "cfg3" else n_total // cfg["blocks"]     # cells per SHARP() call (per block)
        # the RP matmul as BASELINE.json's north_star defines it: the whole stage (the compaction kernel streams X, the apply kernel does
        # the sparse-ternary accumulation and writes E), timed inside the timed region by HIP events on the library's streams
        st = rp_stage_numbers(prof_timed, n_local, m, K, p, args.steps)
        roof = None
        if st:
            traffic, tsrc = None, None
            tf = os.path.join(ROOT, "profiles", "rp_traffic.json")
            if tag == "cfg2" and n_total == CFG2["cells"] and m == CFG2["genes"]:
                if world == 1 and not args.no_traffic:
                    traffic, tsrc = measure_traffic()           # two rocprofv3 --pmc child processes, this process idle meanwhile
                if traffic is None and os.path.exists(tf):
                    tj = json.load(open(tf))
                    traffic = tj.get("hbm_bytes_per_launch")
                    tsrc = "profiles/rp_traffic.json: rocprofv3 --pmc passes of this command on the builder's box (not measured in this run)"
            roof = {"kernel": ("RP matmul stage = rp_pc_kernel (one persistent producer / consumer kernel), per SHARP() call (per block)" if "rp_pc_kernel" in st
                               else "RP matmul stage = rp_compact_kernel + rp_apply_kernel, per SHARP() call (per block)"), "bound": "hbm",
                    "achieved": st["achieved_read"], "peak": 8000.0, "unit": "GB/s", "frac": st["frac_read"],
                    "traffic": traffic, "traffic_source": tsrc,
                    "what": "algorithmic bytes = X read once for all K projectors (cells x genes x 4 B, SURVEY.md 8d: the HBM-read roofline "
                            "north_star names) / stage time from HIP events in the timed region",
                    "stage": st}
        stages = {k: round(v[0], 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        # the two other heavy kernels, for context (from the single-range attribution step)
        others = []
        if tag == "cfg2":
            T_tasks = K * len(range(0, n_total, 2000))
            gms, gcalls = prof.get("corr_dist_gemm", (0.0, 0))
            if gcalls:
                fl = T_tasks * 2000.0 * 2000.0 * p                   # upper triangle of n_t^2 * p * 2 flop per task
                tg = gms * 1e-3
                others.append({"kernel": "gemm_tn_f64_fast_kernel", "bound": "mfma", "achieved": round(fl / tg / 1e12, 1), "peak": 78.6,
                               "unit": "TFLOP/s", "frac": round(fl / tg / 78.6e12, 3), "ms_per_step": round(tg * 1e3, 2),
                               "work": "%d tasks x n_t^2 x p flop (upper triangle), f64 MFMA" % T_tasks})
            hms, hcalls = prof.get("hclust", (0.0, 0))
            if hcalls:
                by = T_tasks * 2000.0 * 2000.0 * 8 * 10.1            # rounds of (read n_a^2 + write n_a'^2) + the round-0 scan
                th = hms * 1e-3
                others.append({"kernel": "hclust_rnn_kernel", "bound": "hbm", "achieved": round(by / th / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                               "frac": round(by / th / 8e12, 3), "ms_per_step": round(th * 1e3, 2),
                               "work": "%d tasks x 10.1 n_t^2 x 8 B as the kernel streams it (every round rewrites the distance matrix "
                                       "compacted); the distance matrices themselves are %d x n_t^2 x 8 B = %.1f GB"
                                       % (T_tasks, T_tasks, T_tasks * 2000.0 * 2000.0 * 8 / 1e9)})
        result = {
            "metric": METRIC,
            "value": round(cells_per_s, 1), "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": {"cfg2": "configs[1]", "cfg3": "configs[2]", "cfg4": "configs[3]"}[tag],
                       "cells_total": n_total, "cells_per_gpu": (sum(int(b.shape[0]) for b in blocks) if sharded else int(dX.shape[0])) if tag != "cfg3" else n_total, "cells_per_block": n_local, "genes": m, "n_RP": K,
                       "reduced_dim": p, "x_storage": "fp32 in HBM (synthetic counts are fp32-exact)",
                       "parallelism": ("%d blocks, block b on GPU b mod %d; one all-gather of the per-block centroid tables" % (len(ncb), world)) if sharded else "single GPU"},
            "roofline": roof,
            "other_kernels": others,
            "kernel_ms_per_step": stages,
            "kernel_ms_note": "per-kernel times from one extra step outside the timed region with one chunk of base-clustering tasks and one task range at a time (SHARP_HC_PIPE=0, SHARP_HC_RANGES=1); the timed steps keep two chunks in flight, so these add up to more than ms_per_step",
            "clusters_found": int(state["n_clusters"]),
            "ari_vs_planted_truth": round(float(ARI(truth, state["pred"])["HA"]), 4),
        }
        if world == 1 and tag == "cfg2" and not args.no_extra and not args.cells and not args.genes:
            del dX
            torch.cuda.empty_cache()
            result["other_configs"], by_cfg = extra_configs(np, torch, sharp_amd, dev, lib, synth_block, unlimited_call, ARI)
            if roof:
                roof["by_config"] = by_cfg
                if by_cfg.get("cfg2_alone"):
                    # `frac` above prices the stage's kernels as they run in the step, a third of them beside the projector draw (slower each,
                    # faster step); the same kernels with the chip to themselves:
                    roof["frac_alone"] = by_cfg["cfg2_alone"]["frac_read"]
                    roof["note"] = ("frac: stage time = main-stream stage + the second stream's compaction beside the projector build (roofline.stage.ms_*); "
                                    "frac_alone: the same stage enqueued alone on an idle chip (by_config.cfg2_alone); SHARP_RP_AHEAD=0 gives the latter in the step, 2.5 ms slower")
            dX = synth_block(0, n_total, m)
        if world == 1 and tag == "cfg2" and not args.no_cpu_baseline:
            result["cpu_baseline"], result["parity"] = cpu_baseline(np, dX, m, K)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


TRAFFIC_CALLS = 8


def traffic_child():
    """The RP matmul stage of the cfg2 workload alone (the same sharp_project_dev calls as rp_stage_alone), as the program
    `rocprofv3 --pmc ...` runs: nothing but the stage's kernels touches the L2 counters."""
    import numpy as np
    import torch

    import sharp_amd
    from sharp_amd import device as dev

    torch.cuda.set_device(0)
    sharp_amd.init(0)
    lib = sharp_amd.lib()
    n, m, K = CFG2["cells"], CFG2["genes"], CFG2["K"]
    p = int(np.ceil(np.log2(n) / 0.04))
    x = torch.empty((n, m), dtype=torch.float32, device="cuda")
    dev.synth_fill(x, DATA_SEED, 0, G_TRUE, N_MARK)
    proj = sharp_amd.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])
    dE = torch.empty((n, K * p), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()
    for _ in range(TRAFFIC_CALLS):
        rc = lib.sharp_project_dev(proj.handle, C.c_void_p(x.data_ptr()), m, n, C.c_longlong(x.stride(0)), 1, C.c_void_p(dE.data_ptr()),
                                   C.c_longlong(K * p))
        if rc:
            raise RuntimeError(lib.sharp_last_error().decode())
    lib.sharp_synchronize()
    return 0


def measure_traffic():
    """roofline.traffic measured in THIS run: HBM bytes of the RP stage's kernels per stage (= per SHARP() call) from the L2's
    memory-side counters, FETCH_SIZE and WRITE_SIZE in separate rocprofv3 passes (they do not fit one pass), each pass a child process
    that runs the stage alone (`--traffic-child`); FETCH_SIZE (KiB of 64-byte requests) doubled, as MI355X_MICROARCH.md prescribes for
    wide coalesced reads on gfx950.  None if rocprofv3 is not on the box or a pass fails."""
    import csv
    import glob
    import re
    import shutil
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None, None
    tot = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            with tempfile.TemporaryDirectory(dir="/tmp") as td:
                env = dict(os.environ, TMPDIR="/tmp")
                r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", td, "--", sys.executable,
                                    os.path.abspath(__file__), "--traffic-child"], cwd="/tmp", env=env, timeout=240,
                                   stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
                if r.returncode != 0:
                    return None, None
                val, launches = 0.0, 0
                for f in glob.glob(os.path.join(td, "**", "*counter_collection.csv"), recursive=True):
                    for row in csv.DictReader(open(f)):
                        k = re.sub(r"\(.*$", "", row["Kernel_Name"])
                        if ("rp_pc_kernel" in k or "rp_compact_kernel" in k or "rp_apply_kernel" in k) and row["Counter_Name"] == counter:
                            val += float(row["Counter_Value"])
                            launches += 1
                if launches == 0:
                    return None, None
                tot[counter] = val * 1024.0 / TRAFFIC_CALLS
        traffic = int(2.0 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"])
        return traffic, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one child process each running the stage alone "
                         "%d times; FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B); read %.2f GB + write %.2f GB per stage"
                         % (TRAFFIC_CALLS, 2.0 * tot["FETCH_SIZE"] / 1e9, tot["WRITE_SIZE"] / 1e9))
    except Exception:
        return None, None


def rp_stage_alone(torch, sharp_amd, dev, lib, x, K, p):
    """The RP matmul stage by itself on a resident block (sharp_project_dev, as tools/bench_rp.py): HIP-event time of the stage."""
    n, m = x.shape
    proj = sharp_amd.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])
    dE = torch.empty((n, K * p), dtype=torch.float64, device="cuda")
    torch.cuda.synchronize()

    def call():
        rc = lib.sharp_project_dev(proj.handle, C.c_void_p(x.data_ptr()), m, n, C.c_longlong(x.stride(0)), 1, C.c_void_p(dE.data_ptr()),
                                   C.c_longlong(K * p))
        if rc:
            raise RuntimeError(lib.sharp_last_error().decode())
    for _ in range(3):
        call()
    dev.profile(True)
    reps = 10
    for _ in range(reps):
        call()
    lib.sharp_synchronize()
    prof = dev.profile_table()
    dev.profile(False)
    proj.close()
    del dE
    return rp_stage_numbers(prof, n, m, K, p, reps)


def extra_configs(np, torch, sharp_amd, dev, lib, synth_block, unlimited_call, ARI):
    """After the timed region of the default run: BASELINE.json's largest single-GPU configuration (cfg3) end to end, and the RP matmul
    stage at the K = 5 shapes (a block of cfg3; one GPU's share of cfg4), so that the driver's box produces these numbers too."""
    out, by_cfg = {}, {}
    # ---- cfg3: 500 000 x 20 000 as 10 blocks, SHARP_unlimited, K = 5
    m, K, B = CFG3["genes"], CFG3["K"], CFG3["blocks"]
    nb = CFG3["cells"] // B
    blocks = [synth_block(b * nb, nb, m) for b in range(B)]
    truth = np.concatenate([dev.synth_labels(DATA_SEED, b * nb, nb, G_TRUE) for b in range(B)])
    torch.cuda.synchronize()
    unlimited_call(blocks, K)                                            # warm-up (workspaces)
    reps = 2
    lib.sharp_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pred, npred, p = unlimited_call(blocks, K)
    lib.sharp_synchronize()
    dt = (time.perf_counter() - t0) / reps
    out["cfg3"] = {"workload": "SHARP_unlimited on synthetic %d cells x %d genes as %d blocks, ensize.K=%d (BASELINE.json configs[2])"
                               % (CFG3["cells"], m, B, K),
                   "value": round(CFG3["cells"] / dt, 1), "unit": "cells/s", "seconds_per_call": round(dt, 4), "calls_timed": reps,
                   "reduced_dim": p, "clusters_found": int(npred), "ari_vs_planted_truth": round(float(ARI(truth, pred)["HA"]), 4)}
    # the cfg2 shape (K = 15, p = 391) the same way: the stage alone on an idle chip, projectors resident -- the kernels' own number, beside the
    # in-step one of `roofline.stage`, where the chunks compacted ahead share the chip with the projector draw and take 1.4x as long
    by_cfg["cfg2_alone"] = rp_stage_alone(torch, sharp_amd, dev, lib, blocks[0], CFG2["K"], int(np.ceil(np.log2(CFG2["cells"]) / 0.04)))
    by_cfg["cfg3_block"] = rp_stage_alone(torch, sharp_amd, dev, lib, blocks[0], K, p)   # (inside the call above the next block's RP stage runs
    del blocks                                                                             #  on a low-priority stream beside the current block's tail)
    torch.cuda.empty_cache()
    # ---- cfg4's per-GPU share at N = 8: one 162 500 x 27 000 block, K = 5, p = 508: the RP stage alone, and the block step
    m, K = CFG4["genes"], CFG4["K"]
    nb, p = CFG4["cells"] // 8, 508
    x = synth_block(0, nb, m)
    proj = sharp_amd.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])
    dev.unlimited_block_dev(x, p, proj.handle, K, RN_SEED)              # warm-up
    lib.sharp_synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        pr, mn, cn = dev.unlimited_block_dev(x, p, proj.handle, K, RN_SEED)
    lib.sharp_synchronize()
    dt = (time.perf_counter() - t0) / reps
    proj.close()
    out["cfg4_share"] = {"workload": "one GPU's block of BASELINE.json configs[3] at N = 8: %d cells x %d genes, ensize.K=%d, p=%d, "
                                     "sharp_unlimited_block_dev (projectors resident)" % (nb, m, K, p),
                         "value": round(nb / dt, 1), "unit": "cells/s", "seconds_per_call": round(dt, 4), "calls_timed": reps,
                         "clusters_found": int(mn.shape[0])}
    by_cfg["cfg4_share"] = rp_stage_alone(torch, sharp_amd, dev, lib, x, K, p)
    # ---- cfg4 whole on ONE GPU: the same eight 162 500-cell blocks the N > 1 runs deal out, one after the other here (140 GB of X
    # resident): the N = 1 point of the strong-scaling curve (`--gpus 1 --config cfg4` times it as the headline value)
    from sharp_amd import dist as sdist

    B = CFG4["blocks"]
    blocks = [x] + [synth_block(b * nb, nb, m) for b in range(1, B)]
    truth = np.concatenate([dev.synth_labels(DATA_SEED, b * nb, nb, G_TRUE) for b in range(B)])
    torch.cuda.synchronize()

    def cfg4_step():
        pg = sdist.global_reduced_dim(nb * B)
        pj = sharp_amd.Projector(m, pg, [50 + RN_SEED + k for k in range(1, K + 1)])
        res, nfin, _ = sdist.unlimited_sharded(blocks, list(range(B)), [nb] * B,
                                               lambda blk, p_, nxt: dev.unlimited_block_dev(blk, p_, pj.handle, K, RN_SEED, next_block=nxt),
                                               dev.unlimited_merge, device="cuda")
        pj.close()
        return np.concatenate([res[b] for b in range(B)]), nfin, pg
    cfg4_step()                                                          # warm-up
    lib.sharp_synchronize()
    t0 = time.perf_counter()
    pred, nfin, pg = cfg4_step()
    lib.sharp_synchronize()
    dt = time.perf_counter() - t0
    out["cfg4_one_gpu"] = {"workload": "SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d cells, block b on GPU b mod 1, ensize.K=%d, "
                                       "rN.seed=%d (BASELINE.json configs[3] on one GPU: the N = 1 point of the curve `--gpus N` draws)"
                                       % (nb * B, m, B, nb, K, RN_SEED),
                           "value": round(nb * B / dt, 1), "unit": "cells/s", "seconds_per_call": round(dt, 4), "calls_timed": 1, "scaling": "strong",
                           "reduced_dim": pg, "clusters_found": int(nfin), "ari_vs_planted_truth": round(float(ARI(truth, pred)["HA"]), 4)}
    del x, blocks
    torch.cuda.empty_cache()
    return out, by_cfg


def cpu_baseline(np, dX, m, K):
    """The oracle (CPU port of the reference path) on a bounded sample of the same workload, on the host
    cores of this box; plus the GPU-vs-oracle label agreement on that sample."""
    from oracle import pyoracle as orc
    from sharp_amd import device as dev
    from sharp_amd.api import ARI

    orc.build()
    present = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))           # the cores this process may run on (a GPU box hands out a share of the host)
    except AttributeError:
        avail = present
    # a sample with at least as many base-clustering tasks as cores (the oracle parallelises over the K*T task grid, one task
    # per thread): 2000-cell folds x K projections, between 4 000 and 16 000 cells
    folds = max(2, min(8, -(-avail // K)))
    ns = 2000 * folds
    Xs = dX[:ns].cpu().numpy().T.astype(np.float64)    # (genes, cells)
    # The oracle parallelises over the K*T task grid and every thread holds a copy of its fold: it is memory-bound, and more threads
    # are not always faster (round 3: 672 cells/s on 120 threads, 947 on 30).  The baseline is the BEST of a few thread counts.
    tried = {}
    ref = None
    for cores in sorted({min(avail, c) for c in (32, 64, 128)}):
        cores = max(1, min(cores, folds * K))
        if cores in tried:
            continue
        t0 = time.perf_counter()
        r = orc.SHARP(Xs, K=K, base_ncells=1, rN_seed=RN_SEED, nthreads=cores, want_view=False)
        tried[cores] = time.perf_counter() - t0
        ref = ref or r
    cores = min(tried, key=tried.get)
    t = tried[cores]
    pred, _ = dev.SHARP_dev(dX[:ns], ensize_K=K, base_ncells=1, rN_seed=RN_SEED)
    ari = float(ARI(ref["pred_clusters"], pred)["HA"])
    base = {"value": round(ns / t, 2), "unit": "cells/s", "cores": cores, "cores_available": avail, "cores_present": present, "kind": "port",
            "threads_tried": {str(c): round(ns / v, 1) for c, v in sorted(tried.items())},
            "sample": "oracle SHARP_large on the first %d cells x %d genes of the same data (%d folds x %d RPs = %d tasks, OpenMP over "
                      "the K*T task grid; best of %s threads = %d, of the %d cores this process may use), %.1f s"
                      % (ns, m, folds, K, folds * K, "/".join(str(c) for c in sorted(tried)), cores, avail, t)}
    return base, {"ari_gpu_vs_oracle_on_sample": round(ari, 4), "sample_cells": ns,
                  "full_size": "tests/test_configs_gpu.py::test_cfg2_full_size_matches_oracle: this workload whole (50 000 x 20 000, K = 15, 375 base tasks), labels identical to the oracle's; ::test_full_size_block_matches_oracle: the same block at K = 5"}


if __name__ == "__main__":
    main()
