#!/usr/bin/env python3
"""Headline benchmark: cells/sec, end-to-end SHARP (fixed genes, n.RP) on MI355X.

Contract: `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched by
`python -m torch.distributed.run --nproc-per-node N ...` (one rank per GPU, RCCL).  W untimed warm-up
steps, then exactly K timed steps bracketed by barrier + device synchronisation on both sides, MAX over
ranks, rank 0 prints ONE JSON line.

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
  N = 1 (default, --config cfg3): BASELINE.json configs[2], the largest single-GPU configuration: SHARP_unlimited on 500 000 cells x
          20 000 genes as 10 blocks of 50 000, ensize.K = 5, p = 474 (per block: projection, 125 base-clustering tasks, 25 wMetaC, sMetaC;
          then the cross-block sMetaC on the blocks' centroid tables).
  N = 1, --config cfg2: BASELINE.json configs[1]: one SHARP() call on 50 000 cells x 20 000 genes, ensize.K = 15 (SHARP_large: projectors,
          RP matmul, 375 base-clustering tasks, 25 wMetaC, sMetaC).  (The default run times it too: "other_configs.cfg2".)
  N > 1, and N = 1 with --config cfg4: BASELINE.json configs[3]: SHARP_unlimited on 1.3 M cells x 27 000 genes, ensize.K = 5, as
          its EIGHT blocks of 162 500 cells whatever N is, block b on GPU b mod N (one per GPU at N = 8; all eight one after the other
          on the one GPU at N = 1: 140 GB of X), p = 508 from the global count: the total problem and its labels are the same for
          every N ("scaling": "strong"), so `--gpus 1 --config cfg4` is the N = 1 point of the curve the N > 1 runs draw (the default
          run reports it under "other_configs.cfg4_one_gpu").  The only data-path collective is the all-gather of the per-block
          centroid table before the final sMetaC (sharp_amd/dist.py).
`ms_per_step` is the step WITHOUT the view outputs; `ms_per_step_forview` the same step with the reference's default `forview = TRUE` /
`viewflag = TRUE` outputs requested (viE and x0: R/SHARP.R:46,717-731,844; R/SHARP_unlimited.R:214-232) and brought to the host.
The JSON carries `roofline` for the RP matmul stage (rp_pc_kernel, rp3.hip; HBM-bound: X is read once for all K projectors)
from HIP events on the library's streams inside the timed region, `roofline.traffic` from two rocprofv3 --pmc child passes over the same
stage, the same stage at the other shapes and on other value kinds (`roofline.by_config`), and `cpu_baseline`: the fp64 CPU oracle (a
port of the reference's R path, not R itself) timed on the host cores on a bounded sample, with its per-stage seconds."""
import argparse
import ctypes as C
import json
import os
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DATA_SEED = 20261003
RN_SEED = 2103
G_TRUE, N_MARK = 12, 1000
N_MARK_CH = 400          # the "CH-decided" data set of SURVEY.md 8d: base max median silhouette 0.26-0.30 <= sil.thre, so which.max(CHind) and the
                         # height-gap rule choose k (R/get_opt_hclust.R:194-210); N_MARK = 1000 gives 0.46-0.55 (silhouette-decided)
CFG2 = dict(cells=50000, genes=20000, K=15)
CFG3 = dict(cells=500000, genes=20000, K=5, blocks=10)
CFG4 = dict(cells=1300000, genes=27000, K=5, blocks=8)     # configs[3]: one block per GPU of an 8-GPU node
METRIC = "cells/sec end-to-end SHARP (fixed genes, n.RP); ARI vs reference labels"
SHAPES = {"cfg2": (50000, 20000, 15, 391), "cfg3": (50000, 20000, 5, 474), "cfg4": (162500, 27000, 5, 508)}    # one block: cells, genes, K, p


def rp_stage_numbers(prof, n, m, K, p, width=4):
    """Roofline numbers of the RP matmul stage from the library's HIP-event table: per SHARP() call (per block) the stage reads
    n*m*width B of X (once for all K projectors; width = the stored width, 4 or 8) and writes n*K*p*4 B of E."""
    cms, ccalls = prof.get("rp_compact", (0.0, 0))
    ams, acalls = prof.get("rp_apply", (0.0, 0))
    sms, scalls = prof.get("rp_stage", (0.0, 0))
    pms, pcalls = prof.get("rp_pc", (0.0, 0))            # the stage as ONE producer / consumer kernel (rp3.hip): the default form
    if not scalls:
        return None
    # chunks compacted beside the projector build (rp_compact_ahead, second stream) are the stage's work too: their time is ADDED
    ahead_ms = prof.get("rp_stage_ahead", (0.0, 0))[0]
    main_ms = sms
    sms += ahead_ms
    t_stage = sms / scalls * 1e-3
    read_b, write_b = n * m * width, n * K * p * 4
    out = {"ms": round(t_stage * 1e3, 4), "cells": n, "genes": m, "n_RP": K, "reduced_dim": p, "stored_width_bytes": width,
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


class Bench:
    """The pieces every workload shares: the library, synthetic blocks, the entry points as ctypes calls."""

    def __init__(self, np, torch, local_rank):
        import sharp_amd
        from sharp_amd import device as dev
        from sharp_amd import dist as sdist
        from sharp_amd.api import ARI

        self.np, self.torch, self.sa, self.dev, self.sdist, self.ARI = np, torch, sharp_amd, dev, sdist, ARI
        sharp_amd.init(local_rank)
        self.lib = sharp_amd.lib()

    def synth_block(self, cell0, n, m, nmark=N_MARK):
        x = self.torch.empty((n, m), dtype=self.torch.float32, device="cuda")
        self.dev.synth_fill(x, DATA_SEED, cell0, G_TRUE, nmark)
        return x

    def labels(self, cell0, n):
        return self.dev.synth_labels(DATA_SEED, cell0, n, G_TRUE)

    def unlimited_call(self, blocks, K, view=False):
        """SHARP_unlimited on resident blocks -> (pred, n_pred, p, viE or None).  view: viewflag = TRUE (R/SHARP_unlimited.R:214-232): above
        1e5 cells viE is E1 reduced to 50 columns by one more sparse projection, taken per block on the device, so only ncells x 50 doubles
        leave the GPU; x0 is the sparse one-hot matrix of the labels."""
        n = sum(int(b.shape[0]) for b in blocks)
        if view and (self._vbuf is None or self._vbuf.shape[0] != n):
            # (an R session allocates the result anew per call; first-touch page faults of a fresh numpy array are host noise, so it is kept)
            self._vbuf = self.np.zeros((n, 50 if n > 1e5 else int(self.np.ceil(self.np.log2(n) / 0.04))))
        pred, npred, p, viE = self.dev.unlimited_dev(blocks, ensize_K=K, rN_seed=RN_SEED, viewflag=view, viE_out=self._vbuf if view else None)
        if view:
            from sharp_amd.api import _one_hot

            self.last_x0 = _one_hot(pred, npred)
        return pred, npred, p, viE

    _vbuf = None

    def rp_stage_alone(self, x, K, p, reps=10):
        """The RP matmul stage by itself on a resident block (sharp_project_dev / _dev64, as tools/bench_rp.py): HIP-event time of the stage."""
        torch, lib, dev = self.torch, self.lib, self.dev
        n, m = x.shape
        f64 = x.dtype == torch.float64
        proj = self.sa.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])
        dE = torch.empty((n, K * p), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()
        entry = lib.sharp_project_dev64 if f64 else lib.sharp_project_dev

        def call():
            rc = entry(proj.handle, C.c_void_p(x.data_ptr()), m, n, C.c_longlong(x.stride(0)), 1, C.c_void_p(dE.data_ptr()), C.c_longlong(K * p))
            if rc:
                raise RuntimeError(lib.sharp_last_error().decode())
        for _ in range(3):
            call()
        dev.profile(True)
        for _ in range(reps):
            call()
        lib.sharp_synchronize()
        prof = dev.profile_table()
        dev.profile(False)
        proj.close()
        del dE
        return rp_stage_numbers(prof, n, m, K, p, 8 if f64 else 4)


def guarded(out, key, fn):
    """one failing side measurement must not cost the run its line"""
    try:
        out[key] = fn()
    except Exception as e:  # noqa: BLE001
        out[key] = {"error": "%s: %s" % (type(e).__name__, e)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", choices=["cfg2", "cfg2_ch", "cfg3", "cfg4"], default="cfg3", help="N = 1 workload (N > 1 always runs cfg4)")
    ap.add_argument("--cells", type=int, default=0, help="override the TOTAL number of cells of the workload (tests)")
    ap.add_argument("--genes", type=int, default=0, help="override the number of genes (tests)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the other configurations / shapes after the timed region")
    ap.add_argument("--no-traffic", action="store_true", help="do not measure roofline.traffic under rocprofv3 (two child processes after the timed region)")
    ap.add_argument("--no-forview", action="store_true", help="skip ms_per_step_forview")
    ap.add_argument("--traffic-child", default="", help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.traffic_child:
        return traffic_child(args.traffic_child)

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
    if world > 1 and "SHARP_HOST_THREADS" not in os.environ:
        # one process per GPU on ONE host: every rank's upload / tail-helper / host-loop pools are sized from its share of the cores
        try:
            os.environ["SHARP_HOST_THREADS"] = str(max(4, len(os.sched_getaffinity(0)) // world))
        except AttributeError:
            pass
    torch.cuda.set_device(local_rank)
    import torch.distributed as dist

    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    xdev = "cuda" if backend == "nccl" else "cpu"      # where the centroid tables are exchanged

    Bn = Bench(np, torch, local_rank)
    sharp_amd, dev, sdist, lib, ARI = Bn.sa, Bn.dev, Bn.sdist, Bn.lib, Bn.ARI

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        lib.sharp_synchronize()

    # ---- workload of this run: synthetic blocks generated on the device (counter-based: identical on CPU and GPU)
    state = {}
    if world > 1 or args.config == "cfg4":
        cfg, tag = dict(CFG4), "cfg4"
    else:
        cfg, tag = {"cfg2": (dict(CFG2), "cfg2"), "cfg2_ch": (dict(CFG2), "cfg2_ch"), "cfg3": (dict(CFG3), "cfg3")}[args.config]
    sharded = tag == "cfg4"                                  # blocks dealt to the ranks (all of them to the one rank at N = 1)
    nmark = N_MARK_CH if tag == "cfg2_ch" else N_MARK
    if args.cells:
        cfg["cells"] = args.cells
    if args.genes:
        cfg["genes"] = args.genes
    n_total, m, K = cfg["cells"], cfg["genes"], cfg["K"]
    full_size = not args.cells and not args.genes
    if sharded:
        # The data set is cut into the EIGHT blocks of configs[3] whatever N is (one block per GPU at N = 8; at N = 2 / 4 a rank runs 4 / 2
        # blocks one after the other, block b on rank b mod N), so every N clusters the same blocks and finds the same labels: strong
        # scaling of one fixed problem.  (Tests shrink the data set: blocks below 5000 cells would leave the SHARP_large path, so then one per rank.)
        B = cfg["blocks"] if n_total // cfg["blocks"] >= 5000 and cfg["blocks"] >= world else world
        bounds = [n_total * b // B for b in range(B + 1)]
        ncb = [bounds[b + 1] - bounds[b] for b in range(B)]
        mine = [b for b in range(B) if sdist.block_owner(b, world) == rank]
        blocks = [Bn.synth_block(bounds[b], ncb[b], m) for b in mine]
        truth = np.concatenate([Bn.labels(bounds[b], ncb[b]) for b in mine])
        n_local = ncb[0]

        def step(view=False):
            p = sdist.global_reduced_dim(n_total)                    # R/SHARP_unlimited.R:65-66: from the GLOBAL cell count
            proj = sharp_amd.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])   # :97-104, regenerated on every rank
            vi = state.setdefault("view_bufs", {})             # (kept between steps: first-touch page faults of fresh arrays are host noise)

            def run_block(blk, p_, nxt):                             # (nxt: this rank's next block, prepared under this one's tail)
                v, kd = None, 0
                if view:
                    kd = 50 if n_total > 1e5 else 0          # R/SHARP_unlimited.R:216-228: above 1e5 cells viE is E1 reduced to 50 columns
                    v = vi.setdefault(blk.data_ptr(), np.zeros((blk.shape[0], kd if kd else p_)))
                return dev.unlimited_block_dev(blk, p_, proj.handle, K, RN_SEED, next_block=nxt, viE=v, view_dim=kd)

            out, nfin, p = sdist.unlimited_sharded(blocks, mine, ncb, run_block, dev.unlimited_merge, device=xdev)
            proj.close()
            state["p"], state["pred"], state["n_clusters"] = p, np.concatenate([out[b] for b in mine]), nfin
            state["crc"] = {int(b): zlib.crc32(np.ascontiguousarray(out[b], np.int32).tobytes()) for b in mine}
        workload = ("SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d cells, block b on GPU b mod %d, ensize.K=%d, rN.seed=%d"
                    % (n_total, m, B, ncb[0], world, K, RN_SEED))
    elif tag in ("cfg2", "cfg2_ch"):
        dX = Bn.synth_block(0, n_total, m, nmark)
        blocks = [dX]
        truth = Bn.labels(0, n_total)
        n_local = n_total

        def step(view=False):
            pred, info = dev.SHARP_dev(dX, ensize_K=K, rN_seed=RN_SEED, forview=view, view_out=state.get("view_out"))
            state["p"], state["pred"], state["n_clusters"] = info["reduced.dim"], pred, info["N.pred_cluster"]
            if view:
                state["view_out"] = info["view_out"]       # (kept between steps: first-touch page faults of fresh arrays are host noise)
        workload = "SHARP() on synthetic %d cells x %d genes%s, ensize.K=%d, SHARP_large path, rN.seed=%d" % (
            n_total, m, " (%d marker genes per planted cluster: the CH-decided data set)" % nmark if tag == "cfg2_ch" else "", K, RN_SEED)
    else:
        B = cfg["blocks"]
        nb = n_total // B
        blocks = [Bn.synth_block(b * nb, nb, m) for b in range(B)]
        truth = np.concatenate([Bn.labels(b * nb, nb) for b in range(B)])
        n_local = nb

        def step(view=False):
            pred, npred, p, _ = Bn.unlimited_call(blocks, K, view=view)
            state["p"], state["pred"], state["n_clusters"] = p, pred, npred
        workload = ("SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d, ensize.K=%d, rN.seed=%d"
                    % (n_total, m, B, nb, K, RN_SEED))
    torch.cuda.synchronize()
    cells_per_gpu = sum(int(b.shape[0]) for b in blocks)

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
    dev.profile(False)
    if world > 1:
        tmax = torch.tensor([dt], dtype=torch.float64, device=xdev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    ms_per_step = dt / args.steps * 1e3
    cells_per_s = n_total * args.steps / dt
    crc_by_block = None
    if sharded:
        # a checksum of every block's final labels, in global block order: equal for every N (the same eight blocks, the same global p)
        parts = [state["crc"]]
        if world > 1:
            parts = [None] * world
            dist.all_gather_object(parts, state["crc"])
        merged = {}
        for d in parts:
            merged.update(d)
        crc_by_block = [merged[b] for b in sorted(merged)]

    # ---- the same step with the reference's default view outputs requested (outside the headline's timed region, same contract)
    ms_forview = None
    if not args.no_forview:
        fsteps = max(2, min(args.steps, 3))
        step(view=True)                                    # two warm-up calls: the view projector and its buffers exist once per tail-helper slot,
        step(view=True)                                    # and which helper takes which block's tail is decided at run time
        barrier()
        t0 = time.perf_counter()
        for _ in range(fsteps):
            step(view=True)
        barrier()
        dtv = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dtv], dtype=torch.float64, device=xdev)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dtv = float(tmax.item())
        ms_forview = dtv / fsteps * 1e3
        Bn._vbuf = None

    prof = prof_timed
    attribution_note = ("per-kernel HIP-event times summed over the timed steps / steps; the blocks' kernels overlap (pipelined chunks, tail helpers), "
                        "so these add up to more than ms_per_step")
    if tag in ("cfg2", "cfg2_ch") and world == 1:
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
        attribution_note = ("per-kernel times from one extra step outside the timed region with one chunk of base-clustering tasks and one task range at a "
                            "time (SHARP_HC_PIPE=0, SHARP_HC_RANGES=1); the timed steps keep two chunks in flight, so these add up to more than ms_per_step")

    if rank == 0:
        p = state["p"]
        # the RP matmul as BASELINE.json's north_star defines it: the whole stage (X streamed once, the sparse-ternary accumulation, E written),
        # timed inside the timed region by HIP events on the library's streams; per block
        st = rp_stage_numbers(prof_timed, n_local, m, K, p)
        roof = None
        if st:
            traffic, tsrc = None, None
            shape_key = {"cfg2": "cfg2", "cfg2_ch": "cfg2", "cfg3": "cfg3", "cfg4": "cfg4"}[tag]
            if full_size and world == 1 and not args.no_traffic:
                del blocks                                       # (the child processes need the HBM; the extras regenerate what they use)
                if tag in ("cfg2", "cfg2_ch"):
                    del dX
                blocks = None
                torch.cuda.empty_cache()
                traffic, tsrc = measure_traffic(shape_key)      # two rocprofv3 --pmc child processes, this process idle meanwhile
            tf = os.path.join(ROOT, "profiles", "rp_traffic.json")
            if traffic is None and os.path.exists(tf):
                tj = json.load(open(tf)).get(shape_key)
                if tj:
                    traffic = tj.get("hbm_bytes_per_launch")
                    tsrc = "profiles/rp_traffic.json: rocprofv3 --pmc passes of this command on the builder's box (not measured in this run)"
            roof = {"kernel": ("RP matmul stage = rp_pc_kernel (one persistent producer / consumer kernel), one launch per block of %d cells" % n_local
                               if "rp_pc_kernel" in st else "RP matmul stage = rp_compact_kernel + rp_apply_kernel, per block"), "bound": "hbm",
                    "achieved": st["achieved_read"], "peak": 8000.0, "unit": "GB/s", "frac": st["frac_read"],
                    "traffic": traffic, "traffic_source": tsrc,
                    "what": "algorithmic bytes per launch = the block's X read once for all K projectors (cells x genes x 4 B, SURVEY.md 8d: the HBM-read "
                            "roofline north_star names) / average launch time from HIP events in the timed region",
                    "stage": st}
        stages = {k: round(v[0] / (args.steps if prof is prof_timed else 1), 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0])}
        result = {
            "metric": METRIC,
            "value": round(cells_per_s, 1), "unit": "cells/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 2),
            "ms_per_step_forview": None if ms_forview is None else round(ms_forview, 2),
            "forview_note": "ms_per_step: labels only; ms_per_step_forview: the same step returning the reference's default view outputs to the host "
                            "(SHARP(): viE n x p and the soft matrix x0, R/SHARP.R:717-731,844; SHARP_unlimited(): viE reduced to 50 columns above 1e5 "
                            "cells and the one-hot x0, R/SHARP_unlimited.R:214-232)",
            "higher_is_better": True, "scaling": "strong" if sharded else "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": workload, "baseline_config": {"cfg2": "configs[1]", "cfg2_ch": "configs[1] shape, CH-decided data set (SURVEY.md 8d)",
                                                                 "cfg3": "configs[2]", "cfg4": "configs[3]"}[tag],
                       "cells_total": n_total, "cells_per_gpu": cells_per_gpu, "cells_per_block": n_local, "genes": m, "n_RP": K, "reduced_dim": p,
                       "x_storage": "fp32 in HBM (synthetic counts are fp32-exact)",
                       "parallelism": ("%d blocks, block b on GPU b mod %d; one all-gather of the per-block centroid tables" % (len(ncb), world)) if sharded else "single GPU"},
            "consistency": {"ms_per_step_x_steps_s": round(ms_per_step * args.steps / 1e3, 3), "timed_region_s": round(dt, 3)},
            # one process per GPU on ONE host: the host cores a rank sizes its upload pool, host loops and tail helpers from (SHARP_HOST_THREADS)
            "host_threads_per_rank": int(os.environ["SHARP_HOST_THREADS"]) if os.environ.get("SHARP_HOST_THREADS") else host_cores_available(),
            "roofline": roof,
            "kernel_ms_per_step": stages,
            "kernel_ms_note": attribution_note,
            "clusters_found": int(state["n_clusters"]),
            "ari_vs_planted_truth": round(float(ARI(truth, state["pred"])["HA"]), 4),
        }
        if crc_by_block is not None:
            result["labels_crc32_by_block"] = crc_by_block
        if sharded:
            # the N > 1 lines (and --config cfg4) run configs[3], a fixed problem ("strong"); the default --gpus 1 line runs configs[2]: the N = 1
            # point of the curve the N > 1 lines draw is that line's other_configs.cfg4_one_gpu (or `--gpus 1 --config cfg4`), not its headline
            result["scaling_curve"] = {"workload": "configs[3], 1.3 M cells x 27 000 genes as eight blocks, whatever N", "n1_point": "the `--gpus 1` line's other_configs.cfg4_one_gpu.value (same blocks on one GPU), or `--gpus 1 --config cfg4`"}
        lv = level_rules(prof_timed, args.steps)
        if lv:
            result["level_rule_per_step"] = lv
        if tag in ("cfg2", "cfg2_ch"):
            result["other_kernels"] = other_kernels(prof, n_total, K, p)
        if world == 1 and full_size:
            if blocks is not None:
                del blocks
                if tag in ("cfg2", "cfg2_ch"):
                    del dX
            torch.cuda.empty_cache()
            if not args.no_extra:
                result["other_configs"], by_cfg = extra_configs(Bn, tag)
                if roof:
                    roof["by_config"] = by_cfg
                c4 = result["other_configs"].get("cfg4_one_gpu", {})
                if "value" in c4:
                    # what `--gpus N` (N > 1) is to be compared with: the same eight blocks of configs[3] on this one GPU
                    result["scaling_curve"] = {"workload": "configs[3] (what --gpus N > 1 runs)", "n1_point_cells_per_s": c4["value"], "n1_point_ms_per_step": c4["ms_per_step"]}
            if not args.no_cpu_baseline:
                cb = {}
                guarded(cb, "r", lambda: cpu_baseline(Bn, tag))
                if isinstance(cb["r"], tuple):
                    result["cpu_baseline"], result["parity"] = cb["r"]
                else:
                    result["cpu_baseline"] = cb["r"]
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def host_cores_available():
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def level_rules(prof, steps):
    """which rule of R/get_opt_hclust.R:162-229 chose the level, per step: base tasks (cells) and meta tasks (wMetaC / sMetaC similarity trees)"""
    out = {}
    for kind in ("base", "meta"):
        d = {r: prof.get("host:level_%s_by_%s" % (kind, r), (0.0, 0))[1] / steps for r in ("silhouette", "CH", "height")}
        if sum(d.values()):
            out[kind] = {k: round(v, 2) for k, v in d.items()}
    return out


def other_kernels(prof, n_total, K, p):
    """the two other heavy kernels of a SHARP_large call, for context (from the single-range attribution step)"""
    others = []
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
    return others


TRAFFIC_CALLS = 8


def traffic_child(shape_key):
    """The RP matmul stage of one block shape alone (the same sharp_project_dev calls as rp_stage_alone), as the program
    `rocprofv3 --pmc ...` runs: nothing but the stage's kernels touches the L2 counters."""
    import numpy as np
    import torch

    import sharp_amd
    from sharp_amd import device as dev

    torch.cuda.set_device(0)
    sharp_amd.init(0)
    lib = sharp_amd.lib()
    n, m, K, p = SHAPES[shape_key]
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


def measure_traffic(shape_key):
    """roofline.traffic measured in THIS run: HBM bytes of the RP stage's kernel per launch (= per block) from the L2's
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
                r = subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", td, "--", os.path.realpath(sys.executable),
                                    os.path.abspath(__file__), "--traffic-child", shape_key], cwd="/tmp", env=env, timeout=240,
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
        return traffic, ("measured in this run: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, one child process each running the stage alone on one %s "
                         "block %d times; FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B); read %.2f GB + write %.2f GB per launch"
                         % (shape_key, TRAFFIC_CALLS, 2.0 * tot["FETCH_SIZE"] / 1e9, tot["WRITE_SIZE"] / 1e9))
    except Exception:
        return None, None


def timed_calls(Bn, fn, calls, warm=1):
    for _ in range(warm):
        fn()
    Bn.lib.sharp_synchronize()
    t0 = time.perf_counter()
    for _ in range(calls):
        r = fn()
    Bn.lib.sharp_synchronize()
    return (time.perf_counter() - t0) / calls, r


def extra_configs(Bn, headline_tag):
    """After the timed region of the default run: the other BASELINE.json configurations a single GPU holds, each under its own warm-up + timed
    calls, and the RP matmul stage at every block shape and on other kinds of values, so that the driver's box produces these numbers too."""
    np, torch, dev, lib, sa, ARI = Bn.np, Bn.torch, Bn.dev, Bn.lib, Bn.sa, Bn.ARI
    out, by_cfg = {}, {}

    # ---- cfg2 (configs[1]) and the CH-decided data set of the same shape: SHARP(), 50 000 x 20 000, K = 15
    def cfg2_like(nmark, calls):
        n, m, K = CFG2["cells"], CFG2["genes"], CFG2["K"]
        x = Bn.synth_block(0, n, m, nmark)
        truth = Bn.labels(0, n)
        dev.profile(True)
        dt, (pred, info) = timed_calls(Bn, lambda: dev.SHARP_dev(x, ensize_K=K, rN_seed=RN_SEED), calls)
        prof = dev.profile_table()
        dev.profile(False)
        keep = {}

        def view_call():
            r = dev.SHARP_dev(x, ensize_K=K, rN_seed=RN_SEED, forview=True, view_out=keep.get("b"))
            keep["b"] = r[1]["view_out"]                   # (result buffers kept between calls: first-touch page faults of fresh arrays are host noise)
            return r
        dtv, _ = timed_calls(Bn, view_call, 3)
        r = {"workload": "SHARP() on synthetic %d cells x %d genes, %d marker genes per planted cluster, ensize.K=%d, SHARP_large (BASELINE.json configs[1]%s)"
                         % (n, m, nmark, K, "" if nmark == N_MARK else " shape; the CH-decided data set of SURVEY.md 8d"),
             "value": round(n / dt, 1), "unit": "cells/s", "ms_per_step": round(dt * 1e3, 2), "ms_per_step_forview": round(dtv * 1e3, 2), "steps": calls, "warmup": 1,
             "reduced_dim": info["reduced.dim"], "clusters_found": int(info["N.pred_cluster"]),
             "ari_vs_planted_truth": round(float(ARI(truth, pred)["HA"]), 4),
             "level_rule_per_step": level_rules(prof, calls + 1),
             "rp_stage_in_step": rp_stage_numbers(prof, n, m, K, info["reduced.dim"])}
        return r, x

    if headline_tag != "cfg2":
        def run_cfg2():
            r, x = cfg2_like(N_MARK, 10)
            by_cfg["cfg2_alone"] = Bn.rp_stage_alone(x, CFG2["K"], SHAPES["cfg2"][3])
            return r
        guarded(out, "cfg2", run_cfg2)
        torch.cuda.empty_cache()
    if headline_tag != "cfg2_ch":
        guarded(out, "cfg2_ch", lambda: cfg2_like(N_MARK_CH, 5)[0])
        torch.cuda.empty_cache()

    # ---- cfg3 when it is not the headline
    if headline_tag != "cfg3":
        def run_cfg3():
            m, K, B = CFG3["genes"], CFG3["K"], CFG3["blocks"]
            nb = CFG3["cells"] // B
            blocks = [Bn.synth_block(b * nb, nb, m) for b in range(B)]
            truth = np.concatenate([Bn.labels(b * nb, nb) for b in range(B)])
            dt, (pred, npred, p, _) = timed_calls(Bn, lambda: Bn.unlimited_call(blocks, K), 3)
            return {"workload": "SHARP_unlimited on synthetic %d cells x %d genes as %d blocks, ensize.K=%d (BASELINE.json configs[2])" % (CFG3["cells"], m, B, K),
                    "value": round(CFG3["cells"] / dt, 1), "unit": "cells/s", "ms_per_step": round(dt * 1e3, 2), "steps": 3, "warmup": 1,
                    "reduced_dim": p, "clusters_found": int(npred), "ari_vs_planted_truth": round(float(ARI(truth, pred)["HA"]), 4)}
        guarded(out, "cfg3", run_cfg3)
        torch.cuda.empty_cache()

    # ---- the RP stage alone on one cfg3 block: counts (the table path), and the other kinds of values the reference meets (R/SHARP.R:110-114:
    #      exp.type = "count" is CPM-normalised to doubles; the README example is TPM): fp64 blocks and the general path of the kernel
    def rp_value_kinds():
        n, m, K, p = SHAPES["cfg3"]
        x = Bn.synth_block(0, n, m)
        by_cfg["cfg3_block"] = Bn.rp_stage_alone(x, K, p)
        by_cfg["cfg2_shape_on_counts"] = by_cfg.get("cfg2_alone") or Bn.rp_stage_alone(x, CFG2["K"], SHAPES["cfg2"][3])
        # (ii) fp32 counts with a UMI-like heavy tail: 0.1 % of the non-zeros raised to 256 .. 4095 (outside the 1024-entry term table)
        xt = x.clone()
        nzmask = xt != 0
        g = torch.Generator(device="cuda"); g.manual_seed(7)
        hit = (torch.rand(xt.shape, device="cuda", generator=g) < 1e-3) & nzmask
        xt[hit] = torch.randint(256, 4096, (int(hit.sum().item()),), device="cuda", generator=g).float()
        r = Bn.rp_stage_alone(xt, K, p)
        r["values"] = "counts, 0.1 % of the non-zeros in 256 .. 4095 (not table values)"
        by_cfg["cfg3_block_heavy_tail"] = r
        del xt, hit
        # (iii) CPM-normalised counts (what SHARP() makes of exp.type = "count"): doubles that do not survive fp32 -> an fp64 block
        xd = x.double()
        xd = xd / xd.sum(1, keepdim=True).clamp_min(1.0) * 1e6
        r = Bn.rp_stage_alone(xd, K, p)
        r["values"] = "CPM-normalised counts as doubles (fp64 block, every non-zero through the general path)"
        by_cfg["cfg3_block_f64"] = r
        r = Bn.rp_stage_alone(xd, CFG2["K"], SHAPES["cfg2"][3])
        r["values"] = "the same fp64 block at the cfg2 shape (K = 15, p = 391)"
        by_cfg["cfg2_shape_f64"] = r
        # (i) TPM-like: the same with per-gene length factors (another set of doubles, same sparsity)
        gl = 0.5 + torch.rand((1, m), device="cuda", generator=g, dtype=torch.float64) * 4.0
        xd = x.double() / gl
        xd = xd / xd.sum(1, keepdim=True).clamp_min(1e-300) * 1e6
        r = Bn.rp_stage_alone(xd, K, p)
        r["values"] = "TPM-like doubles (counts / gene length, scaled to 1e6 per cell; fp64 block)"
        by_cfg["cfg3_block_tpm"] = r
        return True
    g0 = {}
    guarded(g0, "rp_value_kinds", rp_value_kinds)
    if isinstance(g0["rp_value_kinds"], dict):
        by_cfg["rp_value_kinds_error"] = g0["rp_value_kinds"]
    torch.cuda.empty_cache()

    # ---- cfg4's per-GPU share at N = 8: one 162 500 x 27 000 block, K = 5, p = 508: the RP stage alone, the block step, then cfg4 whole on ONE GPU
    def run_cfg4():
        from sharp_amd import dist as sdist

        m, K, B = CFG4["genes"], CFG4["K"], CFG4["blocks"]
        nb, p = CFG4["cells"] // B, 508
        x = Bn.synth_block(0, nb, m)
        proj = sa.Projector(m, p, [50 + RN_SEED + k for k in range(1, K + 1)])
        dt, (pr, mn, cn) = timed_calls(Bn, lambda: dev.unlimited_block_dev(x, p, proj.handle, K, RN_SEED), 3)
        proj.close()
        out["cfg4_share"] = {"workload": "one GPU's block of BASELINE.json configs[3] at N = 8: %d cells x %d genes, ensize.K=%d, p=%d, "
                                         "sharp_unlimited_block_dev (projectors resident)" % (nb, m, K, p),
                             "value": round(nb / dt, 1), "unit": "cells/s", "ms_per_step": round(dt * 1e3, 2), "steps": 3, "warmup": 1,
                             "clusters_found": int(mn.shape[0])}
        by_cfg["cfg4_share"] = Bn.rp_stage_alone(x, K, p)
        # the same eight 162 500-cell blocks the N > 1 runs deal out, one after the other here (140 GB of X resident): the N = 1 point of the
        # strong-scaling curve (`--gpus 1 --config cfg4` times it as the headline value)
        blocks = [x] + [Bn.synth_block(b * nb, nb, m) for b in range(1, B)]
        truth = np.concatenate([Bn.labels(b * nb, nb) for b in range(B)])
        torch.cuda.synchronize()

        def cfg4_step():
            pg = sdist.global_reduced_dim(nb * B)
            pj = sa.Projector(m, pg, [50 + RN_SEED + k for k in range(1, K + 1)])
            res, nfin, _ = sdist.unlimited_sharded(blocks, list(range(B)), [nb] * B,
                                                   lambda blk, p_, nxt: dev.unlimited_block_dev(blk, p_, pj.handle, K, RN_SEED, next_block=nxt),
                                                   dev.unlimited_merge, device="cuda")
            pj.close()
            return np.concatenate([res[b] for b in range(B)]), nfin, pg
        dt, (pred, nfin, pg) = timed_calls(Bn, cfg4_step, 3)
        return {"workload": "SHARP_unlimited on synthetic %d cells x %d genes as %d blocks of %d cells, block b on GPU b mod 1, ensize.K=%d, "
                            "rN.seed=%d (BASELINE.json configs[3] on one GPU: the N = 1 point of the curve `--gpus N` draws)" % (nb * B, m, B, nb, K, RN_SEED),
                "value": round(nb * B / dt, 1), "unit": "cells/s", "ms_per_step": round(dt * 1e3, 2), "steps": 3, "warmup": 1, "scaling": "strong",
                "reduced_dim": pg, "clusters_found": int(nfin), "ari_vs_planted_truth": round(float(ARI(truth, pred)["HA"]), 4)}
    guarded(out, "cfg4_one_gpu", run_cfg4)
    torch.cuda.empty_cache()

    # ---- cfg5 as the reference's documented workflow runs it (R/SHARP_unlimited3.R: a directory of partitions): block FILES streamed into HBM
    guarded(out, "cfg5_streamed", lambda: cfg5_streamed(Bn))
    torch.cuda.empty_cache()

    # ---- host-inclusive: the blocks start on the HOST as an R session holds them (dense fp64 matrices / dgCMatrix-like sparse blocks); never `value`
    guarded(out, "host_inclusive", lambda: host_inclusive(Bn))
    torch.cuda.empty_cache()
    return out, by_cfg


def write_packed_block_from_device(path, x, B):
    """a packed (version 2) block file straight from a resident block of counts below 256: column pointers, 16-bit row indices, 8-bit values"""
    import numpy as np
    import torch

    n, m = x.shape
    nz = x.nonzero()                                                  # (cell, gene), sorted by cell then gene
    cp = np.concatenate([[0], np.cumsum(torch.bincount(nz[:, 0], minlength=n).cpu().numpy())]).astype(np.int64)
    idx = nz[:, 1].to(torch.int32).cpu().numpy().astype(np.uint16)
    vi = x[nz[:, 0], nz[:, 1]].to(torch.int32).cpu().numpy()
    vb = 8 if int(vi.max(initial=0)) <= 255 else 16
    val = vi.astype(np.uint8 if vb == 8 else np.uint16)
    nnz = int(idx.size)
    o_idx, o_val, total = B._packed_layout(n, nnz, 16, vb)
    with open(path, "wb") as fh:
        fh.write(B._HDR2.pack(B.MAGIC, 2, 1, m, n, nnz, 16, vb))
        fh.write(cp.tobytes()); fh.write(b"\0" * (o_idx - cp.nbytes))
        fh.write(idx.tobytes()); fh.write(b"\0" * (o_val - o_idx - idx.nbytes))
        fh.write(val.tobytes()); fh.write(b"\0" * (total - o_val - val.nbytes))
    return total


def cfg5_streamed(Bn, nfiles=20, ndense=3):
    """BASELINE.json configs[4] the way the reference's README runs its 1.3 M-cell example (R/SHARP_unlimited3.R:59-62,103-105: a DIRECTORY of
    partitions): `nfiles` of cfg5's 200 blocks of 50 000 cells x 20 000 genes as block files (the packed format: 4 bytes per non-zero),
    p = 582 as for the whole 10 M cells, read -> pinned memory -> HBM through a ring of buffers while earlier blocks are clustered; the
    blocks that have arrived go together as one pipelined batch.  Reports blocks/s, the file bytes and how much of the reading was hidden under
    the clustering; and the same through dense float32 files (4 GB each: `ndense` of them) for comparison.  Never `value`."""
    import shutil
    import tempfile

    from sharp_amd import blocks as B

    np, torch, sa, dev = Bn.np, Bn.torch, Bn.sa, Bn.dev
    nb, m, K = 50000, CFG3["genes"], 5
    root = None
    for cand in ("/dev/shm", tempfile.gettempdir()):
        try:
            st = os.statvfs(cand)
            if st.f_bavail * st.f_frsize > (nfiles * 0.6 + ndense * 4.2) * 1e9:
                root = cand
                break
        except OSError:
            pass
    if root is None:
        return {"skipped": "no directory with room for the block files"}
    # (the reference orders the partitions by the first number in their FULL path, R/SHARP_unlimited3.R:59-61: no digit in the directory's name)
    import random
    import string
    d = os.path.join(root, "sharpblk_" + "".join(random.choice(string.ascii_lowercase) for _ in range(12)))
    os.mkdir(d)
    out = {"workload": "SHARP_unlimited3 on %d block files of %d cells x %d genes in %s (cfg5 = BASELINE.json configs[4]: 200 such blocks; p = 582 from the 10 M "
                       "cells of the whole), ensize.K=%d, viewflag=FALSE; file reads and PCIe inside the time, never `value`" % (nfiles, nb, m, root, K)}
    try:
        x = torch.empty((nb, m), dtype=torch.float32, device="cuda")
        truth = []
        total = 0
        for b in range(nfiles):
            dev.synth_fill(x, DATA_SEED, b * nb, G_TRUE, N_MARK)
            total += write_packed_block_from_device(os.path.join(d, "part_%d.blk" % (b + 1)), x, B)
            truth.append(Bn.labels(b * nb, nb))
        nd = {"dir": d, "ncells": 10000000, "ngenes": m}
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            res = sa.SHARP_unlimited3(nd, rN_seed=RN_SEED, viewflag=False)
            ts.append(time.perf_counter() - t0)
        t = min(ts[1:])
        t1s = []
        for _ in range(2):
            t0 = time.perf_counter()
            res1 = sa.SHARP_unlimited3(nd, rN_seed=RN_SEED, viewflag=False, group=1)
            t1s.append(time.perf_counter() - t0)
        out["packed_files_block_after_block"] = {"seconds_per_call": round(min(t1s), 3), "blocks_per_s": round(nfiles / min(t1s), 1),
                                                 "labels_equal_grouped_run": bool(np.array_equal(res1["pred_clusters"], res["pred_clusters"]))}
        out["packed_files"] = {"files": nfiles, "file_gb": round(total / 1e9, 2), "seconds_per_call": round(t, 3), "calls_s": [round(v, 3) for v in ts], "steps": 2, "warmup": 1,
                               "blocks_per_s": round(nfiles / t, 1), "cells_per_s": round(nfiles * nb / t, 1), "file_gbps_of_the_call": round(total / 1e9 / t, 2),
                               "read_seconds": round(res["read_seconds"], 3), "consumer_wait_seconds": round(res["wait_seconds"], 3),
                               "expand_seconds": round(res["expand_seconds"], 3), "clustering_calls_seconds": round(res["cluster_seconds"], 3),
                               "read_hidden_fraction": round(1.0 - res["wait_seconds"] / max(res["read_seconds"], 1e-9), 3),
                               "clusters_found": int(res["N.pred_clusters"]),
                               "ari_vs_planted_truth": round(float(Bn.ARI(np.concatenate(truth), res["pred_clusters"])["HA"]), 4)}
        for b in range(nfiles):
            os.remove(os.path.join(d, "part_%d.blk" % (b + 1)))
        hdr = B._HDR.pack(B.MAGIC, 1, 0, m, nb, m)
        for b in range(ndense):
            dev.synth_fill(x, DATA_SEED, b * nb, G_TRUE, N_MARK)
            with open(os.path.join(d, "part_%d.blk" % (b + 1)), "wb") as fh:
                fh.write(hdr)
                x.cpu().numpy().tofile(fh)
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            res = sa.SHARP_unlimited3(nd, rN_seed=RN_SEED, viewflag=False)
            ts.append(time.perf_counter() - t0)
        t = ts[-1]
        out["dense_files"] = {"files": ndense, "file_gb": round(ndense * nb * m * 4 / 1e9, 2), "seconds_per_call": round(t, 3), "blocks_per_s": round(ndense / t, 2),
                              "cells_per_s": round(ndense * nb / t, 1), "file_gbps_of_the_call": round(ndense * nb * m * 4 / 1e9 / t, 2),
                              "read_seconds": round(res["read_seconds"], 3), "consumer_wait_seconds": round(res["wait_seconds"], 3)}
        del x
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def pcie_h2d_gbps(torch, nbytes=1 << 30):
    """pinned host -> device copy rate of this box, measured (what bounds a host-inclusive call from below)"""
    h = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    d = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
    d.copy_(h, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        d.copy_(h, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    return 3 * nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9


def host_inclusive(Bn, B=CFG3["blocks"]):
    """SHARP_unlimited on a LIST OF HOST BLOCKS, cfg3 whole (R/SHARP_unlimited.R:125-143: the data starts in R's memory): the ten blocks as
    dgCMatrix-like CSC blocks and as dense fp64 matrices (as many of the ten as the host's free memory holds twice over: 8 GB each), under a
    warm-up + timed-calls contract; PCIe inside the time, never `value`.  The blocks cross PCIe in the narrowest exact type (counts: 16-bit
    values, 16-bit row indices) into a ring of resident copies while earlier blocks are clustered; those that arrived meanwhile go together as
    one pipelined batch.  tools/bench_host_blocks.py is the longer form (timelines, block after block for comparison)."""
    np, torch, sa = Bn.np, Bn.torch, Bn.sa
    import scipy.sparse as sps

    nb, m, K, _ = SHAPES["cfg3"]
    avail = 0
    for ln in open("/proc/meminfo"):
        if ln.startswith("MemAvailable"):
            avail = int(ln.split()[1]) * 1024
    BD = B
    while BD > 2 and (BD * nb * m * 8 + B * nb * m * 0.13 * 12) * 2 > avail:
        BD -= 1
    dense, sparse = [], []
    for b in range(B):
        x = Bn.synth_block(b * nb, nb, m)
        nz = x.nonzero()                                              # (cell, gene), sorted by cell then gene = CSC order of genes x cells
        indptr = np.concatenate([[0], np.cumsum(torch.bincount(nz[:, 0], minlength=nb).cpu().numpy())]).astype(np.int32)
        sparse.append(sps.csc_matrix((x[nz[:, 0], nz[:, 1]].double().cpu().numpy(), nz[:, 1].int().cpu().numpy(), indptr), shape=(m, nb)))
        if b < BD:
            dense.append(x.double().cpu().numpy().T)                  # (genes, cells) column-major view: R's layout, no copy
        del x, nz
    torch.cuda.empty_cache()
    bw = pcie_h2d_gbps(torch)
    nnz = sum(int(s_.nnz) for s_ in sparse)
    out = {"workload": "SHARP_unlimited on host blocks of %d cells x %d genes (BASELINE.json configs[2] from the caller's memory), ensize.K=%d, viewflag=FALSE; "
                       "PCIe-inclusive, never `value`" % (nb, m, K),
           "dense_host_gb": round(sum(d.nbytes for d in dense) / 1e9, 2), "sparse_host_gb": round(sum(s_.data.nbytes + s_.indices.nbytes + s_.indptr.nbytes for s_ in sparse) / 1e9, 3),
           "pcie_h2d_pinned_gbps_measured": round(bw, 1), "host_memory_available_gb": round(avail / 1e9, 1)}
    ref = None
    for name, blocks in (("sparse_csc", sparse), ("dense_fp64", dense)):
        nblk = len(blocks)
        dt, res = timed_calls(Bn, lambda: sa.SHARP_unlimited(blocks, ensize_K=K, rN_seed=RN_SEED, viewflag=False), 3)
        wire = Bn.lib.sharp_x_wire()
        sent = nnz * (wire // 8 + 2) if blocks is sparse else nblk * nb * m * (wire // 8)
        if nblk == B:
            ref = res["pred_clusters"] if ref is None else ref
        out[name] = {"blocks": nblk, "cells": nblk * nb, "seconds_per_call": round(dt, 4), "cells_per_s": round(nblk * nb / dt, 1), "steps": 3, "warmup": 1,
                     "wire_value_bits": wire, "bytes_over_pcie": int(sent), "pcie_gbps_of_the_call": round(sent / dt / 1e9, 1),
                     "pcie_bound_s": round(sent / (bw * 1e9), 4),
                     "labels_equal_sparse_run": bool(np.array_equal(ref, res["pred_clusters"])) if nblk == B and ref is not None else None}
    out["note"] = ("pcie_bound_s = bytes_over_pcie / the measured pinned copy rate: what the call could not beat on this box; the dense path also "
                   "reads 8 B per value from pageable host memory, which bounds it before PCIe does")
    return out


def cpu_baseline(Bn, tag):
    """The oracle (CPU port of the reference path) on a bounded sample of the same workload, on the host cores of this box, with its
    per-stage seconds; plus the GPU-vs-oracle label agreement (ARI, Hubert-Arabie) on the sample of EVERY configuration."""
    from oracle import pyoracle as orc

    np, dev, ARI = Bn.np, Bn.dev, Bn.ARI
    orc.build()
    present = os.cpu_count() or 1
    try:
        avail = len(os.sched_getaffinity(0))           # the cores this process may run on (a GPU box hands out a share of the host)
    except AttributeError:
        avail = present
    # The oracle parallelises over the K*T task grid of a block and every thread holds a copy of its fold: it is memory-bound, and more
    # threads are not always faster (round 3: 672 cells/s on 120 threads, 947 on 30).
    threads = max(1, min(avail, 32))

    sa = Bn.sa
    margins = {}

    def decisions(tag_, got, want):
        """the two decision logs of a sample: do they agree decision for decision, and how close did any decision come"""
        exact = [0, 1, 2, 3, 4, 5, 6, 7, 12, 13]        # call, rule, chosen k, exact ties, override, levels
        agree = got.shape == want.shape and bool((got[:, exact] == want[:, exact]).all())
        margins[tag_] = {"decisions": int(len(got)), "logs_agree_decision_for_decision": agree, "per_level": sa.decision_margins(got)}

    def unlimited_sample(m, K, nblk, ncell, nmark=N_MARK, tag_=None):
        xs = [Bn.synth_block(b * 50000, ncell, m, nmark) for b in range(nblk)]     # the first cells of the configuration's first blocks
        hs = [x.cpu().numpy().T.astype(np.float64) for x in xs]
        orc.stage_seconds()
        orc.decision_log(True)
        t0 = time.perf_counter()
        ref = orc.SHARP_unlimited(hs, K=K, rN_seed=RN_SEED, nthreads=threads)
        t = time.perf_counter() - t0
        want = orc.last_decisions()
        orc.decision_log(False)
        stages = orc.stage_seconds()
        sa.decision_log(True)
        pred, npred, p, _ = Bn.unlimited_call(xs, K)
        got = sa.last_decisions()
        sa.decision_log(False)
        decisions(tag_, got, want)
        return t, stages, float(ARI(ref["pred_clusters"], pred)["HA"]), bool(np.array_equal(ref["pred_clusters"], pred)), p

    def large_sample(m, K, ncell, nmark, tag_=None):
        x = Bn.synth_block(0, ncell, m, nmark)
        h = x.cpu().numpy().T.astype(np.float64)
        orc.stage_seconds()
        orc.decision_log(True)
        t0 = time.perf_counter()
        ref = orc.SHARP(h, K=K, base_ncells=1, rN_seed=RN_SEED, nthreads=threads, want_view=False)
        t = time.perf_counter() - t0
        want = orc.last_decisions()
        orc.decision_log(False)
        stages = orc.stage_seconds()
        sa.decision_log(True)
        pred, _ = dev.SHARP_dev(x, ensize_K=K, base_ncells=1, rN_seed=RN_SEED)
        got = sa.last_decisions()
        sa.decision_log(False)
        decisions(tag_, got, want)
        return t, stages, float(ARI(ref["pred_clusters"], pred)["HA"]), bool(np.array_equal(ref["pred_clusters"], pred))

    parity = {}
    # cfg3: two blocks of 16 000 cells (8 folds x 5 RPs = 40 tasks per block): 10-15 s of oracle time
    t3, st3, ari3, eq3, p3 = unlimited_sample(CFG3["genes"], CFG3["K"], 2, 16000, tag_="cfg3")
    parity["cfg3"] = {"ari_gpu_vs_oracle_on_sample": round(ari3, 4), "labels_identical": eq3, "sample": "2 blocks x 16000 cells x 20000 genes, K = 5"}
    # cfg2: 8000 cells, K = 15 (60 tasks); and the same on the CH-decided data set
    t2, st2, ari2, eq2 = large_sample(CFG2["genes"], CFG2["K"], 8000, N_MARK, tag_="cfg2")
    parity["cfg2"] = {"ari_gpu_vs_oracle_on_sample": round(ari2, 4), "labels_identical": eq2, "sample": "8000 cells x 20000 genes, K = 15"}
    t2c, st2c, ari2c, eq2c = large_sample(CFG2["genes"], CFG2["K"], 8000, N_MARK_CH, tag_="cfg2_ch")
    parity["cfg2_ch"] = {"ari_gpu_vs_oracle_on_sample": round(ari2c, 4), "labels_identical": eq2c, "sample": "8000 cells x 20000 genes, K = 15, %d marker genes" % N_MARK_CH}
    # cfg4: two blocks of 6000 cells x 27000 genes
    t4, st4, ari4, eq4, _ = unlimited_sample(CFG4["genes"], CFG4["K"], 2, 6000, tag_="cfg4")
    parity["cfg4"] = {"ari_gpu_vs_oracle_on_sample": round(ari4, 4), "labels_identical": eq4, "sample": "2 blocks x 6000 cells x 27000 genes, K = 5"}
    for k_, v_ in margins.items():
        # how close the sample's decisions came to going the other way, per level (sharp_last_decisions): for a decision
        # by the median silhouette with ONE maximum, best - runner-up; exact ties are counted (the reference picks the middle one by `==`);
        # CH decisions: the relative margin; every decision: its distance to the sil.thre switch.  profiles/r06_margins.txt: full-size runs.
        parity[k_]["min_margin"] = v_
    parity["ari_gpu_vs_oracle_on_sample"] = parity[{"cfg2": "cfg2", "cfg2_ch": "cfg2_ch", "cfg3": "cfg3", "cfg4": "cfg4"}[tag]]["ari_gpu_vs_oracle_on_sample"]
    parity["full_size"] = ("tests/test_configs_gpu.py: ::test_cfg2_full_size_matches_oracle (50 000 x 20 000, K = 15, 375 base tasks: labels AND the two decision logs), "
                           "::test_block_of_1e5_cells_no_reshuffle_branch_matches_oracle (cfg4's n >= 1e5 branch): labels identical to the oracle's; "
                           "tools/parity_fullsize.py -> profiles/r06_*_parity.txt: a cfg3 block, one true cfg4 share and the CH-decided data set at full size, with margins")
    per = {"cfg3": (32000, t3, st3), "cfg2": (8000, t2, st2), "cfg2_ch": (8000, t2c, st2c), "cfg4": (12000, t4, st4)}
    ns, t, st = per[tag]
    base = {"value": round(ns / t, 2), "unit": "cells/s", "cores": threads, "cores_available": avail, "cores_present": present, "kind": "port",
            "sample": "the oracle (CPU restatement of the reference's R path, not R) on a sample of this workload: %s; OpenMP over the K*T task grid of a block, "
                      "%d threads of the %d cores this process may use; %.1f s" % (parity[tag]["sample"], threads, avail, t),
            "stage_seconds": st,
            "stage_seconds_note": "rp_matmul / base_clustering are THREAD-seconds summed over the task grid's threads (task_loop_wall is their wall time); the others wall seconds",
            "by_config": {k: {"value": round(v[0] / v[1], 2), "unit": "cells/s", "seconds": round(v[1], 2), "sample_cells": v[0], "stage_seconds": v[2]} for k, v in per.items()}}
    return base, parity


if __name__ == "__main__":
    main()
