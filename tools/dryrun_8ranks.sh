#!/bin/bash
# ON A ONE-GPU BOX: rehearse `bench.py --gpus N` with N ranks (default 5: the box's process guard allows at most six processes on the card, and the launcher counts as one) --
# gloo instead of RCCL, every rank on GPU 0 (SHARP_BENCH_SHARE_GPU=1) -- at reduced size, and check the labels' summary against the N = 1 run
# of the same reduced problem (same eight blocks, same global p: every block's final labels must be identical, compared by checksum).
# No scaling number comes out of this (the ranks share one GPU); it exercises the rank bookkeeping, the all-gather of the centroid tables
# with N ranks, and the per-rank host-thread cap (SHARP_HOST_THREADS = cores / N, set by bench.py).
# usage: tools/dryrun_8ranks.sh [ranks=6] [cells=96000] [genes=6000]
N=${1:-5}; CELLS=${2:-96000}; GENES=${3:-6000}
REPO=$(cd "$(dirname "$0")/.." && pwd); OUT=$REPO/gpurun_out; mkdir -p $OUT
cd $REPO
export HSA_ENABLE_IPC_MODE_LEGACY=0
PORT=$((20000 + RANDOM % 20000))
timeout -k 10 600 python3 bench.py --gpus 1 --config cfg4 --steps 2 --warmup 1 --cells $CELLS --genes $GENES --no-forview > $OUT/dryrun_n1.json 2> $OUT/dryrun_n1.err || { echo "N=1 run failed"; tail -5 $OUT/dryrun_n1.err; exit 1; }
SHARP_BENCH_SHARE_GPU=1 SHARP_BENCH_BACKEND=gloo timeout -k 10 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $PORT \
    bench.py --gpus $N --steps 2 --warmup 1 --cells $CELLS --genes $GENES --no-forview > $OUT/dryrun_nN.json 2> $OUT/dryrun_nN.err || { echo "N=$N run failed"; tail -5 $OUT/dryrun_nN.err; exit 1; }
python3 - "$OUT/dryrun_n1.json" "$OUT/dryrun_nN.json" "$N" <<'PY'
import json, sys
rd = lambda f: json.loads([l for l in open(f) if l.startswith("{")][-1])
a, b, N = rd(sys.argv[1]), rd(sys.argv[2]), int(sys.argv[3])
print("N=1 : %s cells/s, %d clusters, ARI vs truth %.4f, p=%d" % (a["value"], a["clusters_found"], a["ari_vs_planted_truth"], a["config"]["reduced_dim"]))
print("N=%d : %s cells/s (ranks share ONE GPU: not a scaling number), %d clusters, p=%d, n_gpus=%d" % (N, b["value"], b["clusters_found"], b["config"]["reduced_dim"], b["n_gpus"]))
ok = (a["clusters_found"] == b["clusters_found"] and a["config"]["reduced_dim"] == b["config"]["reduced_dim"] and b["n_gpus"] == N
      and a["labels_crc32_by_block"] == b["labels_crc32_by_block"])
print("labels of every block identical at N=1 and N=%d (crc32 per block, global block order): %s" % (N, ok))
sys.exit(0 if ok else 1)
PY
