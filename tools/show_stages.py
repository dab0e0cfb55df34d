import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
top = int(sys.argv[2]) if len(sys.argv) > 2 else 6
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d["kernel_ms_per_step"]
print(tag, d["config"]["cells_per_gpu"], "ms/step", d["ms_per_step"], {x: k[x] for x in list(k)[:top]})
