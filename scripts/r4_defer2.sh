#!/bin/bash
set -u
O=gpurun_out/${1:-r4i}; mkdir -p $O
for q in 4 8 16; do
  GPU_MAX_HW_QUEUES=$q PHD_DEFER_BIG=1 timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_q$q.json 2> $O/bench_q$q.err; echo "bench queues=$q rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_q$q.json"))
print("queues=$q ms/step", d["ms_per_step"], "one stream", d.get("ms_per_step_one_stream"))
for k, v in d.get("other_modes", {}).items():
    if k in ("weights_steady", "config_S"):
        print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "error")})
PY
done
