#!/bin/bash
set -u
O=gpurun_out/${1:-r4h}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not multiproc and not perfect and not murty" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
for d in 1 0; do
  PHD_DEFER_BIG=$d timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_defer$d.json 2> $O/bench_defer$d.err; echo "bench defer=$d rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_defer$d.json"))
print("defer=$d ms/step", d["ms_per_step"], "one stream", d.get("ms_per_step_one_stream"))
print("iso", {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
for k, v in d.get("other_modes", {}).items():
    if k in ("weights_steady", "config_A", "config_S"):
        print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "error")}, {a: round(b * 1e3, 1) for a, b in v.get("kernel_ms_isolated", {}).items()})
PY
done
