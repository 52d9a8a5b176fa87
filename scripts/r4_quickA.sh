#!/bin/bash
set -u
O=gpurun_out/${1:-r4qa}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "not multiproc and not murty" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
export PHD_STAMP_SHAPE=256,128,32
for k in 5 2; do timeout -k 10 200 python scripts/stamps.py steady $k 2>/dev/null | tail -1 | sed "s/^/A kernel $k: /" | tee -a $O/stampsA.log; done
unset PHD_STAMP_SHAPE
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "one stream", d.get("ms_per_step_one_stream"))
print("iso", {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
for k, v in d.get("other_modes", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error")})
PY
