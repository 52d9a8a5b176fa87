#!/bin/bash
# A/B of an environment switch of the library on the default bench line:  scripts/r4_ab.sh <tag> <VAR> <value A> <value B> [pytest -k expr]
set -u
TAG=$1; VAR=$2; A=$3; B=$4; KEXPR=${5:-}
O=gpurun_out/$TAG; mkdir -p $O
if [ -n "$KEXPR" ]; then
  env $VAR=$B timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1; echo "pytest ($VAR=$B) rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
fi
for v in $A $B $A $B; do
  env $VAR=$v timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err; echo "bench $VAR=$v rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_$v.json"))
print("$VAR=$v ms/step", round(d["ms_per_step"], 4), "one stream", round(d.get("ms_per_step_one_stream", 0), 4), "iso", {k: round(x * 1e3, 1) for k, x in d.get("kernel_ms_isolated", {}).items()})
m = d.get("other_modes", {})
print("   steady", round(m.get("weights_steady", {}).get("ms_per_step", 0), 4), "A", round(m.get("config_A", {}).get("ms_per_step", 0), 4), "S", round(m.get("config_S", {}).get("ms_per_step", 0), 3))
PY
done
