#!/bin/bash
# quick GPU check of a kernel change: the tests that pin it, the prune's phase stamps, the default bench line
#   usage: scripts/r4_quick.sh <tag> "<pytest -k expression>" [stamp kernel ids...]
set -u
TAG=${1:-r4q}; KEXPR=${2:-prune}; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for k in "$@"; do timeout -k 10 300 python scripts/stamps.py survey $k 2>/dev/null | tail -2 | sed "s/^/kernel $k: /" | tee -a $O/stamps.log; done
timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("ms/step", d["ms_per_step"], "one stream", d.get("ms_per_step_one_stream"))
print("iso", {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
for k, v in d.get("other_modes", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error")}, {a: round(b * 1e3, 1) for a, b in v.get("kernel_ms_isolated", {}).items()})
PY
