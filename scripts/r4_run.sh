#!/bin/bash
# round-4 GPU session: (optionally) the new tests against round 3's library, the GPU suite, the default bench line
#   usage: scripts/r4_run.sh <tag> [r3check] [notests] [pytest -k expression]
set -u
TAG=${1:-r4}; shift
O=gpurun_out/$TAG; mkdir -p $O
R3=0; TESTS=1; KEXPR=
for a in "$@"; do
  case $a in r3check) R3=1;; notests) TESTS=0;; *) KEXPR=$a;; esac
done
if [ $R3 = 1 ] && [ -f monorfs_amd/csrc/libphdhip_r3.so ]; then
  PHDHIP_SO=$PWD/monorfs_amd/csrc/libphdhip_r3.so timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -m gpu -q -k "prune or plan" > $O/r3check.log 2>&1
  echo "r3check rc=$?" >> $O/r3check.log; tail -15 $O/r3check.log
fi
if [ $TESTS = 1 ]; then
  if [ -n "$KEXPR" ]; then
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1
  else
    timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 > $O/tests.log 2>&1
  fi
  echo "pytest rc=$?" >> $O/tests.log; tail -25 $O/tests.log
  grep -q "pytest rc=0" $O/tests.log || exit 1
fi
timeout -k 10 400 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print("ms/step", d["ms_per_step"], "value", d["value"])
print("iso", d.get("kernel_ms_isolated"))
for k, v in d.get("other_modes", {}).items():
    print(k, {a: b for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error", "kernel_ms_isolated")})
PY
