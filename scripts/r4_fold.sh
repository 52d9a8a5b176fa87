#!/bin/bash
set -u
O=gpurun_out/${1:-r4fold}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "chain or kat or round2 or round4 and not device_path and not perfect" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
for rep in 1 2 3; do for fold in 1 0; do for w in steady survey; do
  r=$(PHD_FOLD_NR=$fold timeout -k 10 200 python bench.py --config A --no-cpu-baseline --no-extra --steps 400 --warmup 20 --weights $w 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "A fold=$fold $w $r" | tee -a $O/ab.log
done; done; done
