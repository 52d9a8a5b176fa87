#!/bin/bash
O=gpurun_out/${1:-r4pre}; mkdir -p $O
for rep in 1 2 3; do
for spec in "20 5 20" "60 5 20" "150 5 20" "400 5 20" "20 5 200" "400 5 200"; do
  set -- $spec
  r=$(PHD_BENCH_PREROLL=$1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --warmup $2 --steps $3 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "preroll $1 warmup $2 steps $3: $r" | tee -a $O/pre.log
done
done
