#!/bin/bash
set -u
O=gpurun_out/${1:-r4ev2}; mkdir -p $O
for rep in 1 2 3; do
for spec in "x_base.so --steps 20 --warmup 5" "x_base.so --steps 20 --warmup 5 --no-events" "libphdhip.so --steps 20 --warmup 5" "libphdhip.so --steps 20 --warmup 5 --events-every 20" "libphdhip.so --steps 200 --warmup 5" "libphdhip.so --steps 200 --warmup 5 --no-events"; do
  set -- $spec; so=$1; shift
  r=$(PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra "$@" 2>$O/err.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],4), {k: round(v*1e3,1) for k,v in d.get('kernel_ms',{}).items()})")
  echo "$spec: $r" | tee -a $O/ev.log
done
done
