#!/bin/bash
# A/B of library builds on config S (and B, to see what the change costs there):  scripts/r4_abS.sh <tag> <so> ...
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
for so in "$@"; do
  for cfg in S B; do
    r=$(PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra --steps 60 --warmup 10 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
    echo "$so $cfg $r" | tee -a $O/ab.log
  done
done
done
