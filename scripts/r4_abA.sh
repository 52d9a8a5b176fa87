#!/bin/bash
# A/B of library builds on config A (steady) and B (survey):  scripts/r4_abA.sh <tag> <so> ...
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
for so in "$@"; do
  rA=$(PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 300 python bench.py --config A --weights steady --no-cpu-baseline --no-extra --steps 400 --warmup 20 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  rB=$(PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 300 python bench.py --no-cpu-baseline --no-extra --steps 100 --warmup 20 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
  echo "$so A $rA B $rB" | tee -a $O/ab.log
done
done
