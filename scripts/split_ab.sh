#!/bin/bash
# Tuning A/B (on the GPU box): the number of sub-range streams of a step, per bench config.
#   scripts/split_ab.sh <tag> <cfg,cfg,...> <split,split,...>
set -u
O=gpurun_out/${1:?tag}; mkdir -p $O
for rep in 1 2; do for cfg in ${2//,/ }; do for sp in ${3//,/ }; do
  PHD_SPLIT=$sp timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline --no-extra --steps 40 --warmup 5 > $O/out.json 2> $O/err.log || { echo "failed $cfg $sp"; tail -3 $O/err.log; exit 1; }
  python -c "
import json
d = json.load(open('$O/out.json')); print('$cfg PHD_SPLIT=$sp ms/step %.4f' % d['ms_per_step'])" | tee -a $O/split.log
done; done; done
