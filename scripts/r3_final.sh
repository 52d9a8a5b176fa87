#!/bin/bash
# round-3 closing run, part 1: every GPU test, the default bench line as the driver runs it, the multi-shard rehearsals
#   usage: scripts/r3_final.sh <tag>
set -u
TAG=${1:-r3fin}; O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
( time timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time || exit 1
echo "bench done"; tail -3 $O/bench.time
for cfg in A B512; do
  timeout -k 10 300 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $cfg --weights steady --steps 40 --warmup 3 > $O/multi8_$cfg.json 2> $O/multi8_$cfg.err || exit 1
done
timeout -k 10 300 python bench.py --single-process --gpus 2 --devices 0,0 --config B1024 --weights steady --steps 40 --warmup 3 > $O/multi2_B1024.json 2> $O/multi2_B1024.err || exit 1
echo "rehearsals done"
