#!/bin/bash
# round-3 baseline on the GPU box: GPU tests, the bench line, the multi-shard rehearsals of the code as round 2 left it
set -u
O=gpurun_out/r3a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
for cfg in A B512; do
  timeout -k 10 200 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $cfg --steps 30 --warmup 3 > $O/multi8_$cfg.json 2> $O/multi8_$cfg.err; echo "multi8 $cfg rc=$?"
done
for cfg in A2048 B4096; do
  timeout -k 10 200 python bench.py --config $cfg --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/single_$cfg.json 2> $O/single_$cfg.err; echo "single $cfg rc=$?"
done
timeout -k 10 200 python bench.py --force-dist --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist.json 2> $O/forcedist.err; echo "forcedist rc=$?"
timeout -k 10 200 python scripts/dist_phases.py > $O/dist_phases.log 2>&1; echo "phases rc=$?"
