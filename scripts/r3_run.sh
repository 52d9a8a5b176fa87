#!/bin/bash
# round-3 GPU session: GPU tests, then the rehearsals of both multi-shard hosts on the one GPU
#   usage: scripts/r3_run.sh <tag> [tests|notests] [pytest -k expression]
set -u
TAG=${1:-r3}; MODE=${2:-tests}; KEXPR=${3:-}
O=gpurun_out/$TAG; mkdir -p $O
python -c "import bench; print('gpus counted without HIP:', bench.count_gpus_without_hip(), 'host threads:', bench.host_threads())" > $O/count.log 2>&1; cat $O/count.log
if [ "$MODE" = tests ]; then
  if [ -n "$KEXPR" ]; then
    timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1
  else
    timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
  fi
  echo "pytest rc=$?" >> $O/tests.log; tail -5 $O/tests.log
  grep -q "pytest rc=0" $O/tests.log || exit 1
fi
for cfg in A B512; do
  timeout -k 10 300 python bench.py --single-process --gpus 8 --devices 0,0,0,0,0,0,0,0 --config $cfg --weights steady --steps 40 --warmup 3 > $O/multi8_$cfg.json 2> $O/multi8_$cfg.err; echo "multi8 $cfg rc=$?"
done
timeout -k 10 300 python bench.py --single-process --gpus 2 --devices 0,0 --config B1024 --weights steady --steps 40 --warmup 3 > $O/multi2_B1024.json 2> $O/multi2_B1024.err; echo "multi2 rc=$?"
timeout -k 10 200 python bench.py --force-dist --weights steady --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist.json 2> $O/forcedist.err; echo "forcedist rc=$?"
timeout -k 10 200 python scripts/dist_phases.py > $O/dist_phases.log 2>&1; echo "phases rc=$?"
timeout -k 10 200 python scripts/global_phase.py 8 steady > $O/global_phase.log 2>&1; echo "global phase rc=$?"; tail -5 $O/global_phase.log
