#!/bin/bash
# Every path bench.py can take through the sharded step, at ONE rank (on the GPU box): the host-plan sequence of round 3, both
# landings, both collectives, ordinary receive buffers. A rehearsal of the fallbacks the ranks agree on at N > 1.
set -u
O=gpurun_out/r5aj; mkdir -p $O
for v in "--host-plan" "--landing allreduce" "--collective allreduce" "--landing flags"; do
  timeout -k 10 200 python bench.py --force-dist $v --weights steady --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/out.json 2> $O/err.txt; rc=$?
  python -c "
import json
d=json.load(open('$O/out.json')); print('$v rc=$rc ms/step %.4f' % d['ms_per_step'], d.get('sharded_step',{}).get('landing'), d.get('sharded_step',{}).get('migration'))" || { tail -5 $O/err.txt; }
done
PHD_COARSE_RECV=1 timeout -k 10 200 python bench.py --force-dist --weights steady --steps 30 --warmup 3 --no-cpu-baseline --no-extra > $O/out.json 2> $O/err.txt; echo "coarse rc=$?"; python -c "
import json
d=json.load(open('$O/out.json')); print('coarse ms/step %.4f' % d['ms_per_step'], d['sharded_step'].get('landing'), d['sharded_step'].get('recv_buffer_finegrained'))"
