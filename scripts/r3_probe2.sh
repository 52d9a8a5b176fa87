#!/bin/bash
O=gpurun_out/${1:-r3f}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "resampl or migration_plan or multi or sharded or kat or soak or sequence" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
for w in 8 1; do timeout -k 10 200 python scripts/nr_stamps.py $w 2>/dev/null | tail -1 | tee -a $O/nr_stamps.log; done
timeout -k 10 200 python scripts/global_phase.py 8 steady > $O/global_phase.log 2>&1; tail -4 $O/global_phase.log
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --weights steady --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steady step %.4f' % d['ms_per_step'], d['kernel_ms'])"
