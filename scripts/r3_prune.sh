#!/bin/bash
# every GPU test, then k_prune_merge's phases (stamps build) and the bench line   usage: scripts/r3_prune.sh <tag>
O=gpurun_out/${1:-r3p}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q ${PYTEST_K:+-k "$PYTEST_K"} > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -3 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
for prof in survey steady; do timeout -k 10 200 python scripts/stamps.py $prof 2 2>/dev/null | tail -1 | sed "s/^/B $prof: /" | tee -a $O/stamps.log; done
PHD_STAMP_SHAPE=256,128,32 timeout -k 10 200 python scripts/stamps.py steady 2 2>/dev/null | tail -1 | sed "s/^/A: /" | tee -a $O/stamps.log
timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --extra-steps 40 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_ms_isolated',{})
print('step %.4f one-stream %.4f ' % (d['ms_per_step'], d.get('ms_per_step_one_stream',0)) + ' '.join('%s %.4f' % (n.replace('k_',''), v) for n,v in k.items()), ' steady %.4f A %.4f S %.3f' % (d['other_modes']['weights_steady']['ms_per_step'], d['other_modes']['config_A']['ms_per_step'], d['other_modes']['config_S']['ms_per_step']))" | tee -a $O/bench.log
