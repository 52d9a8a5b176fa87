#!/bin/bash
# parity tests of the per-particle kernels + bench on both frames + prune stamps
O=gpurun_out/${1:-r3l}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "parity or round2 or kat or soak or full_size or round3" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
for w in steady survey; do
timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --weights $w --extra-steps 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d.get('kernel_ms_isolated',{})
print('%-7s step %.4f one-stream %.4f ' % ('$w', d['ms_per_step'], d.get('ms_per_step_one_stream',0)) + ' '.join('%s %.4f' % (n.replace('k_',''), v) for n,v in k.items()), ' A %.4f' % d['other_modes']['config_A']['ms_per_step'], ' S %.3f' % d['other_modes'].get('config_S',{}).get('ms_per_step',0))" | tee -a $O/bench.log
done
for prof in steady survey; do timeout -k 10 200 python scripts/stamps.py $prof 2 2>/dev/null | tail -1 >> $O/stamps.log; done; cat $O/stamps.log
