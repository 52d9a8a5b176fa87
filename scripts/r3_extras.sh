#!/bin/bash
# A/B of build variants + sizes met by the kernels (run through gpurun)   usage: scripts/r3_extras.sh <tag>
TAG=${1:-r3x}; O=gpurun_out/$TAG; mkdir -p $O
for prof in steady survey; do timeout -k 10 200 python scripts/mapstats.py $prof > $O/mapstats_$prof.log 2>&1; done
for w in survey steady; do [ -n "${SKIP_AB:-}" ] && break
for so in monorfs_amd/csrc/libphdhip.so build/var_*.so; do
	[ -f "$so" ] || continue
	PHDHIP_SO="$PWD/$so" timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --weights $w --extra-steps 20 2>$O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_ms_isolated',{})
print('%-8s %-34s step %.4f one-stream %.4f  ' % ('$w', '$so', d['ms_per_step'], d.get('ms_per_step_one_stream',0)) + ' '.join('%s %.4f' % (n.replace('k_',''), v) for n,v in k.items()), ' A %.4f' % d['other_modes']['config_A']['ms_per_step'])" | tee -a $O/ab.log
done
done
for prof in steady survey; do for k in 2 3; do timeout -k 10 200 python scripts/stamps.py $prof $k 2>/dev/null | tail -2 >> $O/stamps.log; done; done
cat $O/stamps.log
