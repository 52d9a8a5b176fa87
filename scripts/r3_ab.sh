#!/bin/bash
# A/B of build variants (build/var_*.so) against the product library on both frames   usage: scripts/r3_ab.sh <tag>
O=gpurun_out/${1:-ab}; mkdir -p $O
for w in survey steady; do
for so in monorfs_amd/csrc/libphdhip.so build/var_*.so; do
	[ -f "$so" ] || continue
	PHDHIP_SO="$PWD/$so" timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --weights $w --extra-steps 10 2>$O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_ms_isolated',{})
print('%-8s %-34s step %.4f one-stream %.4f  ' % ('$w', '$so', d['ms_per_step'], d.get('ms_per_step_one_stream',0)) + ' '.join('%s %.4f' % (n.replace('k_',''), v) for n,v in k.items()), ' A %.4f' % d['other_modes']['config_A']['ms_per_step'])" | tee -a $O/ab.log
done
done
