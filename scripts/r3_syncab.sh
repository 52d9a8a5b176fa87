#!/bin/bash
# A/B of the synchronous-update cost at config A: product library against build/var_*.so   usage: scripts/r3_syncab.sh <tag>
O=gpurun_out/${1:-syncab}; mkdir -p $O
for so in monorfs_amd/csrc/libphdhip.so build/var_*.so; do
	[ -f "$so" ] || continue
	PHDHIP_SO="$PWD/$so" timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-steps 100 2>$O/ab.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
a=d['other_modes']['config_A']
print('$so', 'B step', round(d['ms_per_step'],4), 'A async', round(a['ms_per_step'],4), 'A synchronous', round(a['ms_per_synchronous_update'],4), 'quasi', d['other_modes']['quasi_set_loglik']['value_ms'], d['other_modes']['quasi_set_loglik']['value_and_gradient_ms'])" | tee -a $O/ab.log
done
