#!/bin/bash
# A/B timing of build variants of libphdhip.so on the GPU box: the product build against builds with other tuning
# macros (-D...), each timed by the isolated-kernel leg of bench.py.   usage: scripts/ab_variants.sh "NAME=MACRO=VAL[,MACRO=VAL]" ...
# Variants are built HERE (hipcc cross-compiles), e.g.
#   python -c "from monorfs_amd import _lib; _lib.build(out='build/var_ef3.so', defines=['PHD_EF_WAVES=3'])"
# and this script, run through gpurun, times every build/var_*.so plus the product library.
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOTDIR"
for so in monorfs_amd/csrc/libphdhip.so build/var_*.so; do
	[ -f "$so" ] || continue
	PHDHIP_SO="$ROOTDIR/$so" timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d.get('kernel_ms_isolated',{})
print('%-40s step %.4f one-stream %.4f  ' % ('$so', d['ms_per_step'], d.get('ms_per_step_one_stream',0)) + ' '.join('%s %.4f' % (n.replace('k_',''), v) for n,v in k.items()), ' A %.4f' % d['other_modes']['config_A']['ms_per_step'])"
done
