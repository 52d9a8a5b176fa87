#!/bin/bash
# Where the one-launch chain (k_particle_chain) stops paying against the separate kernels: ms per step of both at a few
# particle counts (run through gpurun).   usage: scripts/chain_crossover.sh
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$ROOTDIR"
for cfg in A A512 A1024 B512 B1024; do
	for cm in 0 100000; do
		PHD_CHAIN_MAX=$cm timeout -k 10 120 python bench.py --config $cfg --steps 30 --warmup 5 --no-cpu-baseline --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-6s chain_max %-6s  %.4f ms/step' % ('$cfg', '$cm', d['ms_per_step']))"
	done
done
