#!/bin/bash
# The chain's helper workgroups at the reference's real-time particle counts (24 ... 256 particles x 128 components x 32 measurements):
# PHD_DSPLIT_MAX=0 (no helpers) against the default, ms per step (posted) and per synchronous update. On the GPU box.
for cfg in A24 A64 A128 A; do for v in 0 256 0 256; do
PHD_DSPLIT_MAX=$v timeout -k 10 200 python bench.py --config $cfg --weights steady --no-cpu-baseline --no-extra --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg PHD_DSPLIT_MAX=$v ms/step %.5f' % d['ms_per_step'])"
done; done
