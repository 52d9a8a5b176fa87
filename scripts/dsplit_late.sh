#!/bin/bash
# Where the chain's helper workgroups cost: PHD_DSPLIT_MAX=0 (no helpers), PHD_DSPLIT_LATE=0 (helpers), 2 (helpers that leave at once: every main
# keeps its density sums), 3 (helpers that wait but are never picked). Config A, steady. On the GPU box.
for rep in 1 2; do for mode in off 0 2 3; do  # (LATE: 2 = helpers leave at once, 3 = never picked)
if [ $mode = off ]; then export PHD_DSPLIT_MAX=0; unset PHD_DSPLIT_LATE; else export PHD_DSPLIT_MAX=256 PHD_DSPLIT_LATE=$mode; fi
timeout -k 10 200 python bench.py --config A --weights steady --no-cpu-baseline --no-extra --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('mode=$mode ms/step %.5f' % d['ms_per_step'])"
done; done
