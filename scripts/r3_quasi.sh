#!/bin/bash
# quasi / gradient tests, then the default bench line with its quasi leg   usage: scripts/r3_quasi.sh <tag>
O=gpurun_out/${1:-r3m}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "quasi or loopy or loglike or gradient or slab" > $O/tests.log 2>&1; echo "pytest rc=$?" >> $O/tests.log; tail -3 $O/tests.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --extra-steps 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['other_modes']['quasi_set_loglik'], d['other_modes']['config_A']['ms_per_step'])"
