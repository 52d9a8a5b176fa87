#!/bin/bash
O=gpurun_out/${1:-r3j}; mkdir -p $O
for nt in 1024 512 256; do
PHD_NR_THREADS=$nt timeout -k 10 200 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --weights steady --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nr threads $nt: steady step %.4f' % d['ms_per_step'], 'NR %.4f' % d['kernel_ms']['k_normalise_resample'])" | tee -a $O/nr_threads.log
PHD_NR_THREADS=$nt timeout -k 10 200 python bench.py --config A --steps 100 --warmup 5 --no-cpu-baseline --weights steady --no-extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('nr threads $nt: config A step %.4f' % d['ms_per_step'], d['kernel_ms'])" | tee -a $O/nr_threads.log
done
