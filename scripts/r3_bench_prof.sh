#!/bin/bash
# the default bench line as the driver runs it, then the rocprofv3 passes of the same command (scripts/profile_gpu.sh)
TAG=${1:-r03_a}; O=gpurun_out/$TAG; mkdir -p $O
( time timeout -k 10 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err ) 2> $O/bench.time; echo "bench rc=$?"; tail -3 $O/bench.time
bash scripts/profile_gpu.sh $TAG --weights survey > $O/profile.log 2>&1; echo "profile rc=$?"
