#!/bin/bash
# round-3 closing run, part 3: the same rocprofv3 passes for config A (the one-launch chain) and config S, and a kernel trace of
# the default bench command WITH its extra legs (the quasi kernels, config S and A next to the headline)
set -u
bash scripts/profile_gpu.sh r03_b_configA --config A --weights steady > gpurun_out/prof_b4.log 2>&1 || exit 1
echo "config A done"
bash scripts/profile_gpu.sh r03_b_configS --config S --weights survey > gpurun_out/prof_b5.log 2>&1 || exit 1
echo "config S done"
ROOTDIR=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOTDIR/gpurun_out/prof_r03_b_default_extras; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOTDIR/bench.py" --no-cpu-baseline > "$OUT/log.txt" 2>&1 || exit 1
echo "default with extras done"
