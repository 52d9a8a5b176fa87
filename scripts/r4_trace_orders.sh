#!/bin/bash
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
O=$ROOTDIR/gpurun_out/${1:-r4trace}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 80 --warmup 2 --no-cpu-baseline --no-events --no-extra"
unset PHD_PIPELINE PHD_DEVICE_ORDER
export PHD_PIPELINE=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/forkjoin -- python3 $ROOTDIR/bench.py $ARGS > $O/log1.txt 2>&1 || exit 1
export PHD_PIPELINE=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/laststream -- python3 $ROOTDIR/bench.py $ARGS > $O/log2.txt 2>&1 || exit 1
export PHD_DEVICE_ORDER=1
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $O/device -- python3 $ROOTDIR/bench.py $ARGS > $O/log3.txt 2>&1 || exit 1
cd $ROOTDIR
python3 scripts/boundary_from_trace.py $O/forkjoin "fork / join around every step (PHD_PIPELINE=0)" | tee $O/boundary.txt
python3 scripts/boundary_from_trace.py $O/laststream "the end of the step on the stream that finishes last (default)" | tee -a $O/boundary.txt
python3 scripts/boundary_from_trace.py $O/device "device-side order (PHD_DEVICE_ORDER=1)" | tee -a $O/boundary.txt
find $O -name "*.csv" -size +2M -delete
