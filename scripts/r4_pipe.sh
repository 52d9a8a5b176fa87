#!/bin/bash
set -u
O=gpurun_out/${1:-r4pipe}; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -4 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
bash scripts/r4_ab2.sh $1 libphdhip.so:PHD_PIPELINE=0 libphdhip.so:PHD_PIPELINE=1 libphdhip.so:PHD_PIPELINE=1,PHD_EVENT_FLAGS=2
