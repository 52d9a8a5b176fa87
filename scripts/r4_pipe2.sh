#!/bin/bash
set -u
O=gpurun_out/${1:-r4pipe2}; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "back_to_back or full_size or kat or soak or round3" > $O/tests_quick.log 2>&1; echo "pytest quick rc=$?" | tee -a $O/tests_quick.log; tail -4 $O/tests_quick.log
grep -q "pytest quick rc=0" $O/tests_quick.log || exit 1
bash scripts/r4_ab2.sh $1 libphdhip.so:PHD_PIPELINE=0 libphdhip.so:PHD_DEVICE_ORDER=0 libphdhip.so:PHD_DEVICE_ORDER=1
