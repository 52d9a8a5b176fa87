#!/bin/bash
set -u
O=gpurun_out/${1:-r4nr2}; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "resampl or deplet or normalis or launch_shape or kat or slam or soak or back_to_back or chain_ends" > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -3 $O/tests.log
grep -q "pytest rc=0" $O/tests.log || exit 1
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for sh in 256,128,32 2048,16,16 16384,16,16; do PHD_STAMP_SHAPE=$sh PHD_FOLD_NR=0 timeout -k 10 200 python scripts/stamps_nr.py steady 2>/dev/null | tail -1 | tee -a $O/nr.log; done
bash scripts/r4_ab2.sh $1 x_base.so libphdhip.so
