#!/bin/bash
O=gpurun_out/${1:-r4ts}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for w in survey steady; do timeout -k 10 300 python scripts/timeline_step.py $w 2>$O/err.log | tee -a $O/tlstep.log || exit 1; done
PHD_STAMP_SHAPE=256,128,32 timeout -k 10 300 python scripts/timeline_step.py steady 2>$O/err.log | tee -a $O/tlstep.log
