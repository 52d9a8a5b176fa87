#!/bin/bash
O=gpurun_out/${1:-r4ts4}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for spec in "PHD_PIPELINE=0" "PHD_DEVICE_ORDER=0" "PHD_DEVICE_ORDER=1"; do echo "$spec" | tee -a $O/tlstep.log; env $spec timeout -k 10 300 python scripts/timeline_step.py survey 2>$O/err.log | tee -a $O/tlstep.log || exit 1; done
