#!/bin/bash
O=gpurun_out/${1:-r4ts3}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for pl in 0 1; do for ef in 0 2; do echo "PHD_PIPELINE=$pl PHD_EVENT_FLAGS=$ef" | tee -a $O/tlstep.log; PHD_PIPELINE=$pl PHD_EVENT_FLAGS=$ef timeout -k 10 300 python scripts/timeline_step.py survey 2>$O/err.log | tee -a $O/tlstep.log || exit 1; done; done
