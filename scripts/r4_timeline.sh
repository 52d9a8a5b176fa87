#!/bin/bash
O=gpurun_out/${1:-r4tl}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for k in 2 6 4 3; do PHD_SPLIT=1 timeout -k 10 200 python scripts/timeline.py survey $k 2>$O/err$k.log | tee -a $O/timeline.log || exit 1; done
for k in 2 6; do timeout -k 10 200 python scripts/timeline.py survey $k 2>>$O/err$k.log | tee -a $O/timeline.log || exit 1; done
