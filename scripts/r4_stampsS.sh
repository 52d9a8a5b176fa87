#!/bin/bash
O=gpurun_out/${1:-r4s}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
export PHD_STAMP_SHAPE=4096,1024,128
for k in 3 2; do timeout -k 10 300 python scripts/stamps.py survey $k 2>/dev/null | tail -2 | sed "s/^/S kernel $k: /" | tee -a $O/stampsS.log; done
export PHD_STAMP_SHAPE=256,128,32
for k in 5 3 2; do timeout -k 10 200 python scripts/stamps.py steady $k 2>/dev/null | tail -1 | sed "s/^/A kernel $k: /" | tee -a $O/stampsS.log; done
