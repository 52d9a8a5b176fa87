#!/bin/bash
O=gpurun_out/${1:-r4tl2}; mkdir -p $O
rm -f monorfs_amd/csrc/libphdhip_stamps*.so
for k in 6 2 4 3; do PHD_TL_SAVE=$O/tl_k${k}_a.npy PHD_SPLIT=1 timeout -k 10 200 python scripts/timeline.py survey $k 2>$O/err$k.log | tee -a $O/timeline.log || exit 1; done
for k in 6 2; do PHD_TL_STEPS=7 PHD_TL_SAVE=$O/tl_k${k}_b.npy PHD_SPLIT=1 timeout -k 10 200 python scripts/timeline.py survey $k 2>$O/err$k.log | tee -a $O/timeline.log || exit 1; done
for k in 6 2 4; do PHD_TL_SAVE=$O/tl_k${k}_steady.npy PHD_SPLIT=1 timeout -k 10 200 python scripts/timeline.py steady $k 2>$O/err$k.log | tee -a $O/timeline.log || exit 1; done
