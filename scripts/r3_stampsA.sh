#!/bin/bash
O=gpurun_out/${1:-r3u}; mkdir -p $O
export PHD_STAMP_SHAPE=256,128,32
for k in 5 2 3; do timeout -k 10 200 python scripts/stamps.py steady $k 2>/dev/null | tail -1 | sed "s/^/kernel $k: /" | tee -a $O/stampsA.log; done
