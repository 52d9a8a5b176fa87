#!/bin/bash
# A/B of a library variant built as monorfs_amd/csrc/libphdhip_pre.so (here: the helper's pre-staging, before it became the product build) against
# the product library: the helper test on the variant, then config A's frame at 64 / 128 / 256 particles. On the GPU box.
PHDHIP_SO=$PWD/monorfs_amd/csrc/libphdhip_pre.so timeout -k 10 300 python -m pytest tests/test_gpu_round5.py -m gpu -x -q -k "helper_workgroups" > gpurun_out/s2p_t.log 2>&1 || { tail -20 gpurun_out/s2p_t.log; exit 1; }
tail -2 gpurun_out/s2p_t.log
for cfg in A64 A128 A; do for so in libphdhip.so libphdhip_pre.so libphdhip.so libphdhip_pre.so; do
PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 200 python bench.py --config $cfg --weights steady --no-cpu-baseline --no-extra --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg $so ms/step %.5f' % d['ms_per_step'])"
done; done
