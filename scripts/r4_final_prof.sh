#!/bin/bash
# round-4 closing run, part 2: the rocprofv3 passes (scripts/profile_gpu.sh) of the bench command — every kernel alone on the
# chip (PHD_SPLIT=1) on both frames, the default two streams, config S and config A
set -u
PHD_SPLIT=1 bash scripts/profile_gpu.sh r04_b_survey_one_stream --weights survey --no-extra > gpurun_out/prof_r4b1.log 2>&1 || exit 1
echo "one-stream survey done"
PHD_SPLIT=1 bash scripts/profile_gpu.sh r04_b_steady_one_stream --weights steady --no-extra > gpurun_out/prof_r4b2.log 2>&1 || exit 1
echo "one-stream steady done"
bash scripts/profile_gpu.sh r04_b_survey_split2 --weights survey --no-extra > gpurun_out/prof_r4b3.log 2>&1 || exit 1
echo "split2 done"
bash scripts/profile_gpu.sh r04_b_configA --config A --weights steady --no-extra > gpurun_out/prof_r4b4.log 2>&1 || exit 1
echo "config A done"
PHD_SPLIT=1 bash scripts/profile_gpu.sh r04_b_configS --config S --weights survey --no-extra > gpurun_out/prof_r4b5.log 2>&1 || exit 1
echo "config S done"
