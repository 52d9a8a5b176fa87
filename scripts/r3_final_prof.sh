#!/bin/bash
# round-3 closing run, part 2: the rocprofv3 passes (scripts/profile_gpu.sh) of the bench command — every kernel alone on the
# chip (PHD_SPLIT=1) on both frames, then the default two streams
set -u
PHD_SPLIT=1 bash scripts/profile_gpu.sh r03_b_survey_one_stream --weights survey > gpurun_out/prof_b1.log 2>&1 || exit 1
echo "one-stream survey done"
PHD_SPLIT=1 bash scripts/profile_gpu.sh r03_b_steady_one_stream --weights steady > gpurun_out/prof_b2.log 2>&1 || exit 1
echo "one-stream steady done"
bash scripts/profile_gpu.sh r03_b_survey_split2 --weights survey > gpurun_out/prof_b3.log 2>&1 || exit 1
echo "split2 done"
