#!/bin/bash
export PHD_SPLIT=1
bash scripts/profile_gpu.sh r03_a_survey_one_stream --weights survey > gpurun_out/prof_a1.log 2>&1; echo "rc=$?"
bash scripts/profile_gpu.sh r03_a_steady_one_stream --weights steady > gpurun_out/prof_a2.log 2>&1; echo "rc=$?"
