#!/bin/bash
# round-4 profiling session: the perfect-particle test, the two clock probes, the FETCH_SIZE calibration, the prune's phase
# stamps, then the rocprofv3 passes of the bench command on one stream (survey frame)
set -u
TAG=${1:-r4c}; O=$PWD/gpurun_out/$TAG; mkdir -p $O
ROOTDIR=$PWD
timeout -k 10 600 python -m pytest tests/test_gpu_round4.py -m gpu -x -q -k "perfect or device_path" -s > $O/tests.log 2>&1; echo "pytest rc=$?" | tee -a $O/tests.log; tail -5 $O/tests.log
timeout -k 10 120 scripts/probes/bin/fp64_clock > $O/fp64_clock.txt 2>&1; echo "fp64_clock rc=$?"; cat $O/fp64_clock.txt
timeout -k 10 300 python scripts/sweep_clock.py survey > $O/sweep_clock.txt 2> $O/sweep_clock.err; echo "sweep_clock rc=$?"; tail -2 $O/sweep_clock.txt
for k in 2; do timeout -k 10 200 python scripts/stamps.py survey $k 2>/dev/null | tail -2 | tee -a $O/stamps_prune.log; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*" | sort -u > $O/counters.txt; cat $O/counters.txt | head -20
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $O/calib -- $ROOTDIR/scripts/probes/bin/fetch_calib > $O/calib.log 2>&1; echo "calib rc=$?"; tail -3 $O/calib.log )
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/calib/*/*_counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        acc[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: sum(v) / len(v) for c, v in cs.items()})
PY
PHD_SPLIT=1 bash scripts/profile_gpu.sh r04_a_survey_one_stream --weights survey > $O/prof_a1.log 2>&1; echo "profile rc=$?"
