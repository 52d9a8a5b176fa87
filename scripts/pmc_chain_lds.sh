#!/bin/bash
# k_particle_chain at config A with and without helper workgroups: LDS and VALU counters per launch (on the GPU box).
#   scripts/gpurun_retry.sh 600 'bash scripts/pmc_chain_lds.sh'
#   another counter set: PMC_TAG=pmc_chain_wait PMC_COUNTERS='SQ_WAIT_ANY SQ_WAIT_INST_ANY ...' bash scripts/pmc_chain_lds.sh
set -u
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=${PMC_TAG:-pmc_chain_lds}
COUNTERS=${PMC_COUNTERS:-SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES}
OUT=$ROOTDIR/gpurun_out/$TAG; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for v in 0 256; do
  export PHD_DSPLIT_MAX=$v
  timeout -k 10 200 rocprofv3 --pmc $COUNTERS --output-format csv -d "$OUT/d$v" -- python3 "$ROOTDIR/bench.py" --config A --weights steady --steps 40 --warmup 2 --no-cpu-baseline --no-extra --no-events > "$OUT/log$v.txt" 2>&1 || { tail -5 "$OUT/log$v.txt"; exit 1; }
  python3 - "$OUT/d$v" $v <<'PY' | tee -a "$OUT/summary.txt"
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(int))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in acc:
    if "chain" in k:
        print("PHD_DSPLIT_MAX=%s %s per launch:" % (sys.argv[2], k), {c: round(v / n[k][c]) for c, v in sorted(acc[k].items())})
PY
done
