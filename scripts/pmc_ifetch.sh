#!/bin/bash
# One-off PMC pass (on the GPU box): instruction fetch of the per-particle kernels — requests / misses of the instruction cache,
# wave-cycles waiting for an instruction. Counters only (no trace domains).  usage: scripts/pmc_ifetch.sh <tag> [bench.py args...]
set -u
TAG=${1:-ifetch}; shift || true
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOTDIR/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail > "$OUT/avail.txt" 2>&1
grep -i -o -E "\b(SQC?_[A-Z0-9_]*(ICACHE|IFETCH|INST_CACHE|INSTS_SALU|INSTS_SMEM|WAIT_INST|INST_LEVEL)[A-Z0-9_]*)\b" "$OUT/avail.txt" | sort -u > "$OUT/names.txt"
cat "$OUT/names.txt" | tr '\n' ' '; echo
ARGS="--steps 20 --warmup 2 --no-cpu-baseline --no-events --no-extra $*"
PASS1="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_IFETCH"
PHD_SPLIT=1 timeout -k 10 300 rocprofv3 --pmc $PASS1 --output-format csv -d "$OUT/pmc" -- python3 "$ROOTDIR/bench.py" $ARGS > "$OUT/log.txt" 2>&1 || { tail -5 "$OUT/log.txt"; exit 1; }
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", 0))[:8]:
    a = acc[k]; c = max(n[k], 1)
    print("%-40s launches %4d  per launch: icache req %.3g hits %.3g misses %.3g  ifetch %.3g  wave-cycles %.3g  wait-inst %.3g (%.2f)" % (
        k[:40], c, a.get("SQC_ICACHE_REQ", 0) / c, a.get("SQC_ICACHE_HITS", 0) / c, a.get("SQC_ICACHE_MISSES", 0) / c, a.get("SQ_IFETCH", 0) / c,
        a.get("SQ_WAVE_CYCLES", 0) / c, a.get("SQ_WAIT_INST_ANY", 0) / c, a.get("SQ_WAIT_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)))
PY
