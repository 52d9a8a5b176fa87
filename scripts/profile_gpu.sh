#!/bin/bash
# Runs on the GPU box (through gpurun): rocprofv3 kernel trace + three separate PMC passes of the
# same bench command, written under gpurun_out/prof_<tag>/. Summarise afterwards with
# scripts/summarize_profile.py and commit the summaries under profiles/.
#   usage: scripts/profile_gpu.sh <tag> [bench.py args...]
set -u
TAG=${1:-run}; shift || true
ROOTDIR=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOTDIR/gpurun_out/prof_$TAG
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 40 --warmup 2 --no-cpu-baseline --no-events $*"   # (40 steps: the cold first launches then weigh 2 % in the averages, not 10 %)
echo "== kernel trace" > "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$ROOTDIR/bench.py" $ARGS >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc sq" >> "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- python3 "$ROOTDIR/bench.py" $ARGS >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc fetch" >> "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$ROOTDIR/bench.py" $ARGS >> "$OUT/log.txt" 2>&1 || exit 1
echo "== pmc write" >> "$OUT/log.txt"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 "$ROOTDIR/bench.py" $ARGS >> "$OUT/log.txt" 2>&1 || exit 1
find "$OUT" -name '*.csv' | head -20
