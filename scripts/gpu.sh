#!/bin/bash
# One parametrised runner for the GPU box (replaces the per-round r3_*.sh / r4_*.sh one-offs). Runs ON the box, through gpurun:
#   scripts/gpurun_retry.sh 900 'bash scripts/gpu.sh <verb> <tag> [args...]'
# Everything a verb writes goes under gpurun_out/<tag>/ (merged back by gpurun). Steps are joined so that nothing follows a
# GPU step that timed out or failed.
#   tests  <tag> [pytest -k expression]            the GPU suite (or the part -k selects), one process
#   bench  <tag> [bench.py args...]                one bench line + a short summary of it
#   quick  <tag> "<-k expr>" [stamp kernel ids...] tests selected by -k, the phase stamps of the kernels named, the default bench line
#   so-ab  <tag> <cfg,cfg,...> <so> [<so>...]      A/B of library builds (files under monorfs_amd/csrc) on bench configs (B, S, A, Bsteady), two rounds
#   env-ab <tag> <VAR> <a> <b> [-k expr]           A/B of an environment switch of the library on the default bench line
#   stamps <tag> <frame> <kernel id> [...]         scripts/stamps.py (shader-clock phases per workgroup)
#   prof   <tag> [bench.py args...]                scripts/profile_gpu.sh: rocprofv3 kernel trace + the PMC passes (PHD_SPLIT from the environment)
#   final  <tag>                                   the whole suite, the default line, the sharded rehearsals
set -u
VERB=${1:?verb}; TAG=${2:?tag}; shift; shift
O=gpurun_out/$TAG; mkdir -p "$O"

summary() {   # <bench json>
python - "$1" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
print("ms/step %.4f  one stream %.4f  value %.4g  roofline %s %.3f" % (d["ms_per_step"], d.get("ms_per_step_one_stream", 0), d["value"], r.get("kernel"), r.get("frac", 0)))
print("  iso", {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
for k, v in d.get("other_modes", {}).items():
    print("  %s" % k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in v.items() if a in ("ms_per_step", "ms_per_synchronous_update", "value_ms", "value_and_gradient_ms", "us_per_call", "error")},
          {a: round(b * 1e3, 1) for a, b in v.get("kernel_ms_isolated", {}).items()})
if "sharded_step" in d:
    print("  sharded", d["sharded_step"].get("phase_ms"), "host issue us", d.get("host_issue_us_per_step"))
PY
}

case $VERB in
tests)
  timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=8 ${1:+-k "$1"} > $O/tests.log 2>&1; rc=$?
  echo "pytest rc=$rc" | tee -a $O/tests.log; tail -14 $O/tests.log; exit $rc ;;
bench)
  timeout -k 10 500 python bench.py "$@" > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"
  [ $rc -eq 0 ] && summary $O/bench.json; exit $rc ;;
quick)
  KEXPR=${1:-prune}; shift || true
  timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1; rc=$?
  echo "pytest rc=$rc" | tee -a $O/tests.log; tail -4 $O/tests.log; [ $rc -eq 0 ] || exit 1
  rm -f monorfs_amd/csrc/libphdhip_stamps*.so
  for k in "$@"; do timeout -k 10 300 python scripts/stamps.py survey $k 2>/dev/null | tail -2 | sed "s/^/kernel $k: /" | tee -a $O/stamps.log || exit 1; done
  timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; rc=$?; echo "bench rc=$rc"
  [ $rc -eq 0 ] && summary $O/bench.json; exit $rc ;;
so-ab)
  CFGS=${1:?configs}; shift
  for rep in 1 2; do for so in "$@"; do for cfg in ${CFGS//,/ }; do
    case $cfg in Bsteady) A="--config B --weights steady" ;; *) A="--config $cfg" ;; esac
    PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 300 python bench.py $A --no-cpu-baseline --no-extra --steps 100 --warmup 10 > $O/ab_${so%.so}_${cfg}_$rep.json 2> $O/err.log || { echo "$so $cfg failed"; tail -3 $O/err.log; exit 1; }
    python - $O/ab_${so%.so}_${cfg}_$rep.json "$so" $cfg <<'PY' | tee -a $O/ab.log
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], sys.argv[3], "ms/step %.4f" % d["ms_per_step"], "one stream %.4f" % d.get("ms_per_step_one_stream", 0), {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
PY
  done; done; done ;;
env-ab)
  VAR=$1; A=$2; B=$3; KEXPR=${4:-}
  if [ -n "$KEXPR" ]; then
    env $VAR=$B timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "$KEXPR" > $O/tests.log 2>&1; rc=$?
    echo "pytest ($VAR=$B) rc=$rc" | tee -a $O/tests.log; tail -3 $O/tests.log; [ $rc -eq 0 ] || exit 1
  fi
  for v in $A $B $A $B; do
    env $VAR=$v timeout -k 10 500 python bench.py --no-cpu-baseline > $O/bench_$v.json 2> $O/bench_$v.err || { echo "bench $VAR=$v failed"; exit 1; }
    echo "$VAR=$v"; summary $O/bench_$v.json
  done ;;
stamps)
  FRAME=$1; shift
  rm -f monorfs_amd/csrc/libphdhip_stamps*.so
  for k in "$@"; do timeout -k 10 300 python scripts/stamps.py $FRAME $k 2>$O/stamps.err | tail -3 | sed "s/^/kernel $k: /" | tee -a $O/stamps.log || exit 1; done ;;
prof)
  bash scripts/profile_gpu.sh "$TAG" "$@" > $O/prof.log 2>&1; rc=$?; echo "profile rc=$rc"; tail -3 $O/prof.log; exit $rc ;;
final)
  timeout -k 10 1100 python -m pytest tests -m gpu -q --durations=10 > $O/tests.log 2>&1; rc=$?
  echo "pytest rc=$rc" | tee -a $O/tests.log; tail -16 $O/tests.log; [ $rc -eq 0 ] || exit 1
  timeout -k 10 500 python bench.py > $O/bench_default.json 2> $O/bench_default.err || { echo "bench failed"; exit 1; }
  summary $O/bench_default.json
  for w in steady survey; do
    timeout -k 10 200 python bench.py --force-dist --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/forcedist_$w.json 2> $O/forcedist_$w.err || { echo "forcedist $w failed"; exit 1; }
    echo "force-dist $w"; summary $O/forcedist_$w.json
    timeout -k 10 200 python bench.py --weights $w --steps 40 --warmup 3 --no-cpu-baseline --no-extra > $O/plain_$w.json 2> $O/plain_$w.err || exit 1
    echo "plain $w"; summary $O/plain_$w.json
  done ;;
*) echo "unknown verb $VERB"; exit 2 ;;
esac
