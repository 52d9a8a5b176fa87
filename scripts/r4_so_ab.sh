#!/bin/bash
# A/B of library builds on the default bench line:  scripts/r4_so_ab.sh <tag> <so> [<so> ...]   (paths under monorfs_amd/csrc)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
for so in "$@"; do
  PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 400 python bench.py --no-cpu-baseline > $O/bench_${so%.so}_$rep.json 2> $O/bench_${so%.so}_$rep.err; echo "bench $so rc=$?"
  python - <<PY
import json
d = json.load(open("$O/bench_${so%.so}_$rep.json"))
print("$so ms/step", round(d["ms_per_step"], 4), "one stream", round(d.get("ms_per_step_one_stream", 0), 4), "iso", {k: round(x * 1e3, 1) for k, x in d.get("kernel_ms_isolated", {}).items()})
m = d.get("other_modes", {})
print("   steady", round(m.get("weights_steady", {}).get("ms_per_step", 0), 4), "A", round(m.get("config_A", {}).get("ms_per_step", 0), 4), "S", round(m.get("config_S", {}).get("ms_per_step", 0), 3), {k: round(x * 1e3, 1) for k, x in m.get("config_S", {}).get("kernel_ms_isolated", {}).items()})
PY
done
done
