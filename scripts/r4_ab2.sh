#!/bin/bash
# quick A/B of library builds / environments on the headline step only:  scripts/r4_ab2.sh <tag> "<so>[:ENV=V[,ENV=V]]" ...
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2 3; do
for spec in "$@"; do
  so=${spec%%:*}; envs=""; [ "$spec" != "$so" ] && envs=$(echo "${spec#*:}" | tr ',' ' ')
  for w in survey steady; do
    r=$(env $envs PHDHIP_SO=$PWD/monorfs_amd/csrc/$so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extra --steps 200 --warmup 20 --weights $w 2>$O/err.log | python -c "import sys,json; print(round(json.loads(sys.stdin.read())['ms_per_step'],4))")
    echo "$spec $w $r" | tee -a $O/ab.log
  done
done
done
python - <<PY
import collections
d = collections.defaultdict(list)
for l in open("$O/ab.log"):
    s, w, v = l.split(); d[(s, w)].append(float(v))
for k, v in d.items(): print(k, "min %.4f median %.4f" % (min(v), sorted(v)[len(v) // 2]), v)
PY
