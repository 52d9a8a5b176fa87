#!/bin/bash
# A/B of the chain's helper workgroups (PHD_DSPLIT_MAX=0: none) on config A, both weight profiles, posted and synchronous. On the GPU box.
#   scripts/gpurun_retry.sh 600 'bash scripts/dsplit_ab.sh <tag>'
set -u
O=gpurun_out/${1:-dsplit}; mkdir -p $O
for rep in 1 2; do for v in 0 256; do for w in steady survey; do
  PHD_DSPLIT_MAX=$v timeout -k 10 200 python bench.py --config A --weights $w --no-cpu-baseline --no-extra --steps 200 --warmup 20 > $O/A_${w}_${v}_$rep.json 2> $O/err.log || { echo "failed $v $w"; tail -5 $O/err.log; exit 1; }
  python - $O/A_${w}_${v}_$rep.json $v $w <<'PY' | tee -a $O/ab.log
import json, sys
d = json.load(open(sys.argv[1]))
print("PHD_DSPLIT_MAX=%s %s ms/step %.5f" % (sys.argv[2], sys.argv[3], d["ms_per_step"]), {k: round(v * 1e3, 1) for k, v in d.get("kernel_ms_isolated", {}).items()})
PY
done; done; done
