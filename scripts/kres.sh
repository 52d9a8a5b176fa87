#!/bin/bash
# Registers / scratch / occupancy of the kernels, from the compiler's own remarks (no GPU needed):
#   scripts/kres.sh            the whole library (85 s)
#   scripts/kres.sh ep [-D..]  k_emit_finish, k_prune_merge, k_emit_prune alone (a few seconds: PHD_ONLY_EP)
set -u
cd "$(dirname "$0")/.."
MODE=${1:-all}; shift || true
T=$(mktemp -d)
if [ "$MODE" = ep ]; then
  printf '#define PHD_ONLY_EP\n#include "%s/monorfs_amd/csrc/phd_kernels.h"\n' "$PWD" > $T/ep.hip
  SRC=$T/ep.hip
else
  SRC=monorfs_amd/csrc/phdhip.hip
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-unused-value -Wno-unused-result "$@" -Rpass-analysis=kernel-resource-usage -o $T/x.so $SRC 2> $T/log.txt
python scripts/kernel_resources.py $T/log.txt
grep -E "error" $T/log.txt | head
rm -rf $T
