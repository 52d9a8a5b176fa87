#!/bin/bash
# More seeds of the three soaks (on the GPU box), also with the grid kernels forced on every vector length.
#   scripts/soak_more.sh <tag> <first sequence> [multiplier of the sequence counts]
set -u
O=gpurun_out/${1:?tag}; mkdir -p $O; F=${2:-100}; K=${3:-1}
timeout -k 10 1000 python tests/soak.py $((24 * K)) 15 $F > $O/soak.txt 2>&1 || { tail -5 $O/soak.txt; exit 1; }; tail -1 $O/soak.txt
timeout -k 10 1000 python tests/soak_multi.py $((30 * K)) 20 $F > $O/soak_multi.txt 2>&1 || { tail -5 $O/soak_multi.txt; exit 1; }; tail -1 $O/soak_multi.txt
PHD_NR_GRID_MIN=1 PHD_PLAN_GRID_MIN=1 timeout -k 10 1000 python tests/soak_multi.py $((30 * K)) 20 $((F + 30 * K)) > $O/soak_multi_grid.txt 2>&1 || { tail -5 $O/soak_multi_grid.txt; exit 1; }; tail -1 $O/soak_multi_grid.txt
timeout -k 10 1000 python tests/soak_device_path.py $((20 * K)) 15 $F > $O/soak_device.txt 2>&1 || { tail -5 $O/soak_device.txt; exit 1; }; tail -1 $O/soak_device.txt
PHD_NR_GRID_MIN=1 PHD_PLAN_GRID_MIN=1 timeout -k 10 1000 python tests/soak_device_path.py $((20 * K)) 15 $((F + 20 * K)) > $O/soak_device_grid.txt 2>&1 || { tail -5 $O/soak_device_grid.txt; exit 1; }; tail -1 $O/soak_device_grid.txt
