#!/usr/bin/env python3
"""Diagnostic: wall time of the phases of the sharded step on ONE rank (RCCL with world_size 1), each phase
synchronised on its own — an upper bound per phase, to see where the sharded path's overhead over the plain step sits.
    python scripts/dist_phases.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

from bench import DevArray
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
P, Cc, M, seed = CONFIGS["B"]
f = Frame(P, Cc, M, seed, weight_profile="steady")
p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
nav = navigator.PHDNavigator(p, particlecount=P)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
lib, h = nav._lib, nav._h
nav._check(lib.phd_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
gw = torch.as_tensor(DevArray(lib.phd_device_global_weights(h, P), P), device="cuda")
ip = C.POINTER(C.c_int32)
sc, rc = np.zeros(1, np.int32), np.zeros(1, np.int32)
empty = torch.empty(0, dtype=torch.float64, device="cuda")
names = ["local step", "all-gather", "global step", "plan (host)", "pack", "all-to-all", "unpack"]
acc = np.zeros(len(names))


def phase(i, fn):
    t = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    acc[i] += time.perf_counter() - t


N = 30
for it in range(N + 3):
    if it == 3:
        acc[:] = 0
    phase(0, lambda: nav._check(lib.phd_step_local_async(h, 0)))
    phase(1, lambda: dist.all_gather_into_tensor(gw, torch.as_tensor(DevArray(lib.phd_device_local_weights(h), P), device="cuda")))
    phase(2, lambda: nav._check(lib.phd_step_global_async(h, 0, 1, 0.5)))
    phase(3, lambda: nav._check(lib.phd_migration_plan(h, 0, 1, sc.ctypes.data_as(ip), rc.ctypes.data_as(ip))))
    phase(4, lambda: nav._check(lib.phd_migration_pack_async(h)))
    phase(5, lambda: dist.all_to_all_single(empty, empty, [0], [0]))
    phase(6, lambda: nav._check(lib.phd_migration_unpack_async(h)))
for n, a in zip(names, acc):
    print("%-14s %.3f ms" % (n, a / N * 1e3))
print("sum            %.3f ms" % (acc.sum() / N * 1e3))
nav.close()
dist.destroy_process_group()
