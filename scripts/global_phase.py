#!/usr/bin/env python3
"""Diagnostic: device time of the GLOBAL part of a sharded step at the size an 8-GPU run meets it (config C8: 16 384 weights,
2048 particles per rank), on ONE GPU: a handle of 2048 particles plays rank `r` of a world of 8 whose gathered weight vector
is filled by hand. Reports k_normalise_resample (on 16 384 weights), k_plan_migration, k_pack_particles and k_finish_sharded
from the library's own HIP events, and the host's wait for the plan's counts.
    python scripts/global_phase.py [world] [steady|survey]"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from bench import DevArray
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
prof = sys.argv[2] if len(sys.argv) > 2 else "steady"
P, Cc, M, seed = CONFIGS["B"]
f = Frame(P, Cc, M, seed, weight_profile=prof)
p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
nav = navigator.PHDNavigator(p, particlecount=P)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
lib, h = nav._lib, nav._h
Pg = P * world
gw = torch.as_tensor(DevArray(lib.phd_device_global_weights(h, Pg), Pg), device="cuda")
ip = C.POINTER(C.c_int32)
sc, rc = np.zeros(world, np.int32), np.zeros(world, np.int32)
out = {}
for rank in (0, world // 2, world - 1):
    waits = []
    for it in range(14):
        if it == 2:                           # (two untimed trips first: buffers made on first use, cold launches)
            nav.timing_reset(1)
            waits = []
        nav._check(lib.phd_step_local_async(h, 0))
        lw = torch.as_tensor(DevArray(lib.phd_device_local_weights(h), P), device="cuda")
        torch.cuda.synchronize()
        gw.copy_(lw.repeat(world))            # every rank holds the same shard here: the vector an all-gather would deliver
        torch.cuda.synchronize()
        nav._check(lib.phd_step_global_async(h, rank, world, 0.5))
        t0 = time.perf_counter()
        nav._check(lib.phd_migration_plan(h, rank, world, sc.ctypes.data_as(ip), rc.ctypes.data_as(ip)))
        waits.append(time.perf_counter() - t0)
        nav._check(lib.phd_migration_pack_async(h))
        if rc.sum():
            bpp = C.c_int64(0)
            lib.phd_migration_send_buffer(h, C.byref(bpp))
            torch.as_tensor(DevArray(lib.phd_migration_recv_buffer(h), int(rc.sum()) * (bpp.value // 8)), device="cuda").zero_()   # empty records
            torch.cuda.synchronize()
        # (no peers to exchange with: the receive buffer keeps whatever it holds; the finish kernel's cost is what is timed)
        nav._check(lib.phd_migration_unpack_async(h))
        nav.sync()
    t = nav.last_timings()
    out[rank] = {k: round(v * 1e3, 1) for k, v in t.items() if k in ("k_normalise_resample", "k_plan_migration", "k_pack_particles", "k_finish_sharded", "k_push_weights")}
    out[rank]["plan_wait_us(host, includes the global kernels)"] = round(float(np.median(waits)) * 1e6, 1)
    out[rank]["send/recv records"] = (int(sc.sum()), int(rc.sum()))
print("world %d, %d weights, prior weights '%s': device time per launch in us" % (world, Pg, prof))
for r, v in out.items():
    print(" rank %d:" % r, v)
nav.close()
