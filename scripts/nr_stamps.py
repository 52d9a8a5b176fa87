#!/usr/bin/env python3
"""Diagnostic (stamps build only): shader-clock shares of the phases of k_normalise_resample on a gathered weight vector of
world x 2048 weights.   python scripts/nr_stamps.py [world]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, "libphdhip_stamps.so")
os.environ["PHD_STAMP_KERNEL"] = "6"
import torch
from bench import DevArray
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
P, Cc, M, seed = CONFIGS["B"]
f = Frame(P, Cc, M, seed, weight_profile="steady")
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
lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
acc = np.zeros(16)
for it in range(6):
    nav._check(lib.phd_step_local_async(h, 0))
    lw = torch.as_tensor(DevArray(lib.phd_device_local_weights(h), P), device="cuda")
    torch.cuda.synchronize()
    gw.copy_(lw.repeat(world))
    torch.cuda.synchronize()
    nav._check(lib.phd_step_global_async(h, 0, world, 0.5))
    nav._check(lib.phd_migration_plan(h, 0, world, sc.ctypes.data_as(ip), rc.ctypes.data_as(ip)))
    nav._check(lib.phd_migration_unpack_async(h))
    nav.sync()
    out = np.zeros((P, 16))
    lib.phd_debug_stamps(h, out.ctypes.data_as(C.POINTER(C.c_double)))
    if it >= 2:
        acc += out[0]
acc /= 4
names = ["load", "normalise", "stats+argmax", "scan of chunk sums", "speculation loop", "best / fallback", "write 1/P"]
print("world %d (%d weights): cycles per phase" % (world, Pg), {n: int(acc[i + 1] - acc[i]) for i, n in enumerate(names)}, "total", int(acc[7]), "speculation ok", acc[8])
nav.close()
