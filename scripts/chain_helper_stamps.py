#!/usr/bin/env python3
"""Diagnostic (stamped build, -DPHD_STAMPS): k_particle_chain's helper workgroups at config A — how many particles handed their density sums
over, and the hand-over's timing in 100 MHz ticks (10 ns): the main's association behind the hand-over, the helper's wake-up, the
helper's end. On the GPU box: python scripts/chain_helper_stamps.py [steady|survey]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import _lib

so = os.path.join(_lib.CSRC, "libphdhip_stamps.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + ["-DPHD_STAMPS", "-o", so, os.path.join(_lib.CSRC, "phdhip.hip")])
_lib.SO_PATH = so
os.environ["PHD_STAMP_KERNEL"] = "5"
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "steady"
P, Cc, M = 256, 128, 32
f = Frame(P, Cc, M, 1001, weight_profile=prof)
p = prm3d_defaults(P, 600, M)
nav = navigator.PHDNavigator(p, particlecount=P)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
for _ in range(5):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((P, 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
go = out[:, 12] > 0
print(prof, "particles whose helper took the density sums: %d of %d" % (go.sum(), P))
if go.any():
    q = lambda v: "median %.1f  p90 %.1f  max %.1f us" % tuple(np.percentile(v, [50, 90, 100]) / 100)
    print("  main: association behind the hand-over  ", q(out[go, 11]))
    print("  helper: hand-over -> awake              ", q(out[go, 14]))
    print("  helper: hand-over -> its sums are done  ", q(out[go, 15]))
nav.close()
