#!/usr/bin/env python3
"""Diagnostic: where one STEP's time is not inside its four per-particle kernels (-DPHD_STAMPS build, PHD_STAMP_KERNEL=199: every
workgroup of k_sweep, k_emit_prune, k_alpha_assoc and k_alpha_density leaves its start and end on the 100 MHz counter).
Usage on the GPU box: python scripts/timeline_step.py survey"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import _lib

so = os.path.join(_lib.CSRC, "libphdhip_stamps.so")
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + ["-DPHD_STAMPS", "-o", so, os.path.join(_lib.CSRC, "phdhip.hip")])
_lib.SO_PATH = so
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "survey"
os.environ["PHD_STAMP_KERNEL"] = "199"
shape = tuple(int(x) for x in os.environ.get("PHD_STAMP_SHAPE", "2048,512,64").split(","))
f = Frame(shape[0], shape[1], shape[2], 1002, weight_profile=prof)
p = prm3d_defaults(shape[0], max(600, shape[1]), shape[2])
nav = navigator.PHDNavigator(p, particlecount=shape[0])
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
nav.set_all_pairs(True)
for _ in range(100):
    nav.step_async(0.5)
nav.sync()
N = 200
t0 = time.perf_counter()
for _ in range(N):
    nav.step_async(0.5)
nav.sync()
period = (time.perf_counter() - t0) / N * 1e6
out = np.zeros((shape[0], 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
nav.close()
t = out[:, :8] * 0.01
base = t[:, 0].min()
names = ["k_sweep", "k_emit_prune", "k_alpha_assoc", "k_alpha_density"]
h = shape[0] // 2
print("step period %.1f us (%d steps, %s, %s); the last step's launches, us after its first k_sweep workgroup started:" % (period, N, prof, shape))
for half, sl in (("first half-range", slice(0, h)), ("second half-range", slice(h, None))):
    print("  %s: " % half + "; ".join("%s %.1f .. %.1f" % (n, t[sl, 2 * k].min() - base, t[sl, 2 * k + 1].max() - base) for k, n in enumerate(names)))
nr0, nr1 = out[0, 8] * 0.01 - base, out[0, 9] * 0.01 - base
print("  k_normalise_resample %.1f .. %.1f: %.1f us behind the last k_alpha_density workgroup; the next step's first k_sweep workgroup %.1f us behind its end"
      % (nr0, nr1, nr0 - (t[:, 7].max() - base), period - nr1))
span = t[:, 7].max() - base
life = sum((t[:, 2 * k + 1] - t[:, 2 * k]).sum() for k in range(4)) / 1024
print("  first k_sweep start -> last k_alpha_density end: %.1f us; period - that = %.1f us (k_normalise_resample, the launch gaps around it, fork and join); "
      "sum of all lifetimes / 1024 slots = %.1f us" % (span, period - span, life))
