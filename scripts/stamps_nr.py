#!/usr/bin/env python3
"""Diagnostic: phase stamps of k_normalise_resample (one workgroup; -DPHD_STAMPS build, PHD_STAMP_KERNEL=6: it leaves its
stamps in row 0 of the slab behind k_sweep's). Usage on the GPU box: PHD_STAMP_SHAPE=256,128,32 PHD_FOLD_NR=0 python scripts/stamps_nr.py steady"""
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
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "steady"
os.environ["PHD_STAMP_KERNEL"] = "6"
shape = tuple(int(x) for x in os.environ.get("PHD_STAMP_SHAPE", "256,128,32").split(","))
f = Frame(shape[0], shape[1], shape[2], 1002, weight_profile=prof)
p = prm3d_defaults(shape[0], max(600, shape[1]), shape[2])
nav = navigator.PHDNavigator(p, particlecount=shape[0])
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
for _ in range(5):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((shape[0], 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
m = out[0]
names = ["start", "weights staged", "sum", "normalised, squares, best", "prefix sums", "slots", "sources", "weights reset"]
print("k_normalise_resample", prof, shape, "cycles since its first stamp:", ", ".join("%s %d" % (n, v) for n, v in zip(names, m[:8])), "| speculation held:", int(m[8]))
nav.close()
