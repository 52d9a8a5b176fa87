#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of a stamped kernel (build with -DPHD_STAMPS into
monorfs_amd/csrc/libphdhip_stamps.so; never the product build). Usage on the GPU box:
    python scripts/stamps.py [steady|survey]"""
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
f = Frame(2048, 512, 64, 1002, weight_profile=prof)
p = prm3d_defaults(2048, 600, 64)
nav = navigator.PHDNavigator(p, particlecount=2048)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
for _ in range(3):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((2048, 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
m = out.mean(0)
print(prof, "mean cycles per phase:", np.round(m[:8]).astype(int), "sum", int(m[:8].sum()))
nav.close()
