#!/usr/bin/env python3
"""Diagnostic: the DISTRIBUTION over workgroups of a stamped kernel's lifetime (-DPHD_STAMPS build): a kernel ends with its
slowest workgroup. Usage on the GPU box: PHD_STAMP_SHAPE=4096,1024,128 python scripts/stamps_dist.py survey 3"""
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

prof = sys.argv[1] if len(sys.argv) > 1 else "survey"
os.environ["PHD_STAMP_KERNEL"] = sys.argv[2] if len(sys.argv) > 2 else "3"
shape = tuple(int(x) for x in os.environ.get("PHD_STAMP_SHAPE", "2048,512,64").split(","))
f = Frame(shape[0], shape[1], shape[2], 1004 if shape[0] == 4096 else 1002, weight_profile=prof)
p = prm3d_defaults(shape[0], max(600, shape[1]), shape[2])
p.max_quantity = max(600, shape[1])
nav = navigator.PHDNavigator(p, particlecount=shape[0])
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
nav.set_all_pairs(True)
for _ in range(3):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((shape[0], 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
life = out[:, :11].max(axis=1)
print("kernel", os.environ["PHD_STAMP_KERNEL"], prof, shape, "workgroup lifetime in shader cycles: mean %.0f median %.0f p90 %.0f p99 %.0f max %.0f; workgroups above 4 x median: %d of %d; the ten longest: %s"
      % (life.mean(), np.median(life), np.percentile(life, 90), np.percentile(life, 99), life.max(), int((life > 4 * np.median(life)).sum()), len(life),
         np.sort(life)[-10:].astype(int).tolist()))
nav.close()
