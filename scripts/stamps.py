#!/usr/bin/env python3
"""Diagnostic: per-phase shader-clock shares of a stamped kernel (build with -DPHD_STAMPS into
monorfs_amd/csrc/libphdhip_stamps.so; never the product build). Usage on the GPU box:
    python scripts/stamps.py [steady|survey] [kernel id: 2 = k_prune_merge (+ slot 15: the emit body in front of it in the fused launch), 3 = k_alpha_assoc, 7 = the emit body's own stamps]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import _lib

extra = os.environ.get("PHD_STAMP_DEFS", "").split()   # extra -D flags, e.g. "-DPHD_STAMP_COUNTERS" (visit / test counters of k_prune_merge instead of clean timing)
so = os.path.join(_lib.CSRC, "libphdhip_stamps%s.so" % "".join(c for c in "".join(extra) if c.isalnum()))
if not os.path.exists(so):
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + ["-DPHD_STAMPS"] + extra + ["-o", so, os.path.join(_lib.CSRC, "phdhip.hip")])
_lib.SO_PATH = so
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "steady"
if len(sys.argv) > 2:
    os.environ["PHD_STAMP_KERNEL"] = sys.argv[2]   # 2 prune (default), 3 assoc
shape = tuple(int(x) for x in os.environ.get("PHD_STAMP_SHAPE", "2048,512,64").split(","))   # particles, components, measurements
f = Frame(shape[0], shape[1], shape[2], 1002, weight_profile=prof)
p = prm3d_defaults(shape[0], max(600, shape[1]), shape[2])
nav = navigator.PHDNavigator(p, particlecount=shape[0])
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
for _ in range(3):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((shape[0], 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
m = out.mean(0)   # stamp i = cycles since stamp 0 (stamps need not be numbered in time order)
if os.environ.get("PHD_STAMP_KERNEL") == "7":   # the emit body (wave 0 of every workgroup): its own layout
    print(prof, "emit body: cycles %d, of them inside the Kalman path %d in %.2f rounds of wave 0; runs (components with kept pairs) of the particle: %.0f, its queue entries: %.0f"
          % (m[1], m[2], m[3], m[4], m[5]))
    print("   before the queue scan (tables, staging of z and the denominators, pose): %d; the scan with its barriers: %d" % (m[6], m[7]))
    nav.close()
    sys.exit(0)
idx = [i for i in np.argsort(m[:12], kind="stable") if i == 0 or m[i] > 0]
print("counters 12..15 (sum over steps run):", m[12:16])
print(prof, "cycles between stamps:", " ".join("%d->%d:%d" % (a, b, m[b] - m[a]) for a, b in zip(idx[:-1], idx[1:])), "total", int(m[:12].max()))
nav.close()
