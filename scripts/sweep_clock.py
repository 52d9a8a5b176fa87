#!/usr/bin/env python3
"""Diagnostic (-DPHD_STAMPS build, PHD_STAMP_KERNEL=6): the clock MI355X holds under k_sweep — shader cycles (s_memtime)
against the 100 MHz counter (s_memrealtime) over each workgroup's lifetime and over its pair loops alone, after two seconds
of back-to-back steps (MI355X_MICROARCH.md, DVFS give-back item 6). Usage on the GPU box: python scripts/sweep_clock.py [survey|steady]"""
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
os.environ["PHD_STAMP_KERNEL"] = "6"
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame

prof = sys.argv[1] if len(sys.argv) > 1 else "survey"
P, Cc, M = 2048, 512, 64
f = Frame(P, Cc, M, 1002, weight_profile=prof)
p = prm3d_defaults(P, 600, M)
nav = navigator.PHDNavigator(p, particlecount=P)
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
nav.set_all_pairs(True)
nav.set_split(1)
t0 = time.time()
n = 0
while time.time() - t0 < 2.5:
    for _ in range(50):
        nav.step_async(0.5)
    nav.sync()
    n += 50
out = np.zeros((P, 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
clk = out[:, 1] / out[:, 2] * 100e6
pclk = out[:, 3] / np.maximum(out[:, 4], 1) * 100e6
print("%s frame, %d steps warm: k_sweep workgroup lifetime %.0f shader cycles = %.1f us (median); in-kernel clock %.0f MHz (median; p10 %.0f, p90 %.0f); "
      "pair loops %.0f cycles of it (%.0f %%), clock inside them %.0f MHz"
      % (prof, n, np.median(out[:, 1]), np.median(out[:, 2]) / 100.0, np.median(clk) / 1e6, np.percentile(clk, 10) / 1e6, np.percentile(clk, 90) / 1e6,
         np.median(out[:, 3]), 100 * np.median(out[:, 3]) / np.median(out[:, 1]), np.median(pclk) / 1e6))
nav.close()
