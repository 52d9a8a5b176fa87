#!/usr/bin/env python3
"""Diagnostic: the TIMELINE of one launch of a stamped kernel (-DPHD_STAMPS build, PHD_TL_BEGIN / PHD_TL_END): when every workgroup
started and ended (constant 100 MHz counter) and where it ran. Says how much of a launch is ramp, rounds and tail:
    sum of lifetimes / slots   against   last end - first start.
Usage on the GPU box: PHD_SPLIT=1 python scripts/timeline.py survey 2     (2 k_emit_prune, 3 k_alpha_assoc, 4 k_alpha_density, 6 k_sweep)"""
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
kid = int(sys.argv[2]) if len(sys.argv) > 2 else 2
os.environ["PHD_STAMP_KERNEL"] = str(100 + kid)
shape = tuple(int(x) for x in os.environ.get("PHD_STAMP_SHAPE", "2048,512,64").split(","))
slots_per_cu = int(os.environ.get("PHD_TL_SLOTS", "4"))
f = Frame(shape[0], shape[1], shape[2], 1004 if shape[0] == 4096 else 1002, weight_profile=prof)
p = prm3d_defaults(shape[0], max(600, shape[1]), shape[2])
p.max_quantity = max(600, shape[1])
nav = navigator.PHDNavigator(p, particlecount=shape[0])
nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
nav.set_measurements(f.z)
nav.set_frozen(True)
nav.set_all_pairs(True)
for _ in range(int(os.environ.get("PHD_TL_STEPS", "20"))):
    nav.step_async(0.5)
nav.sync()
out = np.zeros((shape[0], 16))
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
nav.close()
if os.environ.get("PHD_TL_SAVE"):
    np.save(os.environ["PHD_TL_SAVE"], out[:, :4])

t0 = out[:, 0] * 0.01   # microseconds
t1 = out[:, 1] * 0.01
hw = out[:, 2].astype(np.int64)
xcc = out[:, 3].astype(np.int64) & 15
cu = (hw >> 8) & 15
sh = (hw >> 12) & 1
se = (hw >> 13) & 7
place = ((xcc * 8 + se) * 2 + sh) * 16 + cu
nsplit = int(os.environ.get("PHD_SPLIT", "0") or 0)
groups = [np.arange(shape[0])]
if nsplit != 1:   # the launches of the two half-ranges are reported apart and together
    h = shape[0] // 2
    groups = [np.arange(shape[0]), np.arange(h), np.arange(h, shape[0])]
for g in groups:
    a, b = t0[g] - t0[g].min(), t1[g] - t0[g].min()
    life = b - a
    span = b.max()
    ncu = len(np.unique(place[g]))
    slots = ncu * slots_per_cu
    order = np.argsort(a)
    print("kernel %d %s %s split %s, workgroups %d..%d: span %.1f us; lifetimes mean %.1f median %.1f p10 %.1f p90 %.1f max %.1f us; "
          "sum of lifetimes / (%d CUs seen x %d) = %.1f us = %.0f %% of the span"
          % (kid, prof, shape, nsplit or "default", g[0], g[-1], span, life.mean(), np.median(life), np.percentile(life, 10), np.percentile(life, 90), life.max(),
             ncu, slots_per_cu, life.sum() / slots, 100 * life.sum() / slots / span))
    print("   starts (us after the first): p1 %.1f p25 %.1f p50 %.1f p51 %.1f p75 %.1f p99 %.1f last %.1f;  ends: first %.1f p50 %.1f p90 %.1f p99 %.1f last %.1f"
          % (tuple(np.percentile(a, q) for q in (1, 25, 50, 51, 75, 99, 100)) + tuple(np.percentile(b, q) for q in (0, 50, 90, 99, 100))))
    # workgroups alive over the span, in 20 bins
    edges = np.linspace(0, span, 21)
    alive = [int(((a < e1) & (b > e0)).sum() * 0 + np.clip(np.minimum(b, e1) - np.maximum(a, e0), 0, None).sum() / (e1 - e0)) for e0, e1 in zip(edges[:-1], edges[1:])]
    print("   workgroups alive (mean per twentieth of the span):", alive)
    first = life[order[: len(g) // 2]]
    second = life[order[len(g) // 2:]]
    print("   lifetimes of the first half to start: mean %.1f; of the second half: mean %.1f; blockIdx of the 8 last to end: %s; workgroups per CU: min %d max %d"
          % (first.mean(), second.mean(), (g[np.argsort(b)[-8:]]).tolist(), np.bincount(place[g])[np.unique(place[g])].min(), np.bincount(place[g]).max()))
