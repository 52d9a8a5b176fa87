#!/usr/bin/env python3
"""Diagnostic (stamps build): shader-clock shares of the phases of the quasi set log-likelihood kernels on bench.py's shape."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import _lib
_lib.SO_PATH = os.path.join(_lib.CSRC, "libphdhip_stamps.so")
if not os.path.exists(_lib.SO_PATH):
    import subprocess
    subprocess.check_call(["/opt/rocm/bin/hipcc"] + _lib.HIPCC_FLAGS + ["-DPHD_STAMPS", "-o", _lib.SO_PATH, os.path.join(_lib.CSRC, "phdhip.hip")], cwd=_lib.CSRC)
os.environ["PHD_STAMP_KERNEL"] = "3"
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import measure_to_map_identity, measure_perfect_identity
P = 2048
p = prm3d_defaults(P, 600, 64)
nav = navigator.PHDNavigator(p, particlecount=P)
qrng = np.random.default_rng(77)
nlm, nz, nq = 50, 50, P
zc = np.stack([qrng.uniform(-300, 300, nlm), qrng.uniform(-220, 220, nlm), qrng.uniform(0.3, 1.8, nlm)], axis=1)
lms = measure_to_map_identity(zc)
qz = measure_perfect_identity(lms[qrng.choice(nlm, size=nz, replace=False)]) + qrng.normal(size=(nz, 3)) * np.sqrt([2.0, 2.0, 1e-3])
qposes = np.tile([0, 0, 0, 1.0, 0, 0, 0], (nq, 1)) + np.concatenate([qrng.normal(size=(nq, 3)) * 2e-3, np.zeros((nq, 1)), qrng.normal(size=(nq, 3)) * 5e-4], axis=1)
nav._lib.phd_debug_stamps.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
for name, fn in (("value", lambda: nav.QuasiSetLogLikelihood(qz, lms, qposes)), ("gradient", lambda: nav.QuasiSetLogLikelihoodGradient(qz, lms, qposes))):
    fn(); fn()
    out = np.zeros((P, 16))
    nav._lib.phd_debug_stamps(nav._h, out.ctypes.data_as(C.POINTER(C.c_double)))
    m = out.mean(0)
    idx = [i for i in np.argsort(m[:11], kind="stable") if i == 0 or m[i] > 0]
    print(name, "cycles between stamps:", " ".join("%d->%d:%d" % (a, b, m[b] - m[a]) for a, b in zip(idx[:-1], idx[1:])), "total", int(m[:11].max()),
          "| clusters per pose %.1f, of them replayed whole %.2f" % (np.mean(out[:, 11] % 1000), np.mean(out[:, 11] // 1000)))
nav.close()
