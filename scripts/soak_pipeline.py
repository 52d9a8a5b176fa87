#!/usr/bin/env python3
"""Soak of the step boundary's three stream orders at full size: config B, un-frozen, N steps posted back to back with a changing
resampling variate (a sync and a read-back every 50), once per order in a process of its own; the read-backs must agree bit for bit.
    python scripts/soak_pipeline.py [steps]            (on the GPU box)"""
import hashlib
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(steps):
    from monorfs_amd import navigator
    from monorfs_amd.abi import prm3d_defaults
    from monorfs_amd.synth import Frame
    f = Frame(2048, 512, 64, 1002, weight_profile="steady")
    p = prm3d_defaults(2048, 600, 64)
    p.max_quantity = 600
    nav = navigator.PHDNavigator(p, particlecount=2048)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    nav.set_measurements(f.z)
    rng = np.random.default_rng(7)
    h = hashlib.sha256()
    for k in range(steps):
        nav.step_async(float(rng.uniform()))
        if (k + 1) % 50 == 0 or k == steps - 1:
            nav.sync()
            src, res = nav.resample_sources()
            h.update(nav.VehicleWeights.tobytes()); h.update(np.asarray(src).tobytes()); h.update(bytes([int(res)])); h.update(str(nav.BestParticle).encode())
            for i in (0, 1023, 1024, 2047):
                for arr in nav.MapModel(i):
                    h.update(np.ascontiguousarray(arr).tobytes())
            print("  step %d: resampled %d best %d digest %s" % (k + 1, res, nav.BestParticle, h.hexdigest()[:16]), flush=True)
    nav.close()
    print("DIGEST", h.hexdigest())


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(int(sys.argv[2]))
        sys.exit(0)
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    digests = {}
    for name, env in (("fork/join", {"PHD_PIPELINE": "0"}), ("last stream ends the step", {"PHD_PIPELINE": "1", "PHD_DEVICE_ORDER": "0"}),
                      ("device order", {"PHD_PIPELINE": "1", "PHD_DEVICE_ORDER": "1"})):
        e = dict(os.environ); e.update(env)
        out = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(steps)], env=e, capture_output=True, text=True, timeout=900)
        print(name, "rc", out.returncode)
        print(out.stdout[-1200:])
        if out.returncode != 0:
            print(out.stderr[-2000:])
            sys.exit(1)
        digests[name] = [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]
    ok = len(set(digests.values())) == 1
    print("soak_pipeline:", "all three orders agree bit for bit over %d steps" % steps if ok else "MISMATCH %s" % digests)
    sys.exit(0 if ok else 1)
