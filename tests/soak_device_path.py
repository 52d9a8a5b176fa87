#!/usr/bin/env python3
"""Soak of the per-rank sharded step without a host wait (GPU box only): `world` handles in one process on one stream play the
sequence bench.py --gpus N runs — local step, "all-gather" (device copies), phd_step_global_device_async, phd_migration_push_async,
phd_migration_unpack_async with the landing flags — against ONE handle holding all particles, bit for bit at every step, over
random sizes (ranks of 16 ... 320 particles, multiples of 64 and not), random resampling numbers and drifting measurements.
    python tests/soak_device_path.py [sequences] [steps]        (PHD_NR_GRID_MIN=1 PHD_PLAN_GRID_MIN=1: the grid kernels on every size)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame
from test_gpu_round4 import _device_path_handles, _device_path_step


def one(seq, nsteps):
    rng = np.random.default_rng(8000 + seq)
    world = int(rng.choice([2, 3, 4, 6]))
    Pl = int(rng.choice([16, 64, 100, 128, 320]))
    C = int(rng.choice([24, 70]))
    M = int(rng.choice([8, 20, 40]))
    f = Frame(Pl * world, C, M, 8100 + seq, weight_profile="steady")
    if rng.random() < 0.5:                      # half of the sequences start depleted: long runs across the rank boundaries at once
        f.weights = rng.random(f.P) ** 10
        f.weights /= f.weights.sum()
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one_ = navigator.PHDNavigator(p1, particlecount=Pl * world)
    one_.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(navigator, f, world, Pl, M)
    flags = all(nv._lib.phd_migration_recv_is_finegrained(nv._h) == 1 for nv in navs)
    if flags:
        for nv in navs:
            nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    nres = 0
    for step in range(nsteps):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * rng.uniform(0.1, 1.0)
        u = float(rng.uniform(0.01, 0.99))
        one_.SlamUpdate(None, z, u_resample=u)
        for nv in navs:
            nv.set_measurements(z)
        _device_path_step(navs, Pl, u)
        nres += int(one_.resample_sources()[1])
        assert np.array_equal(one_.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs])), (seq, step, "weights")
        assert np.array_equal(one_.poses(), np.concatenate([nv.poses() for nv in navs])), (seq, step, "poses")
        for g in rng.choice(Pl * world, size=min(Pl * world, 10), replace=False):
            for x, y in zip(one_.MapModel(int(g)), navs[int(g) // Pl].MapModel(int(g) % Pl)):
                assert np.array_equal(x, y), (seq, step, "map", int(g))
    one_.close()
    for nv in navs:
        nv.close()
    print("sequence %d: %d ranks x %d particles, C=%d M=%d, %d steps, %d resamplings, landing %s ok" % (seq, world, Pl, C, M, nsteps, nres, "flags" if flags else "stream order"), flush=True)


if __name__ == "__main__":
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # (sequence numbers are the seeds)
    for s in range(first, first + nseq):
        one(s, nsteps)
    print("device-path soak ok")
