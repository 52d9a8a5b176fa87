#!/usr/bin/env python3
"""Soak of the multi-device handle (GPU box only): long sequences with odometry, resampling and migration, a handle of
`shards` shards (device 0 listed several times on a one-GPU box) against a single handle, bit for bit at every step,
whole-state downloads included.   python tests/soak_multi.py [sequences] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame


def one(seq, nsteps):
    rng = np.random.default_rng(5000 + seq)
    shards = int(rng.choice([2, 3, 4, 6, 8]))
    P = shards * int(rng.choice([16, 64, 200]))
    C = int(rng.choice([30, 90]))
    M = int(rng.choice([8, 24, 60]))
    f = Frame(P, C, M, 6000 + seq, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
    p.max_quantity = int(rng.choice([120, 600]))
    single = navigator.PHDNavigator(p, particlecount=P)
    multi = navigator.PHDNavigator(p, particlecount=P, devices=[0] * shards)
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    nres = 0
    for step in range(nsteps):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * rng.uniform(0.1, 1.0)
        u = float(rng.uniform(0.01, 0.99))
        reading = rng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.003, 0.003, 0.003]
        noise = rng.normal(0, 1, (P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
        if step % 5 == 4:
            # a batch of steps posted back to back (the multi handle's workers run ahead of the caller), one wait at the end
            us = [float(x) for x in rng.uniform(0.01, 0.99, int(rng.integers(2, 6)))]
            for nav in (single, multi):
                nav.UpdateOdometry(None, reading, noise)
                nav.set_measurements(z)
                for ub in us:
                    nav.step_async(ub)
                nav.sync()
        else:
            for nav in (single, multi):
                nav.UpdateOdometry(None, reading, noise)
                nav.SlamUpdate(None, z, u_resample=u)
        assert np.array_equal(single.VehicleWeights, multi.VehicleWeights), (seq, step, "weights")
        assert np.array_equal(single.poses(), multi.poses()), (seq, step, "poses")
        assert single.BestParticle == multi.BestParticle, (seq, step, "best")
        sa, sb = single.resample_sources(), multi.resample_sources()
        assert sa[1] == sb[1] and np.array_equal(sa[0], sb[0]), (seq, step, "sources")
        nres += int(sa[1])
        for i in rng.choice(P, size=min(P, 12), replace=False):
            for x, y in zip(single.MapModel(int(i)), multi.MapModel(int(i))):
                assert np.array_equal(x, y), (seq, step, "map", int(i))
        if step % 4 == 3:
            (pa, ca, qa, wa), (pb, cb, qb, wb) = single.download_state(600), multi.download_state(600)
            assert np.array_equal(ca, cb) and np.array_equal(qa, qb) and np.array_equal(wa, wb), (seq, step, "download")
            for i in range(P):
                assert np.array_equal(pa[:, i, :ca[i]], pb[:, i, :cb[i]]), (seq, step, "download map", i)
                w, m, c = single.MapModel(i)
                assert np.array_equal(pa[0, i, :ca[i]], w), (seq, step, "download vs getter", i)
    single.close(); multi.close()
    print("sequence %d: %d shards, P=%d C=%d M=%d maxq=%d, %d steps, %d resamplings ok" % (seq, shards, P, C, M, p.max_quantity, nsteps, nres), flush=True)


if __name__ == "__main__":
    nseq = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # (sequence numbers are the seeds)
    for s in range(first, first + nseq):
        one(s, nsteps)
    print("multi soak ok")
