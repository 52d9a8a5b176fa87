#!/usr/bin/env python3
"""Soak: long randomized SlamUpdate sequences, device against the oracle at every step (GPU box only).
    python tests/soak.py [sequences] [steps] [first sequence number]
Every sequence draws its own sizes, pose motion, measurement noise and resampling numbers; the maps evolve (births,
merges, cuts), so the device meets states no fixed fixture has. Stops at the first disagreement."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import orc
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame



def run(nseq, nsteps, first=0, log=print):
    worst = 0.0
    for seq in range(first, first + nseq):
        worst = max(worst, one_sequence(seq, nsteps, log))
    return worst


def one_sequence(seq, nsteps, log):
    worst = 0.0
    rng = np.random.default_rng(9000 + seq)
    P = int(rng.choice([8, 24, 48]))
    C0 = int(rng.choice([20, 60, 140]))
    M = int(rng.choice([6, 20, 40, 70]))
    f = Frame(P, C0, M, 7000 + seq, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
    p.max_quantity = int(rng.choice([80, 600]))
    nav = navigator.PHDNavigator(p, particlecount=P)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    st = orc.State(P, 900)
    st.poses[:] = f.poses
    st.w[:, :C0], st.mean[:, :C0], st.cov[:, :C0], st.n[:] = f.w, f.mean, f.cov, C0
    nres = 0
    for step in range(nsteps):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * rng.uniform(0.1, 1.0)
        if rng.uniform() < 0.3:   # a few measurements wander off: births and misdetections
            k = rng.integers(0, M, max(1, M // 8))
            z[k] += rng.normal(size=(len(k), 3)) * [40.0, 40.0, 0.1]
        u = float(rng.uniform(0.01, 0.99))
        poses = st.poses.copy()
        poses[:, :3] += rng.normal(0, 2e-3, (P, 3))      # the host's motion model between frames
        st.poses[:] = poses
        nav.set_poses(poses)
        best, src, res, _ = orc.slam_update(p, st, z, u=u, threads=8)
        nav.SlamUpdate(None, z, u_resample=u)
        gsrc, gres = nav.resample_sources()
        assert gres == res and np.array_equal(gsrc, src), "seq %d step %d: resampling differs" % (seq, step)
        assert nav.BestParticle == best, "seq %d step %d: best particle" % (seq, step)
        gw = nav.VehicleWeights
        rel = np.max(np.abs(gw - st.weights) / np.maximum(np.abs(st.weights), 1e-300))
        worst = max(worst, rel)
        assert rel < 1e-6, "seq %d step %d: particle weights off by %g" % (seq, step, rel)
        for i in (0, P - 1, int(best)):
            gm, om = nav.MapModel(i), st.map(i)
            assert len(gm[0]) == len(om[0]), "seq %d step %d: map %d has %d components, oracle %d" % (seq, step, i, len(gm[0]), len(om[0]))
            assert np.allclose(gm[0], om[0], rtol=1e-6, atol=1e-12) and np.allclose(gm[1], om[1], rtol=1e-6, atol=1e-9)
        nres += res
    log("sequence %d: P=%d C0=%d M=%d maxq=%d, %d steps, %d resamplings, final map sizes %d..%d ok"
        % (seq, P, C0, M, p.max_quantity, nsteps, nres, st.n.min(), st.n.max()))
    nav.close()
    return worst


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 15
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    print("soak ok, worst particle-weight deviation %.3g" % run(n, k, first))
