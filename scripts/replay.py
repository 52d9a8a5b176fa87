#!/usr/bin/env python3
"""Replay a monorfs record through the HIP solver: the measurements.out / odometry.out stream that
drives the C# solver with `-i=record` drives PHDNavigator here, and estimate.out / maps.out come back in the same
format (Simulation.SaveToFile, Simulation.cs:391-488).

    python scripts/replay.py <record.zip | record dir> [--particles N] [--seed S] [--out DIR] [--oracle]
    python scripts/replay.py --make DIR [--frames K]      # write a small synthetic record first

The host keeps what the reference keeps managed: the motion noise dt * chol(Q) * N(0, I) per particle and the
resampling uniform are drawn here (numpy), the motion step itself runs on the device (phd_update_motion)."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import recordio as rio
from monorfs_amd.abi import prm3d_defaults



def frames_of(rec):
    odo = rio.timed_array_from_descriptor([l for l in rec["odometry.out"].split("\n") if l != ""], 6)
    z = rio.measurements_from_descriptor(rec["measurements.out"], 3)
    zt = {round(t, 9): pts for t, pts in z}
    return [(t, reading, zt.get(round(t, 9), np.zeros((0, 3)))) for t, reading in odo]


def config_of(rec):
    """the record's config.cfg over the defaults (Config.FromRecordFile, Config.cs:127-148), or the defaults"""
    return rio.config_from_descriptor(rec["config.cfg"].splitlines()) if "config.cfg" in rec else rio.default_config()


def params_of(rec, particles, maxm):
    pose, measurer, _ = rio.scene_from_descriptor(rec["scene.world"])
    p = rio.phd_params_from_config(config_of(rec), max_particles=particles, max_components=600, max_measurements=max(maxm, 1))
    if measurer is not None:
        p.measurer[:] = [measurer[0], float(np.float32(measurer[1])), float(np.float32(measurer[2]))] + list(measurer[3:7])
    return p, pose


class DeviceSolver:
    def __init__(self, p, pose, particles):
        from monorfs_amd import navigator
        self.nav = navigator.PHDNavigator(p, particlecount=particles)
        self.nav.reset(pose, (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3))), particles)

    def step(self, reading, noise, z, u):
        self.nav.UpdateOdometry(None, reading, noise)
        if len(z):
            self.nav.SlamUpdate(None, z, u_resample=u)
        b = self.nav.BestParticle
        return self.nav.poses()[b], self.nav.MapModel(b)


def replay(rec, particles, seed, solver_cls):
    frames = frames_of(rec)
    p, pose = params_of(rec, particles, max(len(z) for _, _, z in frames))
    solver = solver_cls(p, pose, particles)
    rng = np.random.default_rng(seed)
    cfg = config_of(rec)
    chol = np.linalg.cholesky(cfg["MotionCovarianceMultiplier"] * np.array(cfg["MotionCovariance"], float))   # PHDNavigator.cs:257-259
    trajectory, estimate, maps = [], [], []
    tprev = frames[0][0]
    for t, reading, z in frames:
        dt = t - tprev
        tprev = t
        noise = dt * (rng.normal(size=(particles, 6)) @ chol.T)     # Util.RandomGaussianVector, Util.cs:173-202
        u = float(rng.uniform(1e-6, 1.0))
        bpose, bmap = solver.step(reading, noise, z, u)
        trajectory.append((t, bpose))
        estimate.append((t, list(trajectory)))
        maps.append((t, bmap))
    return {"estimate.out": rio.serialize_trajectories(estimate), "maps.out": rio.serialize_maps(maps)}


def make_synthetic_record(path, frames=12, landmarks=14, seed=3):
    """a vehicle drifting forward past a handful of landmarks; measurements = MeasurePerfect + N(0, R) of the visible
    ones (+ one clutter point now and then), written exactly as Simulation.SaveToFile would"""
    from monorfs_amd import prm3d
    from monorfs_amd.pose3d import add_odometry
    rng = np.random.default_rng(seed)
    p = prm3d_defaults()
    focal, measurer, ramp = p.measurer[0], list(p.measurer), list(p.visibility_ramp)
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    zs = np.column_stack([rng.uniform(-260, 260, landmarks), rng.uniform(-190, 190, landmarks), rng.uniform(0.5, 1.6, landmarks)])
    lm = np.array([prm3d.measure_to_map(pose, z, focal) for z in zs])
    scene = rio.serialize_scene(pose, p.measurer, lm)
    _, _, lm = rio.scene_from_descriptor(scene)                     # what a reader of the file sees
    odo, meas, traj = [], [], []
    R = np.array(p.R).reshape(3, 3)
    for k in range(frames):
        t = k / 30.0
        reading = np.array([0.004, 0.001 * np.sin(k), 0.006, 0.002, -0.001, 0.0005]) if k else np.zeros(6)
        pose = add_odometry(pose, reading)
        pts = []
        for x in lm:
            z = prm3d.measure_perfect(pose, x, focal)
            if prm3d.fuzzy_visible(z, measurer, ramp) > 0 and rng.uniform() < 0.9:
                pts.append(z + rng.normal(size=3) * np.sqrt(np.diag(R)))
        if k % 4 == 1:
            pts.append([rng.uniform(-300, 300), rng.uniform(-220, 220), rng.uniform(0.3, 1.8)])
        odo.append((t, reading))
        meas.append((t, pts))
        traj.append((t, pose))
    rio.write_record(path, {"scene.world": scene, "odometry.out": rio.serialize_timed_array(odo),
                            "measurements.out": rio.serialize_measurements(meas), "trajectory.out": rio.serialize_timed_array(traj),
                            "tags.out": "0 SLAM mode on", "config.cfg": rio.serialize_config(rio.default_config())})
    return path


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("record", nargs="?")
    ap.add_argument("--make")
    ap.add_argument("--frames", type=int, default=12)
    ap.add_argument("--particles", type=int, default=32)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--out")
    a = ap.parse_args()
    if a.make:
        print("wrote", make_synthetic_record(a.make, a.frames))
        if not a.record:
            return
    rec = rio.read_record(a.record)
    out = replay(rec, a.particles, a.seed, DeviceSolver)
    if a.out:
        rio.write_record(a.out, dict(rec, **out))
    maps = rio.map_history_from_descriptor(out["maps.out"])
    print("replayed %d frames, final map: %d components, expected size %.3f" % (len(maps), len(maps[-1][1][0]), maps[-1][1][0].sum()))


if __name__ == "__main__":
    main()
