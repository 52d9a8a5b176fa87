#!/usr/bin/env python3
"""The headless simulation loop of the reference (`mono-rfs.exe -x -i=simulation -f=scene.world -c=moves.in -g=x.cfg
-p=N`; Simulation.Update, UI/Simulation.cs:560-680) around the HIP solver: a simulated vehicle follows the command file
through the scene, measures (SimulatedVehicle.Measure, SimulatedVehicle.cs:244-300: detection with the fuzzy field of
view, N(0, R) noise, Poisson clutter), the particles take the noisy odometry (phd_update_motion) and the measurements
(phd_slam_update), and the run is written as a record (Simulation.SaveToFile) that `-i=record` / scripts/replay.py can
replay. The random streams are numpy's, not AForge's: runs are statistically, not bitwise, those of the C# program.

    python scripts/simulate.py scene.world moves.in --out run/ [--config x.cfg] [--particles 20] [--seed 1]"""
import argparse
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import recordio as rio
from monorfs_amd.pose3d import add_odometry, diff_odometry, qconj, qmul


class SimulatedVehicle:
    """the measuring side of SimulatedVehicle<PRM3DMeasurer, Pose3D, PixelRangeMeasurement>"""

    def __init__(self, pose, measurer, landmarks, config, rng):
        self.pose = np.asarray(pose, float)
        self.focal, self.rmin, self.rmax = measurer[0], float(np.float32(measurer[1])), float(np.float32(measurer[2]))
        self.left, self.top, self.width, self.height = (int(v) for v in measurer[3:7])
        self.landmarks = np.asarray(landmarks, float).reshape(-1, 3)
        self.R = np.array(config["MeasurementCovariance"], float)
        self.Q = np.array(config["MotionCovariance"], float)
        self.pd, self.ramp = config["DetectionProbability"], list(config["VisibilityRamp"])
        self.clutter_count = config["ClutterDensity"] * self.height * self.width * (self.rmax - self.rmin)   # :111, Volume()
        self.perfect_still = config["PerfectStill"]
        self.odometry_pose, self.ref_odometry = self.pose.copy(), self.pose.copy()
        self.rng = rng

    def measure_perfect(self, x):   # PRM3DMeasurer.MeasurePerfect, PRM3DMeasurer.cs:138-149
        diff = x - self.pose[:3]
        q = self.pose[3:]
        l = qmul(qmul(qconj(q), np.concatenate([[0.0], diff])), q)[1:]
        return np.array([self.focal * l[0] / l[2], self.focal * l[1] / l[2], np.sign(l[2]) * np.linalg.norm(diff)])

    def detection_probability(self, z):   # SimulatedVehicle.cs:324-339 / FuzzyVisibleM, PRM3DMeasurer.cs:277-291
        d = min((z[0] - self.left) / self.ramp[0], (self.left + self.width - z[0]) / self.ramp[0],
                (z[1] - self.top) / self.ramp[1], (self.top + self.height - z[1]) / self.ramp[1],
                (z[2] - self.rmin) / self.ramp[2], (self.rmax - z[2]) / self.ramp[2])
        return self.pd * max(0.0, min(1.0, d))

    def update(self, dt, reading):   # Vehicle.Update / SimulatedVehicle.Update (:190-202)
        self.pose = add_odometry(self.pose, reading)
        self.odometry_pose = add_odometry(self.odometry_pose, reading)
        if not (self.perfect_still and not np.any(reading)):
            noise = dt * (np.linalg.cholesky(self.Q) @ self.rng.normal(size=6))
            self.odometry_pose = add_odometry(self.odometry_pose, noise)

    def read_odometry(self):   # Vehicle.ReadOdometry (:342-352)
        reading = diff_odometry(self.odometry_pose, self.ref_odometry)
        self.odometry_pose, self.ref_odometry = self.pose.copy(), self.pose.copy()
        return reading

    def measure(self):   # SimulatedVehicle.Measure (:244-300)
        out = []
        chol = np.linalg.cholesky(self.R)
        for x in self.landmarks:
            z = self.measure_perfect(x)
            pd = self.detection_probability(z)
            if pd > 0 and self.rng.uniform() < pd:
                out.append(z + chol @ self.rng.normal(size=3))
        nclutter = min(int(self.rng.poisson(self.clutter_count)), int(self.clutter_count * 10)) if self.clutter_count > 0 else 0
        for _ in range(nclutter):   # PRM3DMeasurer.RandomMeasure (:249-256)
            out.append(np.array([self.rng.uniform() * self.width + self.left, self.rng.uniform() * self.height + self.top,
                                 self.rng.uniform() * (self.rmax - self.rmin) + self.rmin]))
        return np.array(out, float).reshape(-1, 3)


def simulate(scene_text, command_lines, config, particles, seed, onlymapping=False, log=None):
    from monorfs_amd import navigator
    rng = np.random.default_rng(seed)
    pose, measurer, landmarks = rio.scene_from_descriptor(scene_text)
    if measurer is None:
        measurer = [575.8156, 0.1, 2.0, -320, -240, 640, 480]   # new PRM3DMeasurer(), PRM3DMeasurer.cs:70-73
    commands = rio.commands_from_descriptor(command_lines)
    explorer = SimulatedVehicle(pose, measurer, landmarks, config, rng)
    p = rio.phd_params_from_config(config, measurer=[measurer[0], float(np.float32(measurer[1])), float(np.float32(measurer[2]))] + list(measurer[3:7]),
                                   max_particles=particles, max_measurements=max(64, 8 * len(landmarks) + 64))
    nav = navigator.PHDNavigator(p, particlecount=particles, onlymapping=onlymapping, pose=pose)
    motion_chol = np.linalg.cholesky(config["MotionCovarianceMultiplier"] * np.array(config["MotionCovariance"], float))
    frame = 1.0 / 30            # FrameElapsed (Manipulator.cs:351, 30 fps)
    measure_elapsed = config["MeasureElapsed"]
    t, last_update = 0.0, 0.0
    trajectory, odometry, measurements, estimate, maps, tags = [(0.0, pose.copy())], [], [], [], [], []
    best_track = [(0.0, pose.copy())]
    for cmd in commands:
        reading = np.asarray(cmd[:6], float)
        if len(cmd) > 6 and cmd[6] != 0:   # Simulation.cs:580-590, 626-634
            if cmd[6] > 0:
                tags.append((t, "SLAM mode on"))
                nav.OnlyMapping = False
                nav.CollapseParticles(particles)    # StartSlam (PHDNavigator.cs:214-217)
            else:
                tags.append((t, "Mapping mode on"))
                nav.OnlyMapping = True
                nav.CollapseParticles(1)            # StartMapping (:224-227)
        origtime = t
        t += frame
        explorer.update(frame, reading)
        corrupt = explorer.read_odometry() if config["UseOdometry"] else np.zeros(6)
        odometry.append((origtime, corrupt))
        n = nav.particle_count
        noise = None if nav.OnlyMapping else frame * (rng.normal(size=(n, 6)) @ motion_chol.T)   # TrackVehicle.UpdateNoisy
        nav.UpdateOdometry(t, corrupt, noise, perfect_still=config["PerfectStill"])
        trajectory.append((t, explorer.pose.copy()))
        if t - last_update >= measure_elapsed - 1e-12:
            z = explorer.measure()
            measurements.append((t, z))
            nav.SlamUpdate(t, z, u_resample=float(rng.uniform(1e-9, 1.0)))
            last_update = t
        best_track.append((t, nav.BestEstimate))
        estimate.append((t, list(best_track)))
        maps.append((t, nav.BestMapModel))
        if log:
            log("t = %.3f: %d measurements, %d particles, %d components in the best map" % (t, len(measurements[-1][1]) if measurements else 0, nav.particle_count, len(maps[-1][1][0])))
    nav.close()
    return {"scene.world": scene_text, "trajectory.out": rio.serialize_timed_array(trajectory),
            "odometry.out": rio.serialize_timed_array(odometry), "measurements.out": rio.serialize_measurements(measurements),
            "estimate.out": rio.serialize_trajectories(estimate), "maps.out": rio.serialize_maps(maps),
            "tags.out": rio.serialize_tags(tags), "config.cfg": rio.serialize_config(config)}


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("scene")
    ap.add_argument("commands")
    ap.add_argument("--config")
    ap.add_argument("--out", required=True, help="record directory or .zip")
    ap.add_argument("--particles", type=int, default=20)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--onlymapping", action="store_true")
    args = ap.parse_args()
    config = rio.default_config()
    if args.config:
        with open(args.config) as fh:
            config = rio.config_from_descriptor(fh.read().splitlines(), config, log=print)
    with open(args.scene) as fh:
        scene = fh.read()
    with open(args.commands) as fh:
        commands = [l for l in fh.read().splitlines() if l.strip()]
    rec = simulate(scene, commands, config, args.particles, args.seed, args.onlymapping, log=print)
    rio.write_record(args.out, rec)
    print("record written to", args.out)


if __name__ == "__main__":
    main()
