"""the reference's record / replay wire formats. The C++ host header is checked by a small program
(tests/recordio_check.cpp), the Python mirror against the same hand-written members of tests/golden/record/."""
import os
import subprocess
import sys

import numpy as np
import pytest

from monorfs_amd import recordio as rio
from monorfs_amd.abi import prm3d_defaults

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = os.path.join(ROOT, "tests", "golden", "record")


def test_cpp_header_parses_and_rewrites_the_record(tmp_path):
    from monorfs_amd import _lib
    so = _lib.build()   # PhdParamsFromConfig starts from phd_default_params (host code of the library; no GPU needed)
    exe = str(tmp_path / "recordio_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "recordio_check.cpp"), so, "-Wl,-rpath," + os.path.dirname(so),
                           "-Wl,-rpath,/opt/rocm/lib"])
    r = subprocess.run([exe, REC], capture_output=True, text=True)
    assert r.returncode == 0 and "recordio ok" in r.stdout, r.stdout + r.stderr


def test_python_mirror_round_trips_every_member(tmp_path):
    rec = rio.read_record(REC)
    pose, params, lm = rio.scene_from_descriptor(rec["scene.world"])
    assert pose[0] == 0.1 and params[0] == 575.816 and lm.shape == (3, 3) and lm[2, 1] == -0.4
    assert rio.serialize_scene(pose, params, lm) == rec["scene.world"]                    # Vehicle.ToString("g6")
    odo = rio.timed_array_from_descriptor(rec["odometry.out"].split("\n"), 6)
    assert odo[2][1][5] == -1.25e-05 and rio.serialize_timed_array(odo) == rec["odometry.out"]
    z = rio.measurements_from_descriptor(rec["measurements.out"], 3)
    assert len(z[0][1]) == 0 and z[2][1][0, 0] == 100.123456789012 and rio.serialize_measurements(z) == rec["measurements.out"]
    maps = rio.map_history_from_descriptor(rec["maps.out"])
    assert maps[1][1][2][0, 2, 2] == 1e12 and rio.serialize_maps(maps) == rec["maps.out"]
    est = rio.trajectory_history_from_descriptor(rec["estimate.out"], 7)
    assert rio.serialize_trajectories(est) == rec["estimate.out"]
    assert len(rio.trajectory_history_from_descriptor(rec["estimate.out"], 7, True)[1][1]) == 2
    # the zip container of Simulation.SaveToFile
    zpath = str(tmp_path / "record.zip")
    rio.write_record(zpath, rec)
    assert rio.read_record(zpath) == rec


@pytest.mark.parametrize("text,dim,msg", [("0.1 1 2 3", 3, "no ':' delimiter"), ("0.1:1 2", 3, "wrong measurement dimension"),
                                          ("x:1 2 3", 3, "missing time"), ("0.1:1 2 y", 3, "invalid point")])
def test_measurement_errors_carry_the_reference_messages(text, dim, msg):
    with pytest.raises(rio.FormatError, match=msg):
        rio.measurements_from_descriptor(text, dim)


@pytest.mark.gpu
@pytest.mark.parametrize("frames,landmarks,particles", [(10, 14, 24), (12, 50, 20)], ids=["14-landmarks", "reference-config-20-particles-50-landmarks"])
def test_replay_of_a_record_matches_the_oracle(tmp_path, frames, landmarks, particles):
    """the same measurements.out / odometry.out stream through the HIP solver and through the oracle: the maps.out /
    estimate.out they write agree number by number (and almost always character by character at g6). The second case is
    BASELINE.json's "C# reference" configuration: 20 particles, ~50 landmarks, noisy odometry, SLAM mode, headless."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import replay
    path = replay.make_synthetic_record(str(tmp_path / "rec.zip"), frames=frames, landmarks=landmarks)
    rec = rio.read_record(path)
    import orc

    class OracleSolver:
        """the CPU oracle behind the interface scripts/replay.py drives its solver through"""

        def __init__(self, p, pose, particles):
            self.p = p
            self.st = orc.State(particles, 900)
            self.st.poses[:] = pose
            self.best = 0

        def step(self, reading, noise, z, u):
            self.st.poses[:] = orc.update_motion(self.st.poses, reading, noise)
            if len(z):
                self.best, _, _, _ = orc.slam_update(self.p, self.st, z, u=u, threads=8)
            return self.st.poses[self.best].copy(), tuple(np.array(x) for x in self.st.map(self.best))   # copies: the slab moves on

    dev = replay.replay(rec, particles, 5, replay.DeviceSolver)
    ref = replay.replay(rec, particles, 5, OracleSolver)
    dmaps, rmaps = rio.map_history_from_descriptor(dev["maps.out"]), rio.map_history_from_descriptor(ref["maps.out"])
    assert len(dmaps) == len(rmaps) == frames
    if landmarks >= 50:
        assert len(dmaps[-1][1][0]) >= 30, "the 50-landmark scene should leave a map of several dozen components"
    for (td, md), (tr, mr) in zip(dmaps, rmaps):
        assert td == tr and len(md[0]) == len(mr[0])
        for a, b in zip(md, mr):
            assert np.allclose(a, b, rtol=2e-5, atol=1e-9)           # g6 keeps six digits
    dest = rio.trajectory_history_from_descriptor(dev["estimate.out"], 7)
    rest = rio.trajectory_history_from_descriptor(ref["estimate.out"], 7)
    for (td, a), (tr, b) in zip(dest, rest):
        assert td == tr and all(np.allclose(x[1], y[1], rtol=2e-5, atol=1e-9) for x, y in zip(a, b))
    assert len(dmaps[-1][1][0]) >= 5                                  # the landmarks in view ended up in the map
    lines = list(zip(dev["maps.out"].split("\n"), ref["maps.out"].split("\n")))
    assert sum(x == y for x, y in lines) >= 0.98 * len(lines)


def test_config_commands_and_tags():
    """Config.FromDescriptor / ToString (Config.cs:155-209, 268-309), FileParser.CommandsFromDescriptor (:263-274),
    TimedMessageFromDescriptor (:237-256) / Manipulator.SerializedTags, and the parameter block a PHDNavigator built
    from that configuration reads (PHDNavigator.cs:257-259)."""
    rec = rio.read_record(REC)
    skipped = []
    cfg = rio.config_from_descriptor(rec["config.cfg"].splitlines(), log=skipped.append)   # File.ReadAllLines
    assert len(skipped) == 1 and "no colon" in skipped[0]                     # reported, then ignored (:162-165)
    assert "SomeFutureField" not in cfg                                       # no such field: ignored
    assert cfg["NParallel"] == 4 and cfg["MaxQuantity"] == 250 and cfg["MinEffectiveParticle"] == 0.3
    assert cfg["MeasurementCovariance"] == [[2.5, 0, 0], [0, 2.5, 0], [0, 0, 0.002]] and cfg["ClutterDensity"] == 3e-7
    assert cfg["GradientClip"] == 10.0 and cfg["BirthCovariance"][1][1] == 1e-2   # missing parameters are left as they were
    assert cfg["MeasureElapsed"] == 0.0333333                                 # whole ticks of 100 ns
    text = rio.serialize_config(cfg)
    assert rio.serialize_config(rio.config_from_descriptor(text.split("\n"))) == text
    assert "MeasurementCovariance: [2.5 0 0; 0 2.5 0; 0 0 0.002]" in text and "ClutterDensity: 3E-07" in text and "PerfectStill: False" in text
    p = rio.phd_params_from_config(cfg, max_particles=8)
    assert list(p.R) == [5.0, 0, 0, 0, 5.0, 0, 0, 0, 0.004]                  # MeasurementCovarianceMultiplier (PHDNavigator.cs:257-259)
    assert p.pd == 0.85 and p.clutter_density == 6e-7 and p.max_quantity == 250 and p.merge_threshold == 0.5
    assert p.min_weight == 0.002 and p.birth_weight == 0.04 and abs(p.visibility_ramp[2] - 0.134164078649987) < 1e-15
    with pytest.raises(rio.FormatError):
        rio.config_from_descriptor(["MaxQuantity: many"])
    with pytest.raises(rio.FormatError):
        rio.phd_params_from_config(dict(cfg, Model="Linear2D"))
    cmds = rio.commands_from_descriptor(["0.01 0 0 0 0 0.002", "0 0 0 0 0 0 1", "0.02 0 0 0 0 0 -1 1 0.5 0.25 2"])
    assert [len(c) for c in cmds] == [6, 7, 11] and cmds[2][6] == -1
    with pytest.raises(rio.FormatError, match="the double descriptor '0.1 x' is malformed"):
        rio.commands_from_descriptor(["0.1 x"])
    tags = rio.timed_message_from_descriptor(rec["tags.out"].split("\n"))
    assert rio.serialize_tags(tags) == rec["tags.out"] and all(isinstance(t, float) for t, _ in tags)
    with pytest.raises(rio.FormatError, match="the TimedMessage descriptor 'soon SLAM mode on' is malformed"):
        rio.timed_message_from_descriptor(["soon SLAM mode on"])


@pytest.mark.gpu
def test_simulation_loop_writes_a_replayable_record(tmp_path):
    """scripts/simulate.py: the reference's headless simulation loop (Simulation.Update) around the HIP solver. A vehicle
    drifts past 12 landmarks for 40 frames; the record it writes parses member by member, the map it ends with holds the
    landmarks it saw, each once (OSPA against them: no cardinality error, 15 cm), and the record replays through scripts/replay.py."""
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import orc
    import replay
    import simulate
    rng = np.random.default_rng(8)
    p = prm3d_defaults()
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    zs = np.column_stack([rng.uniform(-200, 200, 12), rng.uniform(-150, 150, 12), rng.uniform(0.6, 1.5, 12)])
    lm = np.array([orc.measure_to_map(p, pose, z) for z in zs])
    scene = rio.serialize_scene(pose, p.measurer, lm)
    commands = ["0.002 0 0.003 0 0.001 0" + (" 1" if k == 0 else "") for k in range(40)]
    cfg = rio.config_from_descriptor(["MinEffectiveParticle: 0.3", "ClutterDensity: 3E-07"])
    rec = simulate.simulate(scene, commands, cfg, particles=16, seed=4)
    assert set(rec) == set(rio.MEMBERS)
    traj = rio.timed_array_from_descriptor(rec["trajectory.out"].split("\n"), 7)
    odo = rio.timed_array_from_descriptor(rec["odometry.out"].split("\n"), 6)
    z = rio.measurements_from_descriptor(rec["measurements.out"], 3)
    maps = rio.map_history_from_descriptor(rec["maps.out"])
    assert len(traj) == 41 and len(odo) == 40 and len(z) == 40 and len(maps) == 40
    assert rio.timed_message_from_descriptor(rec["tags.out"].split("\n")) == [(0.0, "SLAM mode on")]
    assert rio.config_from_descriptor(rec["config.cfg"].splitlines())["MinEffectiveParticle"] == 0.3
    assert np.allclose(odo[5][1], [0.002, 0, 0.003, 0, 0.001, 0], atol=5e-3)        # the command plus odometry noise
    w, m, c = maps[-1][1]
    est, _ = orc.best_map_estimate((w, m, c))
    def near(pts, zh):   # some measurement within a few sigma of where the landmark should appear
        return len(pts) > 0 and np.min(np.linalg.norm((pts - zh) / [6.0, 6.0, 0.15], axis=1)) < 1
    seen = [x for x in lm if any(near(pts, orc.measure_perfect(p, traj[k + 1][1], x)) for k, (_, pts) in enumerate(z))]
    d, card = orc.ospa(est, np.array(seen))
    assert len(seen) >= 8 and card == 0 and d < 0.15, (d, card, len(est), len(seen))   # every landmark once; the rest is the drift of a 16-particle SLAM run
    path = str(tmp_path / "run.zip")
    rio.write_record(path, rec)
    out = replay.replay(rio.read_record(path), 16, 5, replay.DeviceSolver)
    assert len(rio.map_history_from_descriptor(out["maps.out"])) == 40
