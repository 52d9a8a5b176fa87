"""the reference's record / replay wire formats. The C++ host header is checked by a small program
(tests/recordio_check.cpp), the Python mirror against the same hand-written members of tests/golden/record/."""
import os
import subprocess

import numpy as np
import pytest

from monorfs_amd import recordio as rio

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REC = os.path.join(ROOT, "tests", "golden", "record")


def test_cpp_header_parses_and_rewrites_the_record(tmp_path):
    exe = str(tmp_path / "recordio_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                           os.path.join(ROOT, "tests", "recordio_check.cpp")])
    r = subprocess.run([exe, REC], capture_output=True, text=True)
    assert r.returncode == 0 and "recordio ok" in r.stdout, r.stdout + r.stderr


def test_python_mirror_round_trips_every_member(tmp_path):
    rec = rio.read_record(REC)
    pose, params, lm = rio.scene_from_descriptor(rec["scene.world"])
    assert pose[0] == 0.1 and params[0] == 575.816 and lm.shape == (3, 3) and lm[2, 1] == -0.4
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
def test_replay_of_a_record_matches_the_oracle(tmp_path):
    """the same measurements.out / odometry.out stream through the HIP solver and through the oracle: the maps.out /
    estimate.out they write agree number by number (and almost always character by character at g6)"""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import replay
    path = replay.make_synthetic_record(str(tmp_path / "rec.zip"), frames=10)
    rec = rio.read_record(path)
    dev = replay.replay(rec, 24, 5, replay.DeviceSolver)
    ref = replay.replay(rec, 24, 5, replay.OracleSolver)
    dmaps, rmaps = rio.map_history_from_descriptor(dev["maps.out"]), rio.map_history_from_descriptor(ref["maps.out"])
    assert len(dmaps) == len(rmaps) == 10
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
