"""SURVEY row f3, product side: monorfs_amd/host/Ospa.hpp (its own assignment solver) against the oracle's restatement
of Plot.OSPA (postanalysis/Plot.cs:531-581) on random landmark sets, including the cut-off and unequal sizes."""
import os
import subprocess

import numpy as np

import orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path):
    exe = str(tmp_path / "ospa_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "ospa_check.cpp")])
    return exe


def test_host_ospa_matches_the_oracle(tmp_path):
    exe = build(tmp_path)
    rng = np.random.default_rng(17)
    cases, text = [], []
    for t in range(40):
        na, nb = int(rng.integers(0, 14)), int(rng.integers(0, 14))
        C, P = (1.0, 1.0) if t % 3 else (float(rng.uniform(0.3, 2.0)), float(rng.choice([1.0, 2.0])))
        a = rng.uniform(-1.5, 1.5, (na, 3))
        b = (a[rng.permutation(na)][:nb] + rng.normal(0, 0.05, (min(na, nb), 3))) if t % 2 and na else rng.uniform(-1.5, 1.5, (nb, 3))
        b = np.vstack([b, rng.uniform(-1.5, 1.5, (nb - len(b), 3))]) if len(b) < nb else b
        cases.append((a, b, C, P))
        text.append("%.17g %.17g %d %d\n" % (C, P, len(a), len(b)) + "".join("%.17g %.17g %.17g\n" % tuple(x) for x in np.vstack([a, b])))
    out = subprocess.run([exe], input="".join(text), capture_output=True, text=True, check=True).stdout.split("\n")
    for (a, b, C, P), line in zip(cases, out):
        d, card = (float(v) for v in line.split())
        wd, wcard = orc.ospa(a, b, C, P)
        assert np.isclose(d, wd, rtol=1e-12, atol=1e-12) and np.isclose(card, wcard, rtol=1e-12, atol=1e-15), (len(a), len(b), d, wd)


def test_host_map_error_matches_the_oracle(tmp_path):
    """Plot.MapError (postanalysis/Plot.cs:478-529): the estimate aligned by the pose error at the reference time, OSPA and
    its spatial part; Ospa.hpp against the oracle's restatement, and two properties of the alignment"""
    from monorfs_amd.navigator import pose3d_add
    exe = build(tmp_path)
    rng = np.random.default_rng(23)
    cases, text = [], []
    for t in range(30):
        nv, ne = int(rng.integers(1, 12)), int(rng.integers(0, 12))
        C, P = (1.0, 1.0) if t % 3 else (float(rng.uniform(0.5, 2.0)), 2.0)
        visited = rng.uniform(-1.5, 1.5, (nv, 3))
        true = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1, 4)])
        true[3:] /= np.linalg.norm(true[3:])
        est = pose3d_add(true, rng.normal(0, 1, 6) * [0.05, 0.05, 0.05, 0.03, 0.03, 0.03])
        estimate = np.vstack([visited[:min(nv, ne)] + rng.normal(0, 0.03, (min(nv, ne), 3)), rng.uniform(-1.5, 1.5, (max(0, ne - nv), 3))])
        has = t % 5 != 0
        cases.append((visited, estimate, est if has else None, true if has else None, C, P))
        text.append("%.17g %.17g %d %d %d\n" % (C, P, nv, ne, has) + " ".join("%.17g" % v for v in np.concatenate([est, true])) + "\n"
                    + "".join("%.17g %.17g %.17g\n" % tuple(x) for x in np.vstack([visited, estimate])))
    out = subprocess.run([exe, "maperror"], input="".join(text), capture_output=True, text=True, check=True).stdout.split("\n")
    for (visited, estimate, est, true, C, P), line in zip(cases, out):
        d, sp = (float(v) for v in line.split())
        wd, wsp = orc.map_error(visited, estimate, est, true, C, P)
        assert np.isclose(d, wd, rtol=1e-12, atol=1e-12) and np.isclose(sp, wsp, rtol=1e-9, atol=1e-9, equal_nan=True)
    # equal poses: no alignment, MapError is the plain OSPA; spatial part of equally sized sets is the OSPA itself
    visited = rng.uniform(-1, 1, (6, 3))
    estimate = visited + rng.normal(0, 0.02, (6, 3))
    pose = np.array([0.3, -0.2, 0.1, 1.0, 0, 0, 0])
    d, sp = orc.map_error(visited, estimate, pose, pose)
    assert np.isclose(d, orc.ospa(visited, estimate)[0], rtol=1e-12) and np.isclose(sp, d, rtol=1e-12)
    # a pure translation error of the estimate and its map is undone: the error drops to zero
    shift = np.array([0.2, -0.1, 0.05])
    d, _ = orc.map_error(visited, visited + shift, pose + np.concatenate([shift, np.zeros(4)]), pose)
    assert d < 1e-12


def test_host_visited_map(tmp_path):
    """Plot.VisitedMap (postanalysis/Plot.cs:230-248): every landmark seen with positive weight, once"""
    exe = build(tmp_path)
    lms = np.array([[0, 0, 1.0], [1, 0, 1], [0, 1, 1], [1, 1, 2]])
    frames = [[(0, 1.0), (1, 0.0)], [(1, 1.0), (0, 1.0)], [], [(3, 0.5), (2, 0.0), (1, 1.0)]]
    text = "%d\n" % len(frames) + "".join("%d\n" % len(f) + "".join("%.17g %.17g %.17g %.17g\n" % (*lms[i], w) for i, w in f) for f in frames)
    out = subprocess.run([exe, "visited"], input=text, capture_output=True, text=True, check=True).stdout.split()
    got = np.array([float(v) for v in out]).reshape(-1, 3)
    assert np.array_equal(got, lms[[0, 1, 3]])
