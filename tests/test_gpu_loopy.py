"""SURVEY row f4 on the device: the smoother's pose searches (monorfs_amd/loopy.py) driven by the HIP batches against
the same host code driven by the oracle (tests/loopy_stub.py), and Filter / FilterMissing against the oracle's filter."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from loopy_stub import OracleNav, scene
from monorfs_amd import loopy
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.navigator import pose3d_add
from monorfs_amd.synth import Frame


@pytest.fixture(scope="module")
def nav():
    from monorfs_amd import navigator
    p = prm3d_defaults(max_particles=256, max_components=600, max_measurements=16)
    n = navigator.PHDNavigator(p, particlecount=1)
    yield n
    n.close()


@pytest.mark.parametrize("mode", [0, 1])
def test_gradient_ascent_device_against_oracle(nav, mode):
    rng = np.random.default_rng(21 + mode)
    lin, lm, z = scene(rng, nav.params, 10, 7)
    starts = rng.normal(0, 1, (6, 6)) * [4e-3, 4e-3, 4e-3, 2e-3, 2e-3, 2e-3]
    got, gotv = loopy.LogLikeGradientAscent(nav, starts, z, lm, lin, mode)
    want, wantv = loopy.LogLikeGradientAscent(OracleNav(nav.params), starts, z, lm, lin, mode)
    assert np.allclose(gotv, wantv, rtol=1e-9, atol=1e-8), np.max(np.abs(gotv - wantv))
    assert np.allclose(got, want, rtol=0, atol=1e-9), np.max(np.abs(got - want))
    assert np.any(np.abs(got - starts) > 1e-5)                      # some estimate did climb


def test_fit_gaussian_and_guided_mixture_device_against_oracle(nav):
    rng = np.random.default_rng(31)
    lin, lm, z = scene(rng, nav.params, 6, 5, sigma=0.3)
    stub = OracleNav(nav.params)
    mean, cov = loopy.FitGaussian(nav, np.zeros(6), z, lm, lin, 1)
    wmean, wcov = loopy.FitGaussian(stub, np.zeros(6), z, lm, lin, 1)
    assert np.allclose(mean, wmean, atol=1e-9)
    assert np.allclose(cov, wcov, rtol=1e-4, atol=1e-12)            # pinv of a finite-difference Hessian (eps 1e-5)
    model = (np.full(len(lm), 1.0001), lm, np.broadcast_to(1e-4 * np.eye(3), (len(lm), 3, 3)))
    empty, comps = loopy.GuidedFitMixture(nav, np.zeros(6), z, model, lin, 1)
    wempty, wcomps = loopy.GuidedFitMixture(stub, np.zeros(6), z, model, lin, 1)
    assert np.isclose(empty, wempty, rtol=1e-12)
    assert len(comps) == len(wcomps) >= 1
    for (w, m, c), (ww, wm, wc) in zip(comps, wcomps):
        assert np.allclose(m, wm, atol=1e-9) and np.allclose(c, wc, rtol=1e-4, atol=1e-12)
        assert np.isclose(w, ww, rtol=1e-3)


def test_filter_missing_against_oracle(nav):
    """LoopyPHDNavigator.FilterMissing (:729-762): one-particle mapping-only filter over a trajectory with one factor
    left out, against the oracle's SlamUpdate run over the same frames"""
    rng = np.random.default_rng(41)
    f = Frame(1, 40, 12, 41, weight_profile="steady")
    T = 6
    trajectory, factors = [], []
    for t in range(T):
        pose = pose3d_add(f.poses[0], rng.normal(0, 1, 6) * [3e-3, 3e-3, 3e-3, 1e-3, 1e-3, 1e-3])
        zt = f.z + rng.normal(0, 1, f.z.shape) * [0.5, 0.5, 0.01]
        trajectory.append((0.1 * t, pose))
        factors.append((0.1 * t, zt))
    for index, to in ((2, T), (T, T), (-1, 4)):
        got = loopy.FilterMissing(nav, trajectory, factors, index, to)
        st = orc.State(1, 900)
        to_ = min(T, to)
        idx = to_ if index < 0 else min(to_, index)
        for i in list(range(idx)) + list(range(idx + 1, to_)):
            st.poses[0] = trajectory[i][1]
            orc.slam_update(nav.params, st, factors[i][1], onlymapping=True)
        w, m, c = st.map(0)
        gw, gm, gc = got
        assert len(gw) == len(w) > 0
        o, og = np.argsort(-w, kind="stable"), np.argsort(-gw, kind="stable")
        assert np.allclose(gw[og], w[o], rtol=1e-7) and np.allclose(gm[og], m[o], rtol=1e-7, atol=1e-10)
        assert np.allclose(gc[og], c[o], rtol=1e-6, atol=1e-12)
    assert np.array_equal(loopy.Filter(nav, trajectory, factors)[0], loopy.FilterMissing(nav, trajectory, factors, T, T)[0])


def test_cpp_mirror_matches_python(nav, tmp_path):
    """monorfs_amd/host/Loopy.hpp (tests/loopy_check.cpp) against monorfs_amd/loopy.py on the same scene: the same device
    batches behind the same host arithmetic"""
    import os
    import subprocess
    from monorfs_amd import _lib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    so = _lib.build()
    exe = str(tmp_path / "loopy_check")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(root, "tests", "loopy_check.cpp"), so,
                           "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    rng = np.random.default_rng(51)
    lin, lm, z = scene(rng, nav.params, 9, 6)
    lin = pose3d_add(lin, [0.01, -0.02, 0.005, 0.02, 0.01, -0.03])
    starts = rng.normal(0, 1, (5, 6)) * [4e-3, 4e-3, 4e-3, 2e-3, 2e-3, 2e-3]
    f = Frame(1, 30, 10, 52, weight_profile="steady")
    T = 4
    trajectory = [(0.1 * t, pose3d_add(f.poses[0], rng.normal(0, 1, 6) * 2e-3)) for t in range(T)]
    factors = [(0.1 * t, f.z + rng.normal(0, 1, f.z.shape) * [0.5, 0.5, 0.01]) for t in range(T)]
    path = tmp_path / "scene.txt"
    with open(path, "w") as fh:
        num = lambda a: " ".join("%.17g" % v for v in np.asarray(a, float).ravel())
        fh.write("%d %d %d %d\n%s\n%s\n%s\n%s\n" % (len(lm), len(z), len(starts), T, num(lm), num(z), num(lin), num(starts)))
        for (t, pose), (_, zt) in zip(trajectory, factors):
            fh.write("%.17g %s %d %s\n" % (t, num(pose), len(zt), num(zt)))
    for mode in (0, 1):
        r = subprocess.run([exe, str(path), str(mode)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        rows = [ln.split() for ln in r.stdout.splitlines()]
        asc = np.array([[float(v) for v in row[1:]] for row in rows if row[0] == "ascent"])
        poses, values = loopy.LogLikeGradientAscent(nav, starts, z, lm, lin, mode)
        assert np.allclose(asc[:, 0], values, rtol=1e-12, atol=1e-12) and np.allclose(asc[:, 1:], poses, rtol=0, atol=1e-13)
        grad = np.array([float(v) for row in rows if row[0] == "gradient" for v in row[1:]])
        assert np.allclose(grad, nav.LogLikeGradient(starts[0], z, lm, lin), rtol=1e-6, atol=1e-3)
        rt = np.array([float(v) for row in rows if row[0] == "roundtrip" for v in row[1:]])
        assert np.allclose(rt, starts[0], atol=1e-12)
        if mode == 1:   # the covariance fit and the guided mixture (printed for average_mode 1)
            cov = np.array([float(v) for row in rows if row[0] == "covariance" for v in row[1:]]).reshape(6, 6)
            wcov = loopy.LogLikeFitCovariance(nav, starts[0], z, lm, lin, 1)
            assert np.allclose(cov, wcov, rtol=1e-6, atol=1e-12 * np.abs(wcov).max()), np.max(np.abs(cov - wcov))
            model = (np.full(len(lm), 1.0001), lm, np.broadcast_to(1e-4 * np.eye(3), (len(lm), 3, 3)))
            wempty, wmix = loopy.GuidedFitMixture(nav, starts[0], z, model, lin, 1)
            empty = [float(row[1]) for row in rows if row[0] == "emptyspace"][0]
            mix = np.array([[float(v) for v in row[1:]] for row in rows if row[0] == "mixture"])
            assert np.isclose(empty, wempty, rtol=1e-12)
            assert len(mix) == len(wmix) >= 1
            for got, (ww, wm, _) in zip(mix, wmix):
                assert np.allclose(got[1:], wm, atol=1e-12) and np.isclose(got[0], ww, rtol=1e-5)
        comp = np.array([[float(v) for v in row[1:]] for row in rows if row[0] == "component"])
        w, m, c = loopy.FilterMissing(nav, trajectory, factors, 1, T)
        assert len(comp) == len(w) > 0
        assert np.array_equal(comp[:, 0], w) and np.array_equal(comp[:, 1:4], m) and np.array_equal(comp[:, 4:], c.reshape(-1, 9))
