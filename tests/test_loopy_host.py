"""Host logic of monorfs_amd/loopy.py (LoopyPHDNavigator's pose searches) on the CPU: geometry helpers
against the reference's own test properties, the batched gradient ascent against a literal one-at-a-time restatement of
LoopyPHDNavigator.LogLikeGradientAscent (:916-965), both over an oracle-backed evaluator (tests/loopy_stub.py)."""
import numpy as np
import pytest

import orc
from loopy_stub import OracleNav, scene
from monorfs_amd import loopy
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.navigator import pose3d_add


def test_vector_rotator_quaterniontest():
    """QuaternionTest.VectorRotator / VectorRotatorNone (QuaternionTest.cs:100-130)"""
    unit = lambda v: np.array(v, float) / np.linalg.norm(v)
    for a, b in ((unit([1, 2.3, 3]), unit([4.8, 3, 2])), (unit([1, 2.3, 3]), unit([1, 2.3, 3]))):
        q = loopy.vector_rotator(a, b)
        assert np.allclose(loopy._qmatrix(q) @ a, b, atol=1e-5)


def test_pose3d_subtract_inverts_add():
    """Pose3DTest's round trips (Pose3DTest.cs:65-120): (a + d) - a = d, and against the oracle's Pose3D.Add"""
    rng = np.random.default_rng(4)
    for _ in range(20):
        a = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1, 4)])
        a[3:] /= np.linalg.norm(a[3:])
        d = rng.normal(0, 0.3, 6)
        b = pose3d_add(a, d)
        assert np.allclose(b, orc.pose_add(a, d), atol=1e-14)
        assert np.allclose(loopy.pose3d_subtract(b, a), d, atol=1e-12)


def test_fit_to_measurement_explains_the_pair():
    """PRM3DMeasurer.FitToMeasurement (:224-244): from the fitted pose the landmark is measured as given"""
    rng = np.random.default_rng(5)
    p = prm3d_defaults(4, 600, 8)
    for _ in range(20):
        pose0 = np.concatenate([rng.normal(0, 0.1, 3), [1, 0, 0, 0] + rng.normal(0, 0.05, 4)])
        pose0[3:] /= np.linalg.norm(pose0[3:])
        lm = np.array([rng.uniform(-0.4, 0.4), rng.uniform(-0.3, 0.3), rng.uniform(0.8, 1.5)])
        z = np.array([rng.uniform(-200, 200), rng.uniform(-150, 150), rng.uniform(0.6, 1.6)])
        fit = loopy.fit_to_measurement(p, pose0, z, lm)
        assert np.allclose(orc.measure_perfect(p, fit, lm), z, rtol=1e-9, atol=1e-9)


def test_best_map_estimate_matches_oracle():
    rng = np.random.default_rng(6)
    for _ in range(10):
        n = int(rng.integers(1, 12))
        w = rng.uniform(0.1, 2.6, n)
        m = rng.normal(0, 1, (n, 3))
        c = np.broadcast_to(np.eye(3), (n, 3, 3)).copy()
        want, _ = orc.best_map_estimate((w, m, c))
        assert np.array_equal(loopy.best_map_estimate((w, m, c)), want)


def literal_ascent(nav, initial, z, lm, lin, mode):
    """LoopyPHDNavigator.LogLikeGradientAscent (:916-965), statement by statement, one evaluation per call"""
    pose = np.array(initial, float)
    nextpose7 = pose3d_add(lin, pose)
    prevvalue = -np.inf
    loglike = nav.QuasiSetLogLikelihoodGradient(z, lm, [nextpose7], mode)[0][0]
    while loglike - prevvalue > 1e-3:
        gradient = nav.QuasiSetLogLikelihoodGradient(z, lm, [nextpose7], mode)[1][0]
        gradsize = np.linalg.norm(gradient)
        if gradsize > 10:
            gradient = gradient * (10 / gradsize)
        multiplier = 1e-2
        counter = 0
        while True:
            nextpose = pose + multiplier * gradient
            nextpose7 = pose3d_add(lin, nextpose)
            nextloglike = nav.QuasiSetLogLikelihood(z, lm, [nextpose7])[0]
            multiplier /= 2.0
            counter += 1
            if not (nextloglike < loglike and counter < 16):
                break
        prevvalue = loglike
        if nextloglike > loglike:
            pose = nextpose
            loglike = nextloglike
    return pose, loglike


@pytest.mark.parametrize("mode", [0, 1])
def test_batched_ascent_is_the_literal_loop(mode):
    rng = np.random.default_rng(7 + mode)
    p = prm3d_defaults(64, 600, 16)
    lin, lm, z = scene(rng, p, 8, 6)
    nav = OracleNav(p)
    starts = rng.normal(0, 1, (4, 6)) * [4e-3, 4e-3, 4e-3, 2e-3, 2e-3, 2e-3]
    poses, values = loopy.LogLikeGradientAscent(nav, starts, z, lm, lin, mode)
    batched_calls = nav.calls
    nav.calls = 0
    for a in range(4):
        wp, wv = literal_ascent(nav, starts[a], z, lm, lin, mode)
        assert np.array_equal(poses[a], wp) and values[a] == wv
        assert values[a] >= orc.quasi_set_log_likelihood(p, pose3d_add(lin, starts[a]), lm, z)
    assert batched_calls < nav.calls / 4          # the point of the batch: far fewer round trips
    one, onev = loopy.LogLikeGradientAscent(nav, starts[0], z, lm, lin, mode)
    assert np.array_equal(one, poses[0]) and onev == values[0]


def test_fit_covariance_and_guided_mixture():
    rng = np.random.default_rng(11)
    p = prm3d_defaults(256, 600, 16)
    lin, lm, z = scene(rng, p, 6, 5, sigma=0.3)
    nav = OracleNav(p)
    cov = loopy.LogLikeFitCovariance(nav, np.zeros(6), z, lm, lin, average_mode=1)
    assert cov.shape == (6, 6) and np.allclose(cov, cov.T, atol=1e-12)
    assert np.all(np.linalg.eigvalsh(cov) >= -1e-12)               # pinv of a negative semi-definite Hessian, negated
    model = (np.ones(len(lm)) * 1.0001, lm, np.broadcast_to(1e-4 * np.eye(3), (len(lm), 3, 3)))
    empty, comps = loopy.GuidedFitMixture(nav, np.zeros(6), z, model, lin, average_mode=1)
    far = pose3d_add(np.array([0, 0, 0, 1.0, 0, 0, 0]), np.full(6, 1e5))
    assert empty == orc.quasi_set_log_likelihood(p, far, lm, z)     # nothing within 12 sigma: all clutter
    assert len(comps) >= 1
    for weight, mean, c in comps:
        assert np.isfinite(weight) and weight > 0 and mean.shape == (6,) and c.shape == (6, 6)
        assert orc.quasi_set_log_likelihood(p, pose3d_add(lin, mean), lm, z) > empty
    for i in range(len(comps)):                                     # no two components within Mahalanobis 0.1
        for k in range(i):
            d = comps[k][1] - comps[i][1]
            assert np.sqrt(d @ np.linalg.pinv(comps[k][2]) @ d) >= 0.1


def test_pose3d_odometry_against_the_oracle():
    """monorfs_amd/pose3d.py (used by scripts/simulate.py): Pose3D.AddOdometry / DiffOdometry (Pose3D.cs:314-356) against
    the oracle's restatement and Pose3DTest's round trip"""
    from monorfs_amd.pose3d import add_odometry, diff_odometry
    rng = np.random.default_rng(12)
    for _ in range(20):
        a = np.concatenate([rng.normal(0, 1, 3), rng.normal(0, 1, 4)])
        a[3:] /= np.linalg.norm(a[3:])
        d = rng.normal(0, 0.3, 6)
        b = add_odometry(a, d)
        assert np.allclose(b, orc.add_odometry(a, d), atol=1e-14)
        assert np.allclose(diff_odometry(b, a), orc.diff_odometry(b, a), atol=1e-13)
        assert np.allclose(diff_odometry(b, a), d, atol=1e-12)
