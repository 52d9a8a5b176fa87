"""The reference's own unit-test vectors (PHDNavigatorTest.cs:85-265, transcribed into tests/golden/phdnavigator_kat.json)
put to the DEVICE through the C-ABI: the Linear2D toy model of those tests runs through the same kernels as PRM3D
(include/phdhip.h, PHD_MODEL_LINEAR2D). Matching as in the reference (every expected component found once, 1e-5), plus
the oracle at the parity tolerances."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import PHD_GATE_DISABLED, params_from_dict
from test_oracle_kat import assert_same_set, mix_of

KAT = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "phdnavigator_kat.json")))


def device(params, pose2, model):
    from monorfs_amd import navigator
    p = params_from_dict(params, max_particles=1, max_components=600, max_measurements=8)
    nav = navigator.PHDNavigator(p, particlecount=1, pose=[pose2[0], pose2[1], 0, 1, 0, 0, 0])
    nav.reset(np.array([pose2[0], pose2[1], 0, 1.0, 0, 0, 0]), model, 1)
    return nav, p


def z3(measurements):
    z = np.asarray(measurements, float).reshape(-1, 2)
    return np.column_stack([z, np.zeros(len(z))])


@pytest.mark.parametrize("case", ["predict_initial", "predict_known"])
def test_predict_kats_on_the_device(case):
    """PHDNavigatorTest.PredictInitial / PredictKnown (:85-126)"""
    k = KAT[case]
    nav, p = device(KAT["params"], KAT["pose"], mix_of(k["model"]))
    nav.run_stages(z3(k["measurements"]), with_alpha=False)
    got = nav.PredictConditional(0)
    assert_same_set(got, k["expected"], KAT["tolerance"])
    want = orc.predict(p, KAT["pose"] + [0.0], k["measurements"], mix_of(k["model"]))
    assert len(got[0]) == len(want[0]) and np.allclose(got[0], want[0], rtol=1e-12) and np.allclose(got[1], want[1], rtol=1e-12, atol=1e-15)
    nav.close()


def test_correct_kat_on_the_device():
    """PHDNavigatorTest.Correct (:128-193): all four (measurement, component) pairs, i.e. no radius gate"""
    k = KAT["correct"]
    nav, p = device(dict(KAT["params"], gate_metric=PHD_GATE_DISABLED), KAT["pose"], mix_of(k["model"]))
    nav.run_stages(z3(k["measurements"]), with_alpha=False)
    got = nav.CorrectConditional(0)
    assert_same_set(got, k["expected"], KAT["tolerance"])
    want = orc.correct(p, KAT["pose"] + [0.0], k["measurements"], mix_of(k["model"]))
    keep = want[0] >= p.min_weight
    ow, gw = np.sort(want[0][keep]), np.sort(got[0])
    assert len(gw) == len(ow) and np.allclose(gw, ow, rtol=1e-9)
    nav.close()


def test_prune_kat_on_the_device():
    """PHDNavigatorTest.Prune (:195-265): 13 components -> 5. The device prunes what its correction step emits, so the 13
    are given as a prior seen from a pose far outside the visible square: every detection probability is 0, the
    correction passes them on unchanged (weight (1 - 0) w) and PruneModel gets exactly the test's input."""
    k = KAT["prune"]
    nav, p = device(KAT["params"], [1000.0, -1000.0], mix_of(k["model"]))
    nav.run_stages(np.zeros((0, 3)), with_alpha=False)
    got = nav.PruneModel(0)
    assert_same_set(got, k["expected"], KAT["tolerance"])
    want = orc.prune(p, mix_of(k["model"]))
    assert len(got[0]) == len(want[0]) and np.allclose(got[0], want[0], rtol=1e-9) and np.allclose(got[1], want[1], rtol=1e-9, atol=1e-12)
    assert np.allclose(got[2], want[2], rtol=1e-7, atol=1e-12)
    nav.close()


@pytest.mark.parametrize("maxq", [600, 300])
def test_prune_order_with_nearly_equal_weights(maxq):
    """PruneModel's order (weight descending, List.Sort made stable by the position in `corrected`, PHDNavigator.cs:917-925)
    when weights are equal, differ in the last bit, or differ below what a sort key of k_prune_merge holds (22 bits, then 32
    more in LDS, then the weights themselves): 560 far-apart components (no merge), passed through the correction step
    unchanged as in the test above; the cut at MaxQuantity falls inside a run of equal weights."""
    rng = np.random.default_rng(77)
    n = 560
    base = rng.uniform(0.01, 2.0, 40)
    w = np.empty(n)
    for i in range(n):
        b = base[i % 40]
        kind = (i // 40) % 4
        if kind == 0: w[i] = b                                   # exact ties with the others of the group
        elif kind == 1: w[i] = b * (1 + 3e-6 * (i // 160))       # same 22-bit key, other bits behind it
        elif kind == 2: w[i] = np.nextafter(b, 4.0)              # one ulp above
        else: w[i] = b * (1 + 1e-13 * (1 + i // 160))            # equal in the first 54 bits
    order = rng.permutation(n)
    w = w[order]
    gx, gy = np.meshgrid(np.arange(28) * 5.0, np.arange(20) * 5.0)
    means = np.column_stack([gx.ravel()[:n], gy.ravel()[:n], np.zeros(n)])
    covs = np.broadcast_to(np.diag([0.01, 0.01, 0.01]), (n, 3, 3)).copy()
    params = dict(KAT["params"], max_quantity=maxq)
    nav, p = device(params, [1000.0, -1000.0], (w, means, covs))
    nav.run_stages(np.zeros((0, 3)), with_alpha=False)
    got = nav.PruneModel(0)
    want = orc.prune(p, (w, means, covs))
    assert len(got[0]) == len(want[0]) == min(n, maxq)
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    nav.close()


def test_linear2d_step_against_the_oracle():
    """a whole SlamUpdate of the toy model (reweight and resampling included) on the device against the oracle"""
    rng = np.random.default_rng(61)
    P, C, M = 12, 10, 5
    params = dict(KAT["params"], min_effective_particle=0.5)
    from monorfs_amd import navigator
    p = params_from_dict(params, max_particles=P, max_components=600, max_measurements=8)
    lm = rng.uniform(-4, 4, (C, 2))
    nav = navigator.PHDNavigator(p, particlecount=P, pose=[0, 0, 0, 1, 0, 0, 0])
    st = orc.State(P, 700)
    poses = np.zeros((P, 7))
    poses[:, 3] = 1
    poses[:, :2] = rng.normal(0, 0.02, (P, 2))
    model = (rng.uniform(0.5, 1.1, C), np.column_stack([lm, np.zeros(C)]), np.broadcast_to(np.diag([2e-3, 2e-3, 1e-2]), (C, 3, 3)).copy())
    nav.reset(poses[0], model, P)
    nav.set_poses(poses)
    st.poses[:] = poses
    for i in range(P):
        st.w[i, :C], st.mean[i, :C], st.cov[i, :C], st.n[i] = model[0], model[1], model[2], C
    for step in range(3):
        z = lm[rng.permutation(C)[:M]] + rng.normal(0, 0.02, (M, 2))
        best, src, res, _ = orc.slam_update(p, st, z, u=0.37 + 0.1 * step, threads=2)
        nav.SlamUpdate(None, z3(z), u_resample=0.37 + 0.1 * step)
        gsrc, gres = nav.resample_sources()
        assert gres == res and np.array_equal(gsrc, src) and nav.BestParticle == best
        assert np.allclose(nav.VehicleWeights, st.weights, rtol=1e-6, atol=1e-300)
        for i in (0, P - 1):
            gw, gm, gc = nav.MapModel(i)
            ow, om, oc = st.map(i)
            assert len(gw) == len(ow) and np.allclose(gw, ow, rtol=1e-7) and np.allclose(gm, om, rtol=1e-7, atol=1e-10)
    nav.close()


def test_resample_kat_on_the_device():
    """SimulationTest.resample (SimulationTest.cs:225-270): weights {.11, .28, .31, .01, .29}: for every random number the
    best particle's source is 2, particles 1, 2, 4 always survive, 0 and 3 each die for some — put to phd_resample, and
    every index vector against the oracle's sequential recurrence."""
    from monorfs_amd import navigator
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "resample_kat.json")))
    w = np.array(k["weights"])
    p = params_from_dict(KAT["params"], max_particles=5, max_components=600, max_measurements=8)
    nav = navigator.PHDNavigator(p, particlecount=5)
    missing = {s: 0 for s in k["sometimes_absent"]}
    us = np.concatenate([np.linspace(1e-9, 1 - 1e-9, 1500), np.random.default_rng(7).random(500)])
    for u in us:
        src, best = nav.ResampleParticles(w, float(u))
        osrc, obest = orc.resample(w, float(u))
        assert np.array_equal(src, osrc) and best == obest
        assert src[best] == k["best_source"] and np.all(np.diff(src) >= 0)
        for s in k["always_present"]:
            assert s in src
        for s in missing:
            missing[s] += s not in src
    assert all(cnt > 0 for cnt in missing.values())
    nav.close()


# ---------------------------------------------------------------- GraphCombinatoricsTest.cs on the device
GC = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "graphcombinatorics_kat.json")))


@pytest.fixture(scope="module")
def pairing_nav():
    from monorfs_amd import navigator
    p = params_from_dict(KAT["params"], max_particles=1, max_components=600, max_measurements=8)
    nav = navigator.PHDNavigator(p, particlecount=1)
    yield nav
    nav.close()


def _mat(m):
    return np.array([[float(x) for x in row] for row in m], float)


@pytest.mark.parametrize("case", GC["linear_assignment"], ids=lambda c: c["name"])
def test_hungarian_kats_on_the_device(pairing_nav, case):
    """GraphCombinatoricsTest.LinearAssignment* (:201-255): the optimal assignment = the first pairing of the device's Murty"""
    asg, val = pairing_nav.test_pairing(_mat(case["matrix"]), maxcount=1)
    assert asg[0] == case["expected"]
    assert val[0] == orc.assignment_value(_mat(case["matrix"]), case["expected"])


@pytest.mark.parametrize("case", GC["murty_pairing"], ids=["full_small", "unique"])
def test_murty_order_kats_on_the_device(pairing_nav, case):
    """GraphCombinatoricsTest.MurtyPairing / MurtyPairingUnique (:358-404)"""
    asg, val = pairing_nav.test_pairing(_mat(case["matrix"]))
    assert asg == case["expected"] and np.all(np.diff(val) <= 0)
    oasg, oval = orc.murty(_mat(case["matrix"]))
    assert asg == oasg and np.array_equal(val, oval)


@pytest.mark.parametrize("case", GC["lexicographical"], ids=lambda c: "modelsize%d" % c["modelsize"])
def test_lexicographic_kats_on_the_device(pairing_nav, case):
    """GraphCombinatoricsTest.LexicographicalPairing* (:258-306)"""
    asg, val = pairing_nav.test_pairing(_mat(case["matrix"]), lexicographic=True, modelsize=case["modelsize"])
    assert asg == case["expected"]
    _, oval = orc.lexicographic(_mat(case["matrix"]), case["modelsize"])
    assert np.array_equal(val, oval)


def test_murty_on_a_larger_random_matrix(pairing_nav):
    """beyond the reference's vectors: 9 x 9 with missing entries, the first 200 pairings in order against the oracle"""
    rng = np.random.default_rng(71)
    m = rng.uniform(-20, 0, (9, 9))
    m[rng.uniform(size=m.shape) < 0.3] = -np.inf
    m[np.arange(9), np.arange(9)] = rng.uniform(-5, 0, 9)       # a finite diagonal: solvable
    asg, val = pairing_nav.test_pairing(m)
    oasg, oval = orc.murty(m, maxcount=200)
    assert len(asg) == len(oasg) == 200 and asg == oasg and np.allclose(val, oval, rtol=0, atol=1e-12)


# ---------------------------------------------------------------- LoopyPHDNavigatorTest.LogLike2D on the device
def test_loglike2d_on_the_device():
    """LoopyPHDNavigatorTest.LogLike2D (:351-419), the reference's test of QuasiSetLogLikelihood and its gradient: Linear2D,
    R = 5e-2 I, two measurements, three landmarks, the pose on a 201 x 201 grid over [-1, 1]^2; at every inner grid point
    central differences of the value must equal the analytic gradient within 0.5. The 40 401 poses go to the device in
    batches. With TemperedAverage's weights divided by their sum (average_mode 1) the assertion holds everywhere; as the
    source reads (mode 0) it does not — the same finding as on the oracle (tests/test_oracle_kat.py), to which the
    device is compared point by point."""
    from monorfs_amd import navigator
    params = dict(KAT["params"], R=[[5e-2, 0], [0, 5e-2]])
    p = params_from_dict(params, max_particles=2048, max_components=600, max_measurements=8)
    nav = navigator.PHDNavigator(p, particlecount=1)
    z = np.array([[0, 1, 0], [0.2, 0.6, 0]], float)
    lm = np.array([[0, 1.45, 0], [0, 0.65, 0], [1.0, 0, 0]])
    n = 201
    x = np.array([((i / (n - 1)) - 0.5) / 0.5 for i in range(n)])
    poses = np.zeros((n * n, 7))
    poses[:, 0], poses[:, 1], poses[:, 3] = np.repeat(x, n), np.tile(x, n), 1
    for mode in (1, 0):
        L, G = np.zeros(n * n), np.zeros((n * n, 6))
        for s in range(0, n * n, 2048):
            L[s:s + 2048], G[s:s + 2048] = nav.QuasiSetLogLikelihoodGradient(z, lm, poses[s:s + 2048], average_mode=mode)
        L, G = L.reshape(n, n), G.reshape(n, n, 6)
        assert np.all(G[:, :, 2:] == 0)
        xnum = (L[2:, 1:-1] - L[:-2, 1:-1]) / (x[2:] - x[:-2])[:, None]
        ynum = (L[1:-1, 2:] - L[1:-1, :-2]) / (x[2:] - x[:-2])[None, :]
        bad = (np.abs(xnum - G[1:-1, 1:-1, 0]) > 0.5) | (np.abs(ynum - G[1:-1, 1:-1, 1]) > 0.5)
        if mode == 1:
            assert bad.sum() == 0
        else:
            assert 0.15 * (n - 2) ** 2 < bad.sum() < 0.35 * (n - 2) ** 2
        for i, k in ((0, 0), (100, 100), (37, 151), (200, 3), (120, 80)):   # and the oracle at some points
            v, g = orc.quasi_set_log_likelihood_grad(p, [x[i], x[k], 0, 1, 0, 0, 0], lm[:, :3], z[:, :2], mode)
            assert np.isclose(L[i, k], v, rtol=1e-11, atol=1e-11) and np.allclose(G[i, k, :2], g, rtol=1e-9, atol=1e-9)
        assert np.array_equal(nav.QuasiSetLogLikelihood(z, lm, poses[:2048]), L.reshape(-1)[:2048])
    nav.close()
