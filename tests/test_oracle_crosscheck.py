"""The parts of the path no reference test pins numerically (WeightAlpha, BestMapEstimate,
SetLogLikelihood, PRM3D CorrectConditional, FuzzyVisibleM) are defined by the oracle's reading of the source. This
file holds a SECOND, independent reading in numpy — brute force where the oracle is clever (every permutation
instead of connected components + enumerators; literal list sorting instead of a merge) — and checks that both
readings agree on small random cases."""
import itertools

import numpy as np
import pytest

import orc
from monorfs_amd.abi import prm3d_defaults


# ---- PRM3DMeasurer.cs:138-177, 277-312; Quaternion.cs:155-158, 295-301 ------------------------------------
def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz,
                     aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def unit(pose7):
    q = np.asarray(pose7[3:], float)
    return q / np.linalg.norm(q)          # Pose3D normalises on construction (Pose3D.cs:157-161)


def measure_perfect(p, pose7, lm):
    f = p.measurer[0]
    q = unit(pose7)
    diff = np.asarray(lm, float) - np.asarray(pose7[:3], float)
    local = qmul(qmul(qconj(q), np.array([0.0, *diff])), q)[1:]
    rng = np.sign(local[2]) * np.linalg.norm(diff)
    return np.array([f * local[0] / local[2], f * local[1] / local[2], rng])


def rotation_matrix(q):
    return np.array([qmul(qmul(q, np.array([0.0, *e])), qconj(q))[1:] for e in np.eye(3)]).T


def jacobian_l(p, pose7, lm):
    f = p.measurer[0]
    q = unit(pose7)
    diff = np.asarray(lm, float) - np.asarray(pose7[:3], float)
    x, y, z = qmul(qmul(qconj(q), np.array([0.0, *diff])), q)[1:]
    mag = (1 if z > 0 else -1) * np.sqrt(x * x + y * y + z * z)
    jproj = np.array([[f / z, 0, -f * x / (z * z)], [0, f / z, -f * y / (z * z)], [x / mag, y / mag, z / mag]])
    return jproj @ rotation_matrix(qconj(q))


def measure_to_map(p, pose7, z):
    f = p.measurer[0]
    q = unit(pose7)
    alpha = z[2] / np.sqrt(f * f + z[0] * z[0] + z[1] * z[1])
    diff = np.array([alpha * z[0], alpha * z[1], alpha * f])
    return np.asarray(pose7[:3], float) + qmul(qmul(q, np.array([0.0, *diff])), qconj(q))[1:]


def detection_probability_m(p, z):
    """FuzzyVisibleM (PRM3DMeasurer.cs:277-291) * pd; integer film rectangle, float32 range clip."""
    left, top = int(p.measurer[3]), int(p.measurer[4])
    right, bottom = left + int(p.measurer[5]), top + int(p.measurer[6])
    rmin, rmax = float(np.float32(p.measurer[1])), float(np.float32(p.measurer[2]))
    ramp = p.visibility_ramp
    d = min((z[0] - left) / ramp[0], (right - z[0]) / ramp[0], (z[1] - top) / ramp[1], (bottom - z[1]) / ramp[1],
            (z[2] - rmin) / ramp[2], (rmax - z[2]) / ramp[2])
    return max(0.0, min(1.0, d)) * p.pd


# ---- Gaussian.cs:148-157, 199-204 ------------------------------------------------------------------------------
def multiplier(cov):
    dim = cov.shape[0]
    return (2 * np.pi) ** (-(dim // 2)) / np.sqrt(abs(np.linalg.det(cov)))   # `-mean.Length / 2` is an integer division


def gaussian(x, mean, cov):
    d = np.asarray(x, float) - mean
    return multiplier(cov) * np.exp(-0.5 * d @ np.linalg.inv(cov) @ d)


def mixture(x, mix):
    return sum(w * gaussian(x, m, c) for w, m, c in zip(*mix))


# ---- Map.cs:119-142 --------------------------------------------------------------------------------------------
def best_map_estimate(mix):
    w, m, _ = mix
    size = int(np.sum(w))
    lst = [(float(w[i]), i) for i in range(len(w))]
    lst.sort(key=lambda e: -e[0])                     # python's sort is stable: the canonical order of the build
    picks = []
    for i in range(size):
        wi, src = lst[i]
        picks.append(src)
        lst.append((wi - 1, src))
        lst.sort(key=lambda e: -e[0])
    return np.array([m[s] for s in picks]).reshape(-1, 3), picks


# ---- PHDNavigator.cs:415-515 by brute force --------------------------------------------------------------------
def set_log_likelihood_bruteforce(p, pose7, lm, z, quasi=False):
    J, M = len(lm), len(z)
    n = J + M
    R = np.array(p.R).reshape(3, 3)
    Rinv = np.linalg.inv(R)
    mat = np.full((n, n), -np.inf)
    pd = np.zeros(J)
    gated = np.zeros((J, M), bool)
    for j in range(J):
        zh = measure_perfect(p, pose7, lm[j])
        pd[j] = p.pd if quasi else detection_probability_m(p, zh)
        for k in range(M):
            d = np.sqrt((z[k] - zh) @ Rinv @ (z[k] - zh))
            if d < (12 if quasi else 5):
                gated[j, k] = True
                mat[j, k] = np.log(pd[j]) + np.log(multiplier(R)) - 0.5 * d * d
        mat[j, M + j] = np.log(1 - pd[j])
    for k in range(M):
        mat[J + k, k] = np.log(p.clutter_density)
    # connected components of the detection graph; the (clutter row, misdetection column) quadrant is zero inside one
    label = list(range(n))           # landmarks 0..J-1, measurements J..J+M-1
    changed = True
    while changed:
        changed = False
        for j in range(J):
            for k in range(M):
                if gated[j, k] and label[j] != label[J + k]:
                    label[j] = label[J + k] = min(label[j], label[J + k])
                    changed = True
    for j in range(J):
        for k in range(M):
            if label[j] == label[J + k]:
                mat[J + k, M + j] = 0.0
    sizes = {}
    for v in label:
        sizes[v] = sizes.get(v, 0) + 1
    assert max(sizes.values()) <= 5, "case outside the exact-enumeration regime"
    # every component has <= 5 rows, so the reference enumerates all of its pairings: the sum over the components
    # of their log-sum-exp is the log of the sum over ALL permutations of the full matrix
    with np.errstate(divide="ignore"):
        vals = [sum(mat[i, perm[i]] for i in range(n)) for perm in itertools.permutations(range(n))]
    vals = np.array(vals)
    mx = vals.max()
    return mx + np.log(np.sum(np.exp(vals - mx)))


def random_case(rng, p, J, M):
    """landmarks inside the field of view of a random pose, measurements near some of them + clutter"""
    pose = np.concatenate([rng.normal(0, 0.05, 3), [1.0, 0.0, 0.0, 0.0] + rng.normal(0, 0.03, 4)])
    zs = np.column_stack([rng.uniform(-250, 250, J), rng.uniform(-180, 180, J), rng.uniform(0.4, 1.7, J)])
    lm = np.array([measure_to_map(p, pose, zz) for zz in zs])
    z = []
    for k in range(M):
        if k < J and rng.uniform() < 0.8:
            z.append(zs[k] + rng.normal(0, 1, 3) * np.sqrt(np.diag(np.array(p.R).reshape(3, 3))) * 1.5)
        else:
            z.append([rng.uniform(-300, 300), rng.uniform(-220, 220), rng.uniform(0.3, 1.8)])
    return pose, lm, np.array(z)


@pytest.mark.parametrize("seed", range(12))
def test_set_log_likelihood_against_all_permutations(seed):
    rng = np.random.default_rng(100 + seed)
    p = prm3d_defaults(4, 600, 8)
    J = int(rng.integers(1, 4))
    M = int(rng.integers(1, 6 - J))          # J + M <= 5: 120 permutations at most
    pose, lm, z = random_case(rng, p, J, M)
    if seed % 3 == 0 and J >= 2:            # two landmarks sharing a measurement: a cluster with 3 rows
        lm[1] = lm[0] + rng.normal(0, 1e-3, 3)
    want = set_log_likelihood_bruteforce(p, pose, lm, z)
    got = orc.set_log_likelihood(p, pose, lm, z)[0]
    assert np.isclose(got, want, rtol=1e-10, atol=1e-10), (got, want)


@pytest.mark.parametrize("seed", range(8))
def test_quasi_set_log_likelihood_against_all_permutations(seed):
    """QuasiSetLogLikelihood (PHDNavigator.cs:526-713, value): constant PD, gate 12; landmarks near the border of the
    field of view, where the plain set log-likelihood would fade them out, count fully"""
    rng = np.random.default_rng(500 + seed)
    p = prm3d_defaults(4, 600, 8)
    J = int(rng.integers(1, 4))
    M = int(rng.integers(1, 6 - J))
    pose, lm, z = random_case(rng, p, J, M)
    if seed % 2:                               # a measurement 8 sigma off: inside the gate of 12, outside the gate of 5
        z[0] = measure_perfect(p, pose, lm[0]) + 8 * np.sqrt(np.diag(np.array(p.R).reshape(3, 3))) * [1, 0, 0]
    want = set_log_likelihood_bruteforce(p, pose, lm, z, quasi=True)
    got = orc.quasi_set_log_likelihood(p, pose, lm, z)
    assert np.isclose(got, want, rtol=1e-10, atol=1e-10), (got, want)


def test_measurement_model_against_second_reading():
    rng = np.random.default_rng(5)
    p = prm3d_defaults(4, 600, 8)
    for _ in range(20):
        pose, lm, _ = random_case(rng, p, 3, 1)
        for x in lm:
            assert np.allclose(orc.measure_perfect(p, pose, x), measure_perfect(p, pose, x), rtol=1e-12, atol=1e-12)
            assert np.allclose(orc.jacobian_l(p, pose, x).reshape(3, 3), jacobian_l(p, pose, x), rtol=1e-11, atol=1e-11)
            assert np.isclose(orc.detection_probability(p, pose, x), detection_probability_m(p, measure_perfect(p, pose, x)),
                              rtol=1e-12, atol=1e-15)


def random_mixture(rng, means, wlo, whi):
    n = len(means)
    A = rng.uniform(-0.05, 0.05, (n, 3, 3))
    cov = A @ A.transpose(0, 2, 1) + 1e-4 * np.eye(3)
    return rng.uniform(wlo, whi, n), means + rng.normal(0, 1e-3, (n, 3)), cov


@pytest.mark.parametrize("seed", range(6))
def test_best_map_estimate_literal_list(seed):
    rng = np.random.default_rng(200 + seed)
    n = int(rng.integers(1, 9))
    mix = random_mixture(rng, rng.uniform(-1, 1, (n, 3)), 0.05, 2.6)     # weights above 2: a component picked twice
    if seed == 0:
        mix[0][:] = 0.75                                                # all weights equal: ties keep list order
    want, wsrc = best_map_estimate(mix)
    got, gsrc = orc.best_map_estimate(mix)
    assert list(gsrc) == wsrc and np.array_equal(got, want)


@pytest.mark.parametrize("seed", range(6))
def test_weight_alpha_against_second_reading(seed):
    """alpha = exp( L(Z | J, x) + [sum_j log v_pred(m_j) - sum w_pred] - [sum_j log v_corr(m_j) - sum w_corr] )
    (PHDNavigator.cs:373-393) with the full, ungated mixture densities (Map.cs:192-202)."""
    rng = np.random.default_rng(300 + seed)
    p = prm3d_defaults(4, 600, 8)
    J = int(rng.integers(1, 4))
    M = int(rng.integers(1, 6 - J))
    pose, lm, z = random_case(rng, p, J, M)
    nextra = int(rng.integers(0, 3))
    pred = random_mixture(rng, np.vstack([lm, lm[:nextra] + 0.3]), 0.3, 0.9)
    corr = random_mixture(rng, np.vstack([lm, lm[:nextra] + 0.3]), 0.02, 0.2)
    corr[0][:J] = rng.uniform(1.0, 1.3, J)            # the J landmarks carry the map estimate
    jm, _ = best_map_estimate(corr)
    sll = set_log_likelihood_bruteforce(p, pose, jm, z) if len(jm) + M <= 5 else None
    if sll is None:
        pytest.skip("case outside the exact-enumeration regime")
    plog = sum(np.log(mixture(x, pred)) for x in jm)
    clog = sum(np.log(mixture(x, corr)) for x in jm)
    want = np.exp(sll + (plog - np.sum(pred[0])) - (clog - np.sum(corr[0])))
    got, gsll = orc.weight_alpha(p, pose, z, pred, corr)
    assert np.isclose(gsll, sll, rtol=1e-10, atol=1e-10)
    assert np.isclose(got, want, rtol=1e-9, atol=0), (got, want)


def numpy_correct(p, pose, z, pred, by_value):
    """CorrectConditional (PHDNavigator.cs:829-906) for the pixel-range model, one formula at a time in numpy.
    by_value: look a near component's arrays up through a dictionary keyed by the component's VALUE, as the reference's
    `qindex` (a Dictionary<Gaussian, int>, :854, :868, with Gaussian.Equals / GetHashCode comparing weight, mean and
    covariance exactly, Gaussian.cs:436-489) does: bit-identical components then share the index of the last of them."""
    R = np.array(p.R).reshape(3, 3)
    gate = p.density_distance_threshold               # squared-Euclidean metric by default (Map.Near, :882)
    comps = []
    zh, H, S, PD = [], [], [], []
    qindex = {}
    for n, (w, m, P) in enumerate(zip(*pred)):
        zh.append(measure_perfect(p, pose, m))
        H.append(jacobian_l(p, pose, m))
        S.append(H[-1] @ P @ H[-1].T + R)
        PD.append(detection_probability_m(p, zh[-1]))
        comps.append(((1 - PD[-1]) * w, m, P))                                        # :837-840
        qindex[(float(w), m.tobytes(), P.tobytes())] = n                              # :868
    for zk in z:
        x = measure_to_map(p, pose, zk)
        near = [n for n in range(len(pred[0])) if np.sum((x - pred[1][n]) ** 2) <= gate]
        if by_value:
            near = [qindex[(float(pred[0][n]), pred[1][n].tobytes(), pred[2][n].tobytes())] for n in near]   # int i = qindex[landmark]
        q = [multiplier(S[n]) * np.exp(-0.5 * (zk - zh[n]) @ np.linalg.inv(S[n]) @ (zk - zh[n])) for n in near]
        weightsum = sum(PD[n] * pred[0][n] * qn for n, qn in zip(near, q))            # :886-890
        for n, qn in zip(near, q):
            K = pred[2][n] @ H[n].T @ np.linalg.inv(S[n])
            comps.append((PD[n] * pred[0][n] * qn / (p.clutter_density + weightsum),  # :899
                          pred[1][n] + K @ (zk - zh[n]), (np.eye(3) - K @ H[n]) @ pred[2][n]))
    return comps


@pytest.mark.parametrize("seed", range(4))
def test_prm3d_correct_against_second_reading(seed):
    rng = np.random.default_rng(400 + seed)
    p = prm3d_defaults(4, 600, 8)
    pose, lm, z = random_case(rng, p, 3, 3)
    pred = random_mixture(rng, lm, 0.3, 1.1)
    comps = numpy_correct(p, pose, z, pred, by_value=False)
    ow, om, oc = orc.correct(p, pose, z, pred)
    assert len(ow) == len(comps)
    for i, (w, m, P) in enumerate(comps):
        assert np.isclose(ow[i], w, rtol=1e-9, atol=1e-300), (i, ow[i], w)
        assert np.allclose(om[i], m, rtol=1e-9, atol=1e-12)
        assert np.allclose(oc[i], P, rtol=1e-8, atol=1e-14)


@pytest.mark.parametrize("seed", range(3))
def test_qindex_by_value_aliases_only_equal_numbers(seed):
    """The reference finds a near component's cached arrays through `qindex`, a dictionary keyed by the Gaussian's VALUE
    (PHDNavigator.cs:854, :868; Gaussian.cs:436-489): two bit-identical components (two births from a measurement reported
    twice) both resolve to the LAST of them. Every cached array is a deterministic function of (weight, mean, covariance,
    pose), so the aliased index holds the very numbers the own index would: the corrected map is the same, entry for
    entry, as with the positional indices the oracle and the device use. Shown here on a mixture with a duplicated
    component and a duplicated measurement: by-value and positional readings agree exactly, and with the oracle."""
    rng = np.random.default_rng(450 + seed)
    p = prm3d_defaults(4, 600, 8)
    pose, lm, z = random_case(rng, p, 3, 3)
    w, m, P = random_mixture(rng, lm, 0.3, 1.1)
    pred = (np.concatenate([w, w[1:2]]), np.concatenate([m, m[1:2]]), np.concatenate([P, P[1:2]]))   # component 1 twice
    z = np.concatenate([z, z[:1]])                                                                  # measurement 0 twice
    a = numpy_correct(p, pose, z, pred, by_value=True)
    b = numpy_correct(p, pose, z, pred, by_value=False)
    assert len(a) == len(b)
    for (wa, ma, Pa), (wb, mb, Pb) in zip(a, b):
        assert wa == wb and np.array_equal(ma, mb) and np.array_equal(Pa, Pb)
    ow, om, oc = orc.correct(p, pose, z, pred)
    assert len(ow) == len(a)
    for i, (wi, mi, Pi) in enumerate(a):
        assert np.isclose(ow[i], wi, rtol=1e-9, atol=1e-300) and np.allclose(om[i], mi, rtol=1e-9, atol=1e-12)
        assert np.allclose(oc[i], Pi, rtol=1e-8, atol=1e-14)


def _qexp(w):
    a = np.linalg.norm(w)
    return np.array([1.0, 0, 0, 0]) if a == 0 else np.concatenate([[np.cos(a)], np.sin(a) * w / a])


@pytest.mark.parametrize("seed", range(6))
def test_quasi_gradient_is_the_derivative_the_jacobian_describes(seed):
    """QuasiSetLogLikelihood(..., out gradient) (PHDNavigator.cs:543-713) with MeasurementJacobianP
    (PRM3DMeasurer.cs:185-211): [-R(q*) | -R(q*) [l - t]x] is the derivative of the local landmark coordinates wrt. a
    GLOBAL translation and wrt. a global-frame rotation applied as q' = exp(-w/2) q (not the convention of Pose3D.Add,
    which the restatement does not repair). With weights that sum to one (average_mode 1) the gradient of a case whose
    components are all enumerated must therefore equal central differences of the value along exactly those paths."""
    rng = np.random.default_rng(900 + seed)
    p = prm3d_defaults(4, 600, 8)
    J = int(rng.integers(1, 4))
    M = int(rng.integers(1, 6 - J))
    pose, lm, z = random_case(rng, p, J, M)
    pose[3:] /= np.linalg.norm(pose[3:])
    if seed % 2 and J >= 2:
        lm[1] = lm[0] + rng.normal(0, 1e-3, 3)         # a shared measurement: several pairings carry weight
    value, grad = orc.quasi_set_log_likelihood_grad(p, pose, lm, z, average_mode=1)
    assert np.isclose(value, orc.quasi_set_log_likelihood(p, pose, lm, z), rtol=0, atol=1e-12)
    eps = 1e-6
    num = np.zeros(6)
    for i in range(6):
        l = []
        for s in (1, -1):
            d = np.zeros(6)
            d[i] = s * eps
            q = np.array(qmul(_qexp(-d[3:] / 2), pose[3:]))
            l.append(set_log_likelihood_bruteforce(p, np.concatenate([pose[:3] + d[:3], q]), lm, z, quasi=True))
        num[i] = (l[0] - l[1]) / (2 * eps)
    assert np.allclose(grad, num, rtol=2e-5, atol=2e-4), (grad, num)


def test_tempered_average_as_written():
    """TemperedAverage (MatrixExtensions.cs:400-440) divides the exponentiated weights by the Euclidean norm of the whole
    200-entry array (Accord's vector Normalize), not by their sum, and the entries a previous component left behind
    enter the norm. One landmark between two measurements."""
    p = prm3d_defaults(4, 600, 8)
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    zc = np.array([10.0, -20.0, 1.0])
    # three more landmarks far outside every gate: components of their own, and a map of 4 >= the 3 rows of the
    # component under test, so that LexicographicalPairing walks all of its pairings (`modelsize`, :293-299)
    lm = np.array([measure_to_map(p, pose, zc)] + [measure_to_map(p, pose, zc + [100.0 * (i + 1), 50, 0.2]) for i in range(3)])
    z = np.array([zc + [1.0, 0, 0], zc + [-1.0, 0.5, 0.01]])
    _, g_sum = orc.quasi_set_log_likelihood_grad(p, pose, lm, z, average_mode=1)
    _, g_src = orc.quasi_set_log_likelihood_grad(p, pose, lm, z, average_mode=0)
    # pairings: (lm-z0), (lm-z1), (lm missed)
    Jp = orc.jacobian_p(p, pose, lm[0])
    Rinv = np.linalg.inv(np.array(p.R).reshape(3, 3))
    zh = measure_perfect(p, pose, lm[0])
    d = [(zk - zh) @ Rinv @ Jp for zk in z]
    R = np.array(p.R).reshape(3, 3)
    lv = [np.log(p.pd) + np.log(multiplier(R)) - 0.5 * (zk - zh) @ Rinv @ (zk - zh) + np.log(p.clutter_density) for zk in z]
    lmiss = np.log(1 - p.pd) + 2 * np.log(p.clutter_density)
    e = np.exp(np.array(lv + [lmiss]) - max(lv + [lmiss]))
    want_sum = (e[0] * d[0] + e[1] * d[1]) / e.sum()
    want_src = (e[0] * d[0] + e[1] * d[1]) / np.sqrt((e ** 2).sum())
    assert np.allclose(g_sum, want_sum, rtol=1e-9, atol=1e-9)
    assert np.allclose(g_src, want_src, rtol=1e-9, atol=1e-9)
