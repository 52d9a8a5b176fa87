"""Host mirror of the pose searches of the reference's smoother: the static functions of
LoopyPHDNavigator that evaluate QuasiSetLogLikelihood and its gradient for candidate poses against one map estimate
and one measurement set (LoopyPHDNavigator.cs:718-1021), and Filter / FilterMissing (:713-762).

Every evaluation goes to the device through PHDNavigator.QuasiSetLogLikelihood / QuasiSetLogLikelihoodGradient
(phd_quasi_set_loglik, phd_quasi_set_loglik_grad) as a BATCH: the 16 step sizes of a line search at once, all the
starting guesses of GuidedFitMixture side by side, the 12 gradient evaluations of the covariance fit together. The
results per pose are those of the reference's one-at-a-time loops, evaluations are independent of each other.

Not pinned (third-party code outside the reference tree, Accord.Math 3.0.2): EigenvalueDecomposition of the
finite-difference Hessian, PseudoInverse and PseudoDeterminant of 6 x 6 matrices; numpy stands in, see
LogLikeFitCovariance."""
import math

import numpy as np

from .navigator import pose3d_add

OdoSize = 6
GradientAscentRate = 1e-2   # Config.cs:94
GradientClip = 10.0         # Config.cs:95


# ---------------------------------------------------------------------------------------------- quaternions / Pose3D
def _qmul(a, b):   # Quaternion.cs:295-301
    return np.array([a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]),
                     a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                     a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])


def _qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def _qmatrix(q):   # Quaternion.ToMatrix, Quaternion.cs:327-342
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def _qlog(q):   # Quaternion.Log, Quaternion.cs:204-218
    q = np.asarray(q, float) / np.linalg.norm(q)
    phi = math.acos(min(1.0, max(-1.0, q[0])))
    mag = np.linalg.norm(q[1:])
    if mag < 1e-12:
        return np.zeros(3)
    return phi * (q[1:] / mag)


def vector_rotator(a, b):
    """Quaternion.VectorRotator (Quaternion.cs:281-284): the rotation taking unit vector a into unit vector b"""
    q = np.concatenate([[1 + np.dot(a, b)], np.cross(a, b)])
    return q / np.linalg.norm(q)


def pose3d_subtract(pose7, origin7):
    """Pose3D.Subtract (Pose3D.cs:296-308): the linear vector that takes `origin` into `pose`"""
    pose7, origin7 = np.asarray(pose7, float), np.asarray(origin7, float)
    qo = origin7[3:]
    dq = _qmul(_qconj(qo), pose7[3:])
    dxg = pose7[:3] - origin7[:3]
    dx = _qmul(_qmul(_qconj(qo), np.concatenate([[0.0], dxg])), qo)
    return np.concatenate([dx[1:], 2 * _qlog(dq)])   # dq.ToLinear() = dq.Subtract(Identity) = 2 Log(dq), Quaternion.cs:176-179


def fit_to_measurement(params, pose0, measurement, landmark):
    """PRM3DMeasurer.FitToMeasurement (PRM3DMeasurer.cs:224-244): the pose near pose0 from which `landmark` would be
    measured as `measurement`"""
    pose0 = np.asarray(pose0, float)
    focal = params.measurer[0]
    q0 = pose0[3:]
    diff = np.asarray(landmark, float) - pose0[:3]
    landmarklocal = _qmatrix(_qconj(q0)) @ diff
    px, py, rng = measurement
    ml = np.zeros(3)
    ml[2] = rng / math.sqrt(1 + (px * px + py * py) / (focal * focal))
    ml[0] = px * ml[2] / focal
    ml[1] = py * ml[2] / focal
    align = vector_rotator(landmarklocal / np.linalg.norm(landmarklocal), ml / np.linalg.norm(ml))
    rotation = _qmul(_qconj(align), q0)
    location = np.asarray(landmark, float) - _qmatrix(rotation) @ ml
    return np.concatenate([location, rotation / np.linalg.norm(rotation)])   # new Pose3D normalises (Pose3D.cs:157-161)


def best_map_estimate(model):
    """Map.BestMapEstimate (Map.cs:119-142): (int) ExpectedSize picks from the weight-sorted list, each pick re-entered
    with its weight less one; stable order on ties (the canonical order of the device path). Returns the means [J][3]."""
    w, m, _ = model
    w = np.asarray(w, float)
    size = int(np.sum(w))
    lst = sorted(((w[i], i) for i in range(len(w))), key=lambda e: -e[0])
    picks = []
    for i in range(size):
        wi, src = lst[i]
        picks.append(src)
        lst.append((wi - 1, src))
        lst.sort(key=lambda e: -e[0])
    return np.asarray(m, float).reshape(-1, 3)[picks].reshape(-1, 3)


# ---------------------------------------------------------------------------------------------- batched evaluation
def _batches(n, cap):
    for s in range(0, n, cap):
        yield s, min(n, s + cap)


def _values(nav, measurements, landmarks, poses7):
    poses7 = np.asarray(poses7, float).reshape(-1, 7)
    out = np.empty(len(poses7))
    for s, e in _batches(len(poses7), nav.params.max_particles):
        out[s:e] = nav.QuasiSetLogLikelihood(measurements, landmarks, poses7[s:e])
    return out


def _values_gradients(nav, measurements, landmarks, poses7, average_mode):
    poses7 = np.asarray(poses7, float).reshape(-1, 7)
    out, grad = np.empty(len(poses7)), np.empty((len(poses7), 6))
    for s, e in _batches(len(poses7), nav.params.max_particles):
        out[s:e], grad[s:e] = nav.QuasiSetLogLikelihoodGradient(measurements, landmarks, poses7[s:e], average_mode)
    return out, grad


def LogLikeGradientAscent(nav, initial, measurements, landmarks, linearpoint, average_mode=0):
    """≙ LoopyPHDNavigator.LogLikeGradientAscent (:916-965) for one initial estimate [6] or several [n][6] run side by
    side. Returns (pose, loglike) or (poses[n][6], loglikes[n]).

    Per estimate, exactly the reference's loop: the analytic gradient at the LAST TRIED pose (`nextvehicle`, :933-934 —
    after a rejected step that is not the current one), clipped to GradientClip, step GradientAscentRate halved up to 16
    times until the value does not decrease (:943-951), until an iteration gains no more than 1e-3. The 16 candidate
    steps of an iteration are evaluated as one device batch and the first acceptable one is taken."""
    initial = np.asarray(initial, float)
    single = initial.ndim == 1
    pose = initial.reshape(-1, OdoSize).copy()
    n = len(pose)
    lin = np.asarray(linearpoint, float)
    nextpose7 = np.array([pose3d_add(lin, pose[a]) for a in range(n)])
    loglike, _ = _values_gradients(nav, measurements, landmarks, nextpose7, average_mode)   # :928-929
    prevvalue = np.full(n, -np.inf)
    active = [a for a in range(n) if loglike[a] - prevvalue[a] > 1e-3]
    while active:
        _, grad = _values_gradients(nav, measurements, landmarks, nextpose7[active], average_mode)   # :933-934
        cand6 = np.empty((len(active), 16, OdoSize))
        for u, a in enumerate(active):
            g = grad[u]
            size = np.linalg.norm(g)
            if size > GradientClip:
                g = g * (GradientClip / size)
            multiplier = GradientAscentRate
            for c in range(16):
                cand6[u, c] = pose[a] + multiplier * g
                multiplier /= 2.0
        cand7 = np.array([[pose3d_add(lin, cand6[u, c]) for c in range(16)] for u in range(len(active))])
        vals = _values(nav, measurements, landmarks, cand7.reshape(-1, 7)).reshape(len(active), 16)
        still = []
        for u, a in enumerate(active):
            c = 0
            while True:   # do { ... counter++ } while (nextloglike < loglike && counter < 16)
                nextloglike = vals[u, c]
                c += 1
                if not (nextloglike < loglike[a] and c < 16):
                    break
            nextpose7[a] = cand7[u, c - 1]
            prevvalue[a] = loglike[a]
            if nextloglike > loglike[a]:
                pose[a] = cand6[u, c - 1]
                loglike[a] = nextloglike
            if loglike[a] - prevvalue[a] > 1e-3:
                still.append(a)
        active = still
    return (pose[0], loglike[0]) if single else (pose, loglike)


def LogLikeFitCovariance(nav, pose, measurements, landmarks, linearpoint, average_mode=0):
    """≙ LoopyPHDNavigator.LogLikeFitCovariance (:976-1021): Hessian rows by central differences (eps = 1e-5) of the
    analytic gradient — 12 gradient evaluations, one device batch —, NaN -> zero matrix, positive eigenvalues clipped
    to zero, covariance = pinv(-Hessian).

    The reference decomposes the (not exactly symmetric) finite-difference Hessian with Accord's general
    EigenvalueDecomposition and recombines V D V'; that code is outside the tree. Here the symmetric part is decomposed
    (numpy.linalg.eigh), which is the same thing whenever the Hessian is symmetric. Unpinned, as the module header says."""
    pose = np.asarray(pose, float).reshape(OdoSize)
    lin = np.asarray(linearpoint, float)
    eps = 1e-5
    cand = np.empty((OdoSize, 2, 7))
    for i in range(OdoSize):
        for s, sign in enumerate((1.0, -1.0)):
            d = pose.copy()
            d[i] += sign * eps
            cand[i, s] = pose3d_add(lin, d)
    _, g = _values_gradients(nav, measurements, landmarks, cand.reshape(-1, 7), average_mode)
    g = g.reshape(OdoSize, 2, OdoSize)
    hessian = (g[:, 0] - g[:, 1]) / (2 * eps)
    if np.isnan(hessian).any():
        hessian = np.zeros((OdoSize, OdoSize))
    vals, vecs = np.linalg.eigh(0.5 * (hessian + hessian.T))
    hessian = vecs @ np.diag(np.minimum(0.0, vals)) @ vecs.T
    return np.linalg.pinv(-hessian)


def FitGaussian(nav, pose0, measurements, landmarks, linearpoint, average_mode=0):
    """≙ LoopyPHDNavigator.FitGaussian (:863-869): (mean, covariance) of weight 1"""
    maxpose, _ = LogLikeGradientAscent(nav, pose0, measurements, landmarks, linearpoint, average_mode)
    return maxpose, LogLikeFitCovariance(nav, maxpose, measurements, landmarks, linearpoint, average_mode)


def _pseudo_determinant(m):
    sv = np.linalg.svd(m, compute_uv=False)
    tol = max(m.shape) * np.finfo(float).eps * (sv[0] if len(sv) else 0.0)
    nz = sv[sv > tol]
    return float(np.prod(nz)) if len(nz) else 0.0


def GuidedFitMixture(nav, pose0, measurements, model, linearpoint, average_mode=0):
    """≙ LoopyPHDNavigator.GuidedFitMixture (:777-852). `model` = (w, mean, cov) of the map. Returns
    (emptyspace, [(weight, mean[6], cov[6][6]), ...]).

    As in the reference: guesses = pose0 plus, for every (landmark of the map estimate, measurement) pair, the pose
    that explains the pair if it lies within 0.5 of the initial pose (:790-798); a guess worse than a pose far from
    everything is dropped (:821-823); each remaining guess climbs (all of them side by side here); a maximum within
    Mahalanobis 0.1 of an earlier component is dropped (:827-837); the covariance is fitted at `maxpose`, which the
    reference never updates from pose0 (:800, :839), so it is computed once."""
    pose0 = np.asarray(pose0, float).reshape(OdoSize)
    lin = np.asarray(linearpoint, float)
    z = np.asarray(measurements, float).reshape(-1, 3)
    initpose = pose3d_add(lin, pose0)
    jmap = best_map_estimate(model)
    guesses = [pose0]
    for landmark in jmap:
        for measurement in z:
            guess = fit_to_measurement(nav.params, initpose, measurement, landmark)
            d = pose3d_subtract(guess, initpose)
            if float(np.dot(d, d)) < 0.5 * 0.5:
                guesses.append(pose3d_subtract(guess, lin))
    identity = np.array([0, 0, 0, 1.0, 0, 0, 0])
    infpose = pose3d_add(identity, np.full(OdoSize, 1e5))
    first = _values(nav, z, jmap, np.array([infpose] + [pose3d_add(lin, g) for g in guesses]))
    emptyspace = first[0]
    climbing = [g for g, v in zip(guesses, first[1:]) if not (v - emptyspace < 0)]
    components = []
    if not climbing:
        return emptyspace, components
    poses, values = LogLikeGradientAscent(nav, np.array(climbing), z, jmap, lin, average_mode)
    localcov = None
    for localpose, localmax in zip(poses, values):
        counted = False
        for _, mean, cov in components:
            diff = mean - localpose
            if math.sqrt(max(0.0, diff @ np.linalg.pinv(cov) @ diff)) < 0.1:
                counted = True
                break
        if counted:
            continue
        if localcov is None:
            localcov = LogLikeFitCovariance(nav, pose0, z, jmap, lin, average_mode)
        # Math.Pow(2 pi, -localpose.Length / 2): integer division, exponent -3
        with np.errstate(divide="ignore"):
            logmultiplier = math.log((2 * math.pi) ** (-(OdoSize // 2))) - 0.5 * np.log(_pseudo_determinant(localcov))
        components.append((math.exp(localmax - logmultiplier), localpose, localcov))
    return emptyspace, components


# ---------------------------------------------------------------------------------------------- Filter / FilterMissing
def FilterMissing(nav, trajectory, factors, index, to):
    """≙ LoopyPHDNavigator.FilterMissing (:729-762): a fresh one-particle, mapping-only PHD filter run over the poses
    trajectory[i] = (time, pose7) with the measurement sets factors[i] = (time, [[px, py, range], ...]), frame `index`
    left out, frames from `to` on ignored. Returns BestMapModel (w, mean, cov).

    `nav` is reset (one particle, empty map, mapping only): ≙ InnerFilter = new PHDNavigator(RefVehicle, 1, true)."""
    to = min(len(trajectory), to)
    index = to if index < 0 else min(to, index)
    empty = (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3)))
    nav.OnlyMapping = True
    nav.reset(np.asarray(trajectory[0][1], float) if len(trajectory) else np.array([0, 0, 0, 1.0, 0, 0, 0]), empty, 1)
    for i in list(range(0, index)) + list(range(index + 1, to)):
        nav.set_poses(np.asarray(trajectory[i][1], float).reshape(1, 7))   # InnerFilter.BestEstimate.Pose = trajectory[i].Item2
        nav.SlamUpdate(trajectory[i][0], np.asarray(factors[i][1], float).reshape(-1, 3))
    return nav.BestMapModel


def Filter(nav, trajectory, factors):
    """≙ LoopyPHDNavigator.Filter (:718-721)"""
    return FilterMissing(nav, trajectory, factors, len(trajectory), len(trajectory))
