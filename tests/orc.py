"""ctypes access to the CPU oracle (oracle/libphd_oracle.so). Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from monorfs_amd.abi import PhdParams

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SO = os.path.join(ROOT, "oracle", "libphd_oracle.so")

dp = C.POINTER(C.c_double)
ip = C.POINTER(C.c_int)


def _build():
    src = os.path.join(ROOT, "oracle", "phd_oracle.cpp")
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)


_build()
lib = C.CDLL(_SO)
lib.orc_set_log_likelihood.restype = C.c_double
lib.orc_weight_alpha.restype = C.c_double
lib.orc_assignment_value.restype = C.c_double
lib.orc_log_sum_exp.restype = C.c_double
lib.orc_detection_probability.restype = C.c_double


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(ip)


def pack(mix):
    """mix: (w[n], mean[n,3], cov[n,3,3]) -> contiguous arrays."""
    w, m, c = mix
    return (np.ascontiguousarray(w, np.float64).reshape(-1), np.ascontiguousarray(m, np.float64).reshape(-1, 3),
            np.ascontiguousarray(c, np.float64).reshape(-1, 3, 3))


def _out(cap):
    return np.zeros(cap), np.zeros((cap, 3)), np.zeros((cap, 3, 3))


def _stage(fn, p, pose, z, mix, cap):
    w, m, c = pack(mix)
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    ow, om, oc = _out(cap)
    pose7 = np.zeros(7)
    pose7[:len(pose)] = pose
    if len(pose) < 7:
        pose7[3] = 1.0
    n = fn(C.byref(p), pose7.ctypes.data_as(dp), z.ctypes.data_as(dp), len(z),
           w.ctypes.data_as(dp), m.ctypes.data_as(dp), c.ctypes.data_as(dp), len(w),
           cap, ow.ctypes.data_as(dp), om.ctypes.data_as(dp), oc.ctypes.data_as(dp))
    assert 0 <= n <= cap, "oracle output exceeds capacity"
    return ow[:n], om[:n], oc[:n]


def predict(p, pose, z, mix, cap=4096):
    return _stage(lib.orc_predict, p, pose, z, mix, cap)


def correct(p, pose, z, mix, cap=None):
    n = len(mix[0])
    cap = cap or (n * (len(z) + 1) + 8)
    return _stage(lib.orc_correct, p, pose, z, mix, cap)


def prune(p, mix, cap=None):
    w, m, c = pack(mix)
    cap = cap or (len(w) + 1)
    ow, om, oc = _out(cap)
    n = lib.orc_prune(C.byref(p), w.ctypes.data_as(dp), m.ctypes.data_as(dp), c.ctypes.data_as(dp), len(w),
                      cap, ow.ctypes.data_as(dp), om.ctypes.data_as(dp), oc.ctypes.data_as(dp))
    return ow[:n], om[:n], oc[:n]


def best_map_estimate(mix):
    w, m, c = pack(mix)
    cap = int(max(1, np.sum(w) + 2))
    om = np.zeros((cap, 3))
    src = np.zeros(cap, np.int32)
    n = lib.orc_best_map_estimate(w.ctypes.data_as(dp), m.ctypes.data_as(dp), c.ctypes.data_as(dp), len(w), cap,
                                  om.ctypes.data_as(dp), src.ctypes.data_as(ip))
    return om[:n], src[:n]


def set_log_likelihood(p, pose7, lm, z):
    lm = np.ascontiguousarray(lm, np.float64).reshape(-1, 3)
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    pose7 = np.ascontiguousarray(pose7, np.float64)
    ncl, mx = C.c_int(0), C.c_int(0)
    lmp = lm if len(lm) else np.zeros((1, 3))
    v = lib.orc_set_log_likelihood(C.byref(p), pose7.ctypes.data_as(dp), lmp.ctypes.data_as(dp), len(lm),
                                   z.ctypes.data_as(dp), len(z), C.byref(ncl), C.byref(mx))
    return v, ncl.value, mx.value


def quasi_set_log_likelihood(p, pose7, lm, z):
    """PHDNavigator.QuasiSetLogLikelihood (PHDNavigator.cs:526-531): everything fully visible, gate 12"""
    lm = np.ascontiguousarray(lm, np.float64).reshape(-1, 3)
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    pose7 = np.ascontiguousarray(pose7, np.float64)
    lib.orc_quasi_set_log_likelihood.restype = C.c_double
    lmp = lm if len(lm) else np.zeros((1, 3))
    return lib.orc_quasi_set_log_likelihood(C.byref(p), pose7.ctypes.data_as(dp), lmp.ctypes.data_as(dp), len(lm),
                                            z.ctypes.data_as(dp), len(z))


def pose_add(pose7, delta6):
    """Pose3D.Add (Pose3D.cs:282-291)"""
    a, ap = _d(np.asarray(pose7, float).reshape(7))
    d, dptr = _d(np.asarray(delta6, float).reshape(6))
    out, op = _d(np.zeros(7))
    lib.orc_pose_add(ap, dptr, op)
    return out


def loglike_gradient(p, pose6, linearpoint7, lm, z):
    """LoopyPHDNavigator.LogLikeGradient (LoopyPHDNavigator.cs:876-909)"""
    lm, lmp = _d(np.asarray(lm, float).reshape(-1, 3))
    z, zp = _d(np.asarray(z, float).reshape(-1, p.zdim))
    a, ap = _d(np.asarray(pose6, float).reshape(6))
    l, lp = _d(np.asarray(linearpoint7, float).reshape(7))
    g, gp = _d(np.zeros(6))
    lib.orc_loglike_gradient(C.byref(p), ap, lp, lmp, len(lm), zp, len(z), gp)
    return g


def weight_alpha(p, pose7, z, predicted, corrected):
    pw, pm, pc = pack(predicted)
    cw, cm, cc = pack(corrected)
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    pose7 = np.ascontiguousarray(pose7, np.float64)
    sll = C.c_double(0)
    a = lib.orc_weight_alpha(C.byref(p), pose7.ctypes.data_as(dp), z.ctypes.data_as(dp), len(z),
                             pw.ctypes.data_as(dp), pm.ctypes.data_as(dp), pc.ctypes.data_as(dp), len(pw),
                             cw.ctypes.data_as(dp), cm.ctypes.data_as(dp), cc.ctypes.data_as(dp), len(cw), C.byref(sll))
    return a, sll.value


def resample(w, u):
    w = np.ascontiguousarray(w, np.float64)
    src = np.zeros(len(w), np.int32)
    best = lib.orc_resample(w.ctypes.data_as(dp), len(w), C.c_double(u), src.ctypes.data_as(ip))
    return src, best


def particle_depleted(p, w):
    w = np.ascontiguousarray(w, np.float64)
    return bool(lib.orc_particle_depleted(C.byref(p), w.ctypes.data_as(dp), len(w)))


def hungarian(mat):
    m, mp = _d(mat)
    n = m.shape[0]
    out = np.zeros(n, np.int32)
    ok = lib.orc_hungarian(mp, n, out.ctypes.data_as(ip))
    return out.tolist() if ok else None


def add_odometry(pose7, delta6):
    """Pose3D.AddOdometry (Pose3D.cs:314-333)"""
    a, ap = _d(np.asarray(pose7, float).reshape(7))
    d, dptr = _d(np.asarray(delta6, float).reshape(6))
    out, op = _d(np.zeros(7))
    lib.orc_add_odometry(ap, dptr, op)
    return out


def diff_odometry(pose7, origin7):
    """Pose3D.DiffOdometry (Pose3D.cs:338-356)"""
    a, ap = _d(np.asarray(pose7, float).reshape(7))
    o, optr = _d(np.asarray(origin7, float).reshape(7))
    out, op = _d(np.zeros(6))
    lib.orc_diff_odometry(ap, optr, op)
    return out


def quaternion_ypr(yaw, pitch, roll):
    out, op = _d(np.zeros(4))
    lib.orc_quaternion_ypr.argtypes = [C.c_double, C.c_double, C.c_double, dp]
    lib.orc_quaternion_ypr(yaw, pitch, roll, op)
    return out


def update_motion(poses, reading, noise=None, perfect_still=False):
    """TrackVehicle.UpdateNoisy (TrackVehicle.cs:89-102) over all particles; returns the new poses"""
    out, op = _d(np.array(poses, float).reshape(-1, 7).copy())
    r, rp = _d(np.asarray(reading, float).reshape(6))
    if noise is not None:
        nz, nzp = _d(np.asarray(noise, float).reshape(-1, 6))
    else:
        nzp = None
    lib.orc_update_motion(op, len(out), rp, nzp, int(bool(perfect_still)))
    return out


def ospa(a, b, cutoff=1.0, order=1.0):
    """Plot.OSPA (postanalysis/Plot.cs:531-581): (distance, cardinality part) between two sets of 3-D landmarks;
    cutoff = C, order = P (1 and 1 in SURVEY 8d)."""
    a, ap = _d(np.asarray(a, float).reshape(-1, 3))
    b, bp = _d(np.asarray(b, float).reshape(-1, 3))
    lib.orc_ospa.restype = C.c_double
    lib.orc_ospa.argtypes = [dp, C.c_int, dp, C.c_int, C.c_double, C.c_double, dp]
    card, cp = _d(np.zeros(1))
    d = lib.orc_ospa(ap, len(a), bp, len(b), float(cutoff), float(order), cp)
    return d, card[0]


def assignment_value(mat, matches):
    m, mp = _d(mat)
    a, ap = _i(matches)
    return lib.orc_assignment_value(mp, m.shape[0], ap)


def murty(mat, maxcount=1000):
    m, mp = _d(mat)
    n = m.shape[0]
    asg = np.zeros((maxcount, n), np.int32)
    val = np.zeros(maxcount)
    k = lib.orc_murty(mp, n, maxcount, asg.ctypes.data_as(ip), val.ctypes.data_as(dp))
    return asg[:k].tolist(), val[:k]


def lexicographic(mat, modelsize, maxcount=1000):
    m, mp = _d(mat)
    n = m.shape[0]
    perms = np.zeros((maxcount, n), np.int32)
    val = np.zeros(maxcount)
    k = lib.orc_lexicographic(mp, n, modelsize, maxcount, perms.ctypes.data_as(ip), val.ctypes.data_as(dp))
    return perms[:k].tolist(), val[:k]


def murty_children(forced, eliminated, assignment):
    f, fp = _i(np.array(forced, np.int32).reshape(-1, 2))
    e, ep = _i(np.array(eliminated, np.int32).reshape(-1, 2))
    a, ap = _i(assignment)
    n = len(a)
    stride = len(f) + len(e) + n + 2
    cf = np.zeros((n, stride, 2), np.int32)
    ce = np.zeros((n, stride, 2), np.int32)
    nf = np.zeros(n, np.int32)
    ne = np.zeros(n, np.int32)
    k = lib.orc_murty_children(fp, len(f), ep, len(e), ap, n, stride, cf.ctypes.data_as(ip), nf.ctypes.data_as(ip),
                               ce.ctypes.data_as(ip), ne.ctypes.data_as(ip))
    return [{"forced": cf[c, :nf[c]].tolist(), "eliminated": ce[c, :ne[c]].tolist()} for c in range(k)]


def connected_components(defined, h, w):
    mask = np.zeros((h, w), np.uint8)
    for i, k in defined:
        mask[i, k] = 1
    rl = np.zeros(h, np.int32)
    cl = np.zeros(w, np.int32)
    n = lib.orc_connected_components(mask.ctypes.data_as(C.POINTER(C.c_uint8)), h, w, rl.ctypes.data_as(ip),
                                     cl.ctypes.data_as(ip))
    return n, rl, cl


def probe3(fn, p, pose7, v, nout):
    pose7 = np.ascontiguousarray(pose7, np.float64)
    v = np.ascontiguousarray(v, np.float64)
    out = np.zeros(nout)
    fn(C.byref(p), pose7.ctypes.data_as(dp), v.ctypes.data_as(dp), out.ctypes.data_as(dp))
    return out


def measure_perfect(p, pose7, lm):
    return probe3(lib.orc_measure_perfect, p, pose7, lm, 3)


def measure_to_map(p, pose7, z):
    return probe3(lib.orc_measure_to_map, p, pose7, z, 3)


def jacobian_l(p, pose7, lm):
    return probe3(lib.orc_jacobian_l, p, pose7, lm, 9).reshape(3, 3)


def detection_probability(p, pose7, lm):
    pose7 = np.ascontiguousarray(pose7, np.float64)
    lm = np.ascontiguousarray(lm, np.float64)
    return lib.orc_detection_probability(C.byref(p), pose7.ctypes.data_as(dp), lm.ctypes.data_as(dp))


class State:
    """Flat particle-filter state in the layout orc_slam_update works on."""

    def __init__(self, P, cap):
        self.P, self.cap = P, cap
        self.poses = np.zeros((P, 7))
        self.poses[:, 3] = 1
        self.w = np.zeros((P, cap))
        self.mean = np.zeros((P, cap, 3))
        self.cov = np.zeros((P, cap, 3, 3))
        self.n = np.zeros(P, np.int32)
        self.weights = np.full(P, 1.0 / P)

    def copy(self):
        s = State.__new__(State)
        s.P, s.cap = self.P, self.cap
        for k in ("poses", "w", "mean", "cov", "n", "weights"):
            setattr(s, k, getattr(self, k).copy())
        return s

    def map(self, i):
        n = self.n[i]
        return self.w[i, :n], self.mean[i, :n], self.cov[i, :n]


def slam_update(p, st, z, onlymapping=False, u=0.5, threads=1, stage_times=None):
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    src = np.zeros(st.P, np.int32)
    res = C.c_int(0)
    alpha = np.zeros(st.P)
    stt = np.zeros(4)
    best = lib.orc_slam_update(C.byref(p), st.P, st.poses.ctypes.data_as(dp), st.cap,
                               st.w.ctypes.data_as(dp), st.mean.ctypes.data_as(dp), st.cov.ctypes.data_as(dp),
                               st.n.ctypes.data_as(ip), st.weights.ctypes.data_as(dp),
                               z.ctypes.data_as(dp), len(z), int(onlymapping), C.c_double(u), threads,
                               src.ctypes.data_as(ip), C.byref(res), alpha.ctypes.data_as(dp), stt.ctypes.data_as(dp))
    if best < 0:
        raise RuntimeError("oracle slab overflow (raise cap)")
    if stage_times is not None:
        stage_times[:] = stt
    return best, src, bool(res.value), alpha


def quasi_set_log_likelihood_grad(p, pose7, lm, z, average_mode=0):
    """PHDNavigator.QuasiSetLogLikelihood(..., out gradient) (PHDNavigator.cs:543-548). average_mode 0 = TemperedAverage
    as its source reads (Accord Normalize = Euclidean norm over the whole 200-entry array), 1 = weights / their sum."""
    lm = np.ascontiguousarray(lm, np.float64).reshape(-1, 3)
    z = np.ascontiguousarray(z, np.float64).reshape(-1, p.zdim)
    pose7 = np.ascontiguousarray(pose7, np.float64)
    lib.orc_quasi_set_log_likelihood_grad.restype = C.c_double
    lmp = lm if len(lm) else np.zeros((1, 3))
    g = np.zeros(6)
    v = lib.orc_quasi_set_log_likelihood_grad(C.byref(p), pose7.ctypes.data_as(dp), lmp.ctypes.data_as(dp), len(lm),
                                              z.ctypes.data_as(dp), len(z), g.ctypes.data_as(dp), int(average_mode))
    return v, (g[:2].copy() if p.model == 0 else g)


def jacobian_p(p, pose7, lm):
    """PRM3DMeasurer.MeasurementJacobianP (PRM3DMeasurer.cs:185-211): 3 x 6"""
    a, ap = _d(np.asarray(pose7, float).reshape(7))
    l, lp = _d(np.asarray(lm, float).reshape(3))
    out, op = _d(np.zeros(18))
    lib.orc_jacobian_p(C.byref(p), ap, lp, op)
    return out.reshape(3, 6)


def map_error(visited, estimate, estimated_pose=None, true_pose=None, cutoff=1.0, order=1.0):
    """Plot.MapError, one frame (postanalysis/Plot.cs:489-524): (ospa, spatial part); poses None = no reference frame"""
    v, vp = _d(np.asarray(visited, float).reshape(-1, 3))
    e, ep = _d(np.asarray(estimate, float).reshape(-1, 3))
    has = estimated_pose is not None
    a, ap = _d(np.asarray(estimated_pose if has else [0, 0, 0, 1, 0, 0, 0], float))
    b, bp = _d(np.asarray(true_pose if has else [0, 0, 0, 1, 0, 0, 0], float))
    lib.orc_map_error.restype = C.c_double
    lib.orc_map_error.argtypes = [dp, C.c_int, dp, C.c_int, C.c_int, dp, dp, C.c_double, C.c_double, dp]
    sp, spp = _d(np.zeros(1))
    d = lib.orc_map_error(vp, len(v), ep, len(e), int(has), ap, bp, float(cutoff), float(order), spp)
    return d, sp[0]
