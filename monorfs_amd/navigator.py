"""Python host-side mirror of the reference's solver interface for the PHD path.

`PHDNavigator` here keeps the member names of
    class PHDNavigator<MeasurerT, PoseT, MeasurementT> : Navigator<...>
    (mono-rfs-lib/SLAM/Navigators/PHDNavigator.cs:52-983, Navigator.cs:47-396)
and forwards every one of them to libphdhip.so through the C-ABI of include/phdhip.h — it is what
tests/ and bench.py drive; the C++ twin for native hosts is monorfs_amd/host/PHDNavigator.hpp and the
C# adapter a maintainer would add is bindings/csharp/HipPHDNavigator.cs.

No compute happens in this file: it marshals arrays and raises on a non-zero status."""
import ctypes as C

import numpy as np

from . import _lib
from .abi import (PHD_ERR_ASSOCIATION, PHD_STAGE_CORRECTED, PHD_STAGE_PREDICTED, PHD_STAGE_PRUNED, PhdParams,
                  prm3d_defaults)

dp = _lib.dp
ip = _lib.ip


def pose3d_add(pose7, delta6):
    """Pose3D.Add (Pose3D.cs:282-291): x + q dx q*, q Exp(dr / 2) normalised"""
    x, q = np.asarray(pose7[:3], float), np.asarray(pose7[3:], float)
    lie = 0.5 * np.asarray(delta6[3:], float)
    phi = np.linalg.norm(lie)
    dq = np.array([1.0, 0, 0, 0]) if phi < 1e-12 else np.concatenate([[np.cos(phi)], np.sin(phi) * (lie / phi)])

    def mul(a, b):
        return np.array([a[0] * b[0] - (a[1] * b[1] + a[2] * b[2] + a[3] * b[3]),
                         a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] + a[2] * b[0] + a[3] * b[1] - a[1] * b[3],
                         a[0] * b[3] + a[3] * b[0] + a[1] * b[2] - a[2] * b[1]])
    nq = mul(q, dq)
    nq = nq / np.linalg.norm(nq)
    dl = mul(mul(q, np.concatenate([[0.0], np.asarray(delta6[:3], float)])), q * [1, -1, -1, -1])
    return np.concatenate([x + dl[1:], nq])


class PHDError(RuntimeError):
    """≙ the InvalidOperationException with Data["module"] that Simulation.Update catches
    (Simulation.cs:655-670); `module` is "association" for PHD_ERR_ASSOCIATION."""

    def __init__(self, status, message):
        super().__init__("libphdhip status %d: %s" % (status, message))
        self.status = status
        self.module = "association" if status == PHD_ERR_ASSOCIATION else "phdhip"


def _ptr(a):
    return a.ctypes.data_as(dp)


class PHDNavigator:
    def __init__(self, params: PhdParams = None, particlecount=1, onlymapping=False, device=0,
                 pose=(0, 0, 0, 1, 0, 0, 0), devices=None):
        """≙ new PHDNavigator(vehicle, particlecount, onlymapping) (PHDNavigator.cs:192-208).
        devices: a list of HIP ordinals -> one multi-device handle (phd_create_multi), the particles sharded
        contiguously over them; params.max_particles and the particle count are then totals (multiples of len(devices))."""
        self._lib = _lib.load()
        self.params = params if params is not None else prm3d_defaults(max_particles=max(1, particlecount))
        if devices is not None:
            devs = (C.c_int * len(devices))(*[int(d) for d in devices])
            self._h = self._lib.phd_create_multi(C.byref(self.params), devs, len(devices))
        else:
            self._h = self._lib.phd_create(C.byref(self.params), device)
        if not self._h:
            raise PHDError(-1, self._lib.phd_create_error().decode())
        self.ParticleCount = particlecount
        self.OnlyMapping = bool(onlymapping)
        n = 1 if onlymapping else particlecount   # :201-203
        self.reset(np.asarray(pose, float), (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3))), n)

    # ------------------------------------------------------------------ plumbing
    def _check(self, rc):
        if rc != 0:
            raise PHDError(rc, self._lib.phd_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.phd_destroy(self._h)
            self._h = None

    Dispose = close   # Navigator.Dispose (Navigator.cs:395)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ state
    def reset(self, pose, model, particlecount):
        """≙ PHDNavigator.reset (:245-266)."""
        w, m, c = (np.ascontiguousarray(x, np.float64) for x in model)
        pose = np.ascontiguousarray(pose, np.float64)
        self._check(self._lib.phd_reset(self._h, int(particlecount), _ptr(pose), _ptr(w), _ptr(m), _ptr(c), len(w)))

    def CollapseParticles(self, particlecount):
        """≙ :233-236: every particle becomes a copy of the best one."""
        best = self.BestParticle
        pose = self.poses()[best]
        self.reset(pose, self.MapModel(best), particlecount)

    def ResetMapModel(self):
        """≙ :271-276."""
        empty = (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3)))
        poses, weights = self.poses().copy(), self.VehicleWeights.copy()
        for i in range(self.particle_count):
            self.set_map(i, empty)
        self.set_poses(poses)
        self.set_weights(weights)

    @property
    def particle_count(self):
        return self._lib.phd_particle_count(self._h)

    def set_poses(self, poses):
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        self._check(self._lib.phd_set_poses(self._h, _ptr(poses), len(poses)))

    def set_weights(self, weights):
        weights = np.ascontiguousarray(weights, np.float64)
        self._check(self._lib.phd_set_weights(self._h, _ptr(weights), len(weights)))

    def set_map(self, particle, model):
        w, m, c = (np.ascontiguousarray(x, np.float64) for x in model)
        self._check(self._lib.phd_set_map(self._h, int(particle), _ptr(w), _ptr(m), _ptr(c), len(w)))

    def upload_state(self, planes, counts, poses, weights):
        """planes: [10][P][stride] float64 in the device layout (w, mean xyz, cov xx xy xz yy yz zz)."""
        planes = np.ascontiguousarray(planes, np.float64)
        counts = np.ascontiguousarray(counts, np.int32)
        poses = np.ascontiguousarray(poses, np.float64)
        weights = np.ascontiguousarray(weights, np.float64)
        _, P, stride = planes.shape
        self._check(self._lib.phd_upload_state_soa(self._h, P, stride, _ptr(planes), counts.ctypes.data_as(ip),
                                                   _ptr(poses), _ptr(weights)))

    def download_state(self, stride):
        P = self.particle_count
        planes = np.zeros((10, P, stride))
        counts = np.zeros(P, np.int32)
        poses = np.zeros((P, 7))
        weights = np.zeros(P)
        self._check(self._lib.phd_download_state_soa(self._h, stride, _ptr(planes), counts.ctypes.data_as(ip),
                                                     _ptr(poses), _ptr(weights)))
        return planes, counts, poses, weights

    def poses(self):
        n = C.c_int32(0)
        ptr = self._lib.phd_poses(self._h, C.byref(n))
        if not ptr:
            raise PHDError(-1, self._lib.phd_last_error(self._h).decode())
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).reshape(-1, 7).copy()

    # ------------------------------------------------------------------ Navigator interface
    def Update(self, time, poses):
        """≙ PHDNavigator.Update (:295-314). The motion model and its RNG stay on the host
        (TrackVehicle.UpdateNoisy): the caller hands over the propagated particle poses."""
        self.set_poses(poses)

    def UpdateOdometry(self, time, reading, noise=None, perfect_still=False):
        """≙ PHDNavigator.Update (:295-314) with the motion step on the device (phd_update_motion):
        `reading` is the odometry (dx dy dz dpitch dyaw droll), `noise` the per-particle vectors
        dt * chol(MotionCovariance) * N(0, I) the host drew (TrackVehicle.UpdateNoisy, TrackVehicle.cs:89-102)."""
        reading = np.ascontiguousarray(reading, np.float64).reshape(6)
        if noise is not None:
            noise = np.ascontiguousarray(noise, np.float64).reshape(-1, 6)
            if len(noise) != self.particle_count:
                raise ValueError("one noise vector per particle")
        self._check(self._lib.phd_update_motion(self._h, _ptr(reading), _ptr(noise) if noise is not None else None,
                                                self.particle_count, int(bool(perfect_still))))

    def QuasiSetLogLikelihood(self, measurements, landmarks, poses):
        """≙ static PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose) (PHDNavigator.cs:526-531), batched over
        candidate poses (phd_quasi_set_loglik): returns one value per pose."""
        z = np.ascontiguousarray(measurements, np.float64).reshape(-1, 3)
        lm = np.ascontiguousarray(landmarks, np.float64).reshape(-1, 3)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        out = np.zeros(len(poses))
        self._check(self._lib.phd_quasi_set_loglik(self._h, _ptr(poses), len(poses), _ptr(lm) if len(lm) else None, len(lm),
                                                   _ptr(z) if len(z) else None, len(z), _ptr(out)))
        return out

    def QuasiSetLogLikelihoodGradient(self, measurements, landmarks, poses, average_mode=0):
        """≙ static PHDNavigator.QuasiSetLogLikelihood(measurements, map, pose, out gradient) (PHDNavigator.cs:543-548),
        batched over candidate poses (phd_quasi_set_loglik_grad): returns (values[n], gradients[n][6]).
        average_mode: TemperedAverage as its source reads (0) or with weights that sum to one (1), see include/phdhip.h."""
        z = np.ascontiguousarray(measurements, np.float64).reshape(-1, 3)
        lm = np.ascontiguousarray(landmarks, np.float64).reshape(-1, 3)
        poses = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        out, grad = np.zeros(len(poses)), np.zeros((len(poses), 6))
        self._check(self._lib.phd_quasi_set_loglik_grad(self._h, _ptr(poses), len(poses), _ptr(lm) if len(lm) else None, len(lm),
                                                        _ptr(z) if len(z) else None, len(z), int(average_mode), _ptr(out), _ptr(grad)))
        return out, grad

    def LogLikeGradient(self, pose, measurements, landmarks, linearpoint):
        """≙ LoopyPHDNavigator.LogLikeGradient (LoopyPHDNavigator.cs:876-909): central differences (eps = 1e-5) of the
        quasi set log-likelihood at linearpoint.Add(pose +- eps e_i) — the 12 evaluations go to the device as one batch.
        `pose` may hold several linear poses [n][6]: the 12 n evaluations are still one batch."""
        pose = np.ascontiguousarray(pose, np.float64).reshape(-1, 6)
        eps = 1e-5
        cand = np.empty((len(pose), 6, 2, 7))
        for a in range(len(pose)):
            for i in range(6):
                for s, sign in enumerate((1.0, -1.0)):
                    d = pose[a].copy()
                    d[i] += sign * eps
                    cand[a, i, s] = pose3d_add(linearpoint, d)
        ll = self.QuasiSetLogLikelihood(measurements, landmarks, cand.reshape(-1, 7)).reshape(len(pose), 6, 2)
        g = (ll[:, :, 0] - ll[:, :, 1]) / (2 * eps)
        return g[0] if len(g) == 1 else g

    def SlamUpdate(self, time, measurements, u_resample=0.5):
        """≙ PHDNavigator.SlamUpdate (:323-362)."""
        z = np.ascontiguousarray(measurements, np.float64).reshape(-1, 3)
        self._check(self._lib.phd_slam_update(self._h, _ptr(z), len(z), int(self.OnlyMapping), float(u_resample)))

    @property
    def VehicleWeights(self):
        n = C.c_int32(0)
        ptr = self._lib.phd_weights(self._h, C.byref(n))
        if not ptr:
            raise PHDError(-1, self._lib.phd_last_error(self._h).decode())
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    @property
    def BestParticle(self):
        return self._lib.phd_best_particle(self._h)

    def _map_from(self, fn, *args):
        n = C.c_int32(0)
        w, m, c = dp(), dp(), dp()
        self._check(fn(self._h, *args, C.byref(n), C.byref(w), C.byref(m), C.byref(c)))
        k = n.value
        if k == 0:
            return np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3))
        return (np.ctypeslib.as_array(w, shape=(k,)).copy(), np.ctypeslib.as_array(m, shape=(k * 3,)).reshape(k, 3).copy(),
                np.ctypeslib.as_array(c, shape=(k * 9,)).reshape(k, 3, 3).copy())

    def MapModel(self, particle):
        """≙ MapModels[particle] (:134)."""
        return self._map_from(self._lib.phd_map, int(particle))

    @property
    def BestMapModel(self):
        """≙ :155-161."""
        return self.MapModel(self.BestParticle)

    @property
    def BestEstimate(self):
        """≙ :144-150 — the pose of the best particle."""
        return self.poses()[self.BestParticle]

    def resample_sources(self):
        n = C.c_int32(0)
        r = C.c_uint8(0)
        ptr = self._lib.phd_resample_sources(self._h, C.byref(n), C.byref(r))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy(), bool(r.value)

    def ResampleParticles(self, weights, u):
        """≙ :724-760 on caller-supplied weights; returns (sources, best slot)."""
        weights = np.ascontiguousarray(weights, np.float64)
        src = np.zeros(len(weights), np.int32)
        best = C.c_int32(0)
        self._check(self._lib.phd_resample(self._h, _ptr(weights), len(weights), float(u), src.ctypes.data_as(ip), C.byref(best)))
        return src, best.value

    def ParticleDepleted(self, weights):
        """≙ :768-777."""
        weights = np.ascontiguousarray(weights, np.float64)
        d = C.c_uint8(0)
        self._check(self._lib.phd_particle_depleted(self._h, _ptr(weights), len(weights), C.byref(d)))
        return bool(d.value)

    # ------------------------------------------------------------------ stage-level (unit KAT surface)
    def run_stages(self, measurements, with_alpha=True):
        z = np.ascontiguousarray(measurements, np.float64).reshape(-1, 3)
        self._check(self._lib.phd_stage_run(self._h, _ptr(z), len(z), int(with_alpha)))

    def PredictConditional(self, particle=0):
        return self._map_from(self._lib.phd_stage_map, PHD_STAGE_PREDICTED, int(particle))

    def CorrectConditional(self, particle=0):
        """Corrected components with weight >= MinWeight (the rest cannot survive PruneModel), unsorted."""
        return self._map_from(self._lib.phd_stage_map, PHD_STAGE_CORRECTED, int(particle))

    def PruneModel(self, particle=0):
        return self._map_from(self._lib.phd_stage_map, PHD_STAGE_PRUNED, int(particle))

    def WeightAlpha(self):
        n = C.c_int32(0)
        ptr = self._lib.phd_stage_alpha(self._h, C.byref(n))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def SetLogLikelihood(self):
        n = C.c_int32(0)
        ptr = self._lib.phd_stage_setloglik(self._h, C.byref(n))
        return np.ctypeslib.as_array(ptr, shape=(n.value,)).copy()

    def test_pairing(self, matrix, lexicographic=False, modelsize=0, maxcount=200):
        """the device's MurtyPairing (best first) or LexicographicalPairing(matrix, modelsize) on a dense profit matrix:
        (assignments [k][n], values [k]) — phd_test_pairing, the surface GraphCombinatoricsTest's vectors go through"""
        m = np.ascontiguousarray(matrix, np.float64)
        n = m.shape[0]
        asg = np.zeros((maxcount, n), np.int32)
        val = np.zeros(maxcount)
        cnt = C.c_int(0)
        self._check(self._lib.phd_test_pairing(self._h, _ptr(m), n, 1 if lexicographic else 0, int(modelsize), maxcount,
                                               asg.ctypes.data_as(ip), _ptr(val), C.byref(cnt)))
        k = min(cnt.value, maxcount)
        return asg[:k].tolist(), val[:k].copy()

    # ------------------------------------------------------------------ benchmark surface
    def set_measurements(self, measurements):
        z = np.ascontiguousarray(measurements, np.float64).reshape(-1, 3)
        self._check(self._lib.phd_set_measurements(self._h, _ptr(z), len(z)))

    def step_async(self, u_resample=0.5):
        self._check(self._lib.phd_step_async(self._h, int(self.OnlyMapping), float(u_resample)))

    def sync(self):
        self._check(self._lib.phd_sync(self._h))

    def set_frozen(self, frozen):
        self._check(self._lib.phd_set_frozen(self._h, int(bool(frozen))))

    def set_all_pairs(self, on):
        """SURVEY 8d's benchmark mode: every (component, measurement) pair evaluated, the gate only masks (phd_set_all_pairs)"""
        self._check(self._lib.phd_set_all_pairs(self._h, int(bool(on))))

    def set_split(self, nsplit):
        """Launch a step's per-particle kernels as `nsplit` sub-ranges on concurrent streams (phd_set_split)."""
        self._check(self._lib.phd_set_split(self._h, int(nsplit)))

    def timing_reset(self, enabled=True):
        """enabled: False / 0 off, True / 1 every step, n > 1 every n-th step"""
        self._check(self._lib.phd_timing_reset(self._h, int(enabled)))

    def last_timings(self):
        names = C.POINTER(C.c_char_p)()
        ms = dp()
        n = self._lib.phd_last_timings(self._h, C.byref(names), C.byref(ms))
        return {names[i].decode(): ms[i] for i in range(n)}

    def multi_report(self):
        """diagnostics of a multi-device handle (phd_multi_report): per-phase device time of the sampled steps on the first
        shard's stream, the host's cost of posting / issuing a step, and which shard pairs exchange by direct peer access"""
        out = np.zeros(9)
        p2p = np.zeros(64 * 64, np.uint8)
        n = C.c_int32(0)
        self._check(self._lib.phd_multi_report(self._h, _ptr(out), p2p.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(n)))
        k = n.value
        names = ("local", "gather_wait", "global_and_plan", "pack", "exchange_wait_and_unpack")
        return {"phase_ms": {nm: float(out[i]) for i, nm in enumerate(names)}, "post_us": float(out[5]), "issue_us": float(out[6]), "issue_calls_us": float(out[8]),
                "sampled_steps": int(out[7]), "shards": k, "p2p": p2p[:k * k].reshape(k, k).astype(bool).tolist()}

    def test_migration_plan(self, gsrc, particles_per_rank, world, rank, resampled=True):
        """the migration plan as the device makes it inside a sharded step (phd_test_migration_plan): dict of the lists"""
        g = np.ascontiguousarray(gsrc, np.int32)
        Pl, n = int(particles_per_rank), int(world)
        sc, rc = np.zeros(n, np.int32), np.zeros(n, np.int32)
        sl, code, fs = np.zeros(Pl + 64, np.int32), np.zeros(Pl, np.int32), np.zeros(Pl, np.int32)
        sd = np.zeros((Pl + 64, 2), np.int32)
        ns, nr, st = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        q = lambda a: a.ctypes.data_as(ip)
        self._check(self._lib.phd_test_migration_plan(self._h, q(g), Pl, n, int(rank), int(bool(resampled)), q(sc), q(rc), q(sl), q(code), q(fs), q(sd),
                                                      C.byref(ns), C.byref(nr), C.byref(st)))
        return {"send_counts": sc, "recv_counts": rc, "send_list": sl[:ns.value].copy(), "dst_code": code, "fslot": fs[:nr.value].copy(),
                "send_dst": sd[:ns.value].copy(), "nsend": ns.value, "nrecv": nr.value, "status": st.value}

    def last_timing_counts(self):
        """launches behind each mean of the last last_timings() call (a split step launches each kernel per sub-range)"""
        names = C.POINTER(C.c_char_p)()
        ms = dp()
        n = self._lib.phd_last_timings(self._h, C.byref(names), C.byref(ms))
        cnt = C.POINTER(C.c_int)()
        self._lib.phd_last_timing_counts(self._h, C.byref(cnt))
        return {names[i].decode(): cnt[i] for i in range(n)}
