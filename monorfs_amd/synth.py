"""Synthetic PHD-SLAM frames: the inputs of the parity tests and of bench.py.

Only numpy; no product or oracle code is involved. Shapes follow the reference defaults:
measurer f=575.8156, film (-320,-240,640,480), range clip [0.1f, 2f] (PRM3DMeasurer.cs:70-73);
R = diag(2, 2, 1e-3), PD 0.9, clutter 3e-7 (Config.cs:251-262)."""
import numpy as np

FOCAL = 575.8156

CONFIGS = {   # BASELINE.json configs: particles, components, measurements, seed
    "A": (256, 128, 32, 1001),
    "B": (2048, 512, 64, 1002),
    "C8": (16384, 512, 64, 1003),
    "S": (4096, 1024, 128, 1004),
    # tuning shapes (where the one-launch chain stops paying: phdhip.hip chain_max), not BASELINE configs
    "A24": (24, 128, 32, 1001), "A64": (64, 128, 32, 1001), "A128": (128, 128, 32, 1001),   # (the reference's real-time regime: 20 - 200 particles)
    "A512": (512, 128, 32, 1001), "A1024": (1024, 128, 32, 1001), "B512": (512, 512, 64, 1002), "B1024": (1024, 512, 64, 1002),
    # rehearsal shapes of the multi-shard hosts: what 8 shards of A / B512 hold, as ONE handle
    "A2048": (2048, 128, 32, 1001), "B4096": (4096, 512, 64, 1002),
}


def measure_to_map_identity(z):
    """PRM3DMeasurer.MeasureToMap (PRM3DMeasurer.cs:299-312) for the identity pose."""
    z = np.asarray(z, float)
    alpha = z[..., 2] / np.sqrt(FOCAL ** 2 + z[..., 0] ** 2 + z[..., 1] ** 2)
    return np.stack([alpha * z[..., 0], alpha * z[..., 1], alpha * FOCAL], axis=-1)


def measure_perfect_identity(m):
    """PRM3DMeasurer.MeasurePerfect (PRM3DMeasurer.cs:138-149) for the identity pose."""
    m = np.asarray(m, float)
    rng_ = np.sign(m[..., 2]) * np.linalg.norm(m, axis=-1)
    return np.stack([FOCAL * m[..., 0] / m[..., 2], FOCAL * m[..., 1] / m[..., 2], rng_], axis=-1)


class Frame:
    """One synthetic frame: P particle poses, a C-component prior mixture per particle and M measurements."""

    def __init__(self, P, C, M, seed, detect_fraction=0.9, mean_jitter=1e-2, weight_profile="survey", shard=0):
        """weight_profile: "survey" = w ~ U(0.05, 1.2) for every component (the expected map size
        then exceeds the detections several times over and every WeightAlpha underflows to 0);
        "steady" = detected components U(0.6, 1.2), the others U(0.002, 0.06): a map consistent with the
        frame, finite particle weights, depletion and resampling."""
        rng = np.random.default_rng(seed)
        # per-particle draws of shard (rank) > 0 come from a stream of their own; the shared stream is drawn from all the
        # same, so that the map, the weights and the measurements behind it are those of shard 0 on every rank
        prng = np.random.default_rng([seed, shard]) if shard else None
        self.P, self.C, self.M = P, C, M
        # particle poses: base (identity) + one 30 Hz odometry-noise step, Q = diag(5e-3 x3, 2e-4 x3) (Config.cs:244-249)
        dt = 1.0 / 30
        dloc = rng.normal(size=(P, 3)) * np.sqrt(5e-3) * dt
        drot = rng.normal(size=(P, 3)) * np.sqrt(2e-4) * dt
        if prng is not None:
            dloc = prng.normal(size=(P, 3)) * np.sqrt(5e-3) * dt
            drot = prng.normal(size=(P, 3)) * np.sqrt(2e-4) * dt
        q = np.concatenate([np.ones((P, 1)), 0.5 * drot], axis=1)
        q /= np.linalg.norm(q, axis=1, keepdims=True)
        self.poses = np.concatenate([dloc, q], axis=1)
        # prior: means uniform in the view frustum, shared by the particles up to a small jitter
        zc = np.stack([rng.uniform(-300, 300, C), rng.uniform(-220, 220, C), rng.uniform(0.3, 1.8, C)], axis=1)
        base = measure_to_map_identity(zc)
        A = rng.uniform(-0.05, 0.05, size=(C, 3, 3))
        cov = A @ np.transpose(A, (0, 2, 1)) + 1e-4 * np.eye(3)
        w = rng.uniform(0.05, 1.2, C)
        jitter = rng.normal(size=(P, C, 3))
        if prng is not None:
            jitter = prng.normal(size=(P, C, 3))
        self.mean = base[None] + jitter * mean_jitter
        self.cov = np.broadcast_to(cov, (P, C, 3, 3))
        self.w = np.broadcast_to(w, (P, C))
        self.counts = np.full(P, C, np.int32)
        self.weights = np.full(P, 1.0 / P)
        # measurements: detections of distinct components (+ N(0, R)) and clutter uniform in the FOV
        nd = min(int(np.ceil(detect_fraction * M)), C)
        pick = rng.choice(C, size=nd, replace=False)
        zdet = measure_perfect_identity(base[pick]) + rng.normal(size=(nd, 3)) * np.sqrt([2.0, 2.0, 1e-3])
        nc = M - nd
        zclu = np.stack([rng.uniform(-320, 320, nc), rng.uniform(-240, 240, nc),
                         rng.uniform(float(np.float32(0.1)), 2.0, nc)], axis=1)
        self.z = np.concatenate([zdet, zclu], axis=0)[rng.permutation(M)] if M else np.zeros((0, 3))
        if weight_profile == "steady":
            w = rng.uniform(0.002, 0.06, C)
            w[pick] = rng.uniform(0.6, 1.2, nd)
            self.w = np.broadcast_to(w, (P, C))
        elif weight_profile != "survey":
            raise ValueError("unknown weight profile %r" % (weight_profile,))

    def planes(self, stride=None):
        """[10][P][stride] in the device layout: w, mean xyz, cov xx xy xz yy yz zz."""
        stride = stride or self.C
        out = np.zeros((10, self.P, stride))
        out[0, :, :self.C] = self.w
        for k in range(3):
            out[1 + k, :, :self.C] = self.mean[:, :, k]
        for t, (i, j) in enumerate([(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]):
            out[4 + t, :, :self.C] = self.cov[:, :, i, j]
        return out

    def map(self, i):
        return np.array(self.w[i]), np.array(self.mean[i]), np.array(self.cov[i])


def frame_for(config):
    P, C, M, seed = CONFIGS[config]
    return Frame(P, C, M, seed)
