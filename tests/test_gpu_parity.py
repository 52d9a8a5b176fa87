"""Parity of the HIP path (through the C-ABI of libphdhip.so) with the CPU oracle, on seeded synthetic
frames at sizes the oracle finishes in seconds. FP64 everywhere; tolerances as stated in SURVEY §8d:
  predict / correct : rel 1e-9 + abs 1e-12        prune / merge : rel 1e-7 (order-dependent sums)
  particle weights  : rel 1e-6                    resample indices, BestParticle : exact
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import PHD_GATE_DISABLED, PHD_GATE_EUCLIDEAN, prm3d_defaults
from monorfs_amd.synth import Frame


@pytest.fixture(scope="module")
def nav_mod():
    from monorfs_amd import navigator
    return navigator


def make_nav(navigator, frame, maxq=600, **over):
    p = prm3d_defaults(max_particles=frame.P, max_components=max(maxq, frame.C), max_measurements=max(frame.M, 1))
    p.max_quantity = maxq
    for k, v in over.items():
        setattr(p, k, v)
    nav = navigator.PHDNavigator(p, particlecount=frame.P)
    nav.upload_state(frame.planes(), frame.counts, frame.poses, frame.weights)
    return nav, p


def close(a, b, rtol, atol=1e-12):
    return np.allclose(a, b, rtol=rtol, atol=atol)


def assert_mix_close(got, exp, rtol, what):
    gw, gm, gc = got
    ew, em, ec = exp
    assert len(gw) == len(ew), "%s: %d components, oracle has %d" % (what, len(gw), len(ew))
    assert close(gw, ew, rtol), "%s: weights differ (max rel %g)" % (what, np.max(np.abs(gw - ew) / np.maximum(np.abs(ew), 1e-300)))
    assert close(gm, em, rtol, 1e-11), "%s: means differ" % what
    # the device keeps the upper triangle of the (un-symmetrised) reference covariance
    iu = np.triu_indices(3)
    assert close(gc[:, iu[0], iu[1]], ec[:, iu[0], iu[1]], rtol, 1e-13), "%s: covariances differ" % what


def match_unordered(got, exp, rtol):
    """every oracle component is found exactly once in the device set"""
    gw, gm, gc = got
    ew, em, ec = exp
    assert len(gw) == len(ew), "%d emitted, oracle keeps %d" % (len(gw), len(ew))
    key_g = np.concatenate([gw[:, None], gm], axis=1)
    key_e = np.concatenate([ew[:, None], em], axis=1)
    used = np.zeros(len(gw), bool)
    iu = np.triu_indices(3)
    for i in range(len(ew)):
        d = np.max(np.abs(key_g - key_e[i]) / (np.abs(key_e[i]) + 1e-9), axis=1)
        d[used] = np.inf
        j = int(np.argmin(d))
        assert d[j] < 1e-8, "oracle component %d (w=%g) has no device twin (best %g)" % (i, ew[i], d[j])
        assert close(gc[j][iu], ec[i][iu], rtol, 1e-13)
        used[j] = True


CASES = [
    # P, C, M, seed, profile, overrides
    (6, 40, 12, 11, "steady", {}),
    (5, 130, 32, 12, "survey", {}),
    (4, 300, 70, 13, "steady", {}),                      # two measurement blocks, two component tiles
    (3, 64, 130, 14, "steady", {}),                      # four measurement blocks
    (4, 90, 24, 15, "steady", {"gate_metric": PHD_GATE_EUCLIDEAN}),
    (3, 50, 10, 16, "steady", {"gate_metric": PHD_GATE_DISABLED}),
    (3, 0, 9, 17, "steady", {}),                         # empty map: every measurement is born
    (3, 33, 0, 18, "steady", {}),                        # no measurements
    # MinWeight 0: every gated pair is a candidate, the candidate queue of k_sweep overflows and the
    # all-pairs fallback runs; nothing is cut before the merge
    (2, 30, 24, 19, "steady", {"min_weight": 0.0, "emit_capacity": 4096}),
]


@pytest.mark.parametrize("P,C,M,seed,profile,over", CASES)
def test_stage_parity(nav_mod, P, C, M, seed, profile, over):
    stage_parity(nav_mod, P, C, M, seed, profile, over)


@pytest.mark.parametrize("chain_max", [None, 0])
@pytest.mark.parametrize("P,C,M,seed,profile,over", [CASES[1], (6, 130, 32, 12, "steady", {}), CASES[2], CASES[3], CASES[4], CASES[8]])
def test_stage_parity_in_the_timed_mode(nav_mod, monkeypatch, chain_max, P, C, M, seed, profile, over):
    """the same stages against the oracle with phd_set_all_pairs(1) + phd_set_frozen(1) — the mode bench.py times —, through
    the one-launch chain (the default at these sizes) and through the separate kernels on two streams (PHD_CHAIN_MAX=0)"""
    if chain_max is not None:
        monkeypatch.setenv("PHD_CHAIN_MAX", str(chain_max))
        monkeypatch.setenv("PHD_SPLIT", "2")
    stage_parity(nav_mod, P, C, M, seed, profile, over, timed_mode=True)


def stage_parity(nav_mod, P, C, M, seed, profile, over, timed_mode=False):
    f = Frame(P, C, M, seed, weight_profile=profile) if C > 0 else Frame(P, 1, M, seed, weight_profile="survey")
    if C == 0:
        f.counts[:] = 0
    nav, p = make_nav(nav_mod, f, **over)
    if timed_mode:
        nav.set_frozen(True)
        nav.set_all_pairs(True)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    setll = nav.SetLogLikelihood()
    for i in range(P):
        prior = f.map(i) if C > 0 else (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3)))
        pred = orc.predict(p, f.poses[i], f.z, prior)
        assert_mix_close(nav.PredictConditional(i), pred, 1e-9, "predict[%d]" % i)
        cor = orc.correct(p, f.poses[i], f.z, pred)
        keep = ~(cor[0] < p.min_weight)
        match_unordered(nav.CorrectConditional(i), tuple(x[keep] for x in cor), 1e-9)
        pr = orc.prune(p, cor)
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
        a, sll = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        assert np.isclose(setll[i], sll, rtol=1e-9, atol=1e-9), "set log-likelihood[%d]: %r vs %r" % (i, setll[i], sll)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0), "alpha[%d]: %r vs %r" % (i, alpha[i], a)
    nav.close()


def test_merge_heavy_prune(nav_mod):
    # many near-duplicate components so that the greedy merge absorbs most of them
    rng = np.random.default_rng(3)
    f = Frame(4, 200, 16, 21, weight_profile="steady")
    f.mean = np.array(f.mean)
    centres = f.mean[:, :20]
    f.mean[:, :] = np.repeat(centres, 10, axis=1) + rng.normal(size=f.mean.shape) * 2e-3
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=False)
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        pr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        got = nav.PruneModel(i)
        assert len(pr[0]) < 150
        assert_mix_close(got, pr, 1e-7, "prune[%d]" % i)
    nav.close()


def test_max_quantity_cap(nav_mod):
    f = Frame(3, 256, 32, 22, weight_profile="survey")
    nav, p = make_nav(nav_mod, f, maxq=100)
    nav.run_stages(f.z, with_alpha=True)
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        pr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert len(pr[0]) <= 100
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
    nav.close()


@pytest.mark.parametrize("P,C,M,seed", [(16, 48, 12, 31), (64, 100, 24, 32)])
def test_slam_update_sequence(nav_mod, P, C, M, seed):
    """three full steps (predict, correct, prune, reweight, normalise, resample) against the oracle"""
    f = Frame(P, C, M, seed, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    st = orc.State(P, 700)
    st.poses[:] = f.poses
    st.w[:, :C], st.mean[:, :C], st.cov[:, :C], st.n[:] = f.w, f.mean, f.cov, C
    rng = np.random.default_rng(seed)
    nres = 0
    for step in range(3):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        u = float(rng.uniform(0.05, 0.95))
        best, src, res, _ = orc.slam_update(p, st, z, u=u, threads=4)
        nav.SlamUpdate(None, z, u_resample=u)
        gsrc, gres = nav.resample_sources()
        assert gres == res, "step %d: resampled %r vs oracle %r" % (step, gres, res)
        assert np.array_equal(gsrc, src), "step %d: resample sources differ" % step
        assert nav.BestParticle == best
        assert close(nav.VehicleWeights, st.weights, 1e-6, 1e-300), "step %d: particle weights" % step
        assert close(nav.poses(), st.poses, 1e-15)
        for i in (0, P // 2, P - 1):
            assert_mix_close(nav.MapModel(i), st.map(i), 1e-7, "step %d map[%d]" % (step, i))
        # the acceptance metric of SURVEY 8d: OSPA (C = 1, P = 1; Plot.cs:531-581) between the best particle's map
        # estimate on the device and in the oracle stays within 1e-4
        glm, _ = orc.best_map_estimate(nav.MapModel(nav.BestParticle))
        olm, _ = orc.best_map_estimate(st.map(best))
        d, card = orc.ospa(glm, olm)
        assert card == 0 and d < 1e-4, "step %d: OSPA %g (cardinality part %g)" % (step, d, card)
        nres += res
    assert nres > 0, "the sequence never resampled: the gather path was not exercised"
    nav.close()


def test_only_mapping(nav_mod):
    f = Frame(1, 60, 14, 41, weight_profile="steady")
    p = prm3d_defaults(max_particles=1, max_components=600, max_measurements=f.M)
    nav = nav_mod.PHDNavigator(p, particlecount=20, onlymapping=True)
    assert nav.particle_count == 1
    nav.reset(f.poses[0], f.map(0), 1)
    st = orc.State(1, 700)
    st.poses[:] = f.poses
    st.w[:, :f.C], st.mean[:, :f.C], st.cov[:, :f.C], st.n[:] = f.w, f.mean, f.cov, f.C
    for _ in range(2):
        orc.slam_update(p, st, f.z, onlymapping=True)
        nav.SlamUpdate(None, f.z)
    assert_mix_close(nav.BestMapModel, st.map(0), 1e-7, "mapping")
    assert np.allclose(nav.VehicleWeights, [1.0])
    nav.close()


def test_reset_replicates_and_collapse(nav_mod):
    f = Frame(1, 30, 8, 42, weight_profile="steady")
    p = prm3d_defaults(max_particles=9, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=9)
    nav.reset(f.poses[0], f.map(0), 9)
    assert np.allclose(nav.VehicleWeights, 1.0 / 9)
    for i in (0, 4, 8):
        assert_mix_close(nav.MapModel(i), f.map(0), 1e-15, "reset[%d]" % i)
    assert np.allclose(nav.poses(), np.repeat((f.poses[0] if True else None)[None], 9, 0))
    nav.CollapseParticles(3)
    assert nav.particle_count == 3
    assert_mix_close(nav.MapModel(2), f.map(0), 1e-15, "collapse")
    nav.close()


def test_resample_bit_exact(nav_mod):
    p = prm3d_defaults(max_particles=4, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=4)
    rng = np.random.default_rng(5)
    for P in (5, 257, 2048):
        w = rng.random(P) ** 8
        w /= w.sum()
        for u in (1e-12, 0.25, 0.5, 0.999999):
            src, best = nav.ResampleParticles(w, u)
            osrc, obest = orc.resample(w, u)
            assert np.array_equal(src, osrc) and best == obest
        assert nav.ParticleDepleted(w) == orc.particle_depleted(p, w)
    kat = [0.11, 0.28, 0.31, 0.01, 0.29]   # SimulationTest.cs:225-270
    for u in np.linspace(0.001, 0.999, 97):
        src, best = nav.ResampleParticles(kat, float(u))
        assert src[best] == 2 and {1, 2, 4} <= set(src.tolist())
    nav.close()


def test_capacity_error_is_loud(nav_mod):
    f = Frame(2, 200, 32, 51, weight_profile="survey")
    p = prm3d_defaults(max_particles=2, max_components=600, max_measurements=32)
    p.emit_capacity = 64
    p.max_quantity = 64
    nav = nav_mod.PHDNavigator(p, particlecount=2)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    with pytest.raises(nav_mod.PHDError) as e:
        nav.SlamUpdate(None, f.z)
    assert e.value.status == 2
    nav.close()


def clustered_frame(seed, groups, per_group, M_per_group, spread_px=4.0):
    """landmarks packed a few pixels apart so that the association graph has clusters of more than 5 rows"""
    from monorfs_amd.synth import measure_to_map_identity
    rng = np.random.default_rng(seed)
    zc, zs = [], []
    for _ in range(groups):
        c = np.array([rng.uniform(-250, 250), rng.uniform(-180, 180), rng.uniform(0.5, 1.5)])
        for _ in range(per_group):
            zc.append(c + rng.normal(size=3) * [spread_px, spread_px, 0.02])
        for _ in range(M_per_group):
            zs.append(c + rng.normal(size=3) * [spread_px, spread_px, 0.02])
    zc, zs = np.array(zc), np.array(zs)
    C = len(zc)
    f = Frame(3, C, len(zs), seed, weight_profile="steady")
    base = measure_to_map_identity(zc)
    f.mean = base[None] + rng.normal(size=(f.P, C, 3)) * 1e-3
    cov = np.diag([2e-4, 2e-4, 4e-4])
    f.cov = np.broadcast_to(cov, (f.P, C, 3, 3))
    f.w = np.broadcast_to(rng.uniform(0.9, 1.1, C), (f.P, C))
    f.z = zs
    f.M = len(zs)
    return f


@pytest.mark.parametrize("seed,groups,per_group,mpg", [(61, 2, 3, 3), (62, 3, 4, 3), (63, 1, 6, 7), (64, 4, 2, 4), (65, 2, 5, 5)])
def test_big_association_clusters(nav_mod, seed, groups, per_group, mpg):
    """clusters with more than 5 rows go through Murty's ranked assignments on the device, including
    the reference's early exit on the stale logcomp entry (PHDNavigator.cs:503)"""
    f = clustered_frame(seed, groups, per_group, mpg)
    nav, p = make_nav(nav_mod, f, merge_threshold=1e-3)
    nav.run_stages(f.z, with_alpha=True)
    setll, alpha = nav.SetLogLikelihood(), nav.WeightAlpha()
    biggest = 0
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        pr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
        lm, _ = orc.best_map_estimate(pr)
        v, ncl, mx = orc.set_log_likelihood(p, f.poses[i], lm, f.z)
        biggest = max(biggest, mx)
        assert np.isclose(setll[i], v, rtol=1e-9, atol=1e-9), "set log-likelihood[%d]: %r vs %r (largest cluster %d)" % (i, setll[i], v, mx)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0)
    assert biggest > 5, "the frame did not produce a cluster with more than 5 rows (largest %d)" % biggest
    nav.close()


def test_motion_update_matches_oracle(nav_mod):
    """TrackVehicle.UpdateNoisy on the device (phd_update_motion) against the oracle, including the
    branches of Quaternion.Exp (no rotation) and the PerfectStill shortcut."""
    rng = np.random.default_rng(77)
    f = Frame(300, 4, 3, 71, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    poses = f.poses.copy()
    poses[:, 3:] += rng.normal(0, 0.3, (f.P, 4))
    poses[:, 3:] /= np.linalg.norm(poses[:, 3:], axis=1, keepdims=True)
    nav.set_poses(poses)
    for reading, perfect in ((np.array([0.02, -0.01, 0.03, 0.01, -0.02, 0.015]), False),
                             (np.array([0.1, 0.0, 0.0, 0.0, 0.0, 0.0]), False),       # translation only: Exp takes its identity branch
                             (np.zeros(6), True), (np.zeros(6), False)):
        noise = rng.normal(0, 1, (f.P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
        noise[::7, 3:] = 0                                                            # noise without rotation for some particles
        want = orc.update_motion(poses, reading, noise, perfect)
        nav.UpdateOdometry(None, reading, noise, perfect_still=perfect)
        got = nav.poses()
        assert np.allclose(got, want, rtol=0, atol=1e-14), np.max(np.abs(got - want))
        poses = want
    nav.UpdateOdometry(None, [0.01, 0, 0, 0, 0.02, 0])                                # no noise vector at all
    assert np.allclose(nav.poses(), orc.update_motion(poses, [0.01, 0, 0, 0, 0.02, 0]), rtol=0, atol=1e-14)
    nav.close()


@pytest.mark.parametrize("J,M,seed", [(3, 4, 81), (40, 30, 82), (12, 70, 83), (300, 64, 84), (0, 5, 85), (6, 0, 86)])
def test_quasi_set_log_likelihood_batch(nav_mod, J, M, seed):
    """QuasiSetLogLikelihood for a batch of candidate poses (phd_quasi_set_loglik) against the oracle:
    small and large clusters (the gate of 12 joins more measurements than the gate of 5), a landmark set larger than
    the LDS-resident limit, empty sets."""
    rng = np.random.default_rng(seed)
    f = Frame(48, max(J, 1), max(M, 1), seed, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    lm = f.mean[0, :J].copy()
    z = f.z[:M].copy()
    if M > 3 and J > 3:
        z[1] = z[0] + [3.0, -2.0, 0.01]          # measurements crowding one landmark: clusters beyond 5 rows
        z[2] = z[0] + [-4.0, 1.0, -0.02]
    poses = f.poses.copy()
    poses[:, :3] += rng.normal(0, 5e-3, (f.P, 3))
    poses[:, 3:] += rng.normal(0, 2e-3, (f.P, 4))
    got = nav.QuasiSetLogLikelihood(z, lm, poses)
    want = np.array([orc.quasi_set_log_likelihood(p, poses[i], lm, z) for i in range(f.P)])
    assert np.allclose(got, want, rtol=1e-9, atol=1e-9), np.max(np.abs(got - want))
    nav.close()


@pytest.mark.parametrize("J,M,seed", [(3, 4, 181), (40, 30, 182), (12, 70, 183), (300, 64, 184), (0, 5, 185), (6, 0, 186), (2, 3, 187),
                                      (64, 60, 188), (110, 100, 189)])   # ... more clusters than one run of the ordered pass takes (32), than its LDS copy holds (64)
@pytest.mark.parametrize("mode", [0, 1])
def test_quasi_set_log_likelihood_gradient_batch(nav_mod, J, M, seed, mode):
    """Row f4, gradient part: QuasiSetLogLikelihood(..., out gradient) (PHDNavigator.cs:543-713) for a batch of poses
    (phd_quasi_set_loglik_grad) against the oracle, in both readings of TemperedAverage: clusters of every size (the
    literal enumeration up to 5 rows, Murty beyond, whose cut reads the entries TemperedAverage rewrote), a map below 5
    landmarks (`modelsize` cuts the enumeration), a landmark set beyond the LDS-resident limit, empty sets."""
    rng = np.random.default_rng(seed)
    f = Frame(48, max(J, 1), max(M, 1), seed, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    lm = f.mean[0, :J].copy()
    z = f.z[:M].copy()
    if M > 3 and J > 3:
        z[1] = z[0] + [3.0, -2.0, 0.01]
        z[2] = z[0] + [-4.0, 1.0, -0.02]
    poses = f.poses.copy()
    poses[:, :3] += rng.normal(0, 5e-3, (f.P, 3))
    poses[:, 3:] += rng.normal(0, 2e-3, (f.P, 4))
    got, ggot = nav.QuasiSetLogLikelihoodGradient(z, lm, poses, average_mode=mode)
    want = [orc.quasi_set_log_likelihood_grad(p, poses[i], lm, z, mode) for i in range(f.P)]
    wv, wg = np.array([w[0] for w in want]), np.array([w[1] for w in want])
    assert np.allclose(got, wv, rtol=1e-9, atol=1e-9), np.max(np.abs(got - wv))
    assert np.allclose(ggot, wg, rtol=1e-8, atol=1e-7), np.max(np.abs(ggot - wg))
    if J and M:
        assert np.max(np.abs(wg)) > 1
    nav.close()


def test_loglike_gradient_through_the_batch(nav_mod):
    """LoopyPHDNavigator.LogLikeGradient (LoopyPHDNavigator.cs:876-909): the 12 finite-difference evaluations of a pose
    (here of 5 poses at once) as one device batch, against the oracle's restatement"""
    from monorfs_amd.navigator import pose3d_add
    rng = np.random.default_rng(91)
    f = Frame(64, 30, 20, 91, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    lm, z, lin = f.mean[0, :25].copy(), f.z.copy(), f.poses[0].copy()
    assert np.allclose(pose3d_add(lin, [0.01, -0.02, 0.03, 0.02, -0.01, 0.015]), orc.pose_add(lin, [0.01, -0.02, 0.03, 0.02, -0.01, 0.015]), atol=1e-15)
    poses = rng.normal(0, 1, (5, 6)) * [2e-3, 2e-3, 2e-3, 1e-3, 1e-3, 1e-3]
    got = nav.LogLikeGradient(poses, z, lm, lin)
    for a in range(5):
        want = orc.loglike_gradient(p, poses[a], lin, lm, z)
        assert np.allclose(got[a], want, rtol=1e-4, atol=0.5), (got[a], want)
    assert np.max(np.abs(got)) > 10          # the gradients are far from zero: the comparison means something
    nav.close()


def test_resampling_leaves_the_maps_in_place(nav_mod):
    """A resampling step copies no mixture: the next step reads particle p's map at its source's slot in the bank the
    step before wrote (k_gather_rotate / `inslot`). Steps that follow each other through that indirection must equal
    steps from a state gathered into place after every step (download = k_materialise), bit for bit, and the oracle
    within tolerance; single-map writes and the frozen mode's getters go through the same indirection."""
    P, C, M = 48, 60, 16
    f = Frame(P, C, M, 77, weight_profile="steady")
    nav_a, p = make_nav(nav_mod, f)
    nav_b, _ = make_nav(nav_mod, f)
    st = orc.State(P, 700)
    st.poses[:] = f.poses
    st.w[:, :C], st.mean[:, :C], st.cov[:, :C], st.n[:] = f.w, f.mean, f.cov, C
    rng = np.random.default_rng(78)
    consecutive, last = 0, False
    for step in range(5):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        u = float(rng.uniform(0.05, 0.95))
        _, src, res, _ = orc.slam_update(p, st, z, u=u, threads=4)
        nav_a.SlamUpdate(None, z, u_resample=u)            # only single-map getters in between: the indirection stays
        nav_b.SlamUpdate(None, z, u_resample=u)
        planes_b, counts_b, poses_b, weights_b = nav_b.download_state(600)   # gathers the maps into place
        gsrc, gres = nav_a.resample_sources()
        assert gres == res and np.array_equal(gsrc, src)
        consecutive += int(last)                           # this step read its input through the previous resampling's slots
        last = res
        for i in range(P):
            wa, ma, ca = nav_a.MapModel(i)
            n = counts_b[i]
            assert len(wa) == n and np.array_equal(wa, planes_b[0, i, :n]), "step %d particle %d" % (step, i)
            assert np.array_equal(ma, np.stack([planes_b[1 + t, i, :n] for t in range(3)], axis=1))
            if i % 7 == 0:
                assert_mix_close((wa, ma, ca), st.map(i), 1e-7, "step %d map[%d]" % (step, i))
        assert np.array_equal(nav_a.VehicleWeights, weights_b) and np.array_equal(nav_a.poses(), poses_b)
    assert consecutive >= 1, "no step read its input through the slots of a resampling"
    # a single-map write after a resampling: the particle's slot was shared with the other copies of its source
    nav_a.set_map(3, f.map(0))
    nav_b.set_map(3, f.map(0))
    for i in range(P):
        wa, ma, ca = nav_a.MapModel(i)
        wb, mb, cb = nav_b.MapModel(i)
        assert np.array_equal(wa, wb) and np.array_equal(ma, mb) and np.array_equal(ca, cb)
    # frozen mode: the step's result is read through the resampling sources, the state itself does not move
    before = nav_a.download_state(600)
    nav_a.set_frozen(True)
    nav_b.set_frozen(False)
    z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
    nav_a.SlamUpdate(None, z, u_resample=0.37)
    nav_b.SlamUpdate(None, z, u_resample=0.37)
    assert nav_a.resample_sources()[1] == nav_b.resample_sources()[1]
    for i in range(0, P, 5):
        wa, ma, ca = nav_a.MapModel(i)
        wb, mb, cb = nav_b.MapModel(i)
        assert np.array_equal(wa, wb) and np.array_equal(ma, mb) and np.array_equal(ca, cb)
    assert np.array_equal(nav_a.VehicleWeights, nav_b.VehicleWeights)
    nav_a.set_frozen(False)
    after = nav_a.download_state(600)
    for x, y in zip(before, after):
        assert np.array_equal(x, y)
    nav_a.close()
    nav_b.close()


def test_reweight_far_from_the_origin(nav_mod):
    """A scene 2 km from the origin: Gaussian.Merge's raw second moments (Gaussian.cs:336-344) then cost a pruned
    singleton about eps |m|^2 = 5e-10 of its covariance entries, so k_prune_merge must NOT declare the corrected map's
    misdetection copies equal to their predicted components, and k_alpha_density evaluates them itself — against the oracle.

    The tolerance on alpha here is the reference formula's own conditioning, measured: the rounding residue of
    fl(fl(w (P + m m')) / w) - m m' is a pseudo-random function of the low bits of w, and the weights of the detection
    components go through exp and a 50-term sum (device and oracle agree to ~1e-13, not to the bit — nor would the CLR's
    Math.Exp with glibc's). The oracle is therefore re-run 48 times with those weights moved by a relative 1e-13: its
    alpha scatters with a standard deviation of 2-3e-4 at this distance, whatever the size of the move (1e-15 ... 1e-11
    give the same scatter; at the origin it is 0 and the same check holds to 1e-6 in test_stage_parity), and the device
    must sit within 5 standard deviations. Stage by stage, on identical inputs, everything is checked tightly in
    tests/test_gpu_round2.py::test_far_from_the_origin_stage_by_stage (alpha to 1e-6 there)."""
    P, C, M = 4, 50, 14
    f = Frame(P, C, M, 91, weight_profile="steady")
    shift = np.array([1500.0, -900.0, 1100.0])
    f.poses = f.poses.copy()
    f.poses[:, :3] += shift
    f.mean = f.mean + shift
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    rng = np.random.default_rng(5)
    for i in range(P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        cor = orc.correct(p, f.poses[i], f.z, pred)
        pr = orc.prune(p, cor)
        got = nav.PruneModel(i)
        assert len(got[0]) == len(pr[0])
        assert np.allclose(got[0], pr[0], rtol=1e-7) and np.allclose(got[1], pr[1], rtol=1e-12, atol=1e-9)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        draws = []
        for _ in range(48):
            wp = cor[0] * (1 + 1e-13 * rng.normal(size=len(cor[0])))
            wp[:len(pred[0])] = cor[0][:len(pred[0])]            # the misdetection copies' weights are exact on both sides
            ap, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, orc.prune(p, (wp, cor[1], cor[2])))
            draws.append((ap - a) / a)
        sigma = float(np.std(draws))
        print("alpha[%d]: device %r oracle %r, rel %.3g; scatter of the oracle under 1e-13 weight moves: sigma %.3g" % (i, alpha[i], a, abs(alpha[i] - a) / a, sigma))
        assert abs(alpha[i] - a) / a <= max(1e-6, 5 * sigma), "alpha[%d]: %r vs %r (sigma %g)" % (i, alpha[i], a, sigma)
    nav.close()


@pytest.mark.parametrize("nborn", [5, 30])
def test_births_with_the_density_out_of_the_pair_loop(nav_mod, nborn):
    """k_sweep takes the Explored density out of its pair loop once at most 16 measurements are unexplored and sums
    those component-per-lane: a frame with a few births (the switch happens, the births are decided by the second
    form) and one with more births than that (the switch never happens), 400 components = 4 tiles, against the oracle."""
    P, C, M = 3, 400, 64
    f = Frame(P, C, M, 95, weight_profile="steady")
    p0 = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
    f.w = f.w.copy()
    picked = np.arange(0, 2 * nborn, 2)[:nborn]                     # measurements that must find nothing around them
    for i in range(P):
        for k in picked:
            x = orc.measure_to_map(p0, f.poses[i], f.z[k])
            f.w[i, np.linalg.norm(f.mean[i] - x, axis=1) < 0.45] = 1e-40
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    for i in range(P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        assert len(pred[0]) - C >= nborn
        assert_mix_close(nav.PredictConditional(i), pred, 1e-9, "predict[%d]" % i)
        cor = orc.correct(p, f.poses[i], f.z, pred)
        pr = orc.prune(p, cor)
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0), "alpha[%d]: %r vs %r" % (i, alpha[i], a)
    nav.close()


@pytest.mark.parametrize("dups,indefinite", [(40, False), (10, True)])
def test_prune_crowded_cells_and_rows_without_a_bound(nav_mod, dups, indefinite):
    """k_prune_merge hands a row with more than 32 candidates, and any row whose covariance gives no Euclidean bound (not
    positive definite: it must be tested against every later row), to a whole wave: 5 centres x 40 near-duplicates put
    ~100 rows into one cell; an indefinite covariance takes the second road. Against the oracle."""
    rng = np.random.default_rng(5)
    f = Frame(3, 200, 16, 23, weight_profile="steady")
    f.mean = np.array(f.mean)
    f.cov = np.array(f.cov)
    centres = f.mean[:, :200 // dups]
    f.mean[:, :] = np.repeat(centres, dups, axis=1) + rng.normal(size=f.mean.shape) * 2e-3
    if indefinite:
        f.cov[:, 7] = np.diag([2e-3, -1e-3, 1.5e-3])
        f.cov[:, 90] = np.array([[1e-3, 2e-3, 0], [2e-3, 1e-3, 0], [0, 0, 1e-3]])
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=False)
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        pr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
    nav.close()
