"""GPU parity cases added in round 2 (through the C-ABI, against the CPU oracle):
  * BASELINE config A (256 x 128 x 32) at its stated size, the whole step for every particle;
  * ResampleParticles on weight vectors of 16 384 - 32 768 entries (config C8's global vector; both the LDS-staged and
    the global-memory branch of k_normalise_resample, and the sizes around the switch);
  * the setters right behind an asynchronous step (the bank roles rotate on the device);
  * a failed step (emit capacity) leaves the state as it was;
  * means and covariances of CorrectConditional bit for bit as the oracle's (the reference's arithmetic is compiled
    without FP contraction on the device, as the CLR runs it)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame


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


def oracle_state(f, cap=700):
    st = orc.State(f.P, cap)
    st.poses[:] = f.poses
    st.w[:, :f.C], st.mean[:, :f.C], st.cov[:, :f.C], st.n[:] = f.w, f.mean, f.cov, f.C
    return st


IU = np.triu_indices(3)


def twin_index(means, covs):
    """components by the bits of (mean, covariance upper triangle); a key may hold several (a birth updated by the very
    measurement it was born from keeps its mean when the innovation rounds to zero)"""
    index = {}
    for j in range(len(means)):
        index.setdefault(means[j].tobytes() + np.ascontiguousarray(covs[j][IU]).tobytes(), []).append(j)
    return index


def twin_pop(index, mean, cov):
    lst = index.get(mean.tobytes() + np.ascontiguousarray(cov[IU]).tobytes())
    return lst.pop() if lst else None


def assert_map_close(got, exp, rtol, what):
    (gw, gm, gc), (ew, em, ec) = got, exp
    assert len(gw) == len(ew), "%s: %d components, oracle has %d" % (what, len(gw), len(ew))
    assert np.allclose(gw, ew, rtol=rtol, atol=1e-12), what
    assert np.allclose(gm, em, rtol=rtol, atol=1e-11), what
    assert np.allclose(gc[:, IU[0], IU[1]], ec[:, IU[0], IU[1]], rtol=rtol, atol=1e-13), what


def test_config_A_full_size_step(nav_mod):
    """BASELINE config A, 256 particles x 128 components x 32 measurements: two full steps, every particle against the
    oracle (weights 1e-6, resampling sources and BestParticle exact, every map 1e-7, OSPA of the best map 1e-4)."""
    P, C, M, seed = CONFIGS["A"]
    f = Frame(P, C, M, seed, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    st = oracle_state(f)
    rng = np.random.default_rng(seed)
    for step in range(2):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.2 * step
        u = float(rng.uniform(0.05, 0.95))
        best, src, res, _ = orc.slam_update(p, st, z, u=u, threads=8)
        nav.SlamUpdate(None, z, u_resample=u)
        gsrc, gres = nav.resample_sources()
        assert gres == res and np.array_equal(gsrc, src), "step %d: resampling differs" % step
        assert nav.BestParticle == best
        assert np.allclose(nav.VehicleWeights, st.weights, rtol=1e-6, atol=1e-300)
        for i in range(P):
            assert_map_close(nav.MapModel(i), st.map(i), 1e-7, "step %d map[%d]" % (step, i))
        glm, _ = orc.best_map_estimate(nav.MapModel(nav.BestParticle))
        olm, _ = orc.best_map_estimate(st.map(best))
        d, card = orc.ospa(glm, olm)
        assert card == 0 and d < 1e-4
    nav.close()


@pytest.mark.parametrize("P", [16384, 18300, 18400, 18432, 20000, 32768])
def test_resample_large_vectors(nav_mod, P):
    """ResampleParticles / ParticleDepleted on the global weight vector of a sharded run (C8: 16 384) and beyond, bit-exact
    against the sequential recurrence: the LDS-staged branch (up to ~18 350 weights beside the kernel's static arrays),
    the sizes around the switch, and the global-memory branch."""
    p = prm3d_defaults(max_particles=4, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=4)
    rng = np.random.default_rng(P)
    for power, u in ((8, 0.5), (1, 0.25), (30, 0.999999), (2, 1e-12)):
        w = rng.random(P) ** power
        w /= w.sum()
        src, best = nav.ResampleParticles(w, u)
        osrc, obest = orc.resample(w, u)
        assert np.array_equal(src, osrc), "P=%d power=%d u=%g: %d sources differ" % (P, power, u, np.count_nonzero(src != osrc))
        assert best == obest
        assert nav.ParticleDepleted(w) == orc.particle_depleted(p, w)
    nav.close()


def test_setters_right_behind_an_asynchronous_step(nav_mod):
    """phd_step_async rotates the bank roles on the device; phd_update_motion / phd_set_poses / phd_set_weights called
    right behind it (no phd_sync, no getter) must land in the NEW current state. Same sequence with a synchronisation
    after every call, and the oracle, as the references."""
    f = Frame(48, 60, 14, 207, weight_profile="steady")
    rng = np.random.default_rng(3)
    reading = np.array([0.02, -0.01, 0.03, 0.01, -0.02, 0.015])
    noise = rng.normal(0, 1, (f.P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
    z2 = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
    neww = rng.uniform(0.5, 1.5, f.P)
    neww /= neww.sum()

    def run(sync):
        nav, p = make_nav(nav_mod, f)
        nav.set_measurements(f.z)
        nav.step_async(0.37)
        if sync:
            nav.sync()
        nav.UpdateOdometry(None, reading, noise)
        if sync:
            nav.sync()
        nav.set_weights(neww)
        if sync:
            nav.sync()
        nav.set_measurements(z2)
        nav.step_async(0.61)
        nav.sync()
        out = (nav.VehicleWeights, nav.poses(), nav.resample_sources(), [nav.MapModel(i) for i in (0, 17, f.P - 1)], nav.BestParticle)
        nav.close()
        return out, p

    (wa, pa, sa, ma, ba), p = run(False)
    (wb, pb, sb, mb, bb), _ = run(True)
    assert np.array_equal(wa, wb) and np.array_equal(pa, pb) and np.array_equal(sa[0], sb[0]) and sa[1] == sb[1] and ba == bb
    for x, y in zip(ma, mb):
        assert all(np.array_equal(a, b) for a, b in zip(x, y))
    # and against the oracle
    st = oracle_state(f)
    orc.slam_update(p, st, f.z, u=0.37, threads=4)
    st.poses[:] = orc.update_motion(st.poses, reading, noise, False)
    st.weights[:] = neww
    best, src, res, _ = orc.slam_update(p, st, z2, u=0.61, threads=4)
    assert np.array_equal(sa[0], src) and sa[1] == res and ba == best
    assert np.allclose(wa, st.weights, rtol=1e-6, atol=1e-300)
    assert np.allclose(pa, st.poses, rtol=0, atol=1e-14)

    # phd_set_poses the same way
    nav, p = make_nav(nav_mod, f)
    nav.set_measurements(f.z)
    nav.step_async(0.37)
    poses2 = f.poses.copy()
    poses2[:, :3] += 0.01
    nav.set_poses(poses2)           # no synchronisation in between
    assert np.array_equal(nav.poses(), poses2)
    nav.close()


def test_a_failed_step_leaves_the_state_as_it_was(nav_mod):
    """A step whose corrected mixture outgrows emit_capacity reports PHD_ERR_CAPACITY and is dropped as a whole: the
    particle set (weights, poses, maps, BestParticle) is the one before the step — also for a second step queued behind
    the failed one —, and a step that fits runs from it."""
    small = Frame(6, 40, 6, 52, weight_profile="steady")
    p = prm3d_defaults(max_particles=small.P, max_components=600, max_measurements=32)
    p.emit_capacity = 64
    p.max_quantity = 64
    nav = nav_mod.PHDNavigator(p, particlecount=small.P)
    nav.upload_state(small.planes(), small.counts, small.poses, small.weights)
    nav.SlamUpdate(None, small.z, u_resample=0.3)           # a step that fits

    def snapshot():
        return (nav.VehicleWeights, nav.poses(), nav.BestParticle, [nav.MapModel(i) for i in range(small.P)])

    before = snapshot()
    rng = np.random.default_rng(1)
    many = np.column_stack([rng.uniform(-300, 300, 32), rng.uniform(-220, 220, 32), rng.uniform(0.3, 1.8, 32)])   # 32 births
    for attempt in range(2):                                 # twice: the failed phd_sync clears the flag
        nav.set_measurements(many)
        nav.step_async(0.5)
        nav.step_async(0.5)                                  # queued behind the failed one: dropped too
        with pytest.raises(nav_mod.PHDError) as e:
            nav.sync()
        assert e.value.status == 2
        after = snapshot()
        assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1]) and before[2] == after[2]
        for x, y in zip(before[3], after[3]):
            assert all(np.array_equal(a, b) for a, b in zip(x, y))
    nav.SlamUpdate(None, small.z[:3], u_resample=0.4)       # the handle still works, from the kept state
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    nav.close()


def test_corrected_means_and_covariances_are_bit_exact(nav_mod):
    """The measurement model, S^-1, the Kalman gain and update run without FP contraction on the device, in the
    reference's order of operations: every corrected component's mean and covariance equal the oracle's bit for bit
    (weights go through exp and the per-measurement sums: 1e-9)."""
    f = Frame(4, 150, 40, 211, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=False)
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        ew, em, ec = orc.correct(p, f.poses[i], f.z, pred)
        keep = ew >= p.min_weight
        ew, em, ec = ew[keep], em[keep], ec[keep]
        gw, gm, gc = nav.CorrectConditional(i)
        assert len(gw) == len(ew)
        # pair the two sets by the bits of mean and covariance (exact match expected), then compare the weights
        index = twin_index(gm, gc)
        nexact = 0
        for k in range(len(ew)):
            j = twin_pop(index, em[k], ec[k])
            assert j is not None, "particle %d: oracle component %d (w=%g) has no device twin with the same mean and covariance bits" % (i, k, ew[k])
            assert np.isclose(gw[j], ew[k], rtol=1e-9, atol=0)
            nexact += gw[j] == ew[k]
        assert nexact >= np.count_nonzero(ew < 0.2 * f.w[i].max()) // 4   # the misdetection copies inside the field of view: PD = 0.9 exactly
    nav.close()


def test_far_from_the_origin_stage_by_stage(nav_mod):
    """A scene 2 km from the origin, where Gaussian.Merge's raw second moments (Gaussian.cs:336-344) cost a pruned
    singleton eps |m|^2 ~ 5e-10 of its covariance entries whatever the machine: each stage is checked on the inputs the
    device itself produced, so that no stage's tolerance hides behind the conditioning of another.
      * CorrectConditional: means / covariances bit-exact, weights 1e-9 (as above);
      * PruneModel + Merge: the oracle's prune of the DEVICE's corrected list equals the device's pruned map to 1e-12
        (same arithmetic, no contraction: in practice bit for bit);
      * WeightAlpha: the oracle's alpha for (oracle predicted, DEVICE pruned map) within 1e-6 of the device's."""
    P, C, M = 4, 50, 14
    f = Frame(P, C, M, 91, weight_profile="steady")
    shift = np.array([1500.0, -900.0, 1100.0])
    f.poses = f.poses.copy()
    f.poses[:, :3] += shift
    f.mean = f.mean + shift
    nav, p = make_nav(nav_mod, f)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    for i in range(P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        ew, em, ec = orc.correct(p, f.poses[i], f.z, pred)
        keep = ew >= p.min_weight
        gw, gm, gc = nav.CorrectConditional(i)
        assert len(gw) == np.count_nonzero(keep)
        index = twin_index(gm, gc)
        order = []
        for k in np.flatnonzero(keep):                          # the device's list in the reference's (canonical) order
            j = twin_pop(index, em[k], ec[k])
            assert j is not None and np.isclose(gw[j], ew[k], rtol=1e-9), "particle %d: oracle component %d has no bit-identical device twin" % (i, k)
            order.append(j)
        dev_corrected = (gw[order], gm[order], ec[keep])        # full covariances from the oracle: equal bits in the upper triangle
        opr = orc.prune(p, dev_corrected)
        got = nav.PruneModel(i)
        assert len(got[0]) == len(opr[0])
        assert np.allclose(got[0], opr[0], rtol=1e-12) and np.allclose(got[1], opr[1], rtol=1e-13, atol=0)
        assert np.allclose(got[2][:, IU[0], IU[1]], opr[2][:, IU[0], IU[1]], rtol=1e-9, atol=1e-16)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, opr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0), "alpha[%d]: %r vs %r" % (i, alpha[i], a)
    nav.close()


# ---- association clusters beyond 32 rows (MurtyPairing has no size limit, GraphCombinatorics.cs:241-272) ---------------
@pytest.mark.parametrize("n", [33, 40, 64, 65, 100, 128, 129, 200])
def test_murty_pairing_beyond_32_rows(nav_mod, n):
    """The device's MurtyPairing on dense n x n matrices with missing entries, n up to 64 in the per-particle workspace
    (one row and column per lane), beyond it in a slab block (two or four per lane): the first 40 pairings, assignments
    and values, equal the oracle's (GraphCombinatorics.cs:64-272 restated) exactly."""
    p = prm3d_defaults(max_particles=1, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=1)
    rng = np.random.default_rng(700 + n)
    m = rng.uniform(-20, 0, (n, n))
    m[rng.uniform(size=m.shape) < 0.6] = -np.inf
    m[np.arange(n), np.arange(n)] = rng.uniform(-5, 0, n)       # a finite diagonal: solvable
    asg, val = nav.test_pairing(m, maxcount=40)
    oasg, oval = orc.murty(m, maxcount=40)
    assert len(asg) == len(oasg) == 40
    assert asg == oasg
    assert np.array_equal(val, oval)
    # no perfect matching at all: the first node is yielded unsolved with value -inf and has no children (:245-249, :473)
    m2 = m.copy()
    m2[n // 2, :] = -np.inf
    asg2, val2 = nav.test_pairing(m2, maxcount=5)
    assert len(asg2) == 1 and val2[0] == -np.inf and all(a == -1 for a in asg2[0])
    nav.close()


@pytest.mark.parametrize("seed,per_group,mpg,P,rows", [(66, 20, 20, 2, 41), (67, 30, 60, 1, 88), (69, 60, 150, 1, 172)])
def test_association_clusters_of_40_to_170_rows(nav_mod, seed, per_group, mpg, P, rows):
    """One cluster of 41 rows (the per-particle workspace, a row per lane), of 88 and of 172 rows (the association slab,
    two and four rows per lane): set log-likelihood and particle weight against the oracle, through the whole stage chain."""
    from test_gpu_parity import clustered_frame
    f = clustered_frame(seed, 1, per_group, mpg, spread_px=3.0)
    f.P = P
    f.poses, f.mean, f.cov, f.w, f.counts, f.weights = f.poses[:P], f.mean[:P], f.cov[:P], f.w[:P], f.counts[:P], np.full(P, 1.0 / P)
    nav, p = make_nav(nav_mod, f, merge_threshold=1e-3, emit_capacity=12000)
    nav.run_stages(f.z, with_alpha=True)
    setll, alpha = nav.SetLogLikelihood(), nav.WeightAlpha()
    biggest = 0
    for i in range(P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        pr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert_map_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
        lm, _ = orc.best_map_estimate(pr)
        v, ncl, mx = orc.set_log_likelihood(p, f.poses[i], lm, f.z)
        biggest = max(biggest, mx)
        assert np.isclose(setll[i], v, rtol=1e-9, atol=1e-9), "set log-likelihood[%d]: %r vs %r (largest cluster %d)" % (i, setll[i], v, mx)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0)
    assert biggest == rows, "the frame did not produce the large cluster it was built for (largest %d)" % biggest
    nav.close()


def test_association_workspace_exhausted_is_an_error_that_keeps_the_state(nav_mod):
    """Without an association slab a cluster of more than 64 rows cannot be solved: PHD_ERR_ASSOCIATION, the state stays
    (the step is dropped as a whole); with the slab back the same step runs."""
    from test_gpu_parity import clustered_frame
    f = clustered_frame(67, 1, 30, 60, spread_px=3.0)
    nav, p = make_nav(nav_mod, f, merge_threshold=1e-3, emit_capacity=12000)
    before = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])
    nav._check(nav._lib.phd_set_association_workspace(nav._h, 0))
    with pytest.raises(nav_mod.PHDError) as e:
        nav.SlamUpdate(None, f.z)
    assert e.value.status == 3 and e.value.module == "association"
    after = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])
    assert np.array_equal(before[0], after[0])
    for x, y in zip(before[1], after[1]):
        assert all(np.array_equal(a, b) for a, b in zip(x, y))
    nav._check(nav._lib.phd_set_association_workspace(nav._h, 64 << 20))
    nav.SlamUpdate(None, f.z)
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    nav.close()


def test_one_launch_chain_equals_the_separate_kernels(nav_mod, monkeypatch):
    """Small particle sets run predict / correct / prune / reweight as one launch (k_particle_chain: the five kernels'
    bodies back to back in the particle's workgroup); the same steps through the five separate launches must give the
    same bits."""
    f = Frame(40, 90, 20, 321, weight_profile="steady")
    rng = np.random.default_rng(9)
    zs = [f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3 for _ in range(3)]

    def run(chain_max):
        monkeypatch.setenv("PHD_CHAIN_MAX", str(chain_max))
        nav, _ = make_nav(nav_mod, f)
        out = []
        for k, z in enumerate(zs):
            nav.SlamUpdate(None, z, u_resample=0.2 + 0.3 * k)
            out.append((nav.VehicleWeights, nav.resample_sources(), nav.BestParticle, [nav.MapModel(i) for i in (0, 13, 39)]))
        nav.close()
        return out

    for (wa, sa, ba, ma), (wb, sb, bb, mb) in zip(run(0), run(1000)):
        assert np.array_equal(wa, wb) and np.array_equal(sa[0], sb[0]) and sa[1] == sb[1] and ba == bb
        for x, y in zip(ma, mb):
            assert all(np.array_equal(u, v) for u, v in zip(x, y))


def test_a_cluster_beyond_256_rows_is_an_error_that_keeps_the_state(nav_mod):
    """MurtyPairing has no size limit in the reference; the device solves clusters of up to 256 rows. One clump of ~25
    landmarks and 250 measurements is beyond that: PHD_ERR_ASSOCIATION, the particle set stays as it was."""
    from test_gpu_parity import clustered_frame
    f = clustered_frame(70, 1, 60, 250, spread_px=3.0)
    nav, p = make_nav(nav_mod, f, merge_threshold=1e-3, emit_capacity=24000)
    before = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])
    with pytest.raises(nav_mod.PHDError) as e:
        nav.SlamUpdate(None, f.z)
    assert e.value.status == 3 and e.value.module == "association"
    after = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])
    assert np.array_equal(before[0], after[0])
    for x, y in zip(before[1], after[1]):
        assert all(np.array_equal(a, b) for a, b in zip(x, y))
    nav.SlamUpdate(None, f.z[:40])          # a frame the solver takes runs from the kept state
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    nav.close()


def test_whole_state_download_behind_a_resampling_step_of_many_particles(nav_mod):
    """phd_download_state_soa gathers the mixtures a resampling step left behind slot numbers (k_materialise), here with
    more workgroups than the device holds at once; checked against the per-particle getter, which reads through the
    slots. (The gather must not depend on anything a workgroup of the same launch writes. It once did — workgroup 0 set
    the role INMIX — and a late workgroup on a CU that had not yet cached the roles copied a stale map; seen once in
    test_multi_handle_equals_single_handle, too rare to provoke on purpose.)"""
    f = Frame(6000, 12, 6, 4242, weight_profile="steady")
    f.weights = np.random.default_rng(4242).random(f.P) ** 40   # depleted: the step resamples
    f.weights /= f.weights.sum()
    nav, p = make_nav(nav_mod, f, maxq=40)
    nav.SlamUpdate(None, f.z, u_resample=0.37)
    src, resampled = nav.resample_sources()
    assert resampled and len(np.unique(src)) < f.P
    maps = {i: nav.MapModel(i) for i in list(range(0, f.P, 37)) + [f.P - 1]}
    planes, counts, poses, weights = nav.download_state(40)
    for i, (w, m, c) in maps.items():
        assert counts[i] == len(w)
        assert np.array_equal(planes[0, i, :counts[i]], w)
        assert np.array_equal(planes[1:4, i, :counts[i]].T, m)
        assert np.array_equal(planes[4:10, i, :counts[i]].T, c[:, IU[0], IU[1]])
    nav.close()
