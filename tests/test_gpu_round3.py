"""GPU cases added in round 3 (through the C-ABI):
  * the migration plan as the device makes it (k_plan_migration) against its host statement (phd_plan_migration), and the
    consistency of the sender-side record numbers with the receiver-side lists;
  * a multi-device handle of eight shards, steps posted back to back without waiting, its diagnostics;
  * the holes the round-2 review named: MurtyPairing at 192 / 193 / 255 / 256 rows, the candidate-queue overflow of
    k_emit_finish (every gated pair queued) at 64 and 128 measurements, a quasi batch with clusters of more than 64 rows
    followed by a step that needs the association slab, non-finite input rejected at the boundary."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame
from test_gpu_round2 import assert_map_close, make_nav, oracle_state

ip = C.POINTER(C.c_int32)


@pytest.fixture(scope="module")
def nav_mod():
    from monorfs_amd import navigator
    return navigator


def host_plan(lib, gsrc, Pl, world, rank):
    g = np.ascontiguousarray(gsrc, np.int32)
    sc, rc = np.zeros(world, np.int32), np.zeros(world, np.int32)
    sl, code = np.zeros(Pl * max(world - 1, 1), np.int32), np.zeros(Pl, np.int32)
    q = lambda a: a.ctypes.data_as(ip)
    nrecv = lib.phd_plan_migration(q(g), Pl, world, rank, q(sc), q(rc), q(sl), q(code))
    return sc, rc, sl[:int(sc.sum())], code, nrecv


def check_plans(nav, gsrc, Pl, world):
    """every rank's device plan against the host plan; then the senders' record numbers against the receivers' lists"""
    plans = []
    for rank in range(world):
        d = nav.test_migration_plan(gsrc, Pl, world, rank)
        sc, rc, sl, code, nrecv = host_plan(nav._lib, gsrc, Pl, world, rank)
        assert d["status"] == 0
        assert np.array_equal(d["send_counts"], sc) and np.array_equal(d["recv_counts"], rc), "rank %d counts" % rank
        assert d["nsend"] == sc.sum() and d["nrecv"] == nrecv == rc.sum()
        assert np.array_equal(d["send_list"], sl), "rank %d send list" % rank
        assert np.array_equal(d["dst_code"], code), "rank %d slot codes" % rank
        # arrivals go to the first slots no local particle keeps as its source, ascending
        used = np.zeros(Pl, bool)
        used[code[code >= 0]] = True
        assert np.array_equal(d["fslot"], np.flatnonzero(~used)[:nrecv]), "rank %d free slots" % rank
        plans.append(d)
    for t in range(world):   # receiver t: record k of its buffer is the particle its slots with code -(k + 1) take
        want = {}
        for i in range(Pl):
            c = plans[t]["dst_code"][i]
            if c < 0:
                want.setdefault(-c - 1, gsrc[t * Pl + i])
                assert want[-c - 1] == gsrc[t * Pl + i]
        got = {}
        for s in range(world):
            for k in range(plans[s]["nsend"]):
                dst, recno = plans[s]["send_dst"][k]
                if dst == t:
                    assert recno not in got, "two records for one place in rank %d's buffer" % t
                    got[int(recno)] = s * Pl + int(plans[s]["send_list"][k])
        assert got == {k: int(v) for k, v in want.items()}, "rank %d: the records pushed to it are not the ones its slots read" % t


@pytest.mark.parametrize("Pl,world,power", [(40, 3, 30), (2048, 8, 60), (1, 5, 3), (7, 64, 8), (300, 1, 20), (1030, 2, 200), (2048, 8, 2)])
def test_device_migration_plan_equals_the_host_plan(nav_mod, Pl, world, power):
    p = prm3d_defaults(max_particles=Pl, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=Pl)
    rng = np.random.default_rng(Pl * 100 + world)
    Pg = Pl * world
    for trial in range(3):
        w = rng.random(Pg) ** power + 1e-300   # depleted: few sources, long runs, many of them across rank boundaries
        w /= w.sum()
        gsrc, _ = nav.ResampleParticles(w, float(rng.uniform(0.01, 0.99)))
        assert np.all(np.diff(gsrc) >= 0)
        check_plans(nav, gsrc, Pl, world)
    # every slot from one particle; the identity (nothing moves)
    check_plans(nav, np.full(Pg, Pg // 2, np.int32), Pl, world)
    check_plans(nav, np.arange(Pg, dtype=np.int32), Pl, world)
    # not resampled: nothing is planned; a decreasing vector is not a resampling result
    d = nav.test_migration_plan(np.arange(Pg, dtype=np.int32), Pl, world, 0, resampled=False)
    assert d["status"] == 0 and d["nsend"] == 0 and d["nrecv"] == 0 and not d["send_counts"].any()
    if Pg > 1:
        bad = np.arange(Pg, dtype=np.int32)[::-1].copy()
        assert nav.test_migration_plan(bad, Pl, world, world - 1)["status"] == 2
    nav.close()


def same_state(a, b, maps):
    assert np.array_equal(a.VehicleWeights, b.VehicleWeights)
    assert np.array_equal(a.poses(), b.poses())
    assert a.BestParticle == b.BestParticle
    sa, sb = a.resample_sources(), b.resample_sources()
    assert sa[1] == sb[1] and np.array_equal(sa[0], sb[0])
    for i in maps:
        for x, y in zip(a.MapModel(i), b.MapModel(i)):
            assert np.array_equal(x, y), "map %d differs" % i


def test_multi_handle_of_eight_shards_with_steps_posted_back_to_back(nav_mod):
    """Eight shards behind one handle (eight worker threads, peer stores for the weights and the migrating particles, the
    plan on the device): phd_step_async only posts the step, so a batch of steps is queued without a single wait and the
    result is still, bit for bit, the single handle's. Then the diagnostics of the multi-device host."""
    shards, P = 8, 8 * 20
    f = Frame(P, 70, 18, 808, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=18)
    single = nav_mod.PHDNavigator(p, particlecount=P)
    multi = nav_mod.PHDNavigator(p, particlecount=P, devices=[0] * shards)
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
        nav.set_measurements(f.z)
    multi.timing_reset(2)
    us = [0.31, 0.77, 0.12, 0.5, 0.93, 0.05]
    nres = 0
    for u in us:
        single.step_async(u)
        multi.step_async(u)          # returns at once: nothing of the step has to have run
    single.sync()
    multi.sync()
    same_state(single, multi, range(P))
    # ... and step by step, with odometry in between
    rng = np.random.default_rng(8)
    for step in range(3):
        reading = rng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.003, 0.003, 0.003]
        noise = rng.normal(0, 1, (P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
        u = float(rng.uniform(0.05, 0.95))
        for nav in (single, multi):
            nav.UpdateOdometry(None, reading, noise)
            nav.SlamUpdate(None, f.z[: 18 - 3 * step], u_resample=u)
        same_state(single, multi, range(0, P, 7))
        nres += single.resample_sources()[1]
    assert nres >= 1, "the sequence did not resample: the migration was not exercised"
    rep = multi.multi_report()
    assert rep["shards"] == shards and all(all(r) for r in rep["p2p"])
    assert rep["sampled_steps"] >= 2 and rep["phase_ms"]["local"] > 0 and rep["issue_us"] > 0
    assert rep["post_us"] < 2000, "posting a step took %.0f us: phd_step_async waited for something" % rep["post_us"]
    single.close()
    multi.close()


def test_multi_handle_on_distinct_devices(nav_mod):
    """the same on two or more real devices (peer stores over xGMI); skipped on a one-GPU box"""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("needs two GPUs")
    shards, P = ndev, ndev * 32
    f = Frame(P, 70, 18, 909, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=18)
    single = nav_mod.PHDNavigator(p, particlecount=P)
    multi = nav_mod.PHDNavigator(p, particlecount=P, devices=list(range(shards)))
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    rng = np.random.default_rng(9)
    for step in range(6):
        u = float(rng.uniform(0.05, 0.95))
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        for nav in (single, multi):
            nav.SlamUpdate(None, z, u_resample=u)
        same_state(single, multi, range(P))
    assert all(all(r) for r in multi.multi_report()["p2p"])
    single.close()
    multi.close()


# ---- MurtyPairing at the edges of the rows-per-lane formula (64 NT rows: 128 -> 2, 192 / 193 -> 3 / 4, 256 -> 4) -------------
@pytest.mark.parametrize("n", [191, 192, 193, 255, 256])
def test_murty_pairing_at_the_slab_block_edges(nav_mod, n):
    p = prm3d_defaults(max_particles=1, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=1)
    rng = np.random.default_rng(1700 + n)
    m = rng.uniform(-20, 0, (n, n))
    m[rng.uniform(size=m.shape) < 0.6] = -np.inf
    m[np.arange(n), np.arange(n)] = rng.uniform(-5, 0, n)
    asg, val = nav.test_pairing(m, maxcount=25)
    oasg, oval = orc.murty(m, maxcount=25)
    assert len(asg) == len(oasg) == 25
    assert asg == oasg
    assert np.array_equal(val, oval)
    nav.close()


# ---- k_emit_finish when a wave's candidate segment overflows: the full second sweep --------------------------------------------
@pytest.mark.parametrize("C,M", [(208, 64), (136, 128)])
def test_emit_finish_with_an_overflowing_candidate_queue(nav_mod, C, M):
    """MinWeight tiny and the radius gate off: every pair whose weight does not underflow is a candidate, a wave's segment
    (a quarter of 16 (max_components + measurements) entries) overflows and k_emit_finish takes every pair again instead of
    the queue. Corrected list and pruned map against the oracle."""
    from monorfs_amd.abi import PHD_GATE_DISABLED
    from monorfs_amd.synth import measure_perfect_identity, measure_to_map_identity
    rng = np.random.default_rng(C + M)
    f = Frame(2, C, M, 5000 + M, weight_profile="steady")
    # a compact scene: every measurement within a few sigma of every component, so that no pair underflows
    zc = np.stack([rng.uniform(-40, 40, C), rng.uniform(-30, 30, C), rng.uniform(0.95, 1.05, C)], axis=1)
    base = measure_to_map_identity(zc)
    f.mean = base[None] + rng.normal(size=(f.P, C, 3)) * 1e-3
    f.w = np.broadcast_to(rng.uniform(0.3, 1.0, C), (f.P, C))
    f.z = measure_perfect_identity(base[rng.choice(C, M, replace=False)]) + rng.normal(size=(M, 3)) * np.sqrt([2.0, 2.0, 1e-3])
    cap = 640
    segcap = 16 * (cap + max(64, (M + 63) // 64 * 64)) // 4
    assert (C // 4) * M > segcap, "the frame cannot overflow a wave's segment"
    nav, p = make_nav(nav_mod, f, min_weight=1e-290, gate_metric=PHD_GATE_DISABLED, emit_capacity=C * (M + 1) + 64)
    nav.run_stages(f.z, with_alpha=False)
    for i in range(f.P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        cor = orc.correct(p, f.poses[i], f.z, pred)
        keep = cor[0] >= p.min_weight
        gw, gm, gc = nav.CorrectConditional(i)
        assert len(gw) == keep.sum() > segcap, "corrected[%d]: %d entries, oracle %d" % (i, len(gw), keep.sum())
        assert np.isclose(np.sort(gw).sum(), np.sort(cor[0][keep]).sum(), rtol=1e-9)
        assert_map_close(nav.PruneModel(i), orc.prune(p, cor), 1e-7, "prune[%d]" % i)
    nav.close()


# ---- ADVICE (round 2): the association slab's counter behind a quasi batch ------------------------------------------------------
def test_a_quasi_batch_with_big_clusters_leaves_the_association_slab_to_the_next_step(nav_mod):
    """phd_quasi_set_loglik takes slab blocks for its clusters of more than 64 rows; the next step's association kernel must
    find the slab free again (the counter used to be reset only by a step's own last kernel). A slab that holds one such
    cluster, a batch that uses it up, then a step with a cluster of 88 rows: it must run."""
    from test_gpu_parity import clustered_frame
    f = clustered_frame(67, 1, 30, 60, spread_px=3.0)
    nav, p = make_nav(nav_mod, f, merge_threshold=1e-3, emit_capacity=12000)
    lm = f.mean[0][:30]
    poses = np.tile(f.poses[0], (3, 1))
    nav._check(nav._lib.phd_set_association_workspace(nav._h, 0))
    with pytest.raises(nav_mod.PHDError) as e:           # (the batch does need the slab: a cluster of more than 64 rows)
        nav.QuasiSetLogLikelihood(f.z, lm, poses)
    assert e.value.status == 3
    # a block for a cluster of ~90 rows takes ~0.25 MB: the batch (three poses) takes three, the step (three particles) three —
    # 1 MiB holds either, not both
    nav._check(nav._lib.phd_set_association_workspace(nav._h, 1 << 20))
    v = nav.QuasiSetLogLikelihood(f.z, lm, poses)       # gate 12 sigma: one cluster of 90 rows per pose -> a slab block each
    assert np.isclose(v[0], orc.quasi_set_log_likelihood(p, poses[0], lm, f.z), rtol=1e-9, atol=1e-9)
    nav.run_stages(f.z, with_alpha=True)                # a stage run takes three blocks as well, and gives them back
    nav.SlamUpdate(None, f.z)                           # would fail with PHD_ERR_ASSOCIATION if the earlier blocks were still counted
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    v2 = nav.QuasiSetLogLikelihood(f.z, lm, poses)
    assert np.array_equal(v, v2)
    nav.close()


# ---- non-finite input is rejected at the boundary -------------------------------------------------------------------------------
def test_non_finite_input_is_a_bad_argument(nav_mod):
    """PHDNavigator.cs:886-890 lets a NaN term poison a weight sum; the device's pair loops count a NaN exponent as 0.
    The two never meet: measurements, poses, weights, odometry and maps that are not finite do not cross the ABI."""
    f = Frame(4, 30, 8, 17, weight_profile="steady")
    nav, p = make_nav(nav_mod, f)
    nav.SlamUpdate(None, f.z, u_resample=0.4)
    before = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])

    def rejected(fn):
        with pytest.raises(nav_mod.PHDError) as e:
            fn()
        assert e.value.status == 1, e.value

    z = f.z.copy(); z[3, 1] = np.nan
    rejected(lambda: nav.SlamUpdate(None, z))
    z[3, 1] = np.inf
    rejected(lambda: nav.set_measurements(z))
    poses = nav.poses(); poses[2, 4] = np.nan
    rejected(lambda: nav.set_poses(poses))
    w = nav.VehicleWeights; w[0] = -np.inf
    rejected(lambda: nav.set_weights(w))
    rejected(lambda: nav.UpdateOdometry(None, [0, 0, np.nan, 0, 0, 0]))
    noise = np.zeros((f.P, 6)); noise[1, 5] = np.inf
    rejected(lambda: nav.UpdateOdometry(None, np.zeros(6), noise))
    mw, mm, mc = f.map(0)
    mm = mm.copy(); mm[5, 0] = np.nan
    rejected(lambda: nav.set_map(1, (mw, mm, mc)))
    planes = f.planes(); planes[7, 2, 3] = np.nan
    rejected(lambda: nav.upload_state(planes, f.counts, f.poses, f.weights))
    rejected(lambda: nav.reset(np.array([0, 0, 0, np.nan, 0, 0, 0.0]), f.map(0), 4))
    rejected(lambda: nav.QuasiSetLogLikelihood(z, f.mean[0][:5], f.poses[:1]))
    after = (nav.VehicleWeights, [nav.MapModel(i) for i in range(f.P)])
    assert np.array_equal(before[0], after[0])
    for x, y in zip(before[1], after[1]):
        assert all(np.array_equal(a, b) for a, b in zip(x, y))
    nav.SlamUpdate(None, f.z, u_resample=0.6)   # and the handle goes on
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    nav.close()


def test_resample_every_launch_shape_and_degenerate_weights(nav_mod):
    """k_normalise_resample decides a particle's slots from prefix sums with an error margin and falls back to the
    recurrence itself (PHDNavigator.cs:724-760) when a boundary is closer than the margin. Sizes around every change of its
    launch shape (256 / 512 / 1024 threads, chunks of 1, 2, ... weights), and weight vectors that sit ON the boundaries
    (exactly uniform: every slot boundary is a tie, the fallback runs), vectors of mostly zeros, one particle holding
    everything, u at both ends of [0, 1): sources and BestParticle bit-exact against the sequential recurrence."""
    p = prm3d_defaults(max_particles=4, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=4)
    rng = np.random.default_rng(11)
    sizes = [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 4095, 4096, 4097, 8192, 12345]
    for P in sizes:
        vectors = {"uniform": np.full(P, 1.0 / P)}
        w = rng.random(P) ** 6
        vectors["random^6"] = w / w.sum()
        w = rng.random(P) * (rng.random(P) < 0.1)
        w[rng.integers(P)] += 1e-3
        vectors["nine tenths zero"] = w / w.sum()
        for at in sorted({0, P // 2, P - 1}):
            w = np.zeros(P)
            w[at] = 1.0
            vectors["all in %d" % at] = w
        w = np.where(np.arange(P) % 2 == 0, 1.0, 1e-12)
        vectors["alternating"] = w / w.sum()
        for name, w in vectors.items():
            for u in (0.0, 2.0 ** -60, 0.5, 1.0 - 2.0 ** -53):
                src, best = nav.ResampleParticles(w, u)
                osrc, obest = orc.resample(w, u)
                assert np.array_equal(src, osrc), "P=%d %s u=%g: %d sources differ" % (P, name, u, np.count_nonzero(src != osrc))
                assert best == obest, "P=%d %s u=%g: best %d, oracle %d" % (P, name, u, best, obest)
            assert nav.ParticleDepleted(w) == orc.particle_depleted(p, w), "P=%d %s" % (P, name)
    nav.close()


@pytest.mark.parametrize("holder", [0, 37, 63])
def test_multi_handle_when_one_particle_takes_every_slot(nav_mod, holder):
    """The extreme of the migration: one particle holds all the weight, so after the step every slot of every shard has
    it as its source — each other shard receives ONE record that all its slots share, the holder's shard sends to everybody.
    Then a second step on the cloned set. Bit for bit the single handle's."""
    shards, P = 4, 64
    f = Frame(P, 60, 14, 909 + holder, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=14)
    single = nav_mod.PHDNavigator(p, particlecount=P)
    multi = nav_mod.PHDNavigator(p, particlecount=P, devices=[0] * shards)
    w = np.zeros(P)
    w[holder] = 1.0
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, w)
        nav.SlamUpdate(None, f.z, u_resample=0.41)
    src, resampled = single.resample_sources()
    assert resampled and np.all(src == holder)
    same_state(single, multi, range(P))
    rng = np.random.default_rng(holder)
    noise = rng.normal(0, 1, (P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
    for nav in (single, multi):
        nav.UpdateOdometry(None, np.zeros(6), noise)
        nav.SlamUpdate(None, f.z[:9], u_resample=0.83)
    same_state(single, multi, range(P))
    single.close()
    multi.close()
