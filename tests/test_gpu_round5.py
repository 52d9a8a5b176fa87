"""Round 5: what the round changed or closed, through the C-ABI (ctypes -> libphdhip.so) like every other GPU test."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import PHD_ERR_BAD_ARGUMENT, prm3d_defaults
from monorfs_amd.synth import Frame


@pytest.fixture(scope="module")
def nav_mod():
    from monorfs_amd import navigator
    return navigator


def _handle(nav_mod, f, maxq=600):
    p = prm3d_defaults(max_particles=f.P, max_components=max(maxq, f.C), max_measurements=max(f.M, 1))
    p.max_quantity = maxq
    nav = nav_mod.PHDNavigator(p, particlecount=f.P)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    return nav, p


def test_a_failed_ipc_open_leaves_the_handle_without_peers(nav_mod):
    """phd_migration_ipc_open closes the mappings of an earlier call before it opens the new ones: when an open then fails,
    the handle must be left WITHOUT peers (push / global-device step refuse) — not with a table of closed mappings the pack
    kernel would store into."""
    f = Frame(32, 40, 12, 501, weight_profile="steady")
    a, _ = _handle(nav_mod, f)
    b, _ = _handle(nav_mod, f)
    lib = a._lib
    a.set_measurements(f.z)
    recv = (C.c_void_p * 2)(lib.phd_migration_recv_buffer(a._h), lib.phd_migration_recv_buffer(b._h))
    a._check(lib.phd_migration_set_peers(a._h, recv, 0, 2))
    a._check(lib.phd_step_local_async(a._h, 0))
    a.sync()
    bad = b"\x00" * 128                       # two handles no runtime ever exported
    rc = lib.phd_migration_ipc_open(a._h, bad, 0, 2)
    assert rc != 0, "a made-up IPC handle was opened"
    assert b"cannot be opened" in lib.phd_last_error(a._h)
    assert lib.phd_migration_push_async(a._h) == PHD_ERR_BAD_ARGUMENT
    assert lib.phd_step_global_device_async(a._h, 0, 2, C.c_double(0.5), 0) == PHD_ERR_BAD_ARGUMENT
    # ... and the handle is still good for everything else, and for a new set of peers
    a._check(lib.phd_migration_set_peers(a._h, recv, 0, 2))
    a._check(lib.phd_step_local_async(a._h, 0))
    a.sync()
    a.close()
    b.close()


@pytest.mark.parametrize("M", [24, 70])
def test_emit_and_prune_as_one_launch_or_two(nav_mod, monkeypatch, M):
    """k_emit_finish + k_prune_merge as ONE launch (k_emit_prune, the default up to 64 measurements) or as two (the default
    beyond): PHD_FUSE_EP=0 / 1 force either — the same bodies, the same bits, on frames on both sides of the default's
    boundary, through the separate kernels on two streams (PHD_CHAIN_MAX=0: the one-launch chain has no such boundary)."""
    f = Frame(40, 150, M, 502, weight_profile="steady")
    monkeypatch.setenv("PHD_CHAIN_MAX", "0")
    monkeypatch.setenv("PHD_SPLIT", "2")
    got = {}
    for mode in ("default", "0", "1"):
        if mode == "default":
            monkeypatch.delenv("PHD_FUSE_EP", raising=False)
        else:
            monkeypatch.setenv("PHD_FUSE_EP", mode)
        nav, p = _handle(nav_mod, f)
        for step in range(3):
            nav.SlamUpdate(None, f.z + 0.05 * step, u_resample=0.3 + 0.2 * step)
        got[mode] = (nav.VehicleWeights, nav.resample_sources()[0], [nav.MapModel(i) for i in (0, 7, f.P - 1)])
        if mode == "default":     # ... and the default against the oracle, so that "equal" means "right"
            st = orc.State(f.P, 700)
            st.poses[:] = f.poses
            st.w[:, :f.C], st.mean[:, :f.C], st.cov[:, :f.C], st.n[:] = f.w, f.mean, f.cov, f.C
            for step in range(3):
                _, src, _, _ = orc.slam_update(p, st, f.z + 0.05 * step, u=0.3 + 0.2 * step, threads=4)
            assert np.array_equal(got[mode][1], src)
            assert np.allclose(got[mode][0], st.weights, rtol=1e-6, atol=1e-300)
        nav.close()
    for mode in ("0", "1"):
        assert np.array_equal(got[mode][0], got["default"][0]) and np.array_equal(got[mode][1], got["default"][1]), "PHD_FUSE_EP=%s" % mode
        for a, b in zip(got[mode][2], got["default"][2]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), "PHD_FUSE_EP=%s" % mode


# ---- the landing flags: no second collective in the sharded step (phd_migration_set_landing) -------------------------------------
def _depleted_frame(world, Pl, Cc, M, seed):
    f = Frame(Pl * world, Cc, M, seed, weight_profile="steady")
    f.weights = np.random.default_rng(world).random(f.P) ** 12      # depleted from the start: the first step resamples, long runs cross the rank boundaries
    f.weights /= f.weights.sum()
    return f


@pytest.mark.parametrize("world", [2, 3])
def test_landing_flags_sharded_step_equals_single_handle(nav_mod, world):
    """The device-path sequence with phd_migration_set_landing(1): k_post_landing behind every push, k_finish_sharded waiting for
    the flags of the ranks it takes records from — `world` handles in one process on one stream (every push is enqueued before
    any unpack, so the waits find their flags), bit for bit the single handle, over steps that resample and migrate and one that
    does not (no records: nobody waits)."""
    from test_gpu_round4 import _device_path_handles, _device_path_step
    Pl, Cc, M = 56, 70, 18
    f = _depleted_frame(world, Pl, Cc, M, 5100 + world)
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one = nav_mod.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(nav_mod, f, world, Pl, M)
    for nv in navs:
        if nv._lib.phd_migration_recv_is_finegrained(nv._h) != 1:
            pytest.skip("no fine-grained receive buffers on this box")
        nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    nres = 0
    for u in (0.31, 0.77, 0.12):
        one.SlamUpdate(None, f.z, u_resample=u)
        _device_path_step(navs, Pl, u)
        nres += int(one.resample_sources()[1])
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs]))
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs]))
        for g in list(range(0, Pl * world, 7)) + [Pl * world - 1]:
            a_, b_ = one.MapModel(g), navs[g // Pl].MapModel(g % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "particle %d" % g
    assert nres >= 1, "the sequence did not resample: the flags were never waited for"
    one.close()
    for nv in navs:
        nv.close()


def test_a_landing_flag_that_never_comes_is_a_bounded_wait(nav_mod, monkeypatch):
    """rank 0 ends a resampling step whose records rank 1 never pushed: its k_finish_sharded gives up after the bound (1.5 s here) and phd_sync
    reports the step as failed (a peer died) — a bounded wait, not a hung device"""
    import time
    from test_gpu_round4 import _Dev, _device_path_handles
    import torch
    monkeypatch.setenv("PHD_LANDING_TIMEOUT_MS", "1500")
    world, Pl, Cc, M = 2, 56, 70, 18
    f = _depleted_frame(world, Pl, Cc, M, 5102)
    f.weights[Pl:] *= 50.0            # the heavy particles live on rank 1: rank 0's slots take records from there
    f.weights /= f.weights.sum()
    navs = _device_path_handles(nav_mod, f, world, Pl, M)
    lib = navs[0]._lib
    for nv in navs:
        if lib.phd_migration_recv_is_finegrained(nv._h) != 1:
            pytest.skip("no fine-grained receive buffers on this box")
        nv._check(lib.phd_migration_set_landing(nv._h, 1))
    for nv in navs:
        nv._check(lib.phd_step_local_async(nv._h, 0))
    allw = torch.cat([torch.as_tensor(_Dev(lib.phd_device_local_weights(nv._h), Pl + 1), device="cuda") for nv in navs])
    for nv in navs:
        torch.as_tensor(_Dev(lib.phd_device_gather_buffer(nv._h, world), world * (Pl + 1)), device="cuda").copy_(allw)
    for r, nv in enumerate(navs):
        nv._check(lib.phd_step_global_device_async(nv._h, r, world, C.c_double(0.31), 0))
    navs[0]._check(lib.phd_migration_push_async(navs[0]._h))        # rank 1 never pushes
    navs[0]._check(lib.phd_migration_unpack_async(navs[0]._h))
    t0 = time.time()
    with pytest.raises(nav_mod.PHDError):
        navs[0].sync()
    assert 1.0 < time.time() - t0 < 30.0
    for nv in navs:
        nv.close()


# ---- normalise / BestParticle / depletion / resampling over a grid of workgroups (k_nr_*, phd_resample.h) ---------------------------
def _resample_vectors(P, rng):
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
    return vectors


def test_grid_resampling_at_every_size_and_on_degenerate_weights(nav_mod, monkeypatch):
    """The grid kernels forced on EVERY vector length (PHD_NR_GRID_MIN=1; by default they take 4096 .. 65 536 weights): one
    workgroup and many, lengths around the workgroup and wave boundaries, vectors that sit on the slot boundaries (uniform:
    the margin test fails and one wave replays the recurrence), mostly zeros, one particle holding everything, u at both ends —
    sources and BestParticle bit-exact against the sequential recurrence (PHDNavigator.cs:724-760), the depletion test too."""
    monkeypatch.setenv("PHD_NR_GRID_MIN", "1")
    p = prm3d_defaults(max_particles=4, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=4)
    rng = np.random.default_rng(12)
    for P in [1, 2, 3, 63, 64, 65, 255, 256, 257, 511, 512, 513, 1000, 1024, 1025, 4097, 12345, 16384, 65536]:
        for name, w in _resample_vectors(P, rng).items():
            for u in (0.0, 2.0 ** -60, 0.5, 1.0 - 2.0 ** -53):
                src, best = nav.ResampleParticles(w, u)
                osrc, obest = orc.resample(w, u)
                assert np.array_equal(src, osrc), "P=%d %s u=%g: %d sources differ" % (P, name, u, np.count_nonzero(src != osrc))
                assert best == obest, "P=%d %s u=%g: best %d, oracle %d" % (P, name, u, best, obest)
            assert nav.ParticleDepleted(w) == orc.particle_depleted(p, w), "P=%d %s" % (P, name)
    nav.close()


def test_vectors_beyond_the_grid_take_the_one_workgroup_kernel(nav_mod):
    """65 537 weights and more: the grid's statistics block holds 256 workgroups — the one-workgroup kernel's global-memory walk it is"""
    p = prm3d_defaults(max_particles=4, max_components=600, max_measurements=8)
    nav = nav_mod.PHDNavigator(p, particlecount=4)
    rng = np.random.default_rng(13)
    for P in (65537, 70000):
        w = rng.random(P) ** 8
        w /= w.sum()
        src, best = nav.ResampleParticles(w, 0.37)
        osrc, obest = orc.resample(w, 0.37)
        assert np.array_equal(src, osrc) and best == obest
    nav.close()


@pytest.mark.parametrize("frozen", [False, True])
def test_steps_that_end_on_the_grid_kernels(nav_mod, monkeypatch, frozen):
    """Whole SlamUpdate steps whose end (normalise, BestParticle, depletion, resampling, the gather of the small arrays, the
    rotation of the bank roles) runs on the grid kernels (forced: PHD_NR_GRID_MIN=64) against the oracle and against a handle
    that ends its steps on the one-workgroup kernel: resampling decision, sources and BestParticle exact, weights to the
    tolerance of two summation orders (1e-12 between the kernels, 1e-6 against the oracle), maps bit for bit."""
    f = Frame(300, 50, 12, 520, weight_profile="steady")
    f.weights = np.random.default_rng(5).random(f.P) ** 8     # a depleted set: the first step resamples (the frozen runs repeat that step)
    f.weights /= f.weights.sum()
    one, p = _handle(nav_mod, f)
    monkeypatch.setenv("PHD_NR_GRID_MIN", "64")
    grid, _ = _handle(nav_mod, f)
    st = orc.State(f.P, 700)
    st.poses[:] = f.poses
    st.w[:, :f.C], st.mean[:, :f.C], st.cov[:, :f.C], st.n[:] = f.w, f.mean, f.cov, f.C
    st.weights[:] = f.weights
    rng = np.random.default_rng(521)
    nres = 0
    for step in range(4):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        u = float(rng.uniform(0.05, 0.95))
        if frozen:          # the frozen step leaves the state where it was: every step starts from the same one, on both handles
            one.set_frozen(True)
            grid.set_frozen(True)
        one.SlamUpdate(None, z, u_resample=u)
        grid.SlamUpdate(None, z, u_resample=u)
        (s1, r1), (s2, r2) = one.resample_sources(), grid.resample_sources()
        assert r1 == r2 and np.array_equal(s1, s2), "step %d: the two kernels resample differently" % step
        assert one.BestParticle == grid.BestParticle
        assert np.allclose(one.VehicleWeights, grid.VehicleWeights, rtol=1e-12, atol=0)
        for i in (0, 17, 150, f.P - 1):
            assert all(np.array_equal(x, y) for x, y in zip(one.MapModel(i), grid.MapModel(i))), "step %d map %d" % (step, i)
        assert np.array_equal(one.poses(), grid.poses())
        if not frozen:
            best, src, res, _ = orc.slam_update(p, st, z, u=u, threads=4)
            assert res == r2 and np.array_equal(src, s2) and best == grid.BestParticle
            assert np.allclose(grid.VehicleWeights, st.weights, rtol=1e-6, atol=1e-300)
        nres += int(r2)
    assert nres >= 1, "no step resampled"
    one.close()
    grid.close()


# ---- the migration plan over a grid of workgroups (k_plan_count / k_plan_lists) ---------------------------------------------------
@pytest.mark.parametrize("Pl,world,power", [(64, 2, 30), (64, 64, 8), (128, 3, 60), (448, 5, 2), (2048, 8, 60), (2048, 8, 2), (1024, 64, 20)])
def test_grid_migration_plan_equals_the_host_plan(nav_mod, monkeypatch, Pl, world, power):
    """phd_plan_migration (the host statement of the plan) against the two grid kernels, forced on every size whose ranks hold a
    multiple of 64 particles (PHD_PLAN_GRID_MIN=1; by default they take global vectors of 4096 .. 65 536 slots): every rank's
    lists, the senders' record numbers against the receivers' slot codes, the free slots; several plans one after the other
    (the accumulators alternate between two sets that the kernels clear for each other); a vector that is no resampling result."""
    from test_gpu_round3 import check_plans
    monkeypatch.setenv("PHD_PLAN_GRID_MIN", "1")
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
    check_plans(nav, np.full(Pg, Pg // 2, np.int32), Pl, world)          # every slot from one particle
    check_plans(nav, np.arange(Pg, dtype=np.int32), Pl, world)           # the identity: nothing moves
    d = nav.test_migration_plan(np.arange(Pg, dtype=np.int32), Pl, world, 0, resampled=False)
    assert d["status"] == 0 and d["nsend"] == 0 and d["nrecv"] == 0 and not d["send_counts"].any()
    # steps that do not resample BETWEEN steps that do (one, then two in a row: both parities of the alternating accumulator sets):
    # the launch that returns early must still clear the set the plan before it filled (found by tests/soak_multi.py with the
    # grid kernels forced: the counts of a resampling step met again two plans later)
    for quiet in (1, 2, 1):
        w = rng.random(Pg) ** power + 1e-300
        gsrc, _ = nav.ResampleParticles(w / w.sum(), float(rng.uniform(0.01, 0.99)))
        check_plans(nav, gsrc, Pl, world)
        for _ in range(quiet):
            d = nav.test_migration_plan(gsrc, Pl, world, world - 1, resampled=False)
            assert d["status"] == 0 and d["nsend"] == 0 and d["nrecv"] == 0
    w = rng.random(Pg) ** power + 1e-300
    gsrc, _ = nav.ResampleParticles(w / w.sum(), 0.37)
    check_plans(nav, gsrc, Pl, world)
    bad = np.arange(Pg, dtype=np.int32)[::-1].copy()
    assert nav.test_migration_plan(bad, Pl, world, world - 1)["status"] == 2
    w = rng.random(Pg) ** power + 1e-300                                   # ... and a good plan again behind the refused one
    gsrc, _ = nav.ResampleParticles(w / w.sum(), 0.5)
    check_plans(nav, gsrc, Pl, world)
    nav.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_steps_on_the_grid_kernels_equal_the_single_handle(nav_mod, monkeypatch, world):
    """the device-path sharded sequence with BOTH global kernels on their grid versions (forced: 64 particles per rank are far
    below the default thresholds) and the landing flags: bit for bit the single handle (which ends its steps on the grid
    resampling kernels too: the sums' shape depends on the vector's length and the kernel)"""
    from test_gpu_round4 import _device_path_handles, _device_path_step
    monkeypatch.setenv("PHD_PLAN_GRID_MIN", "1")
    monkeypatch.setenv("PHD_NR_GRID_MIN", "1")
    Pl, Cc, M = 64, 70, 18
    f = _depleted_frame(world, Pl, Cc, M, 5300 + world)
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one = nav_mod.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(nav_mod, f, world, Pl, M)
    for nv in navs:
        if nv._lib.phd_migration_recv_is_finegrained(nv._h) == 1:
            nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    nres = 0
    for u in (0.31, 0.77, 0.12, 0.6):
        one.SlamUpdate(None, f.z, u_resample=u)
        _device_path_step(navs, Pl, u)
        nres += int(one.resample_sources()[1])
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs]))
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs]))
        for g in list(range(0, Pl * world, 5)) + [Pl * world - 1]:
            a_, b_ = one.MapModel(g), navs[g // Pl].MapModel(g % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "particle %d" % g
    assert nres >= 1
    one.close()
    for nv in navs:
        nv.close()


# ---- the emit body's runs: long ones (cut at four measurements), a queue longer than one chunk ---------------------------------------
@pytest.mark.parametrize("C,M,targets,chain_max", [(60, 64, 3, None), (60, 64, 3, 0), (600, 64, 4, 0), (300, 128, 12, 0)])
def test_emit_runs_longer_than_four_and_queues_longer_than_a_chunk(nav_mod, monkeypatch, C, M, targets, chain_max):
    """Every measurement is a noisy sighting of one of a FEW components, so a component is queued with a dozen or more measurements
    (a lane's run is cut at four: the component's part is computed again for the next four) and, with 600 components x 64
    measurements on a small clutter density, a wave queues more than the 512 pairs of one chunk. Corrected (as a set) and pruned
    mixtures, set log-likelihood and alpha against the oracle, through the chain and through the separate kernels."""
    from monorfs_amd.synth import measure_perfect_identity
    from test_gpu_parity import assert_mix_close, match_unordered
    if chain_max is not None:
        monkeypatch.setenv("PHD_CHAIN_MAX", str(chain_max))
        monkeypatch.setenv("PHD_SPLIT", "2")
    P = 3
    f = Frame(P, C, M, 540 + C + M, weight_profile="steady")
    rng = np.random.default_rng(541)
    pick = rng.choice(C, targets, replace=False)
    base = np.array(f.mean[0])                           # (particle 0's means: the others differ by the frame's small jitter)
    z = measure_perfect_identity(base[pick[rng.integers(targets, size=M)]]) + rng.normal(size=(M, 3)) * np.sqrt([2.0, 2.0, 1e-3])
    f.z = z
    w = np.array(f.w)
    w[:, pick] = rng.uniform(0.6, 1.2, targets)          # the sighted components are heavy: their updates reach MinWeight
    f.w = w
    nav, p = _handle(nav_mod, f, maxq=max(600, C))
    nav.run_stages(f.z, with_alpha=True)
    alpha, setll = nav.WeightAlpha(), nav.SetLogLikelihood()
    longest = 0
    for i in range(P):
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        cor = orc.correct(p, f.poses[i], f.z, pred)
        keep = ~(cor[0] < p.min_weight)
        npred = len(pred[0])
        det = np.flatnonzero(keep[npred:])
        if len(det):
            longest = max(longest, int(np.bincount(det % npred).max()))   # (the corrected list is measurement-major: index = np + k np + c)
        match_unordered(nav.CorrectConditional(i), tuple(x[keep] for x in cor), 1e-9)
        pr = orc.prune(p, cor)
        assert_mix_close(nav.PruneModel(i), pr, 1e-7, "prune[%d]" % i)
        a, sll = orc.weight_alpha(p, f.poses[i], f.z, pred, pr)
        assert np.isclose(setll[i], sll, rtol=1e-9, atol=1e-9)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0)
    assert longest > 4, "no component met more than four measurements (%d): the frame does not exercise the cut" % longest
    nav.close()


def test_sharded_steps_at_the_sizes_the_grid_kernels_take_by_default(nav_mod):
    """8 ranks x 1024 particles = 8192 slots: the default thresholds (no environment switch) put the resampling on its grid kernels
    with the plan's counting in their last launch, then k_plan_lists — against ONE handle of 8192 particles (whose steps end on the
    same grid resampling): weights, poses, sampled maps bit for bit over steps that resample and one that does not, landing flags on."""
    from test_gpu_round4 import _device_path_handles, _device_path_step
    world, Pl, Cc, M = 8, 1024, 24, 10
    f = _depleted_frame(world, Pl, Cc, M, 5400)
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=96, max_measurements=M)
    p1.max_quantity = 96
    one = nav_mod.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(nav_mod, f, world, Pl, M, maxc=96, maxq=96)
    for nv in navs:
        if nv._lib.phd_migration_recv_is_finegrained(nv._h) == 1:
            nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    rng = np.random.default_rng(5401)
    nres = 0
    for step, u in enumerate((0.31, 0.77, 0.12, 0.6, 0.45)):
        one.SlamUpdate(None, f.z, u_resample=u)
        _device_path_step(navs, Pl, u)
        nres += int(one.resample_sources()[1])
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs])), "step %d" % step
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs])), "step %d" % step
        for g in rng.choice(Pl * world, 24, replace=False):
            a_, b_ = one.MapModel(int(g)), navs[int(g) // Pl].MapModel(int(g) % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "step %d particle %d" % (step, g)
    assert 1 <= nres, "no step resampled"
    one.close()
    for nv in navs:
        nv.close()


def test_landing_flags_are_refused_on_ordinary_receive_buffers(nav_mod, monkeypatch):
    """phd_migration_set_landing(1) needs fine-grained receive buffers (a peer's store must become visible while the receiver's
    kernel runs): with PHD_COARSE_RECV=1 — bench.py's fallback when IPC on fine-grained memory is refused — the call fails loudly
    and the handle keeps the caller's collective as landing barrier (0 is always accepted)."""
    monkeypatch.setenv("PHD_COARSE_RECV", "1")
    f = Frame(32, 40, 12, 503, weight_profile="steady")
    nav, _ = _handle(nav_mod, f)
    lib = nav._lib
    assert lib.phd_migration_recv_is_finegrained(nav._h) == 0
    with pytest.raises(Exception):
        nav._check(lib.phd_migration_set_landing(nav._h, 1))
    nav._check(lib.phd_migration_set_landing(nav._h, 0))
    nav.close()


@pytest.mark.parametrize("inline", ["0", "1"])
def test_landing_wait_in_its_own_wave_and_inside_the_unpack_kernel(nav_mod, monkeypatch, inline):
    """the two places of the receiver's wait (k_wait_landing in front of k_finish_sharded, the default; PHD_LANDING_INLINE=1: inside
    its workgroups) give the same sharded steps, bit for bit the single handle"""
    from test_gpu_round4 import _device_path_handles, _device_path_step
    monkeypatch.setenv("PHD_LANDING_INLINE", inline)
    world, Pl, Cc, M = 3, 64, 70, 18
    f = _depleted_frame(world, Pl, Cc, M, 5500)
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one = nav_mod.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(nav_mod, f, world, Pl, M)
    for nv in navs:
        if nv._lib.phd_migration_recv_is_finegrained(nv._h) != 1:
            pytest.skip("no fine-grained receive buffers on this box")
        nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    nres = 0
    for u in (0.31, 0.77, 0.12):
        one.SlamUpdate(None, f.z, u_resample=u)
        _device_path_step(navs, Pl, u)
        nres += int(one.resample_sources()[1])
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs]))
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs]))
        for g in range(0, Pl * world, 7):
            assert all(np.array_equal(x, y) for x, y in zip(one.MapModel(g), navs[g // Pl].MapModel(g % Pl))), "particle %d" % g
    assert nres >= 1
    one.close()
    for nv in navs:
        nv.close()


# ---- two workgroups per particle in the one-launch chain: the density sums of WeightAlpha on a helper workgroup, beside the association -----
@pytest.mark.parametrize("shape", [(40, 150, 24, "steady"), (256, 128, 32, "steady"), (256, 128, 32, "survey"), (24, 400, 40, "survey"),
                                   (7, 60, 5, "steady"), (100, 128, 32, "steady")])
def test_helper_workgroups_of_the_chain_change_no_bit(nav_mod, monkeypatch, shape):
    """k_particle_chain with a helper workgroup per particle (PHD_DSPLIT_MAX particles and fewer): the helper runs alpha_density_body beside
    the main workgroup's association and the two meet in alpha_meet — the same body and the same last line as without helpers, so
    the SAME bits whether the helper takes the sums (helper in time, on the main's XCD), the main keeps them (PHD_DSPLIT_LATE=1:
    every helper reports 0.5 ms late; 3: helpers nobody picks) or there are no helpers at all (PHD_DSPLIT_MAX=0). The first mode against
    the oracle, so that "equal" means "right"; a frame whose map estimate outgrows the densities' LDS arrays (more than 128 landmarks:
    never handed over) among them."""
    P, Cc, M, prof = shape
    f = Frame(P, Cc, M, 511, weight_profile=prof)
    got = {}
    for mode in ("helpers", "again", "late", "unpicked", "off", "other-end", "numbers-start-again"):
        monkeypatch.setenv("PHD_DSPLIT_MAX", "0" if mode == "off" else "256")
        # (the launches' numbers are 27 bits long: the handle of the last mode starts two launches short of their end)
        monkeypatch.setenv("PHD_DSPLIT_SEQ0", str(0x07ffffff - 3) if mode == "numbers-start-again" else "0")
        if mode == "other-end":   # the step's end inside the chain's launch (PHD_FOLD_NR=1; the default is a launch behind it): the finishers' count
            monkeypatch.setenv("PHD_FOLD_NR", "1")
        monkeypatch.setenv("PHD_DSPLIT_LATE", {"late": "1", "unpicked": "3"}.get(mode, "0"))
        nav, p = _handle(nav_mod, f)
        nav.run_stages(f.z)               # (the stages of one update, for WeightAlpha's values themselves; the state stays)
        alphas = [nav.WeightAlpha(), nav.SetLogLikelihood()]
        for step in range(3):
            nav.SlamUpdate(None, f.z + 0.05 * step, u_resample=0.3 + 0.2 * step)
        got[mode] = (nav.VehicleWeights, nav.resample_sources()[0], [nav.MapModel(i) for i in (0, P // 2, P - 1)], alphas)
        if mode == "helpers":
            st = orc.State(P, 700)
            st.poses[:] = f.poses
            st.w[:, :Cc], st.mean[:, :Cc], st.cov[:, :Cc], st.n[:] = f.w, f.mean, f.cov, Cc
            for step in range(3):
                _, src, _, _ = orc.slam_update(p, st, f.z + 0.05 * step, u=0.3 + 0.2 * step, threads=4)
            assert np.array_equal(got[mode][1], src)
            assert np.allclose(got[mode][0], st.weights, rtol=1e-6, atol=1e-300)
        nav.close()
    monkeypatch.delenv("PHD_FOLD_NR", raising=False)
    monkeypatch.delenv("PHD_DSPLIT_SEQ0", raising=False)
    for mode in ("again", "late", "unpicked", "off", "other-end", "numbers-start-again"):
        assert np.array_equal(got[mode][0], got["helpers"][0]) and np.array_equal(got[mode][1], got["helpers"][1]), mode
        for a, b in zip(got[mode][3], got["helpers"][3]):
            assert np.array_equal(a, b), mode
        for a, b in zip(got[mode][2], got["helpers"][2]):
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), mode
