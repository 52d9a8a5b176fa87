"""GPU cases added in round 4 (through the C-ABI), the branches the round-3 review found untested:
  * k_prune_merge's ordering when hundreds of weights are bit-equal: the histogram bin of the ranked path overflows
    (more than 512 entries in one bin), the cut at MaxQuantity falls inside such a run, the kept entries outgrow the
    ranked path's arrays (its one-pass and its several-pass fallbacks) — each against orc.prune, order exact;
  * the migration plan's un-staged path (a global source vector that does not fit the plan kernel's LDS);
  * a generated-scene counterpart of the reference's one end-to-end acceptance, SimulationTest.perfectparticle
    (Test/SimulationTest.cs:151-223): the particle that is handed the true pose wins most often."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import params_from_dict, prm3d_defaults
from test_gpu_kat import KAT
from test_gpu_round3 import check_plans


@pytest.fixture(scope="module")
def nav_mod():
    from monorfs_amd import navigator
    return navigator


def far_grid(n, spacing=5.0):
    """n means on a plane grid, far enough apart that nothing merges (covariance 0.01 I, MergeThreshold 0.3)"""
    side = int(np.ceil(np.sqrt(n)))
    gx, gy = np.meshgrid(np.arange(side) * spacing, np.arange(side) * spacing)
    means = np.column_stack([gx.ravel()[:n], gy.ravel()[:n], np.zeros(n)])
    covs = np.broadcast_to(np.diag([0.01, 0.01, 0.01]), (n, 3, 3)).copy()
    return means, covs


def prune_through_the_device(nav_mod, w, maxq, max_components):
    """PruneModel of (w, far-apart components) on the device: the mixture is the prior of a Linear2D particle far outside
    the visible square, so every detection probability is 0 and the correction step passes it on unchanged (as
    test_gpu_kat.test_prune_kat_on_the_device does with the reference's own vector)."""
    n = len(w)
    means, covs = far_grid(n)
    params = dict(KAT["params"], max_quantity=maxq)
    p = params_from_dict(params, max_particles=1, max_components=max_components, max_measurements=8)
    pose = [100000.0, -100000.0]
    nav = nav_mod.PHDNavigator(p, particlecount=1, pose=[pose[0], pose[1], 0, 1, 0, 0, 0])
    nav.reset(np.array([pose[0], pose[1], 0, 1.0, 0, 0, 0]), (w, means, covs), 1)
    nav.run_stages(np.zeros((0, 3)), with_alpha=False)
    cw, _, _ = nav.CorrectConditional(0)
    assert len(cw) == n and np.array_equal(np.sort(cw), np.sort(w)), "the correction step did not pass the mixture on unchanged"
    got = nav.PruneModel(0)
    want = orc.prune(p, (w, means, covs))
    nav.close()
    assert len(got[0]) == len(want[0]) == min(n, maxq)
    # no merge happens: weights and means leave PruneModel as they came (Merge of one component), in the reference's order
    assert np.array_equal(got[0], want[0]), "weights / order: first difference at %d" % int(np.flatnonzero(got[0] != want[0])[0])
    assert np.array_equal(got[1], want[1]), "means / order: first difference at row %d" % int(np.flatnonzero((got[1] != want[1]).any(axis=1))[0])
    return got


@pytest.mark.parametrize("n,maxq,cap", [(600, 600, 640), (1100, 1024, 1152), (1100, 600, 1152)])
def test_prune_with_hundreds_of_bit_equal_weights(nav_mod, n, maxq, cap):
    """a unit-weight map (what BestMapEstimate hands out, Map.cs:134) seen from outside the field of view: n bit-equal
    weights in ONE histogram bin (more than the 512 the ranked path ranks inside a bin) — List.Sort's order made stable by
    the canonical index (PHDNavigator.cs:920), exact"""
    got = prune_through_the_device(nav_mod, np.ones(n), maxq, cap)
    assert np.all(got[0] == 1.0)


def test_prune_with_513_equal_weights_straddling_the_cut(nav_mod):
    """200 heavier distinct weights, 513 bit-equal ones, 187 lighter: MaxQuantity 400 cuts inside the run of equals (the
    first 200 of them, by canonical index, survive)"""
    rng = np.random.default_rng(4)
    w = np.concatenate([rng.uniform(0.6, 0.9, 200), np.full(513, 0.5), rng.uniform(0.01, 0.4, 187)])
    w = w[rng.permutation(len(w))]
    got = prune_through_the_device(nav_mod, w, 400, 960)
    assert np.count_nonzero(got[0] == 0.5) == 200


@pytest.mark.parametrize("n,heavy", [(2000, 500), (2500, 0), (1400, 100)])
def test_prune_when_the_kept_set_outgrows_the_ranked_path(nav_mod, n, heavy):
    """the entries from the threshold bin up are more than the ranked path's arrays hold (capN = 1365 at MaxQuantity 600):
    distinct weights that share one histogram bin (same exponent and leading five mantissa bits), with MaxQuantity cutting
    inside it — (2000, 500): 2000 entries at or above the threshold bin, the one-pass sort of those; (2500, 0): more than the
    sort width too, the several-pass sort; (1400, 100): just past capN"""
    rng = np.random.default_rng(n)
    crowd = 1.0 + rng.permutation(n - heavy) * (1.0 / 32 / (n - heavy + 1))   # distinct, all in [1, 1 + 1 / 32)
    w = np.concatenate([rng.uniform(1.5, 3.0, heavy), crowd])
    w = w[rng.permutation(n)]
    prune_through_the_device(nav_mod, w, 600, ((n + 63) // 64) * 64)


def test_prune_when_small_bins_outgrow_the_ranked_path(nav_mod):
    """three bins of 500 distinct weights each, MaxQuantity 1024: no bin is crowded (500 <= 512) but the entries from the
    threshold bin up (1500) are more than the ranked path's arrays hold (1365): its one-pass fallback, by capacity alone"""
    rng = np.random.default_rng(15)
    w = np.concatenate([b * (1.0 + rng.permutation(500) * (1.0 / 32 / 501)) for b in (1.0, 1.0 + 1.0 / 32, 1.0 + 2.0 / 32)])
    w = w[rng.permutation(len(w))]
    prune_through_the_device(nav_mod, w, 1024, 1536)


# ---- ADVICE (round 3): the plan kernel's un-staged path (source vector read from global memory) -------------------------------
@pytest.mark.parametrize("Pl,world,power", [(8192, 8, 40), (2048, 32, 12), (5000, 9, 3)])
def test_device_migration_plan_with_a_vector_beyond_its_lds(nav_mod, Pl, world, power):
    """more than ~37 K global particles: k_plan_migration reads the source vector from global memory (the slot before from
    the neighbouring lane) instead of staging it"""
    p = prm3d_defaults(max_particles=Pl, max_components=64, max_measurements=8)
    p.max_quantity = 64
    nav = nav_mod.PHDNavigator(p, particlecount=Pl)
    rng = np.random.default_rng(Pl + world)
    Pg = Pl * world
    for trial in range(2):
        w = rng.random(Pg) ** power + 1e-300
        w /= w.sum()
        gsrc = orc.resample(w, float(rng.uniform(0.01, 0.99)))[0].astype(np.int32)
        assert np.all(np.diff(gsrc) >= 0)
        check_plans(nav, gsrc, Pl, world)
    check_plans(nav, np.full(Pg, Pg - 1, np.int32), Pl, world)
    check_plans(nav, np.arange(Pg, dtype=np.int32), Pl, world)
    bad = np.arange(Pg, dtype=np.int32)
    bad[Pg // 2], bad[Pg // 2 + 1] = bad[Pg // 2 + 1], bad[Pg // 2]
    assert nav.test_migration_plan(bad, Pl, world, world // 2)["status"] == 2
    nav.close()


# ---- SimulationTest.perfectparticle on a generated scene ------------------------------------------------------------------------
def room_scene(rng, n=40, depth=0.9, half=0.9):
    """landmarks on the walls of a small room around the origin, most of them ahead of the camera (+z): close enough that
    one frame of odometry noise (2.4 mm) moves the pixels about as much as the measurement noise does"""
    pts = []
    while len(pts) < n:
        wall = rng.choice(4, p=[0.55, 0.05, 0.2, 0.2])
        a, b = rng.uniform(-half, half), rng.uniform(-0.6, 0.6)
        pts.append([[a, b, depth], [a, b, -depth], [depth, b, a], [-depth, b, a]][wall])
    return np.array(pts)


@pytest.mark.slow
def test_perfect_particle_wins_most_often(nav_mod):
    """SimulationTest.perfectparticle (Test/SimulationTest.cs:151-223) with the device as the navigator: in every frame one
    particle is handed the true pose and the map it built so far; over a run it should be the best particle at least as
    often as any other, in more than half of the runs (the reference's assertion, its 20 runs and 20 particles; 60 frames
    on a generated room — assets/map.world and movroom.in are not in the reference's repository). The CPU oracle plays
    the same runs from the same random numbers: best particle, resampling decision and sources must agree in every frame,
    so the two reach the same verdict by the same history."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from simulate import SimulatedVehicle
    from monorfs_amd import recordio as rio
    nparticles, iterations, nloops = 20, 20, 60
    config = rio.default_config()
    measurer = [575.8156, 0.1, 2.0, -320, -240, 640, 480]
    frame = 1.0 / 30
    motion_chol = np.linalg.cholesky(config["MotionCovarianceMultiplier"] * np.array(config["MotionCovariance"], float))
    empty = (np.zeros(0), np.zeros((0, 3)), np.zeros((0, 3, 3)))
    success, rounds_all, resamplings = 0, [], 0
    for h in range(iterations):
        rng = np.random.default_rng(1000 + h)
        landmarks = room_scene(rng)
        pose0 = np.array([0, 0, 0, 1.0, 0, 0, 0])
        explorer = SimulatedVehicle(pose0, measurer, landmarks, config, rng)
        p = rio.phd_params_from_config(config, measurer=[measurer[0], float(np.float32(measurer[1])), float(np.float32(measurer[2]))] + measurer[3:],
                                       max_particles=nparticles, max_measurements=128)
        nav = nav_mod.PHDNavigator(p, particlecount=nparticles, pose=pose0)
        st = orc.State(nparticles, 700)
        st.poses[:] = pose0
        good, ogood = empty, empty
        nbest = np.zeros(nparticles, int)
        missed = False
        for loop in range(nloops):
            reading = np.array([0, 0, 0.004, 0, 0.02 * np.sin(loop / 9.0), 0])   # ds forward, a slow yaw to and fro
            explorer.update(frame, reading)
            noise = frame * (rng.normal(size=(nparticles, 6)) @ motion_chol.T)
            nav.UpdateOdometry(None, reading, noise)
            poses = nav.poses()
            assert np.allclose(poses, orc.update_motion(st.poses, reading, noise), rtol=0, atol=1e-14), "run %d frame %d: the motion step" % (h, loop)
            st.poses[:] = poses                     # (the oracle goes on from the device's poses: its sin / cos differ in the last bit)
            poses[0] = explorer.pose                # updatehook: the perfect particle
            nav.set_poses(poses)
            nav.set_map(0, good)
            st.poses[0] = explorer.pose
            k = len(ogood[0])
            st.n[0] = k
            st.w[0, :k], st.mean[0, :k], st.cov[0, :k] = ogood
            z = explorer.measure()
            u = float(rng.uniform(1e-9, 1.0))
            nav.SlamUpdate(None, z, u_resample=u)
            prev = st.weights.copy()
            obest, osrc, ores, oalpha = orc.slam_update(p, st, z, u=u, threads=4)
            src, res = nav.resample_sources()
            assert res == ores and np.array_equal(src, osrc), "run %d frame %d: the device and the oracle part ways" % (h, loop)
            top = np.sort(prev * oalpha)[::-1]
            if top[0] - top[1] > 1e-9 * top[0]:     # (the first frames: twenty particles with empty maps weigh the same to the last bits, the argmax is anybody's)
                assert nav.BestParticle == obest, "run %d frame %d: best particle %d, oracle %d" % (h, loop, nav.BestParticle, obest)
            resamplings += int(res)
            poses = nav.poses()
            found = np.flatnonzero((poses == explorer.pose).all(axis=1))
            if len(found):
                good = nav.MapModel(int(found[0]))
                ogood = tuple(x.copy() for x in st.map(int(found[0])))
            else:
                missed = True                       # lost in the resampling: the run is judged on the frames before
            if not missed:
                nbest[nav.BestParticle] += 1
        nav.close()
        rounds_all.append(int(nbest.sum()))
        success += int(np.all(nbest[0] >= nbest))
    print("perfectparticle: rounds per run", rounds_all, "resamplings", resamplings, "success rate", success / iterations)
    assert resamplings > iterations, "hardly a resampling: the runs do not exercise what the test is about"
    assert success / iterations > 0.5


# ---- the sharded step without a host wait, shards in one process (phd_migration_set_peers) ---------------------------------------
class _Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def _device_path_handles(nav_mod, f, world, Pl, M, over=None, maxc=600, maxq=600):
    import ctypes as C
    import torch
    navs = []
    planes = f.planes()
    for r in range(world):
        pr = prm3d_defaults(max_particles=Pl, max_components=maxc, max_measurements=M)
        pr.max_quantity = maxq
        for k, v in (over or {}).get(r, {}).items():
            setattr(pr, k, v)
        nv = nav_mod.PHDNavigator(pr, particlecount=Pl)
        sl = slice(r * Pl, (r + 1) * Pl)
        nv.upload_state(planes[:, sl], f.counts[sl], f.poses[sl], f.weights[sl])
        nv.set_measurements(f.z)
        nv._check(nv._lib.phd_set_stream(nv._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
        navs.append(nv)
    lib = navs[0]._lib
    recv = (C.c_void_p * world)(*[lib.phd_migration_recv_buffer(nv._h) for nv in navs])
    for r, nv in enumerate(navs):
        nv._check(lib.phd_migration_set_peers(nv._h, recv, r, world))
    return navs


def _device_path_step(navs, Pl, u, sync=True):
    """one sharded step of every handle, the all-gather played by device copies; every enqueue of every rank before any wait"""
    import ctypes as C
    import torch
    lib, world = navs[0]._lib, len(navs)
    for nv in navs:
        nv._check(lib.phd_step_local_async(nv._h, 0))
    lws = [torch.as_tensor(_Dev(lib.phd_device_local_weights(nv._h), Pl + 1), device="cuda") for nv in navs]
    allw = torch.cat(lws)
    for nv in navs:
        torch.as_tensor(_Dev(lib.phd_device_gather_buffer(nv._h, world), world * (Pl + 1)), device="cuda").copy_(allw)
    for r, nv in enumerate(navs):
        nv._check(lib.phd_step_global_device_async(nv._h, r, world, C.c_double(u), 0))
        nv._check(lib.phd_migration_push_async(nv._h))
    for nv in navs:                 # (one stream orders everything: all pushes are behind us, as behind the landing barrier)
        nv._check(lib.phd_migration_unpack_async(nv._h))
    if sync:
        for nv in navs:
            nv.sync()


@pytest.mark.parametrize("world", [2, 5])
def test_device_path_sharded_step_equals_single_handle(nav_mod, world):
    """phd_step_global_device_async + phd_migration_push_async (the sequence bench.py --gpus N runs, here with `world`
    handles in one process on one GPU): weights, poses, maps bit for bit those of one handle holding all particles, over
    steps that resample and migrate; the steps are enqueued without a single host wait in between."""
    from monorfs_amd.synth import Frame
    Pl, Cc, M = 56, 70, 18
    f = Frame(Pl * world, Cc, M, 4100 + world, weight_profile="steady")
    f.weights = np.random.default_rng(world).random(f.P) ** 12      # depleted from the start: the first step resamples, long runs cross the rank boundaries
    f.weights /= f.weights.sum()
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one = nav_mod.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    one.set_measurements(f.z)
    navs = _device_path_handles(nav_mod, f, world, Pl, M)
    nres = 0
    us = [0.31, 0.77, 0.12, 0.55]
    one.SlamUpdate(None, f.z, u_resample=us[0])
    _device_path_step(navs, Pl, us[0])
    nres += int(one.resample_sources()[1])
    for u in (0.44, 0.91):          # two steps back to back, nothing waited for in between
        one.step_async(u)
        _device_path_step(navs, Pl, u, sync=False)
    one.sync()
    for nv in navs:
        nv.sync()
    for u in us[2:] + [None]:
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs]))
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs]))
        for g in list(range(0, Pl * world, 9)) + [Pl * world - 1]:
            a_, b_ = one.MapModel(g), navs[g // Pl].MapModel(g % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "particle %d" % g
        nres += int(one.resample_sources()[1])
        if u is not None:
            one.SlamUpdate(None, f.z, u_resample=u)
            _device_path_step(navs, Pl, u)
    assert nres >= 1, "the sequence did not resample: the migration was not exercised"
    one.close()
    for nv in navs:
        nv.close()


def test_device_path_a_flag_on_one_rank_drops_the_step_on_all(nav_mod):
    """the status words travel with the weights: rank 1's emit capacity is too small for its corrected mixtures, its step
    raises the flag — every rank drops the step (state as before it), rank 1's phd_sync names the capacity, the others say
    that another rank dropped it; then the handles go on"""
    from monorfs_amd.synth import Frame
    world, Pl, Cc, M = 3, 24, 60, 16
    f = Frame(Pl * world, Cc, M, 4242, weight_profile="steady")
    navs = _device_path_handles(nav_mod, f, world, Pl, M, over={1: {"emit_capacity": 40}}, maxc=64, maxq=40)   # rank 1: 64 emit slots for ~80 entries
    before = [(nv.VehicleWeights, nv.poses(), nv.MapModel(3)) for nv in navs]
    _device_path_step(navs, Pl, 0.4, sync=False)
    status = []
    for nv in navs:
        with pytest.raises(nav_mod.PHDError) as e:
            nv.sync()
        status.append(e.value.status)
    assert status[1] == 2 and status[0] == -1 and status[2] == -1, status
    for nv, (w, q, m) in zip(navs, before):
        assert np.array_equal(nv.VehicleWeights, w) and np.array_equal(nv.poses(), q)
        assert all(np.array_equal(x, y) for x, y in zip(nv.MapModel(3), m))
    # without measurements nothing is emitted beyond the copies: the step fits, and runs on every rank
    for nv in navs:
        nv.set_measurements(np.zeros((0, 3)))
    _device_path_step(navs, Pl, 0.4)
    for nv in navs:
        nv.close()


@pytest.mark.parametrize("grid_nr", [False, True])
def test_deferred_replay_of_big_clusters_is_the_same_step(nav_mod, monkeypatch, grid_nr):
    """PHD_DEFER_BIG=1 (read when a handle is created): the particles whose association has a cluster of more than ten rows
    are listed, their ordered replay runs inside the launch of the densities, alpha is finished by the resampling kernel —
    weights, sources and maps bit for bit those of the default path, on a frame with clusters of more than ten rows.
    grid_nr: the step ends on the grid resampling kernels (round 5; forced on this small set), whose first launch finishes alpha."""
    from test_gpu_parity import clustered_frame
    from test_gpu_round2 import make_nav
    f = clustered_frame(91, 3, 8, 7, spread_px=4.0)        # three groups of 8 landmarks and 7 measurements a few pixels apart: clusters of up to 15 rows
    if grid_nr:
        monkeypatch.setenv("PHD_NR_GRID_MIN", "1")
    results = []
    for defer in ("0", "1"):
        monkeypatch.setenv("PHD_DEFER_BIG", defer)
        monkeypatch.setenv("PHD_CHAIN_MAX", "0")      # (the separate kernels, not the one-launch chain of small particle sets)
        nav, p = make_nav(nav_mod, f, merge_threshold=1e-3, emit_capacity=12000)
        out = []
        for u in (0.3, 0.8):
            nav.SlamUpdate(None, f.z, u_resample=u)
            out.append((nav.VehicleWeights, nav.resample_sources(), nav.BestParticle, [nav.MapModel(i) for i in (0, f.P - 1)]))
        nav.run_stages(f.z, with_alpha=True)
        out.append((nav.WeightAlpha(), nav.SetLogLikelihood()))
        results.append(out)
        nav.close()
    a, b = results
    for (wa, sa, ba, ma), (wb, sb, bb, mb) in zip(a[:2], b[:2]):
        assert np.array_equal(wa, wb) and sa[1] == sb[1] and np.array_equal(sa[0], sb[0]) and ba == bb
        for x, y in zip(ma, mb):
            assert all(np.array_equal(u, v) for u, v in zip(x, y))
    assert np.array_equal(a[2][0], b[2][0]) and np.array_equal(a[2][1], b[2][1])
    assert np.all(np.isfinite(a[2][1]))


@pytest.mark.parametrize("shape,profile", [((256, 128, 32), "steady"), ((256, 128, 32), "survey"), ((96, 40, 20), "steady"), ((512, 64, 70), "steady")])
def test_chain_ends_the_step_itself_as_the_separate_launch_does(nav_mod, monkeypatch, shape, profile):
    """Small particle sets: the last workgroup of k_particle_chain to take its ticket runs k_normalise_resample's body (no launch
    of its own). Weights, resampling sources, BestParticle, poses and maps after several un-frozen steps — resampled and not,
    localising and mapping-only — are bit for bit those of the separate launch (PHD_FOLD_NR=0, read when a handle is created)."""
    from monorfs_amd.synth import Frame
    from test_gpu_round2 import make_nav
    f = Frame(shape[0], shape[1], shape[2], 1002, weight_profile=profile)
    results = []
    for fold in ("1", "0"):
        monkeypatch.setenv("PHD_FOLD_NR", fold)
        nav, p = make_nav(nav_mod, f)
        out = []
        for k, u in enumerate((0.3, 0.8, 0.05, 0.6, 0.95)):
            nav.OnlyMapping = (k == 3)
            nav.SlamUpdate(None, f.z, u_resample=u)
            src, resampled = nav.resample_sources()
            out.append((nav.VehicleWeights.copy(), np.array(src).copy(), resampled, nav.BestParticle, [nav.MapModel(i) for i in (0, f.P // 2, f.P - 1)]))
        results.append(out)
        nav.close()
    a, b = results
    if profile == "steady":
        assert any(s[2] for s in a)   # (the resampling branch was taken)
    for (wa, sa, ra, ba, ma), (wb, sb, rb, bb, mb) in zip(a, b):
        assert np.array_equal(wa, wb) and ra == rb and np.array_equal(sa, sb) and ba == bb
        for x, y in zip(ma, mb):
            assert all(np.array_equal(u, v) for u, v in zip(x, y))


@pytest.mark.parametrize("mode", ["device", "events"])
def test_steps_posted_back_to_back_end_on_the_stream_that_finishes_last(nav_mod, monkeypatch, mode):
    """Two sub-range streams (1024 particles and more), phd_step_async after phd_step_async: no fork between the steps, the end of
    the step behind the k_alpha_density of the stream that finishes last — ordered by events (the default, PHD_DEVICE_ORDER=0) or on
    the device (tickets + k_gate, PHD_DEVICE_ORDER=1: an option). Twelve un-frozen steps, resampled and not, a mapping-only step and an upload in between: weights,
    sources, BestParticle and maps bit for bit those of a handle that forks and joins around every step (PHD_PIPELINE=0)."""
    from monorfs_amd.synth import Frame
    from test_gpu_round2 import make_nav
    f = Frame(1024, 48, 24, 1002, weight_profile="steady")
    results = []
    for env in ({"PHD_PIPELINE": "0"}, {"PHD_PIPELINE": "1", "PHD_DEVICE_ORDER": "1" if mode == "device" else "0"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        nav, p = make_nav(nav_mod, f)
        nav.set_measurements(f.z)
        out = []
        for k, u in enumerate((0.3, 0.8, 0.05, 0.6, 0.95, 0.2, 0.5, 0.7, 0.1, 0.9, 0.4, 0.65)):
            nav.OnlyMapping = (k == 4)
            nav.step_async(u)
            if k == 7:
                nav.set_measurements(f.z[::-1].copy())   # (something else on the stream between two steps: the next one forks again)
            if k in (2, 9, 11):
                nav.sync()
                src, resampled = nav.resample_sources()
                out.append((nav.VehicleWeights.copy(), np.array(src).copy(), resampled, nav.BestParticle, [nav.MapModel(i) for i in (0, 511, 512, 1023)]))
        results.append(out)
        nav.close()
    a, b = results
    for (wa, sa, ra, ba, ma), (wb, sb, rb, bb, mb) in zip(a, b):
        assert np.array_equal(wa, wb) and ra == rb and np.array_equal(sa, sb) and ba == bb
        for x, y in zip(ma, mb):
            assert all(np.array_equal(u, v) for u, v in zip(x, y))
