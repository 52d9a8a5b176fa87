"""BASELINE.json's full sizes on the GPU, through the C-ABI: size-independent properties of the step plus an
oracle check on a sample of particles (the whole oracle step at these sizes takes minutes on the host).
  B: 2048 particles x 512 components x 64 measurements      S: 4096 x 1024 x 128, MaxQuantity 1024"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import CONFIGS, Frame


def setup(cfg, profile):
    from monorfs_amd import navigator
    P, C, M, seed = CONFIGS[cfg]
    f = Frame(P, C, M, seed, weight_profile=profile)
    maxq = max(600, C)
    p = prm3d_defaults(max_particles=P, max_components=maxq, max_measurements=M)
    p.max_quantity = maxq
    nav = navigator.PHDNavigator(p, particlecount=P)
    nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    return nav, p, f


@pytest.mark.parametrize("cfg,profile", [("B", "steady"), ("B", "survey"), ("S", "steady"), ("S", "survey")])
def test_full_size_properties_and_sample(cfg, profile):
    nav, p, f = setup(cfg, profile)
    nav.run_stages(f.z, with_alpha=True)
    alpha = nav.WeightAlpha()
    assert alpha.shape == (f.P,) and np.all(np.isfinite(alpha)) and np.all(alpha >= 0)
    rng = np.random.default_rng(1)
    for i in rng.choice(f.P, 6, replace=False):
        cw, cm, cc = nav.CorrectConditional(i)
        pw, pm, pc = nav.PruneModel(i)
        # every emitted weight reaches MinWeight; the pruned map holds at most MaxQuantity components, in
        # non-increasing order of the candidates' weights; merging conserves the weight of what it keeps
        assert np.all(cw >= p.min_weight)
        assert len(pw) <= p.max_quantity
        kept = np.sort(cw)[::-1][:p.max_quantity]
        assert np.isclose(pw.sum(), kept.sum(), rtol=1e-12)
        # covariances stay symmetric positive definite
        assert np.all(np.linalg.eigvalsh(pc) > 0)
        # oracle on this particle
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        opr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert len(opr[0]) == len(pw)
        assert np.allclose(pw, opr[0], rtol=1e-7) and np.allclose(pm, opr[1], rtol=1e-7, atol=1e-11)
        iu = np.triu_indices(3)
        assert np.allclose(pc[:, iu[0], iu[1]], opr[2][:, iu[0], iu[1]], rtol=1e-7, atol=1e-13)
        a, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, opr)
        assert np.isclose(alpha[i], a, rtol=1e-6, atol=0)
    nav.close()


def test_full_size_step_resamples_consistently():
    nav, p, f = setup("B", "steady")
    nav.SlamUpdate(None, f.z, u_resample=0.41)
    w = nav.VehicleWeights
    src, res = nav.resample_sources()
    assert res, "the steady frame is expected to deplete the particle set"
    assert np.all(np.diff(src) >= 0) and src.min() >= 0 and src.max() < f.P     # systematic resampling is monotone
    assert np.allclose(w, 1.0 / f.P)
    # a resampled slot holds a copy of its source: equal maps for equal sources
    dup = np.flatnonzero(np.diff(src) == 0)
    assert len(dup) > 0
    i = int(dup[0])
    a, b = nav.MapModel(i), nav.MapModel(i + 1)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    assert np.array_equal(nav.poses()[i], nav.poses()[i + 1])
    # idempotence of the frozen step: the same input gives the same output, bit for bit
    nav2, _, _ = setup("B", "steady")
    nav2.SlamUpdate(None, f.z, u_resample=0.41)
    src2, _ = nav2.resample_sources()
    assert np.array_equal(src, src2)
    m1, m2 = nav.MapModel(7), nav2.MapModel(7)
    assert all(np.array_equal(x, y) for x, y in zip(m1, m2))
    nav.close()
    nav2.close()


class _Dev:
    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_equals_single_handle(world):
    """The multi-GPU step sequence (local step, all-gather of the weights, global normalise/resample on every
    rank, plan, pack, all-to-all, unpack) played by `world` handles on ONE GPU, the exchanges done with
    device-to-device copies, must leave exactly the state a single handle holding all particles computes."""
    import ctypes as C
    import torch
    from monorfs_amd import navigator
    Pl, Cc, M = 96, 80, 20
    f = Frame(Pl * world, Cc, M, 77, weight_profile="steady")
    p1 = prm3d_defaults(max_particles=Pl * world, max_components=600, max_measurements=M)
    one = navigator.PHDNavigator(p1, particlecount=Pl * world)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = []
    planes = f.planes()
    for r in range(world):
        pr = prm3d_defaults(max_particles=Pl, max_components=600, max_measurements=M)
        nv = navigator.PHDNavigator(pr, particlecount=Pl)
        sl = slice(r * Pl, (r + 1) * Pl)
        nv.upload_state(planes[:, sl], f.counts[sl], f.poses[sl], f.weights[sl])
        nv.set_measurements(f.z)
        # kernels of every handle and the copies that stand in for the collectives share torch's current stream
        nv._check(nv._lib.phd_set_stream(nv._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
        navs.append(nv)
    lib = navs[0]._lib
    ip = C.POINTER(C.c_int32)
    Pg = Pl * world
    resampled = []
    for step in range(2):
        u = 0.3 + 0.2 * step
        one.SlamUpdate(None, f.z, u_resample=u)
        resampled.append(one.resample_sources()[1])
        # local steps + "all-gather"
        gws = [torch.as_tensor(_Dev(lib.phd_device_global_weights(nv._h, Pg), Pg), device="cuda") for nv in navs]
        for nv in navs:
            nv._check(lib.phd_step_local_async(nv._h, 0))
        lws = [torch.as_tensor(_Dev(lib.phd_device_local_weights(nv._h), Pl), device="cuda").clone() for nv in navs]
        for g in gws:
            g.copy_(torch.cat(lws))
        torch.cuda.synchronize()
        sc, rc, sends, recvs, recs = [], [], [], [], []
        for r, nv in enumerate(navs):
            nv._check(lib.phd_step_global_async(nv._h, r, world, C.c_double(u)))
            s, q = np.zeros(world, np.int32), np.zeros(world, np.int32)
            nv._check(lib.phd_migration_plan(nv._h, r, world, s.ctypes.data_as(ip), q.ctypes.data_as(ip)))
            nv._check(lib.phd_migration_pack_async(nv._h))
            sc.append(s); rc.append(q)
        for r, nv in enumerate(navs):
            bpp = C.c_int64(0)
            sp = lib.phd_migration_send_buffer(nv._h, C.byref(bpp))
            rp = lib.phd_migration_recv_buffer(nv._h)
            rec = bpp.value // 8
            recs.append(rec)
            sends.append(torch.as_tensor(_Dev(sp, max(int(sc[r].sum()), 1) * rec), device="cuda") if sp else None)
            recvs.append(torch.as_tensor(_Dev(rp, max(int(rc[r].sum()), 1) * rec), device="cuda") if rp else None)
        # "all-to-all": rank a's block for rank b lands in b's receive buffer after the blocks of ranks < a
        for b in range(world):
            off = 0
            for a_ in range(world):
                if a_ == b:
                    continue
                n = int(sc[a_][b])
                assert n == int(rc[b][a_])
                if n:
                    so = int(sum(sc[a_][x] for x in range(b) if x != a_))
                    recvs[b][off * recs[b]:(off + n) * recs[b]].copy_(sends[a_][so * recs[a_]:(so + n) * recs[a_]])
                off += n
        torch.cuda.synchronize()
        for nv in navs:
            nv._check(lib.phd_migration_unpack_async(nv._h))
            nv.sync()
        # compare with the single handle
        w_one = one.VehicleWeights
        w_sh = np.concatenate([nv.VehicleWeights for nv in navs])
        assert np.array_equal(w_one, w_sh), "step %d: weights differ" % step
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs]))
        for g in (0, Pl - 1, Pl, Pg - 1):
            a_, b_ = one.MapModel(g), navs[g // Pl].MapModel(g % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "step %d particle %d" % (step, g)
    assert any(resampled), "no step resampled: the migration path was not exercised"
    one.close()
    for nv in navs:
        nv.close()


def test_sharded_step_after_a_single_handle_resampling():
    """A single-handle step that resampled leaves the maps behind the slot indirection, and so does the sharded sequence
    (local sources are not copied, arrivals go to free slots of the bank the step wrote): the two kinds of step can
    follow each other. One handle takes [steps, step, step], the other [steps, sharded step with one rank, step]: the
    same state, bit for bit."""
    import ctypes as C
    import torch
    from monorfs_amd import navigator
    P, Cc, M = 64, 60, 16
    f = Frame(P, Cc, M, 88, weight_profile="steady")
    navs = []
    for _ in range(2):
        pr = prm3d_defaults(max_particles=P, max_components=600, max_measurements=M)
        nv = navigator.PHDNavigator(pr, particlecount=P)
        nv.upload_state(f.planes(), f.counts, f.poses, f.weights)
        navs.append(nv)
    a, b = navs
    lib = b._lib
    ip = C.POINTER(C.c_int32)
    rng = np.random.default_rng(89)
    for _ in range(6):                                   # ordinary steps until one resamples
        z1 = f.z + rng.normal(size=f.z.shape) * [0.5, 0.5, 0.01]
        a.SlamUpdate(None, z1, u_resample=0.41)
        b.SlamUpdate(None, z1, u_resample=0.41)
        if a.resample_sources()[1]:
            break
    assert a.resample_sources()[1], "no step resampled: this test would mean nothing"
    z2 = f.z + 0.2
    a.SlamUpdate(None, z2, u_resample=0.27)
    b.set_measurements(z2)
    b._check(lib.phd_set_stream(b._h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
    gw = torch.as_tensor(_Dev(lib.phd_device_global_weights(b._h, P), P), device="cuda")
    b._check(lib.phd_step_local_async(b._h, 0))
    gw.copy_(torch.as_tensor(_Dev(lib.phd_device_local_weights(b._h), P), device="cuda"))
    b._check(lib.phd_step_global_async(b._h, 0, 1, C.c_double(0.27)))
    s, q = np.zeros(1, np.int32), np.zeros(1, np.int32)
    b._check(lib.phd_migration_plan(b._h, 0, 1, s.ctypes.data_as(ip), q.ctypes.data_as(ip)))
    b._check(lib.phd_migration_pack_async(b._h))
    b._check(lib.phd_migration_unpack_async(b._h))
    b.sync()
    assert np.array_equal(a.VehicleWeights, b.VehicleWeights) and np.array_equal(a.poses(), b.poses())
    assert a.BestParticle == b.BestParticle
    for g in range(0, P, 3):
        assert all(np.array_equal(x, y) for x, y in zip(a.MapModel(g), b.MapModel(g))), "particle %d" % g
    # and onwards from the sharded step's (materialised) state with an ordinary one
    b._check(lib.phd_set_stream(b._h, None, 0))
    a.SlamUpdate(None, f.z, u_resample=0.63)
    b.SlamUpdate(None, f.z, u_resample=0.63)
    assert np.array_equal(a.VehicleWeights, b.VehicleWeights)
    for g in range(0, P, 5):
        assert all(np.array_equal(x, y) for x, y in zip(a.MapModel(g), b.MapModel(g)))
    a.close()
    b.close()


def _state_of(nav, particles):
    src, res = nav.resample_sources()
    return {"w": nav.VehicleWeights, "src": src, "res": res, "best": nav.BestParticle,
            "maps": [nav.MapModel(int(i)) for i in particles]}


@pytest.mark.parametrize("cfg,profile", [("B", "survey"), ("B", "steady"), ("S", "survey"), ("S", "steady")])
def test_timed_mode_equals_the_gated_mode(cfg, profile):
    """The mode `bench.py` times — phd_set_all_pairs(1) + phd_set_frozen(1), the default two-stream split, steps posted back
    to back — against the default gated mode on the same frame: the gate only selects what is summed
    (PHDNavigator.cs:882-890; a masked pair adds an exact zero), so particle weights, resampling decision and sources,
    BestParticle and sampled maps must be BIT-identical after a SlamUpdate, and alpha, the set log-likelihood and the pruned
    maps of run_stages too. The oracle sample of test_full_size_properties_and_sample is repeated in all-pairs mode."""
    gated, p, f = setup(cfg, profile)
    timed, _, _ = setup(cfg, profile)
    timed.set_measurements(f.z)
    timed.set_frozen(True)
    timed.set_all_pairs(True)
    rng = np.random.default_rng(5)
    sample = rng.choice(f.P, 3, replace=False)
    # one SlamUpdate in the default mode | five steps posted back to back in the timed mode (frozen: each sees the same input)
    gated.SlamUpdate(None, f.z, u_resample=0.41)
    for _ in range(5):
        timed.step_async(0.41)
    timed.sync()
    a, b = _state_of(gated, sample), _state_of(timed, sample)
    assert a["res"] == b["res"] and a["best"] == b["best"]
    assert np.array_equal(a["src"], b["src"]), "resampling sources differ between the gated and the all-pairs mode"
    assert np.array_equal(a["w"], b["w"]), "particle weights differ between the gated and the all-pairs mode"
    for i, ma, mb in zip(sample, a["maps"], b["maps"]):
        assert all(np.array_equal(x, y) for x, y in zip(ma, mb)), "map of particle %d" % i
    # the timed mode's synchronous entry point too (what `ms_per_synchronous_update` times)
    timed.SlamUpdate(None, f.z, u_resample=0.41)
    c = _state_of(timed, sample)
    assert np.array_equal(a["w"], c["w"]) and np.array_equal(a["src"], c["src"])
    gated.close()
    # stage by stage, all-pairs | gated, from the same (frozen: untouched) state
    gated, _, _ = setup(cfg, profile)
    gated.run_stages(f.z, with_alpha=True)
    timed.run_stages(f.z, with_alpha=True)
    assert np.array_equal(gated.WeightAlpha(), timed.WeightAlpha()), "alpha differs between the modes"
    assert np.array_equal(gated.SetLogLikelihood(), timed.SetLogLikelihood())
    alpha = timed.WeightAlpha()
    for i in sample:
        i = int(i)
        for stage in ("PredictConditional", "CorrectConditional", "PruneModel"):
            ga, ta = getattr(gated, stage)(i), getattr(timed, stage)(i)
            if stage == "CorrectConditional":   # (an unsorted list: its order is that of the queue's atomics — the same entries, bit for bit)
                ga, ta = (tuple(x[np.lexsort((m[1][:, 2], m[1][:, 1], m[1][:, 0], m[0]))] for x in m) for m in (ga, ta))
            assert all(np.array_equal(x, y) for x, y in zip(ga, ta)), "%s[%d]" % (stage, i)
        # the oracle on this particle, against the all-pairs run
        pw, pm, pc = timed.PruneModel(i)
        pred = orc.predict(p, f.poses[i], f.z, f.map(i))
        opr = orc.prune(p, orc.correct(p, f.poses[i], f.z, pred))
        assert len(opr[0]) == len(pw)
        assert np.allclose(pw, opr[0], rtol=1e-7) and np.allclose(pm, opr[1], rtol=1e-7, atol=1e-11)
        iu = np.triu_indices(3)
        assert np.allclose(pc[:, iu[0], iu[1]], opr[2][:, iu[0], iu[1]], rtol=1e-7, atol=1e-13)
        al, _ = orc.weight_alpha(p, f.poses[i], f.z, pred, opr)
        assert np.isclose(alpha[i], al, rtol=1e-6, atol=0)
    gated.close()
    timed.close()


def test_config_c8_at_full_size_on_one_gpu():
    """BASELINE config C8 at its full size — 8 ranks x 2048 particles x 512 components x 64 measurements — rehearsed on ONE GPU: eight
    handles in one process play the per-rank sequence of `bench.py --gpus 8` (local step, the weights gathered by device copies,
    the 16 384-slot resampling and the migration plan on their grid kernels, records stored straight into the peers' receive
    buffers, landing flags, unpack) against ONE handle holding all 16 384 particles: particle weights, poses and sampled maps bit
    for bit over steps that resample (long runs of one source cross the rank boundaries) — what no 8-GPU node was ever available
    to show, minus the wires."""
    import sys, os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from monorfs_amd import navigator
    from test_gpu_round4 import _device_path_handles, _device_path_step
    world, (Pg, Cc, M, seed) = 8, CONFIGS["C8"]
    Pl = Pg // world
    f = Frame(Pg, Cc, M, seed, weight_profile="steady")
    f.weights = np.random.default_rng(8).random(f.P) ** 12     # depleted from the start: the first step resamples
    f.weights /= f.weights.sum()
    p1 = prm3d_defaults(max_particles=Pg, max_components=600, max_measurements=M)
    one = navigator.PHDNavigator(p1, particlecount=Pg)
    one.upload_state(f.planes(), f.counts, f.poses, f.weights)
    navs = _device_path_handles(navigator, f, world, Pl, M)
    flags = all(nv._lib.phd_migration_recv_is_finegrained(nv._h) == 1 for nv in navs)
    for nv in navs:
        if flags:
            nv._check(nv._lib.phd_migration_set_landing(nv._h, 1))
    rng = np.random.default_rng(seed)
    nres = 0
    for step, u in enumerate((0.37, 0.81, 0.09)):
        one.SlamUpdate(None, f.z, u_resample=u)
        _device_path_step(navs, Pl, u)
        src, res = one.resample_sources()
        nres += int(res)
        assert np.array_equal(one.VehicleWeights, np.concatenate([nv.VehicleWeights for nv in navs])), "step %d" % step
        assert np.array_equal(one.poses(), np.concatenate([nv.poses() for nv in navs])), "step %d" % step
        for g in list(rng.choice(Pg, 12, replace=False)) + [0, Pl - 1, Pl, Pg - 1]:
            a_, b_ = one.MapModel(int(g)), navs[int(g) // Pl].MapModel(int(g) % Pl)
            assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "step %d particle %d" % (step, g)
        if res:   # records did cross the rank boundaries
            own = np.arange(Pg) // Pl
            assert np.any(np.asarray(src) // Pl != own), "step %d: nothing migrated" % step
    assert nres >= 1, "no step resampled"
    one.close()
    for nv in navs:
        nv.close()


def test_multi_device_handle_at_config_c8():
    """The library's own multi-device handle (phd_create over a device list: what a single-process host — the C# one — uses for
    N GPUs) at config C8's full size: 8 shards x 2048 particles x 512 components x 64 measurements, device 0 listed eight times on a
    one-GPU box, against a single handle of 16 384 particles: weights, poses, BestParticle, resampling sources and sampled maps bit
    for bit over steps that resample, a batch of steps posted back to back included."""
    from monorfs_amd import navigator
    Pg, Cc, M, seed = CONFIGS["C8"]
    f = Frame(Pg, Cc, M, seed, weight_profile="steady")
    f.weights = np.random.default_rng(9).random(f.P) ** 12
    f.weights /= f.weights.sum()
    p = prm3d_defaults(max_particles=Pg, max_components=600, max_measurements=M)
    single = navigator.PHDNavigator(p, particlecount=Pg)
    multi = navigator.PHDNavigator(p, particlecount=Pg, devices=[0] * 8)
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    rng = np.random.default_rng(seed + 1)
    nres = 0
    for step in range(3):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        if step == 2:
            us = [0.21, 0.66, 0.93]
            for nav in (single, multi):
                nav.set_measurements(z)
                for ub in us:
                    nav.step_async(ub)
                nav.sync()
        else:
            u = float(rng.uniform(0.05, 0.95))
            single.SlamUpdate(None, z, u_resample=u)
            multi.SlamUpdate(None, z, u_resample=u)
        assert np.array_equal(single.VehicleWeights, multi.VehicleWeights), "step %d: weights" % step
        assert np.array_equal(single.poses(), multi.poses()), "step %d: poses" % step
        assert single.BestParticle == multi.BestParticle, "step %d: BestParticle" % step
        sa, sb = single.resample_sources(), multi.resample_sources()
        assert sa[1] == sb[1] and np.array_equal(sa[0], sb[0]), "step %d: sources" % step
        nres += int(sa[1])
        for g in list(rng.choice(Pg, 10, replace=False)) + [0, 2047, 2048, Pg - 1]:
            assert all(np.array_equal(x, y) for x, y in zip(single.MapModel(int(g)), multi.MapModel(int(g)))), "step %d particle %d" % (step, g)
    assert nres >= 1, "no step resampled"
    single.close(); multi.close()
