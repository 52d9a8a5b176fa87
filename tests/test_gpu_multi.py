"""The multi-device handle (phd_create_multi: one handle, one caller thread, a shard per listed device) against a single
handle holding all particles: bit for bit, over sequences that resample and migrate particles between shards. On the
one-GPU box the device list names device 0 several times (each shard is still its own set of buffers and its own
stream; the exchanges are the same peer copies)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import orc
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame


@pytest.fixture(scope="module")
def nav_mod():
    from monorfs_amd import navigator
    return navigator


def both(nav_mod, f, shards, maxq=600):
    p = prm3d_defaults(max_particles=f.P, max_components=max(maxq, f.C), max_measurements=max(f.M, 1))
    p.max_quantity = maxq
    single = nav_mod.PHDNavigator(p, particlecount=f.P)
    multi = nav_mod.PHDNavigator(p, particlecount=f.P, devices=[0] * shards)
    for nav in (single, multi):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
    return single, multi, p


def assert_same_state(a, b, P, maps):
    assert np.array_equal(a.VehicleWeights, b.VehicleWeights)
    assert np.array_equal(a.poses(), b.poses())
    assert a.BestParticle == b.BestParticle
    sa, sb = a.resample_sources(), b.resample_sources()
    assert sa[1] == sb[1] and np.array_equal(sa[0], sb[0])
    for i in maps:
        for x, y in zip(a.MapModel(i), b.MapModel(i)):
            assert np.array_equal(x, y), "map %d differs" % i


@pytest.mark.parametrize("shards", [2, 3, 4])
def test_multi_handle_equals_single_handle(nav_mod, shards):
    """five steps with odometry in between; the steady frame depletes the particle set, so particles migrate between the
    shards in most steps"""
    P = 24 * shards
    f = Frame(P, 70, 18, 300 + shards, weight_profile="steady")
    single, multi, p = both(nav_mod, f, shards)
    assert multi.particle_count == P
    rng = np.random.default_rng(shards)
    nres = 0
    for step in range(5):
        z = f.z + rng.normal(size=f.z.shape) * np.sqrt([2.0, 2.0, 1e-3]) * 0.3
        u = float(rng.uniform(0.05, 0.95))
        reading = rng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.003, 0.003, 0.003]
        noise = rng.normal(0, 1, (P, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
        for nav in (single, multi):
            nav.UpdateOdometry(None, reading, noise)
            nav.SlamUpdate(None, z, u_resample=u)
        assert_same_state(single, multi, P, range(P))
        nres += single.resample_sources()[1]
    assert nres >= 2, "the sequence hardly resampled: the migration was not exercised"
    # whole-state download, then the same state uploaded into fresh handles continues identically
    (pa, ca, qa, wa), (pb, cb, qb, wb) = single.download_state(600), multi.download_state(600)
    assert np.array_equal(ca, cb) and np.array_equal(qa, qb) and np.array_equal(wa, wb)
    for i in range(P):                                          # (slots beyond a particle's count hold leftovers)
        assert np.array_equal(pa[:, i, :ca[i]], pb[:, i, :cb[i]])
    single.close()
    multi.close()


def test_multi_handle_against_the_oracle_and_its_other_calls(nav_mod):
    P, shards = 32, 2
    f = Frame(P, 60, 14, 411, weight_profile="steady")
    single, multi, p = both(nav_mod, f, shards)
    st = orc.State(P, 700)
    st.poses[:] = f.poses
    st.w[:, :f.C], st.mean[:, :f.C], st.cov[:, :f.C], st.n[:] = f.w, f.mean, f.cov, f.C
    best, src, res, _ = orc.slam_update(p, st, f.z, u=0.37, threads=4)
    multi.SlamUpdate(None, f.z, u_resample=0.37)
    gsrc, gres = multi.resample_sources()
    assert gres == res and np.array_equal(gsrc, src) and multi.BestParticle == best
    assert np.allclose(multi.VehicleWeights, st.weights, rtol=1e-6, atol=1e-300)
    # set_map / set_weights / set_poses / reset / collapse through the multi handle
    single.SlamUpdate(None, f.z, u_resample=0.37)
    w2 = np.linspace(1, 2, P)
    w2 /= w2.sum()
    poses2 = single.poses() + 0.01
    for nav in (single, multi):
        nav.set_weights(w2)
        nav.set_poses(poses2)
        nav.set_map(P - 3, f.map(5))
        nav.SlamUpdate(None, f.z[:9], u_resample=0.8)
    assert_same_state(single, multi, P, [0, P // 2 - 1, P // 2, P - 3, P - 1])
    for nav in (single, multi):
        nav.CollapseParticles(P)
        nav.SlamUpdate(None, f.z, u_resample=0.2)
    assert_same_state(single, multi, P, [0, P - 1])
    # what a multi handle does not offer says so
    with pytest.raises(nav_mod.PHDError):
        multi.run_stages(f.z)
    # a failed step (emit capacity) on a multi handle keeps the state, like on a single one
    single.close()
    multi.close()


def test_multi_handle_only_mapping(nav_mod):
    P = 8
    f = Frame(P, 50, 12, 512, weight_profile="steady")
    p = prm3d_defaults(max_particles=P, max_components=600, max_measurements=12)
    a = nav_mod.PHDNavigator(p, particlecount=P)
    b = nav_mod.PHDNavigator(p, particlecount=P, devices=[0, 0])
    for nav in (a, b):
        nav.upload_state(f.planes(), f.counts, f.poses, f.weights)
        nav.OnlyMapping = True
        nav.SlamUpdate(None, f.z)
        nav.SlamUpdate(None, f.z[:7])
    assert np.array_equal(a.VehicleWeights, b.VehicleWeights)
    for i in range(P):
        for x, y in zip(a.MapModel(i), b.MapModel(i)):
            assert np.array_equal(x, y)
    a.close()
    b.close()


def test_multi_handle_rejects_unequal_shards(nav_mod):
    p = prm3d_defaults(max_particles=10, max_components=600, max_measurements=8)
    with pytest.raises(nav_mod.PHDError):
        nav_mod.PHDNavigator(p, particlecount=10, devices=[0, 0, 0])


def test_multi_handle_failed_step_keeps_the_state(nav_mod):
    """emit capacity overflow on a multi-device handle: PHD_ERR_CAPACITY, nothing rotated on any shard"""
    small = Frame(8, 40, 6, 52, weight_profile="steady")
    p = prm3d_defaults(max_particles=small.P, max_components=600, max_measurements=32)
    p.emit_capacity = 64
    p.max_quantity = 64
    nav = nav_mod.PHDNavigator(p, particlecount=small.P, devices=[0, 0])
    nav.upload_state(small.planes(), small.counts, small.poses, small.weights)
    nav.SlamUpdate(None, small.z, u_resample=0.3)
    before = (nav.VehicleWeights, nav.poses(), [nav.MapModel(i) for i in range(small.P)])
    rng = np.random.default_rng(1)
    many = np.column_stack([rng.uniform(-300, 300, 32), rng.uniform(-220, 220, 32), rng.uniform(0.3, 1.8, 32)])
    with pytest.raises(nav_mod.PHDError) as e:
        nav.SlamUpdate(None, many)
    assert e.value.status == 2
    after = (nav.VehicleWeights, nav.poses(), [nav.MapModel(i) for i in range(small.P)])
    assert np.array_equal(before[0], after[0]) and np.array_equal(before[1], after[1])
    for x, y in zip(before[2], after[2]):
        assert all(np.array_equal(a, b) for a, b in zip(x, y))
    nav.SlamUpdate(None, small.z[:3], u_resample=0.4)
    assert np.isclose(nav.VehicleWeights.sum(), 1.0)
    nav.close()
