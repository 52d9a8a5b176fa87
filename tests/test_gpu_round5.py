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
