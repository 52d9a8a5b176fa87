"""The N > 1 path on CPU: two processes over gloo play two ranks of the sharded particle filter.
What is rank-dependent in the product — the all-gather of the local weights, the identical global
normalise/resample decision on every rank and the particle migration plan (phd_plan_migration, pure
host logic of libphdhip.so) — is exercised end to end with stand-in particle payloads; the kernels
themselves are covered by the -m gpu tests."""
import ctypes as C
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def worker(rank, world, port, Pl, seed, out):
    import orc
    from monorfs_amd import _lib
    from monorfs_amd.abi import prm3d_defaults
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = _lib.load()
    ip = _lib.ip
    try:
        rng = np.random.default_rng([seed, rank])
        # local un-normalised weights (a few dominant particles so that the filter is depleted)
        lw = np.full(Pl, 1e-6)
        lw[rng.choice(Pl, 3, replace=False)] = rng.uniform(0.5, 1.0, 3)
        local = torch.from_numpy(lw.copy())
        gathered = [torch.zeros(Pl, dtype=torch.float64) for _ in range(world)]
        dist.all_gather(gathered, local)                       # the one exchange of the path (PHDNavigator.cs:343-358)
        gw = torch.cat(gathered).numpy()
        # every rank takes the same decision from the same vector
        w = gw / (gw.sum() if gw.sum() != 0 else 1.0)
        params = prm3d_defaults()
        depleted = orc.particle_depleted(params, w)
        assert depleted
        gsrc, best = orc.resample(w, 0.37)
        digest = torch.tensor([float(np.dot(gsrc, np.arange(len(gsrc)) % 97)), float(best)], dtype=torch.float64)
        ref = digest.clone()
        dist.broadcast(ref, 0)
        assert torch.equal(ref, digest), "ranks disagree on the resampling"

        # migration plan from the product library (no GPU involved)
        sc = np.zeros(world, np.int32)
        rc = np.zeros(world, np.int32)
        sl = np.zeros(Pl * max(world - 1, 1), np.int32)
        code = np.zeros(Pl, np.int32)
        g = np.ascontiguousarray(gsrc, np.int32)
        nrecv = lib.phd_plan_migration(g.ctypes.data_as(ip), Pl, world, rank, sc.ctypes.data_as(ip), rc.ctypes.data_as(ip),
                                       sl.ctypes.data_as(ip), code.ctypes.data_as(ip))
        assert nrecv == rc.sum()
        # stand-in particle payload: [global id, 10 * global id]
        first = rank * Pl
        payload = np.stack([np.arange(first, first + Pl), 10.0 * np.arange(first, first + Pl)], axis=1).astype(np.float64)
        send = payload[sl[:sc.sum()]]
        recv = np.zeros((int(rc.sum()), 2))
        # all-to-all with point-to-point calls (what ncclSend/ncclRecv groups do over xGMI)
        reqs, so, ro = [], 0, 0
        rbufs = {}
        for r in range(world):
            if r == rank:
                continue
            if sc[r]:
                reqs.append(dist.isend(torch.from_numpy(np.ascontiguousarray(send[so:so + sc[r]])), r))
            so += sc[r]
            if rc[r]:
                rbufs[r] = (torch.zeros((int(rc[r]), 2), dtype=torch.float64), ro)
                reqs.append(dist.irecv(rbufs[r][0], r))
            ro += rc[r]
        for q in reqs:
            q.wait()
        for r, (buf, off) in rbufs.items():
            recv[off:off + len(buf)] = buf.numpy()
        # unpack: every local slot must now hold the particle the global resample chose for it
        newp = np.zeros((Pl, 2))
        for i in range(Pl):
            newp[i] = payload[code[i]] if code[i] >= 0 else recv[-(code[i] + 1)]
        assert np.array_equal(newp[:, 0], gsrc[first:first + Pl].astype(float))
        assert np.array_equal(newp[:, 1], 10.0 * gsrc[first:first + Pl])
        moved = torch.tensor([float(sc.sum())], dtype=torch.float64)
        dist.all_reduce(moved)
        if rank == 0:
            out.put(("ok", int(moved.item())))
    except Exception as e:   # surface the failure to the parent
        out.put(("fail", "rank %d: %r" % (rank, e)))
        raise
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("Pl,seed", [(64, 1), (257, 2)])
def test_two_ranks_shard_resample_and_migrate(Pl, seed):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, 2, port, Pl, seed, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    status, info = q.get(timeout=5)
    assert status == "ok", info
    assert info > 0, "no particle crossed the rank boundary: the migration path was not exercised"


def test_plan_is_consistent_for_four_ranks():
    from monorfs_amd import _lib
    lib = _lib.load()
    ip = _lib.ip
    rng = np.random.default_rng(9)
    world, Pl = 4, 50
    gsrc = np.sort(rng.integers(0, world * Pl, world * Pl)).astype(np.int32)
    plans = []
    for rank in range(world):
        sc, rc = np.zeros(world, np.int32), np.zeros(world, np.int32)
        sl, code = np.zeros(Pl * (world - 1), np.int32), np.zeros(Pl, np.int32)
        lib.phd_plan_migration(gsrc.ctypes.data_as(ip), Pl, world, rank, sc.ctypes.data_as(ip), rc.ctypes.data_as(ip),
                               sl.ctypes.data_as(ip), code.ctypes.data_as(ip))
        plans.append((sc, rc, sl, code))
    for a in range(world):
        for b in range(world):
            assert plans[a][0][b] == plans[b][1][a], "send count a->b must equal receive count b<-a"
    # simulate the exchange
    for rank in range(world):
        sc, rc, sl, code = plans[rank]
        recv = []
        for src_rank in range(world):
            if src_rank == rank:
                continue
            ssc, _, ssl, _ = plans[src_rank]
            off = int(sum(ssc[r] for r in range(rank) if r != src_rank))
            recv.extend((ssl[off:off + ssc[rank]] + src_rank * Pl).tolist())
        for i in range(Pl):
            got = rank * Pl + code[i] if code[i] >= 0 else recv[-(code[i] + 1)]
            assert got == gsrc[rank * Pl + i]


def test_the_frames_of_all_ranks_share_map_and_measurements():
    """bench.py gives every rank Frame(..., shard=rank): the same map (weights, covariances, base means), the same
    measurements, particle poses and mean jitter of its own; rank 0's frame is the single-GPU frame."""
    from monorfs_amd.synth import Frame
    a = Frame(48, 20, 9, 1002, weight_profile="steady")
    b = Frame(48, 20, 9, 1002, weight_profile="steady", shard=3)
    c = Frame(48, 20, 9, 1002, weight_profile="steady", shard=5)
    for f in (b, c):
        assert np.array_equal(a.z, f.z) and np.array_equal(a.w, f.w) and np.array_equal(a.cov, f.cov)
        assert np.allclose(a.mean, f.mean, atol=0.1) and not np.array_equal(a.mean, f.mean)
        assert not np.array_equal(a.poses, f.poses)
    assert not np.array_equal(b.poses, c.poses)
    a0 = Frame(48, 20, 9, 1002, weight_profile="steady", shard=0)
    assert np.array_equal(a.mean, a0.mean) and np.array_equal(a.poses, a0.poses)
