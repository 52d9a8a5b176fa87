"""Worker of tests/test_gpu_multiproc.py: one rank of a sharded SlamUpdate sequence on a real device. All ranks share
cuda:0 (RCCL refuses two ranks on one GPU), so the collectives of the step run on gloo through host copies of the library's
device buffers, with exactly the call sequence bench.py uses over RCCL. Rank 0 also runs the whole particle set in one handle
and compares.
  mode "device" (argv[1], the default; what bench.py --gpus N runs): no host wait — the weights (+ status words) are
      all-gathered, the plan stays on the device, every rank's kernels store the migrating particles straight into the other
      PROCESSES' receive buffers (opened through hipIpcMemHandle), a barrier stands in for the one-word all-reduce;
  mode "flags": the same with phd_migration_set_landing(1) — nothing at all between push and unpack: the senders leave a
      step-stamped flag behind their records, the receiver's k_finish_sharded waits for it on the device (round 5);
  mode "host": round 3's sequence — the host waits for the plan's split sizes and moves the records with all_to_all_single.
  argv[2] "big": every rank holds a shard of config C8's size (2048 x 512 x 64); with four ranks the 8192-slot global vector
      puts the resampling and the plan on their grid kernels."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from monorfs_amd import navigator
from monorfs_amd.abi import prm3d_defaults
from monorfs_amd.synth import Frame


class Dev:
    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


def dev(ptr, n):
    return torch.as_tensor(Dev(ptr, n), device="cuda")


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    Pl, Cc, M, steps = 40, 60, 16, 3
    if len(sys.argv) > 2 and sys.argv[2] == "big":   # a rank's shard at config C8's size: 2048 particles x 512 components x 64 measurements
        Pl, Cc, M, steps = 2048, 512, 64, 3
    if len(sys.argv) > 3:
        steps = int(sys.argv[3])          # (a longer sequence, by hand: tests run three steps)
    Pg = Pl * world
    f = Frame(Pg, Cc, M, 77, weight_profile="steady")
    planes = f.planes()
    sl = slice(rank * Pl, (rank + 1) * Pl)
    p = prm3d_defaults(max_particles=Pl, max_components=600, max_measurements=M)
    nav = navigator.PHDNavigator(p, particlecount=Pl)
    nav.upload_state(planes[:, sl], f.counts[sl], f.poses[sl], f.weights[sl])
    nav.set_measurements(f.z)
    lib, h = nav._lib, nav._h
    nav._check(lib.phd_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
    ip = C.POINTER(C.c_int32)
    scounts, rcounts = np.zeros(world, np.int32), np.zeros(world, np.int32)
    one = None
    if rank == 0:
        p1 = prm3d_defaults(max_particles=Pg, max_components=600, max_measurements=M)
        one = navigator.PHDNavigator(p1, particlecount=Pg)
        one.upload_state(planes, f.counts, f.poses, f.weights)
    nresampled = 0
    mode = sys.argv[1] if len(sys.argv) > 1 else "device"
    flags = mode == "flags"
    if flags:
        mode = "device"
    if mode == "device":
        graw = dev(lib.phd_device_gather_buffer(h, world), world * (Pl + 1))
        hbuf = C.create_string_buffer(64)
        nav._check(lib.phd_migration_ipc_export(h, hbuf, None))
        handles = [None] * world
        dist.all_gather_object(handles, bytes(hbuf.raw))
        nav._check(lib.phd_migration_ipc_open(h, b"".join(handles), rank, world))
        if flags:
            assert lib.phd_migration_recv_is_finegrained(h) == 1, "no fine-grained receive buffer on this box: the flags need one"
            nav._check(lib.phd_migration_set_landing(h, 1))
    for step in range(steps):
        u = (0.3 + 0.2 * step) % 1.0
        nav._check(lib.phd_step_local_async(h, 0))
        if mode == "device":
            lw = dev(lib.phd_device_local_weights(h), Pl + 1).cpu()
            g_host = torch.empty(world * (Pl + 1), dtype=torch.float64)
            dist.all_gather_into_tensor(g_host, lw)
            graw.copy_(g_host)
            nav._check(lib.phd_step_global_device_async(h, rank, world, C.c_double(u), 0))
            nav._check(lib.phd_migration_push_async(h))
            if not flags:
                torch.cuda.synchronize()
                dist.barrier()                  # every rank's records have landed (bench.py --landing allreduce: a one-word all-reduce on the stream)
            nav._check(lib.phd_migration_unpack_async(h))
            nav.sync()
            ns = nr = 0
            w_all = [None] * world
            dist.gather_object((nav.VehicleWeights, nav.poses(), [nav.MapModel(i) for i in (0, Pl // 2, Pl - 1)], 0, 0),
                               w_all if rank == 0 else None, dst=0)
            if rank == 0:
                one.SlamUpdate(None, f.z, u_resample=u)
                nresampled += bool(one.resample_sources()[1])
                assert np.array_equal(one.VehicleWeights, np.concatenate([x[0] for x in w_all])), "step %d: weights differ" % step
                assert np.array_equal(one.poses(), np.concatenate([x[1] for x in w_all])), "step %d: poses differ" % step
                for r in range(world):
                    for j, i in enumerate((0, Pl // 2, Pl - 1)):
                        a_, b_ = one.MapModel(r * Pl + i), w_all[r][2][j]
                        assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "step %d rank %d particle %d" % (step, r, i)
                print("step %d ok (device plan, IPC push%s)" % (step, ", landing flags" if flags else ""), flush=True)
            continue
        lw = dev(lib.phd_device_local_weights(h), Pl).cpu()
        gw_host = torch.empty(Pg, dtype=torch.float64)
        dist.all_gather_into_tensor(gw_host, lw)
        dev(lib.phd_device_global_weights(h, Pg), Pg).copy_(gw_host)
        nav._check(lib.phd_step_global_async(h, rank, world, C.c_double(u)))
        nav._check(lib.phd_migration_plan(h, rank, world, scounts.ctypes.data_as(ip), rcounts.ctypes.data_as(ip)))
        nav._check(lib.phd_migration_pack_async(h))
        bpp = C.c_int64(0)
        sptr = lib.phd_migration_send_buffer(h, C.byref(bpp))
        rptr = lib.phd_migration_recv_buffer(h)
        rec = bpp.value // 8
        ns, nr = int(scounts.sum()), int(rcounts.sum())
        send = dev(sptr, ns * rec).cpu() if ns else torch.empty(0, dtype=torch.float64)
        recv = torch.empty(nr * rec, dtype=torch.float64)
        dist.all_to_all_single(recv, send, (rcounts * rec).tolist(), (scounts * rec).tolist())
        if nr:
            dev(rptr, nr * rec).copy_(recv)
        nav._check(lib.phd_migration_unpack_async(h))
        nav.sync()
        # everything of the sharded state goes to rank 0
        w_all = [None] * world
        dist.gather_object((nav.VehicleWeights, nav.poses(), [nav.MapModel(i) for i in (0, Pl // 2, Pl - 1)], int(ns), int(nr)),
                           w_all if rank == 0 else None, dst=0)
        if rank == 0:
            one.SlamUpdate(None, f.z, u_resample=u)
            nresampled += bool(one.resample_sources()[1])
            assert np.array_equal(one.VehicleWeights, np.concatenate([x[0] for x in w_all])), "step %d: weights differ" % step
            assert np.array_equal(one.poses(), np.concatenate([x[1] for x in w_all])), "step %d: poses differ" % step
            for r in range(world):
                for j, i in enumerate((0, Pl // 2, Pl - 1)):
                    a_, b_ = one.MapModel(r * Pl + i), w_all[r][2][j]
                    assert all(np.array_equal(x, y) for x, y in zip(a_, b_)), "step %d rank %d particle %d" % (step, r, i)
            moved = sum(x[3] for x in w_all)
            assert moved == sum(x[4] for x in w_all)
            print("step %d ok: %d particles migrated between ranks" % (step, moved), flush=True)
    if rank == 0:
        assert nresampled > 0, "no step resampled: the migration path was not exercised"
        print("multiproc ok: %d ranks x %d particles x %d components x %d measurements" % (world, Pl, Cc, M))
        one.close()
    nav.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
