#!/usr/bin/env python3
"""PHD-updates/s benchmark of the RB-PHD-SLAM inner loop on MI355X (BASELINE.json's metric).

A step = one PHDNavigator.SlamUpdate (predict, correct, prune, reweight, normalise, depletion test,
resample + particle copy when depleted) over one synthetic frame; the state and the measurements are
resident in HBM before the timed region. N = 1 runs BASELINE config B (2048 particles x 512
components x 64 measurements); N > 1 keeps 2048 particles per GPU (weak scaling, config C8 at N = 8):
one process per GPU, RCCL all-gather of the particle weights, all-to-all of migrating particles.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python bench.py --gpus N ...                      starts its own N ranks (torch.distributed.run, one per GPU)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_SUSTAINED_TFLOPS = 59.6   # v_fma_f64 over the whole chip at four waves per SIMD, the occupancy of the dense kernels: measured with the
                               # in-kernel clock beside it (2.14 GHz under that loop; scripts/probes/fp64_clock.hip, profiles/r04_fp64_clock.txt;
                               # 62.3 at eight waves per SIMD; round 1's 53 came from a colder, shorter run)
FP64_SPEC_TFLOPS = 78.6        # the datasheet FP64 vector rate (4 cycles per wave instruction, 1024 SIMDs, 2.4 GHz)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling
# SURVEY §8d's protocol is ">= 20 warm + >= 100 timed steps": the first steps behind the creation of a handle run slower than
# the steady state (the device comes out of idle). Round 4, 20 timed steps behind 5 warm-up steps: 0.669 ms per step with 20 steps
# rolled first, 0.653 with 60, 0.653 with 150, 0.656 with 400 (200 timed steps: 0.647 either way; round 4). The state
# is therefore rolled this many untimed steps BEFORE the --warmup steps (reported as "preroll_steps"; 40 ms of device time); the
# timed region is still exactly --steps steps between two barriers.
PREROLL_STEPS = int(os.environ.get("PHD_BENCH_PREROLL", "60"))


def committed_source(relpath):
    """provenance of a figure copied from a committed profile file rather than measured by this run (the driver's run collects
    no PMC counters): the file and its git blob hash (= `git hash-object <file>`, computed here: the GPU box holds no .git)"""
    import hashlib
    try:
        data = open(os.path.join(ROOT, relpath), "rb").read()
        blob = hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()
    except Exception:
        blob = None
    return {"file": relpath, "git_blob": blob, "measured_in_this_run": False}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="B", help="BASELINE config name: A, B, S (per-GPU shard of C8 = B)")
    ap.add_argument("--weights", default="survey", choices=["steady", "survey"],
                    help="prior weight profile of the synthetic frame (monorfs_amd/synth.py): survey = SURVEY 8d's literal U(0.05, 1.2), "
                         "the headline; steady = a map consistent with the frame (finite particle weights, resampling every step)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="particles in the CPU-baseline sample (0 = auto)")
    ap.add_argument("--no-events", action="store_true", help="skip the per-kernel HIP events")
    ap.add_argument("--events-every", type=int, default=8, help="time the launches of every n-th step of the timed region (each event costs the device a few microseconds)")
    ap.add_argument("--force-dist", action="store_true", help="take the sharded (RCCL) step path even with one rank (rehearsal)")
    ap.add_argument("--collective", default="allgather", choices=["allgather", "allreduce"],
                    help="how the particle weights meet (sharded step): allgather = RCCL all-gather of P + 1 doubles per rank (SURVEY 8e's recommendation, the default); "
                         "allreduce = the north star's wording, one RCCL all-reduce(sum) over the zero-padded global vector (every element has one non-zero "
                         "contributor, so the result is the gathered vector, bit for bit)")
    ap.add_argument("--landing", default="flags", choices=["flags", "allreduce"],
                    help="sharded step, how a receiver learns that the migrating particles have landed: flags = the senders leave a step-stamped word "
                         "behind their records in the receiver's fine-grained buffer and k_finish_sharded waits for it on the device — ONE collective per "
                         "step, the weights (round 5, the default; falls back to allreduce when a rank has no fine-grained buffer); allreduce = a one-word "
                         "RCCL all-reduce on the stream between push and unpack (rounds 3 - 4)")
    ap.add_argument("--host-plan", action="store_true",
                    help="sharded step as in round 3: the host waits for the plan's split sizes and moves the migrating particles with all_to_all_single "
                         "(default: no host wait — peer stores through IPC-opened receive buffers, a one-word all-reduce as the landing barrier)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra legs of the N = 1 run (isolated kernel times, other modes)")
    ap.add_argument("--extra-steps", type=int, default=100, help="timed steps of every extra leg (SURVEY 8d: >= 100)")
    ap.add_argument("--single-process", action="store_true",
                    help="N GPUs behind ONE handle in this process (phd_create_multi: peer copies instead of RCCL, the path a C# host drives); "
                         "2048 particles per device as in the sharded run")
    ap.add_argument("--devices", default="", help="with --single-process: the HIP ordinals, comma separated (default 0..N-1; a device may repeat: rehearsal on one GPU)")
    return ap.parse_args()


def launch_ranks(args):
    """`bench.py --gpus N` started as ONE process: start the N ranks as children (torch.distributed.run, one per GPU) and
    pass rank 0's JSON line through. Nothing here touches a GPU (counting devices does not initialise HIP on this image);
    the current process is never re-executed."""
    have = count_gpus_without_hip(args.gpus)
    if have < args.gpus:
        raise SystemExit("bench.py --gpus %d: only %d GPU(s) visible to this process; refusing to report a %d-GPU number "
                         "from fewer devices" % (args.gpus, have, args.gpus))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("bench.py: starting %d ranks: %s" % (args.gpus, " ".join(cmd)), file=sys.stderr)
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit("bench.py --gpus %d: the rank processes exited with status %d (no number reported)" % (args.gpus, rc))
    return 0


def count_gpus_without_hip(want=1):
    """GPUs this process would see, counted without loading or initialising the HIP runtime: the parent of the rank
    processes must stay GPU-free (a process that has initialised the GPU must not start other programs on this pool).
    KFD's topology in sysfs lists every node of the host; a GPU node has simd_count > 0 and counts when its DRM render node
    can be opened from here. HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES narrow the list as the
    runtime would."""
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in sorted(os.listdir(base)):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue   # (a node of another container: not ours)
            if int(props.get("simd_count", "0")) <= 0:
                continue   # a CPU node
            # sysfs shows every GPU of the host; the ones this container may use are those whose render node opens
            # (opening the DRM node is not a HIP call and initialises nothing)
            minor = int(props.get("drm_render_minor", "-1"))
            try:
                fd = os.open("/dev/dri/renderD%d" % minor, os.O_RDWR)
                os.close(fd)
                n += 1
            except OSError:
                pass
    except OSError:
        n = 0
    if n < want:
        # fewer than asked for by that count (no KFD topology to read, or render nodes named otherwise): ask a short-lived
        # child, which may initialise whatever it likes, before refusing anything
        try:
            out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True, timeout=300)
            n = max(n, int(out.stdout.strip().splitlines()[-1]))
        except Exception:
            pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def host_threads():
    """threads the CPU baseline may use: the cores this process is allowed on, capped by the container's CPU quota"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    quota = None
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    quota = float(txt[0]) / float(txt[1])
            else:
                q = float(txt[0])
                if q > 0:
                    quota = q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().split()[0])
            break
        except (OSError, ValueError, IndexError):
            continue
    used = n if quota is None else max(1, min(n, int(quota + 0.5)))
    return used, n, quota


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class DevArray:
    """__cuda_array_interface__ view of a device pointer owned by libphdhip (for torch.distributed)."""

    def __init__(self, ptr, n, typestr="<f8"):
        self.__cuda_array_interface__ = {"shape": (n,), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def cpu_baseline(frame, params, sample, threads):
    """The CPU oracle (C++ restatement of the C# algorithm) timed on the host cores, on a bounded sample of
    the same workload. A reported baseline: it only times the checker, nothing of it is shipped."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    P = min(sample, frame.P)
    cap = max(700, frame.C + 128)

    def state(n):
        st = orc.State(n, cap)
        st.poses[:] = frame.poses[:n]
        st.w[:, :frame.C], st.mean[:, :frame.C], st.cov[:, :frame.C] = frame.w[:n], frame.mean[:n], frame.cov[:n]
        st.n[:] = frame.C
        return st

    orc.slam_update(params, state(min(P, threads)), frame.z, u=0.5, threads=threads)   # warm-up
    stages = np.zeros(4)
    times = []
    elapsed = 0.0
    while (len(times) < 100 or elapsed < 8.0) and elapsed < 25.0 and len(times) < 400:   # >= 100 steps, about 10-25 s of CPU work, every step from the same input
        st = state(P)
        stt = np.zeros(4)
        t0 = time.perf_counter()
        orc.slam_update(params, st, frame.z, u=0.5, threads=threads, stage_times=stt)
        times.append(time.perf_counter() - t0)
        elapsed += times[-1]
        stages += stt
    steps = len(times)
    med = float(np.median(times))
    p10, p90 = float(np.percentile(times, 10)), float(np.percentile(times, 90))
    # the same step on 1 thread and on 8 (the reference's Parallel.For runs on NParallel = 8 threads, Config.cs:46), on
    # proportionally smaller particle samples (about a second each)
    by_threads = {}
    for t in (1, 8, 16):
        if t >= threads:
            continue
        n = min(P, 48 * t)
        st = state(n)
        t0 = time.perf_counter()
        orc.slam_update(params, st, frame.z, u=0.5, threads=t)
        by_threads[str(t)] = n * frame.C * frame.M / (time.perf_counter() - t0)
    by_threads[str(threads)] = P * frame.C * frame.M / med
    import shutil
    runtimes = {name: shutil.which(name) for name in ("mono", "dotnet", "mcs", "csc", "msbuild", "xbuild")}
    return {"value": P * frame.C * frame.M / med, "unit": "PHD updates/s", "cores": threads, "cpu_model": cpu_model(), "kind": "port",
            "by_threads": by_threads, "steps": steps, "statistic": "median step time", "mean_value": steps * P * frame.C * frame.M / elapsed,
            "step_ms": {"median": med * 1e3, "p10": p10 * 1e3, "p90": p90 * 1e3},
            "value_p10_p90": [P * frame.C * frame.M / p90, P * frame.C * frame.M / p10],
            # SURVEY 8d / BASELINE.md 2.2: the reference's own C# path can only be timed where a C# runtime exists
            "csharp_runtime": ("absent (no mono / dotnet / mcs / csc / msbuild / xbuild on this box's PATH): the reference's mono-rfs.exe cannot be run here"
                               if not any(runtimes.values()) else {k: v for k, v in runtimes.items() if v}),
            "sample": "%d steps of %d of the %d particles (C=%d, M=%d), oracle/phd_oracle.cpp with OpenMP over particles on %d threads, %.1f s; "
                      "a restatement of the C# algorithm in C++, not the C# runtime"
                      % (steps, P, frame.P, frame.C, frame.M, threads, elapsed),
            "stage_share": {k: float(v / stages.sum()) for k, v in zip(("predict", "correct", "prune", "reweight"), stages)}}


def main():
    args = parse()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.single_process:
        return single_process(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args)   # plain `python bench.py --gpus N`: this process only starts the ranks
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py --gpus %d but WORLD_SIZE=%d: launch as many ranks as GPUs asked for" % (args.gpus, world))

    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the PHD path has no CPU fallback")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU of its own (LOCAL_RANK %d, %d device(s) visible)" % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    # Native libraries (RCCL prints a banner) write to file descriptor 1: keep it for the one JSON line and send
    # everything else to stderr.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        joined = torch.ones(1, dtype=torch.int32, device="cuda")
        dist.all_reduce(joined)    # every rank is there and RCCL works, or this raises / times out
        if int(joined.item()) != args.gpus:
            raise SystemExit("bench.py: %d ranks joined, --gpus %d" % (int(joined.item()), args.gpus))

    from monorfs_amd import navigator
    from monorfs_amd.abi import prm3d_defaults
    from monorfs_amd.synth import CONFIGS, Frame

    P, Cc, M, seed = CONFIGS[args.config]
    frame = Frame(P, Cc, M, seed, weight_profile=args.weights, shard=rank)   # same map + frame, own particles
    maxq = max(600, Cc)
    params = prm3d_defaults(max_particles=P, max_components=maxq, max_measurements=M)
    params.max_quantity = maxq   # MaxQuantity = max(600, C)
    nav = navigator.PHDNavigator(params, particlecount=P, device=local_rank)
    nav.upload_state(frame.planes(), frame.counts, frame.poses, frame.weights)
    nav.set_measurements(frame.z)
    nav.set_frozen(True)     # steady state: every step sees the same P x C x M input
    nav.set_all_pairs(True)  # benchmark mode: all C x M pairs evaluated, the gate only masks — the unit count is exact
    lib, h = nav._lib, nav._h

    sharded_info = None
    if use_dist:
        # the library launches on torch's current stream: RCCL collectives and kernels are ordered by the stream
        nav._check(lib.phd_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
        Pg = P * world
        lw_ptr = lib.phd_device_local_weights(h)          # fixed addresses: the export buffer and the migration buffers
        if not args.host_plan:
            # every rank's receive buffer, opened by every other rank (hipIpcMemHandle): the senders' kernels store the
            # migrating particles straight into them. If ANY rank cannot export or open (a runtime that refuses IPC on
            # fine-grained memory: retried with an ordinary allocation), every rank takes round 3's host-plan sequence instead —
            # a number from the slower path is better than none, and the line says which path ran.
            ipc_note = None
            for attempt in ("finegrained", "coarse"):
                ok = 1
                hbuf = C.create_string_buffer(64)
                try:
                    nav._check(lib.phd_migration_ipc_export(h, hbuf, None))
                except Exception as e:
                    ok, ipc_note = 0, "export (%s): %s" % (attempt, e)
                handles = [None] * world
                dist.all_gather_object(handles, bytes(hbuf.raw))
                if ok:
                    try:
                        nav._check(lib.phd_migration_ipc_open(h, b"".join(handles), rank, world))
                    except Exception as e:
                        ok, ipc_note = 0, "open (%s): %s" % (attempt, e)
                agreed = torch.tensor([ok], dtype=torch.int32, device="cuda")
                dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
                if int(agreed.item()) == 1:
                    break
                if attempt == "finegrained":
                    # a fresh handle with ordinary receive buffers (the allocation class is fixed when the buffers are made)
                    os.environ["PHD_COARSE_RECV"] = "1"
                    nav.close()
                    nav = navigator.PHDNavigator(params, particlecount=P, device=local_rank)
                    nav.upload_state(frame.planes(), frame.counts, frame.poses, frame.weights)
                    nav.set_measurements(frame.z)
                    nav.set_frozen(True)
                    nav.set_all_pairs(True)
                    lib, h = nav._lib, nav._h
                    nav._check(lib.phd_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
                    lw_ptr = lib.phd_device_local_weights(h)
                else:
                    args.host_plan = True
                    print("bench.py rank %d: the IPC path is not available (%s): host-plan sequence" % (rank, ipc_note), file=sys.stderr)
        if args.host_plan:
            gw = torch.as_tensor(DevArray(lib.phd_device_global_weights(h, Pg), Pg), device="cuda")
            scounts = np.zeros(world, np.int32)
            rcounts = np.zeros(world, np.int32)
            ip = C.POINTER(C.c_int32)
            bpp = C.c_int64(0)
            mig_send = lib.phd_migration_send_buffer(h, C.byref(bpp))
            mig_recv = lib.phd_migration_recv_buffer(h)
            mig_rec = bpp.value // 8
            sharded_info = {"plan": "device, its split sizes waited for by the host", "migration": "all_to_all_single", "collective": "allgather"}
        else:
            graw = torch.as_tensor(DevArray(lib.phd_device_gather_buffer(h, world), world * (P + 1)), device="cuda")
            lw1 = torch.as_tensor(DevArray(lw_ptr, P + 1), device="cuda")
            token = torch.zeros(1, dtype=torch.float32, device="cuda")
            # the landing flags need fine-grained receive buffers on EVERY rank (a peer's store must be visible while the receiver's
            # kernel runs): agreed between the ranks, like the IPC path itself
            fg = torch.tensor([1 if lib.phd_migration_recv_is_finegrained(h) == 1 else 0], dtype=torch.int32, device="cuda")
            dist.all_reduce(fg, op=dist.ReduceOp.MIN)
            if args.landing == "flags" and int(fg.item()) != 1:
                args.landing = "allreduce"
            if args.landing == "flags":
                nav._check(lib.phd_migration_set_landing(h, 1))
            sharded_info = {"plan": "device", "migration": "peer stores into IPC-opened receive buffers", "landing": args.landing,
                            "collectives_per_step": 1 if args.landing == "flags" else 2,
                            "collective": args.collective, "recv_buffer_finegrained": bool(lib.phd_migration_recv_is_finegrained(h) == 1),
                            "p2p": [bool(r == local_rank or torch.cuda.can_device_access_peer(local_rank, r)) for r in range(torch.cuda.device_count())][:max(world, 1)]}

    cache = {}

    def dev_tensor(ptr, n):
        """torch view of a library-owned device buffer (cached: the pointers only change when a buffer grows)"""
        key = (int(ptr), int(n))
        if key not in cache:
            cache[key] = torch.as_tensor(DevArray(ptr, n), device="cuda")
        return cache[key]

    empty = torch.empty(0, dtype=torch.float64, device="cuda") if use_dist else None

    phase_ev = None   # (set for a few steps behind the timed region: torch events at the phase boundaries of a sharded step)

    def mark(i):
        if phase_ev is not None:
            phase_ev[-1][i].record()

    def step(u=0.5):
        if not use_dist:
            nav.step_async(u)
            return
        if not args.host_plan:
            # nothing here waits for the device: five enqueues and two collectives per step, all on one stream
            mark(0)
            nav._check(lib.phd_step_local_async(h, 0))
            mark(1)
            if args.collective == "allgather":
                dist.all_gather_into_tensor(graw, lw1)
            else:
                graw.zero_()
                graw[rank * (P + 1):(rank + 1) * (P + 1)].copy_(lw1)
                dist.all_reduce(graw)
            mark(2)
            nav._check(lib.phd_step_global_device_async(h, rank, world, u, 0))
            mark(3)
            nav._check(lib.phd_migration_push_async(h))
            mark(4)
            if args.landing == "allreduce":
                dist.all_reduce(token)   # stream-ordered behind the push: once it has run, every rank's records have landed
            mark(5)                      # (--landing flags: the push left flags in the receivers' buffers; k_finish_sharded waits for them)
            nav._check(lib.phd_migration_unpack_async(h))
            mark(6)
            return
        nav._check(lib.phd_step_local_async(h, 0))
        dist.all_gather_into_tensor(gw, dev_tensor(lw_ptr, P))
        nav._check(lib.phd_step_global_async(h, rank, world, u))
        # the plan was made on the device; the host only waits for its 2 n split sizes (pinned memory, no stream sync)
        nav._check(lib.phd_migration_plan(h, rank, world, scounts.ctypes.data_as(ip), rcounts.ctypes.data_as(ip)))
        if lib.phd_last_resampled(h) != 0:   # (every rank knows the same flag: no resampling, no exchange)
            nav._check(lib.phd_migration_pack_async(h))
            ns, nr = int(scounts.sum()), int(rcounts.sum())
            send = dev_tensor(mig_send, ns * mig_rec) if ns else empty
            recv = dev_tensor(mig_recv, nr * mig_rec) if nr else empty
            dist.all_to_all_single(recv, send, (rcounts * mig_rec).tolist(), (scounts * mig_rec).tolist())
        nav._check(lib.phd_migration_unpack_async(h))

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        nav.sync()

    nav.timing_reset(False)
    if use_dist and not args.host_plan and args.landing == "flags" and (world > 1 or os.environ.get("PHD_BENCH_LANDING_TRIAL")):   # (the variable: rehearsal of this block at one rank)
        # The landing flags have run between processes sharing ONE GPU (tests/test_gpu_multiproc.py) — never, before this very
        # run, between GPUs. A few steps with a short bound first: should a peer's flag not become visible on this node, every
        # rank learns it here (phd_sync reports the timed-out wait), all ranks switch to the one-word all-reduce together, and the
        # state — undefined after a timed-out wait — is uploaded again. The line says which landing ran.
        os.environ["PHD_LANDING_TIMEOUT_MS"] = "3000"
        nav._check(lib.phd_migration_set_landing(h, 1))
        ok = 1
        try:
            for _ in range(4):
                step()
            torch.cuda.synchronize()
            nav.sync()
        except Exception as e:
            ok = 0
            print("bench.py rank %d: landing flags failed their trial (%s): one-word all-reduce instead" % (rank, e), file=sys.stderr)
        agreed = torch.tensor([ok], dtype=torch.int32, device="cuda")
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)
        os.environ["PHD_LANDING_TIMEOUT_MS"] = "10000"
        if int(agreed.item()) == 1:
            nav._check(lib.phd_migration_set_landing(h, 1))
        else:
            args.landing = "allreduce"
            sharded_info["landing"] = "allreduce (the flags' trial failed on this node)"
            sharded_info["collectives_per_step"] = 2
            nav._check(lib.phd_migration_set_landing(h, 0))
            nav.set_frozen(False)
            nav.upload_state(frame.planes(), frame.counts, frame.poses, frame.weights)
            nav.set_measurements(frame.z)
            nav.set_frozen(True)
            nav._check(lib.phd_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream), 1))
    for _ in range(PREROLL_STEPS + args.warmup):
        step()
    barrier()
    nav.timing_reset(0 if args.no_events else max(1, min(args.events_every, 255)))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    host_issue = time.perf_counter() - t0      # the host's own time to issue the steps (it runs ahead of the device)
    barrier()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    kernels = nav.last_timings()
    launches = nav.last_timing_counts() if kernels else {}
    if use_dist and not args.host_plan:
        # per-phase device time of the sharded step (torch events on the stream, 16 steps behind the timed region), and
        # what the collectives cost on this node, alone on the stream
        phase_ev = []
        for _ in range(16):
            phase_ev.append([torch.cuda.Event(enable_timing=True) for _ in range(7)])
            step()
        barrier()
        names = ("local_step", "weights_collective", "global_resample_and_plan", "push", "landing_barrier", "unpack")
        sharded_info["phase_ms"] = {nm: float(np.mean([e[i].elapsed_time(e[i + 1]) for e in phase_ev])) for i, nm in enumerate(names)}
        phase_ev = None

        def probe(fn, n=20):
            fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize()
            return a.elapsed_time(b) / n * 1e3
        stats = torch.zeros(3, dtype=torch.float64, device="cuda")
        big = torch.zeros(world * (P + 1), dtype=torch.float64, device="cuda")
        sharded_info["rccl_probe_us"] = {
            "all_gather_P_plus_1_doubles": probe(lambda: dist.all_gather_into_tensor(big, lw1)),
            "all_reduce_sum_3_doubles (SURVEY 8e's {sum w, sum w^2, max} variant: G-dependent order, normalisation constants only)": probe(lambda: dist.all_reduce(stats)),
            "all_reduce_sum_zero_padded_global_vector": probe(lambda: dist.all_reduce(big)),
            "all_reduce_1_word (the landing barrier)": probe(lambda: dist.all_reduce(token))}

    # ---- extra legs of the single-GPU run, all outside the timed region ------------------------------------------------
    # (1) every kernel alone on the chip: the same step on ONE stream (phd_set_split(1)), HIP events around every launch.
    #     In the timed region the two particle halves run on concurrent streams, so an event pair there also spans the
    #     other half's kernels; the roofline figure is taken from these non-overlapped launches.
    iso, iso_ms = {}, None
    extra = world == 1 and not use_dist and not args.no_extra and not args.no_events
    if extra:
        nav.set_split(1)
        for _ in range(2):
            step()
        barrier()
        nav.timing_reset(1)
        t1 = time.perf_counter()
        for _ in range(8):
            step()
        barrier()
        iso_ms = (time.perf_counter() - t1) / 8 * 1e3
        iso = nav.last_timings()
        nav.timing_reset(False)
        nav.set_split(0)

    def timed(steps, warm=20):
        for _ in range(warm):
            step()
        barrier()
        t2 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier()
        return (time.perf_counter() - t2) / steps * 1e3

    def short_run(fr, steps):
        """ms per step of a frozen run on another frame that fits the handle (an extra, recorded next to the headline):
        20 warm + `steps` timed steps"""
        nav.set_frozen(False)
        nav.upload_state(fr.planes(), fr.counts, fr.poses, fr.weights)
        nav.set_measurements(fr.z)
        nav.set_frozen(True)
        return timed(steps)

    modes = {}
    xs = max(1, args.extra_steps)
    if extra and args.config == "B":
        # SURVEY 8d "realistic-gate": the reference evaluates only the pairs inside the radius gate of Map.Near
        # (PHDNavigator.cs:882); the headline ran in benchmark mode (all C x M pairs, the gate only masks). The same frame,
        # the same state, with the all-pairs mode off: k_sweep skips every group of 64 pairs that lies outside the gate as a
        # whole. A RUN, not an estimate; next to it the share of gated pairs on this frame (host side, 64 particles).
        from monorfs_amd.synth import measure_to_map_identity
        nav.set_all_pairs(False)
        mg = timed(xs)
        nav.set_all_pairs(True)
        samp = min(P, 64)
        x = measure_to_map_identity(frame.z)                                    # base pose = identity up to 1e-3: an estimate
        d2 = ((frame.mean[:samp, :, None, :] - x[None, None, :, :]) ** 2).sum(-1)
        share = float((d2 <= params.density_distance_threshold).mean())
        modes["realistic_gate"] = {"ms_per_step": mg, "value": P * Cc * M / (mg * 1e-3), "unit": "PHD updates/s (all P x C x M pairs counted)", "steps": xs,
                                   "gated_pair_share": share, "gated_pair_updates_per_s": share * P * Cc * M / (mg * 1e-3),
                                   "note": "measured with phd_set_all_pairs(0): groups of 64 pairs outside the correct-step gate are skipped; on this "
                                           "frame %.0f %% of the pairs are inside the gate and hardly a group is empty" % (100 * share)}
        other = "survey" if args.weights == "steady" else "steady"
        fo = Frame(P, Cc, M, seed, weight_profile=other)
        mo = short_run(fo, xs)
        modes["weights_" + other] = {"ms_per_step": mo, "value": P * Cc * M / (mo * 1e-3), "unit": "PHD updates/s", "steps": xs,
                                     "note": "SURVEY 8d's literal prior weights U(0.05, 1.2): every WeightAlpha underflows, no resampling" if other == "survey"
                                             else "the variant frame: detected components U(0.6, 1.2), the others U(0.002, 0.06) — a map consistent with "
                                                  "the frame, finite particle weights, depletion and resampling in every step"}
    if extra:
        Pa, Ca, Ma, seeda = CONFIGS["A"]
        if P >= Pa and Cc >= Ca and M >= Ma:
            # BASELINE config A (256 x 128 x 32), the latency-bound regime, on the same handle
            for prof in ("steady", "survey"):
                fa = Frame(Pa, Ca, Ma, seeda, weight_profile=prof)
                ma = short_run(fa, xs)
                # ... and as the reference's host calls it: one synchronous SlamUpdate per frame (phd_slam_update = measurements
                # uploaded, the step, phd_sync with its status read-back), from Python through ctypes
                t3 = time.perf_counter()
                for _ in range(xs):
                    nav.SlamUpdate(None, fa.z, u_resample=0.5)
                msync = (time.perf_counter() - t3) / xs * 1e3
                modes["config_A" if prof == "steady" else "config_A_survey"] = {
                    "ms_per_step": ma, "value": Pa * Ca * Ma / (ma * 1e-3), "unit": "PHD updates/s", "steps": xs,
                    "ms_per_synchronous_update": msync,
                    "workload": "%d particles x %d components x %d measurements, prior weights '%s'" % (Pa, Ca, Ma, prof)}

        # QuasiSetLogLikelihood of a batch of candidate poses against one landmark set and one measurement
        # set (the smoother's shape: PHDNavigator.cs:526-548, LoopyPHDNavigator.cs:876-909) through the C-ABI, host arrays in
        # and out (PCIe-inclusive): 50 landmarks seen from the base pose, their detections + clutter, poses jittered around it
        try:
            from monorfs_amd.synth import measure_to_map_identity, measure_perfect_identity
            qrng = np.random.default_rng(77)
            nlm, nz, nq = 50, min(M, 64), P
            zc = np.stack([qrng.uniform(-300, 300, nlm), qrng.uniform(-220, 220, nlm), qrng.uniform(0.3, 1.8, nlm)], axis=1)
            lms = measure_to_map_identity(zc)
            qz = measure_perfect_identity(lms[qrng.choice(nlm, size=min(nz, nlm), replace=False)]) + qrng.normal(size=(min(nz, nlm), 3)) * np.sqrt([2.0, 2.0, 1e-3])
            qposes = np.tile([0, 0, 0, 1.0, 0, 0, 0], (nq, 1)) + np.concatenate([qrng.normal(size=(nq, 3)) * 2e-3, np.zeros((nq, 1)), qrng.normal(size=(nq, 3)) * 5e-4], axis=1)
            nav.QuasiSetLogLikelihood(qz, lms, qposes)
            t1 = time.perf_counter()
            for _ in range(5):
                nav.QuasiSetLogLikelihood(qz, lms, qposes)
            tv = (time.perf_counter() - t1) / 5
            nav.QuasiSetLogLikelihoodGradient(qz, lms, qposes)
            t1 = time.perf_counter()
            for _ in range(5):
                nav.QuasiSetLogLikelihoodGradient(qz, lms, qposes)
            tg = (time.perf_counter() - t1) / 5
            modes["quasi_set_loglik"] = {"poses": nq, "landmarks": nlm, "measurements": int(len(qz)), "value_ms": tv * 1e3, "value_poses_per_s": nq / tv,
                                         "value_and_gradient_ms": tg * 1e3, "gradient_poses_per_s": nq / tg,
                                         "note": "SURVEY row f4, one call per batch through the C-ABI with host arrays (transfers included)"}
        except Exception as e:   # an extra leg must not take the headline line with it
            modes["quasi_set_loglik"] = {"error": str(e)}
        # the particle motion step on the device (odometry reading + one noise vector per particle from the
        # host's RNG, TrackVehicle.cs:89-102), host arrays in, asynchronous
        try:
            mrng = np.random.default_rng(78)
            reading = mrng.normal(0, 1, 6) * [0.01, 0.01, 0.01, 0.003, 0.003, 0.003]
            nmo = nav.particle_count   # (the handle holds the last extra leg's frame)
            noise = mrng.normal(0, 1, (nmo, 6)) * [5e-3, 5e-3, 5e-3, 2e-4, 2e-4, 2e-4]
            nav.set_frozen(False)
            nav.UpdateOdometry(None, reading, noise)
            nav.sync()
            t1 = time.perf_counter()
            for _ in range(20):
                nav.UpdateOdometry(None, reading, noise)
            nav.sync()
            tm = (time.perf_counter() - t1) / 20
            nav.set_frozen(True)
            modes["motion_update"] = {"particles": nmo, "us_per_call": tm * 1e6, "particles_per_s": nmo / tm,
                                      "note": "SURVEY row f1, phd_update_motion with host arrays (the upload of the noise vectors included)"}
        except Exception as e:
            modes["motion_update"] = {"error": str(e)}
        # BASELINE's stress configuration S (4096 x 1024 x 128, MaxQuantity 1024: prune / merge compaction of up to 132 k
        # corrected components per particle every step), in a handle of its own, benchmark mode, frozen state
        if args.config == "B":
            try:
                Ps, Cs, Ms, seeds = CONFIGS["S"]
                fs = Frame(Ps, Cs, Ms, seeds, weight_profile=args.weights)
                ps = prm3d_defaults(max_particles=Ps, max_components=max(600, Cs), max_measurements=Ms)
                ps.max_quantity = max(600, Cs)
                navs = navigator.PHDNavigator(ps, particlecount=Ps, device=local_rank)
                navs.upload_state(fs.planes(), fs.counts, fs.poses, fs.weights)
                navs.set_measurements(fs.z)
                navs.set_frozen(True)
                navs.set_all_pairs(True)
                navs.timing_reset(False)
                for _ in range(10):
                    navs.step_async(0.5)
                navs.sync()
                nS = max(20, xs // 4)
                t1 = time.perf_counter()
                for _ in range(nS):
                    navs.step_async(0.5)
                navs.sync()
                msS = (time.perf_counter() - t1) / nS * 1e3
                navs.timing_reset(1)
                navs.set_split(1)
                for _ in range(4):
                    navs.step_async(0.5)
                navs.sync()
                kS = navs.last_timings()
                modes["config_S"] = {"ms_per_step": msS, "value": Ps * Cs * Ms / (msS * 1e-3), "unit": "PHD updates/s", "steps": nS,
                                     "workload": "%d particles x %d components x %d measurements, MaxQuantity %d, prior weights '%s'" % (Ps, Cs, Ms, ps.max_quantity, args.weights),
                                     "kernel_ms_isolated": kS, "algorithmic_bytes_per_step": 160.0 * Ps * Cs,
                                     "achieved_GBs_whole_step": 160.0 * Ps * Cs / (msS * 1e-3) / 1e9}
                navs.close()
                del fs
            except Exception as e:
                modes["config_S"] = {"error": str(e)}

    if rank == 0:
        units = P * world * Cc * M * args.steps
        ms = elapsed / args.steps * 1e3
        out = {"metric": "PHD updates/sec (particles x components x measurements)", "value": units / elapsed,
               "unit": "PHD updates/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": PREROLL_STEPS,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f64", "data": "synthetic",
               "frame_to_quote": ("`value` is measured on SURVEY 8d's LITERAL frame ('survey': every prior weight U(0.05, 1.2)): quote it as the metric of record, "
                                  "knowing what that frame is — the map's expected size is five times the measurements, every WeightAlpha underflows, all "
                                  "particle weights end at 0 (PHDNavigator.cs:344's guard) and the step never resamples. other_modes.weights_steady is the same "
                                  "configuration on a map consistent with the frame (finite weights, depletion, resampling and the particle copy in EVERY step): "
                                  "quote that one for what a converged filter does per frame."),
               "config": {"workload": "RB-PHD-SLAM SlamUpdate, BASELINE config %s: %d particles/GPU x %d components x %d measurements, "
                                      "PRM3D pixel-range model, prior weights '%s', state resident in HBM"
                                      % (args.config, P, Cc, M, args.weights),
                          "particles_per_gpu": P, "components": Cc, "measurements": M, "max_quantity": maxq,
                          "parallelism": "particles sharded x%d" % world}}
        out["host_issue_us_per_step"] = host_issue / args.steps * 1e6   # time inside the step calls, one caller thread
        if use_dist:
            out["rccl_ranks"] = dist.get_world_size()
            out["sharded_step"] = sharded_info
        if kernels:
            # the step's per-particle kernels run once per particle sub-range (phd_set_split): a kernel's cost per
            # step is its mean launch duration x launches per step, and the dominant kernel is the largest of those
            timed_steps = len(range(0, args.steps, max(1, min(args.events_every, 255))))
            per_step = {k: launches[k] / timed_steps for k in kernels}
            src = iso if iso else kernels                  # non-overlapped launches when they were taken
            per_launch_particles = {k: (P if iso else P / per_step[k]) for k in src}
            dom = max(src, key=lambda k: src[k] * (1 if iso else per_step[k]))
            particles_per_launch = per_launch_particles[dom]
            alg_bytes = 160.0 * particles_per_launch * Cc   # 80 B/component read + 80 B written, per particle
            achieved = alg_bytes / (src[dom] * 1e-3) / 1e9
            traffic = None
            traffic_source = None
            for tname in ("r05_hbm_traffic.json", "r04_hbm_traffic.json", "r03_hbm_traffic.json", "r02_hbm_traffic.json", "r01_hbm_traffic.json"):
                tfile = os.path.join(ROOT, "profiles", tname)
                if os.path.exists(tfile):
                    try:
                        traffic = json.load(open(tfile)).get(args.config, {}).get(dom)
                    except Exception:
                        traffic = None
                    if traffic is not None:
                        traffic_source = committed_source("profiles/" + tname)
                        if not iso and per_step[dom] > 1 and not tname.startswith("r01"):
                            traffic = traffic / per_step[dom]   # r02 figures are per whole-range launch
                        break
            # (every kernel by the same rule, for the reader who wants another one than the dominant: the step's algorithmic
            # bytes over that kernel's launch duration — k_sweep was the dominant kernel until k_emit_finish and k_prune_merge
            # became one launch)
            per_kernel = {k: {"kernel_ms": src[k], "frac": 160.0 * per_launch_particles[k] * Cc / (src[k] * 1e-3) / 1e9 / HBM_PEAK_GBS}
                          for k in src if k.startswith("k_") and src[k] > 0 and k != "k_normalise_resample"}   # (one workgroup for all particles: no per-particle stream)
            out["roofline"] = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source, "per_kernel": per_kernel,
                               "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": src[dom],
                               "particles_per_launch": particles_per_launch,
                               "launches_per_step": 1 if iso else per_step[dom],
                               "kernel_ms_source": "HIP events around non-overlapped launches (one stream, 8 steps after the timed region)" if iso
                                                   else "HIP events inside the timed region (particle halves on concurrent streams: launches overlap)"}
            if iso:
                out["kernel_ms_isolated"] = iso
                out["ms_per_step_one_stream"] = iso_ms
                out["step_bytes"] = {"algorithmic_bytes_per_step": 160.0 * P * Cc, "achieved_GBs_whole_step": 160.0 * P * Cc / (ms * 1e-3) / 1e9,
                                     "frac_of_peak": 160.0 * P * Cc / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
            out["kernel_ms"] = kernels
            out["kernel_launches_per_step"] = per_step
            out["kernel_ms_sampling"] = "HIP events on every %d-th step of the timed region (%d of %d steps)" % (max(1, min(args.events_every, 255)), timed_steps, args.steps)
            # The step is bound by vector-ALU issue, not by HBM: the wave-level VALU instructions of one
            # step (SQ_INSTS_VALU of the committed profile, per particle) against the rate at which the chip sustains
            # FP64 FMAs (scripts/probes/fp64_clock.hip: 59.6 TFLOP/s = 4.66e11 wave instructions/s at four waves per SIMD).
            for vname in ("r05_valu_insts.json", "r04_valu_insts.json", "r03_valu_insts.json", "r02_valu_insts.json", "r01_valu_insts.json"):
                vfile = os.path.join(ROOT, "profiles", vname)
                if not os.path.exists(vfile):
                    continue
                try:
                    valu = json.load(open(vfile)).get(args.config, {})
                    insts = sum(v.get("per_particle", 0.0) * P + v.get("per_launch", 0.0) for v in valu.values())
                    rate = FP64_SUSTAINED_TFLOPS * 1e12 / 128.0
                    if insts > 0:
                        spec = FP64_SPEC_TFLOPS * 1e12 / 128.0
                        out["valu_issue"] = {"wave_instructions_per_step": insts, "sustained_wave_instructions_per_s": rate,
                                             "bound_ms": insts / rate * 1e3, "frac": insts / rate * 1e3 / ms,
                                             "spec_wave_instructions_per_s": spec, "bound_ms_at_spec": insts / spec * 1e3, "frac_at_spec": insts / spec * 1e3 / ms,
                                             "flop_model": {"flop_per_update": 60, "tflops_delivered": 60.0 * P * Cc * M / (ms * 1e-3) / 1e12,
                                                            "frac_of_fp64_vector_peak": 60.0 * P * Cc * M / (ms * 1e-3) / 1e12 / FP64_SPEC_TFLOPS},
                                             "in_kernel_clock_mhz": {"k_sweep": 2375, "v_fma_f64 loop at 4 waves per SIMD": 2144, "measured_in_this_run": False,
                                                                     "source": "profiles/r04_sweep_clock.txt, profiles/r04_fp64_clock.txt (s_memtime / s_memrealtime)"},
                                             "source": "profiles/%s (rocprofv3 SQ_INSTS_VALU), profiles/r04_fp64_clock.txt" % vname,
                                             "valu_source": committed_source("profiles/" + vname)}
                        break
                except Exception:
                    pass
        if modes:
            out["other_modes"] = modes
        if world == 1 and not args.no_cpu_baseline:
            threads, allowed, quota = host_threads()   # every core this process may use
            sample = args.cpu_sample or min(P, max(256, 4 * threads))
            out["cpu_baseline"] = cpu_baseline(frame, params, sample, threads)
            out["cpu_baseline"]["cores_allowed"] = allowed
            out["cpu_baseline"]["cpu_quota"] = quota
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    nav.close()
    if use_dist:
        dist.destroy_process_group()
    return 0


def single_process(args):
    """`--single-process`: the same weak-scaling workload through one multi-device handle (phd_create_multi) — the host a C#
    caller drives: one worker thread per shard issues the steps, peer stores move weights and migrating particles, the
    caller only posts. The line carries the host's cost per step (time inside phd_step_async, time a worker needs to issue
    a step), the phases of a step on the first shard's stream, and which device pairs are connected peer to peer. When every
    shard sits on the same device (a rehearsal on one GPU) the same particle set is also run through ONE single-device
    handle, for the ratio."""
    import torch
    devices = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devices) != args.gpus:
        raise SystemExit("--devices must name --gpus devices")
    if max(devices) >= torch.cuda.device_count():
        raise SystemExit("bench.py --single-process: device %d asked for, %d visible" % (max(devices), torch.cuda.device_count()))
    from monorfs_amd import navigator
    from monorfs_amd.abi import prm3d_defaults
    from monorfs_amd.synth import CONFIGS, Frame
    P, Cc, M, seed = CONFIGS[args.config]
    n = len(devices)
    frames = [Frame(P, Cc, M, seed, weight_profile=args.weights, shard=r) for r in range(n)]
    maxq = max(600, Cc)
    params = prm3d_defaults(max_particles=P * n, max_components=maxq, max_measurements=M)
    params.max_quantity = maxq
    json_fd = os.dup(1)
    os.dup2(2, 1)
    planes = np.concatenate([f.planes() for f in frames], axis=1)
    counts = np.concatenate([f.counts for f in frames])
    poses = np.concatenate([f.poses for f in frames])
    weights = np.full(P * n, 1.0 / (P * n))

    def run(nav):
        nav.upload_state(planes, counts, poses, weights)
        nav.set_measurements(frames[0].z)
        nav.set_frozen(True)
        nav.set_all_pairs(True)
        nav.timing_reset(False)
        for _ in range(PREROLL_STEPS + args.warmup):
            nav.step_async(0.5)
        nav.sync()
        nav.timing_reset(max(1, min(args.events_every, 255)))
        t0 = time.perf_counter()
        for _ in range(args.steps):
            nav.step_async(0.5)
        t_posted = time.perf_counter() - t0
        nav.sync()
        return time.perf_counter() - t0, t_posted

    nav = navigator.PHDNavigator(params, particlecount=P * n, devices=devices)
    elapsed, t_posted = run(nav)
    rep = nav.multi_report()
    kernels = nav.last_timings()
    resampled = bool(nav.resample_sources()[1])
    nav.close()
    out = {"metric": "PHD updates/sec (particles x components x measurements)", "value": P * n * Cc * M * args.steps / elapsed,
           "unit": "PHD updates/s", "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "preroll_steps": PREROLL_STEPS, "ms_per_step": elapsed / args.steps * 1e3,
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": "RB-PHD-SLAM SlamUpdate, BASELINE config %s: %d particles/shard x %d components x %d measurements, prior weights '%s', one "
                                  "multi-device handle (phd_create_multi: a worker thread per shard, peer stores, plan on the device), devices %s"
                                  % (args.config, P, Cc, M, args.weights, devices),
                      "particles_per_gpu": P, "components": Cc, "measurements": M, "max_quantity": maxq,
                      "parallelism": "particles sharded x%d in one process" % n},
           "multi_host": {"post_us_per_step": rep["post_us"], "worker_issue_us_per_step": rep["issue_us"], "worker_hip_calls_us_per_step": rep["issue_calls_us"],
                          "posting_all_steps_ms": t_posted * 1e3, "phase_ms_first_shard": rep["phase_ms"], "phase_samples": rep["sampled_steps"],
                          "p2p": rep["p2p"], "resampled_every_step": resampled,
                          "note": "post = the caller's thread inside phd_step_async (it only posts); worker issue = host time one shard's thread needs to issue "
                                  "a step (launches, two events, 2 (n - 1) stream waits); phases = device time on the first shard's stream"},
           "kernel_ms_first_shard": kernels}
    if len(set(devices)) == 1:
        p1 = prm3d_defaults(max_particles=P * n, max_components=maxq, max_measurements=M)
        p1.max_quantity = maxq
        one = navigator.PHDNavigator(p1, particlecount=P * n, device=devices[0])
        e1, _ = run(one)
        one.close()
        out["single_handle_same_particles"] = {"ms_per_step": e1 / args.steps * 1e3, "particles": P * n,
                                               "multi_over_single": elapsed / e1,
                                               "note": "all shards share device %d: the same %d particles in ONE single-device handle, same steps" % (devices[0], P * n)}
    os.write(json_fd, (json.dumps(out) + "\n").encode())
    return 0


def units_per_s(P, C, M, elapsed, steps, world):
    return P * world * C * M * steps / elapsed


if __name__ == "__main__":
    sys.exit(main())
