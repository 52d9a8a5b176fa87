"""Two / three real processes, one rank each, sharing the GPU: the sharded SlamUpdate (collectives on gloo, staged through
host memory because RCCL refuses two ranks on one device) against a single handle holding all particles — in the sequence
bench.py runs (plan on the device, migrating particles stored into the other processes' IPC-opened buffers, no host wait) and
with the landing flags in place of the barrier behind the push (round 5: nothing between push and unpack), and in round 3's
(host-side split sizes, all_to_all_single)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, *args, **more_env):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world), **more_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29530 + world),
                        os.path.join(ROOT, "tests", "dist_gpu_worker.py")] + list(args), capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "multiproc ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world,mode", [(2, "device"), (3, "device"), (2, "flags"), (3, "flags"), (2, "host"), (3, "host")])
def test_sharded_sequence_in_real_processes(world, mode):
    _run(world, mode)


@pytest.mark.parametrize("mode", ["device", "flags"])
def test_four_processes_with_shards_of_config_c8(mode):
    """Four processes, each a rank's shard of config C8 (2048 particles x 512 components x 64 measurements): the 8192-slot global
    vector is resampled and planned by the grid kernels at their default thresholds, the records go into the other PROCESSES'
    buffers — against one handle of 8192 particles in rank 0, bit for bit. "flags": nothing between push and unpack but the landing
    flags, waited for by ONE wave in front of k_finish_sharded (k_wait_landing). (With the wait inside k_finish_sharded's 2048
    workgroups — PHD_LANDING_INLINE=1 — this case ran into the wait's bound on every try: the ranks share ONE GPU here, and a
    waiting grid holds the slots the senders' kernels need.)"""
    assert "4 ranks x 2048 particles x 512 components x 64 measurements" in _run(4, mode, "big")


def test_sharded_sequence_with_ordinary_receive_buffers():
    """bench.py's first fallback (a runtime that refuses IPC on fine-grained memory): PHD_COARSE_RECV=1 makes the receive buffers
    ordinary device allocations — exported, opened and stored into by the other processes all the same, with the barrier behind
    the push (the landing flags need fine-grained memory: phd_migration_set_landing refuses)."""
    _run(3, "device", PHD_COARSE_RECV="1")
