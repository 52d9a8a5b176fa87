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


def _run(world, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29530 + world),
                        os.path.join(ROOT, "tests", "dist_gpu_worker.py")] + list(args), capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "multiproc ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


@pytest.mark.parametrize("world,mode", [(2, "device"), (3, "device"), (2, "flags"), (3, "flags"), (2, "host"), (3, "host")])
def test_sharded_sequence_in_real_processes(world, mode):
    _run(world, mode)


def test_four_processes_with_shards_of_config_c8():
    """Four processes, each a rank's shard of config C8 (2048 particles x 512 components x 64 measurements): the 8192-slot global
    vector is resampled and planned by the grid kernels at their default thresholds, the records go into the other PROCESSES'
    buffers — against one handle of 8192 particles in rank 0, bit for bit. With a barrier behind the push, not the landing flags:
    the ranks share ONE GPU here, and a rank whose 2048 workgroups of k_finish_sharded wait for a flag hold every slot the
    sender's kernels would need (the wait then runs into its bound — seen; between GPUs a waiting rank holds only its own)."""
    assert "4 ranks x 2048 particles x 512 components x 64 measurements" in _run(4, "device", "big")
