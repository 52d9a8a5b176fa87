"""Two real processes, one rank each, sharing the GPU: the sharded SlamUpdate with its two collectives (gloo, staged
through host memory because RCCL refuses two ranks on one device) against a single handle holding all particles."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_sequence_in_real_processes(world):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29530 + world))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29530 + world),
                        os.path.join(ROOT, "tests", "dist_gpu_worker.py")], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0 and "multiproc ok" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
