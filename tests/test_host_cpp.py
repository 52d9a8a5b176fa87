"""The C++ host mirror compiles against include/phdhip.h, links libphdhip.so and behaves: without a GPU
it fails loudly, with one it runs."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp):
    from monorfs_amd import _lib
    so = _lib.build()
    exe = os.path.join(tmp, "host_smoke")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-o", exe, os.path.join(ROOT, "tests", "host_smoke.cpp"),
                           so, "-Wl,-rpath," + os.path.dirname(so), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_cpp_host_fails_loudly_without_gpu(tmp_path):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu test")
    r = subprocess.run([build(str(tmp_path))], capture_output=True, text=True)
    assert r.returncode == 3, r.stdout + r.stderr
    assert "no HIP device" in r.stdout


@pytest.mark.gpu
def test_cpp_host_runs_on_gpu(tmp_path):
    r = subprocess.run([build(str(tmp_path))], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host smoke ok" in r.stdout
