"""CPU tier: partitioning / gather bookkeeping of the C++ multi-GPU batch driver (tools/batch_plan.hpp), and that the
driver itself builds and refuses to run without a GPU (no CPU path)."""
import os
import subprocess

from conftest import PKG, ROOT


def test_batch_plan_partitioning(tmp_path):
    exe = str(tmp_path / "batch_plan_test")
    src = os.path.join(ROOT, "tests", "cpp", "batch_plan_test.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr


def test_pano_batch_builds_and_needs_a_gpu():
    import torch
    subprocess.run(["make", "-C", PKG, "-j8", "examples"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(PKG, "tools", "pano_batch")
    assert os.path.exists(exe)
    r = subprocess.run([exe, "-pairs", "2", "-size", "bogus"], capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([exe, "-pairs", "2", "-size", "256x256"], capture_output=True, text=True)
        assert r.returncode == 1 and "no HIP device" in r.stderr
