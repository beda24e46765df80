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


def test_pano_batch_sha256_matches_hashlib(tmp_path):
    """the C++ SHA-256 that pano_batch -golden compares strips with, at the padding edge cases and on a few MB"""
    import hashlib
    import numpy as np
    subprocess.run(["make", "-C", PKG, "-j8", "examples"], check=True, stdout=subprocess.DEVNULL)
    exe = os.path.join(PKG, "tools", "pano_batch")
    rng = np.random.default_rng(1)
    for n in (0, 1, 55, 56, 63, 64, 65, 119, 120, 3000017):
        data = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        f = tmp_path / ("b%d" % n)
        f.write_bytes(data)
        r = subprocess.run([exe, "-sha256", str(f)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.strip() == hashlib.sha256(data).hexdigest(), n


def test_sidecar_fixture_matches_the_npz_fixtures():
    """tests/golden/dense_9000x4000.sha256.txt (what the C++ driver reads) says what the .npz fixtures say, for seeds 1234..1241"""
    import numpy as np
    lines = [ln.split() for ln in open(os.path.join(ROOT, "tests", "golden", "dense_9000x4000.sha256.txt")) if not ln.startswith("#")]
    assert [int(ln[0]) for ln in lines] == list(range(1234, 1242))
    for ln in lines:
        seed = int(ln[0])
        g = np.load(os.path.join(ROOT, "tests", "golden", "dense_9000x4000%s.npz" % ("" if seed == 1234 else "_s%d" % seed)))
        assert ln[1:] == [str(v) for v in g["sha_inputs"]] + [str(v) for v in g["sha_outputs"]]
