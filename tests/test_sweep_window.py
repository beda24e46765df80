"""CPU tier: geometry of the swept window (host logic of the v2 sweep launch) checked by a small C++ program."""
import os
import subprocess

from conftest import ROOT


def test_sweep_window_geometry(tmp_path):
    exe = str(tmp_path / "sweep_window_test")
    src = os.path.join(ROOT, "tests", "cpp", "sweep_window_test.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-o", exe, src], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.startswith("ok ")
