import importlib.util
import os
import sys

import pytest

# several contexts / lanes drive 4 HIP streams each: the runtime's default of 4 hardware queues would serialise them.
# Read by the HIP runtime at initialisation, i.e. before the first test touches the GPU.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "panorama-opticalflow_amd")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def load_pkg_module(name):
    """The package directory name has a hyphen (not importable); load its modules by path."""
    modname = "pano_amd_" + name
    if modname in sys.modules:
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(PKG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    import orc as _orc
    _orc.build()
    return _orc


@pytest.fixture(scope="session")
def synth():
    return load_pkg_module("synth")


def _torch_hip_first():
    """The PyTorch-ROCm wheel bundles its own libamdhip64; libpanoflow.so links /opt/rocm's.  Both runtimes live in one process, and
    torch's only finds the GPU if it is initialised BEFORE the system one (seen on the GPU box when a -k selection ran the
    numpy-only stage tests first: 'No HIP GPUs are available' in the first torch test after them).  Tests that hand torch tensors to the
    library therefore start torch's runtime before the library's first context, whatever the test order."""
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.init()
    except ImportError:
        pass


@pytest.fixture(scope="session")
def pf():
    """ctypes binding of the product C-ABI (include/panoflow.h)."""
    _torch_hip_first()
    return load_pkg_module("pyabi")
