"""CPU-side checks of the product boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/panoflow.h declares, and fails loudly (no fallback) when there is no GPU."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "panoflow.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(pf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported(pf):
    pf.build()
    lib = ctypes.CDLL(pf.SO_PATH)
    names = _declared_symbols()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), "include/panoflow.h declares %s but libpanoflow.so does not export it" % n
    assert set(pf.EXPORTS) == set(names)


def test_algorithm_names(pf):
    assert pf.max_percentage_by_name("pixflow_low") == 0
    assert pf.max_percentage_by_name("pixflow_search_20") == 20
    with pytest.raises(pf.PanoflowError):
        pf.max_percentage_by_name("pixflow_ultra")  # reference: VrCamException (PixFlow.hpp:499)


def test_geometry_matches_survey(pf):
    # level pixels / levels / sweep critical path (steps per direction)
    assert pf.level_pixels(512, 512) == (376212, 23, 2 * 4888)
    assert pf.level_pixels(2000, 4000) == (11582676, 37, 2 * 30363)
    assert pf.level_pixels(9000, 4000) == (52110962, 42, 2 * 68652)
    assert abs(pf.algorithmic_bytes(2000, 4000) / 1e9 - 6.30) < 0.01


def test_no_gpu_fails_loudly(pf):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pf.PanoflowError):
        pf.Context(0)
