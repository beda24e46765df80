"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle): the oracle
must keep reproducing them (CPU), and the HIP path must reproduce them through the C ABI (GPU)."""
import os

import numpy as np
import pytest

from conftest import ROOT

G = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def gflow():
    return np.load(os.path.join(G, "flow_160x128.npz"))


@pytest.fixture(scope="module")
def gstitch():
    return np.load(os.path.join(G, "stitch_240x200.npz"))


def test_synth_inputs_reproducible(gflow, synth):
    L, R, blend = synth.make_pair_np(160, 128, 1234)
    assert np.array_equal(L, gflow["L"]) and np.array_equal(R, gflow["R"]) and np.array_equal(blend, gflow["blend"])


@pytest.mark.parametrize("name,mp", [("low", 0), ("s20", 20)])
def test_oracle_matches_golden_flow(gflow, orc, name, mp):
    f0, f1 = orc.flow_bidir(gflow["L"], gflow["R"], mp)
    assert np.array_equal(f0, gflow["flowLR_" + name]) and np.array_equal(f1, gflow["flowRL_" + name])
    assert np.array_equal(orc.combine_novel_views(gflow["L"], gflow["R"], f0, f1, gflow["blend"]), gflow["merged_" + name])
    # the search preset must actually change something on this pair
    if mp:
        assert not np.array_equal(f0, gflow["flowLR_low"])


def test_oracle_matches_golden_stitch(gstitch, orc):
    mp_, ovl, ovr, bl, md = orc.stitch_prepare(gstitch["L"], gstitch["R"], True)
    assert np.array_equal(mp_, gstitch["map"]) and np.array_equal(bl, gstitch["blend"]) and np.array_equal(md, gstitch["mergedDis"])
    assert np.array_equal(orc.stitch_gather(gstitch["L"], gstitch["R"], gstitch["merged"], mp_), gstitch["final"])


@pytest.mark.gpu
@pytest.mark.parametrize("name,mp", [("low", 0), ("s20", 20)])
def test_gpu_matches_golden_flow(gflow, pf, name, mp):
    ctx = pf.Context(0)
    out, f0, f1 = ctx.novel_view(gflow["L"], gflow["R"], mp, gflow["blend"])
    assert np.array_equal(f0, gflow["flowLR_" + name]) and np.array_equal(f1, gflow["flowRL_" + name])
    assert np.array_equal(out, gflow["merged_" + name]), "%d blended bytes differ from the golden fixture" % int((out != gflow["merged_" + name]).sum())
    ctx.close()


@pytest.mark.gpu
def test_gpu_matches_golden_stitch(gstitch, pf):
    ctx = pf.Context(0)
    mp_, ovl, ovr, bl, md = ctx.stitch_prepare(gstitch["L"], gstitch["R"])
    assert np.array_equal(mp_, gstitch["map"]) and np.array_equal(bl, gstitch["blend"]) and np.array_equal(md, gstitch["mergedDis"])
    assert np.array_equal(ctx.stitch_gather(gstitch["L"], gstitch["R"], gstitch["merged"], gstitch["map"]), gstitch["final"])
    ctx.close()
