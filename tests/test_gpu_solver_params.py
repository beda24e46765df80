"""Round 6: the reference's PixFlow<P> takes its coefficients as constructor arguments (CPU/PixFlow.hpp:46-68);
its factory only ever passes one set (:459-497).  The HIP path now takes them at run time too (pf_set_solver_params): non-preset sets
against the oracle (whose Params are run-time as well, orc.set_params), bit for bit -- the same bar as every other parity test."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# (pyr_scale_factor, smoothness_coef, vertical_regularization_coef, horizontal_regularization_coef, gradient_step_size)
SET_POW2 = (0.8, 0.002, 0.02, 0.005, 0.25)     # a power-of-two step size: the sweeps' fast exact forms (one fused multiply-add)
SET_ODD = (0.85, 0.0005, 0.0, 0.03, 0.7)       # any other step size: every sweep step through the IEEE sequence; one coefficient exactly zero
KEYS = ("pyr_scale_factor", "smoothness_coef", "vertical_regularization_coef", "horizontal_regularization_coef", "gradient_step_size")


@pytest.fixture()
def ctx(pf):
    c = pf.Context(0)
    yield c
    c.close()


@pytest.fixture()
def orc_params(orc):
    yield orc
    orc.reset_params()


@pytest.mark.parametrize("pset", [SET_POW2, SET_ODD], ids=["step_pow2", "step_odd"])
@pytest.mark.parametrize("cols,rows,max_pct", [(512, 512, 0), (700, 520, 20)])
def test_non_preset_parameters_vs_oracle(ctx, orc_params, synth, pset, cols, rows, max_pct):
    orc = orc_params
    L, R, blend = synth.make_pair_np(cols, rows, 99 + cols)
    preset = orc.flow_bidir(L, R, max_pct)
    orc.set_params(*pset)
    want = orc.flow_bidir(L, R, max_pct)
    assert not np.array_equal(want[0], preset[0])           # the parameters do change the solution
    ctx.set_solver_params(**dict(zip(KEYS, pset)))
    got = ctx.flow_bidir(L, R, max_pct)
    for d in range(2):
        assert np.array_equal(got[d], want[d]), "direction %d: max |dflow| %g" % (d, np.abs(got[d] - want[d]).max())
    # blend on those flows, and the single-direction entry point with a hint
    out, _, _ = ctx.novel_view(L, R, max_pct, blend)
    assert np.array_equal(out, orc.combine_novel_views(L, R, want[0], want[1], blend))
    one = ctx.flow(L[:, :300].copy(), R[:, :300].copy(), max_pct, 3)
    assert np.array_equal(one, orc.compute_optical_flow(L[:, :300].copy(), R[:, :300].copy(), max_pct, 3))
    # ... and back: the presets' bits again
    ctx.set_solver_params()
    orc.reset_params()
    back = ctx.flow_bidir(L, R, max_pct)
    assert np.array_equal(back[0], preset[0]) and np.array_equal(back[1], preset[1])


@pytest.mark.parametrize("pset", [SET_POW2, SET_ODD], ids=["step_pow2", "step_odd"])
def test_non_preset_parameters_in_the_throughput_form_and_in_batches(pf, orc_params, synth, pset):
    """The batch entry point (its lanes inherit the owner's parameters) with every sweep launch in the throughput form."""
    orc = orc_params
    cols, rows, mp = 420, 360, 0
    pairs = [synth.make_pair_np(cols, rows, 500 + i) for i in range(3)]
    orc.set_params(*pset)
    want = []
    for L, R, blend in pairs:
        f0, f1 = orc.flow_bidir(L, R, mp)
        want.append((f0, f1, orc.combine_novel_views(L, R, f0, f1, blend)))
    c = pf.Context(0, sweep_wide=2)
    c.set_solver_params(**dict(zip(KEYS, pset)))
    n = cols * rows
    d = [{"L": c.dev_alloc(n * 4), "R": c.dev_alloc(n * 4), "b": c.dev_alloc(n * 4), "o": c.dev_alloc(n * 4), "f0": c.dev_alloc(n * 8), "f1": c.dev_alloc(n * 8)} for _ in pairs]
    for k, (L, R, blend) in zip(d, pairs):
        c.upload(k["L"], L); c.upload(k["R"], R); c.upload(k["b"], blend)
    for in_flight in (3, 1):
        c.novel_view_batch_dev([k["L"] for k in d], [k["R"] for k in d], cols, rows, mp, [k["b"] for k in d], [k["o"] for k in d], [k["f0"] for k in d], [k["f1"] for k in d], in_flight=in_flight)
        for k, w in zip(d, want):
            assert np.array_equal(c.download(np.empty((rows, cols, 2), np.float32), k["f0"]), w[0])
            assert np.array_equal(c.download(np.empty((rows, cols, 2), np.float32), k["f1"]), w[1])
            assert np.array_equal(c.download(np.empty((rows, cols, 4), np.uint8), k["o"]), w[2])
    c.close()


def test_stage_level_with_parameters_in_all_sweep_forms(pf, orc_params, synth):
    """One level (blurred flow, two sweeps, two medians, diffusion) on explicit planes: latency form, throughput form, and the lab build's
    independent v1 kernel, all with a non-preset set."""
    orc = orc_params
    rng = np.random.default_rng(5)
    w, h = 233, 141
    L, R, _ = synth.make_pair_np(2 * w, 2 * h, 31)
    I0, a0 = orc.preprocess(L); I1, a1 = orc.preprocess(R)
    I0, a0, I1, a1 = I0[:h, :w].copy(), a0[:h, :w].copy(), I1[:h, :w].copy(), a1[:h, :w].copy()
    fin = (rng.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
    for pset in (SET_POW2, SET_ODD):
        orc.set_params(*pset)
        want = orc.level(I0, I1, a0, a1, fin, 0, 0)
        for kw in (dict(), dict(sweep_wide=2), dict(exp=True, sweep_impl=1)):
            c = pf.Context(0, **kw)
            c.set_solver_params(**dict(zip(KEYS, pset)))
            got = c.stage_level(I0, I1, a0, a1, fin, 0, 0)
            c.close()
            assert np.array_equal(got, want), (pset, kw, float(np.abs(got - want).max()))


def test_parameter_validation(ctx, pf):
    assert ctx.solver_params()["gradient_step_size"] == 0.5 and abs(ctx.solver_params()["pyr_scale_factor"] - 0.9) < 1e-7
    for bad in (dict(downscale_factor=0.4), dict(smoothness_coef=-0.001), dict(vertical_regularization_coef=float("nan")), dict(gradient_step_size=float("inf")),
                dict(gradient_step_size=-0.5), dict(pyr_scale_factor=1.0), dict(pyr_scale_factor=0.1), dict(horizontal_regularization_coef=-1.0)):
        with pytest.raises(pf.PanoflowError):
            ctx.set_solver_params(**bad)
        assert ctx.solver_params()["gradient_step_size"] == 0.5      # a rejected set changes nothing
    ctx.set_solver_params(gradient_step_size=0.0)                    # "no gradient step" is a legitimate (if odd) request
    assert ctx.solver_params()["gradient_step_size"] == 0.0


def test_cpp_dropin_pixflow_with_custom_coefficients(orc_params, synth, tmp_path):
    """PixFlow<0>(pyrScaleFactor, smoothnessCoef, ...) through panorama-opticalflow_amd/include/PixFlow.hpp: the constructor arguments reach the
    kernels, and a named algorithm used right afterwards on the same (per-thread) context gets the factory's presets again."""
    from conftest import PKG
    orc = orc_params
    exe = os.path.join(PKG, "examples", "custom_flow")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    cols, rows = 384, 300
    L, R, _ = synth.make_pair_np(cols, rows, 8)
    L.tofile(tmp_path / "L.bgra"); R.tofile(tmp_path / "R.bgra")
    subprocess.check_call([exe, str(cols), str(rows), str(tmp_path / "L.bgra"), str(tmp_path / "R.bgra"), "3"] + ["%.9g" % v for v in SET_ODD] + [str(tmp_path / "o.f32")])
    got = np.fromfile(tmp_path / "o.f32", dtype=np.float32).reshape(2, rows, cols, 2)
    preset = orc.compute_optical_flow(L, R, 0, 3)
    orc.set_params(*SET_ODD)
    want = orc.compute_optical_flow(L, R, 0, 3)
    assert np.array_equal(got[0], want) and np.array_equal(got[1], preset) and not np.array_equal(want, preset)
    # a negative coefficient is refused (VrCamException -> exit code 1)
    assert subprocess.call([exe, str(cols), str(rows), str(tmp_path / "L.bgra"), str(tmp_path / "R.bgra"), "3", "0.9", "0.001", "-1", "0.01", "0.5", str(tmp_path / "x")]) == 1
