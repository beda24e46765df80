"""GPU end-to-end parity through the C ABI vs the oracle on the same seeded inputs (config 1 size).
Stated tolerance: max per-pixel |dflow| <= 1e-2 px, blended PSNR >= 50 dB; the
measured result is bit-identical flows, which the test asserts (and reports the deltas if not)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx(pf):
    c = pf.Context(0)
    yield c
    c.close()


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


@pytest.fixture(scope="module")
def pair512(synth):
    return synth.make_pair_np(512, 512, 1234)


@pytest.mark.parametrize("alg", ["pixflow_low", "pixflow_search_20"])
def test_flow_bidir_and_blend_512(ctx, orc, pf, pair512, alg):
    L, R, blend = pair512
    mp = pf.max_percentage_by_name(alg)
    rLR, rRL = orc.flow_bidir(L, R, mp)
    rout = orc.combine_novel_views(L, R, rLR, rRL, blend)
    out, fLR, fRL = ctx.novel_view(L, R, mp, blend)
    dLR = np.abs(fLR - rLR).max(); dRL = np.abs(fRL - rRL).max()
    print("max|dflow| LR %g RL %g" % (dLR, dRL))
    assert dLR <= 1e-2 and dRL <= 1e-2
    assert np.array_equal(fLR, rLR) and np.array_equal(fRL, rRL)
    # blended strip: byte-identical (tanhf / exp with the host libm's roundings on the device, csrc/libm_exact.hpp)
    assert np.array_equal(out, rout), "%d blended bytes differ (PSNR %.2f dB)" % (int((out != rout).sum()), _psnr(out, rout))


@pytest.mark.parametrize("cols,rows,scale,alg", [(900, 520, 6.0, "pixflow_low"), (520, 900, 10.0, "pixflow_search_20")])
def test_large_displacement_pair_vs_oracle(ctx, orc, pf, synth, cols, rows, scale, alg):
    """Round 5: a whole bidirectional solve + blend on a pair whose displacement field is 6x / 10x the benchmark scene's (up to ~90 px at full
    resolution: the sweeps' gather windows follow offsets of tens of texels at the fine levels, drift with them and are cut back at the
    borders), both band orientations (wide and tall pair), lone-pair entry point and a batch of three in the throughput form: the oracle's bits."""
    import torch
    L, R, blend, _ = synth.make_pair(cols, rows, 77, "cpu", disp_scale=scale)
    L, R, blend = L.numpy(), R.numpy(), blend.numpy()
    mp = pf.max_percentage_by_name(alg)
    rLR, rRL = orc.flow_bidir(L, R, mp)
    assert np.abs(rLR).max() > 3.5 * scale            # the solve does follow the large field
    rout = orc.combine_novel_views(L, R, rLR, rRL, blend)
    out, fLR, fRL = ctx.novel_view(L, R, mp, blend)
    assert np.array_equal(fLR, rLR) and np.array_equal(fRL, rRL), "flows differ: max %g" % max(np.abs(fLR - rLR).max(), np.abs(fRL - rRL).max())
    assert np.array_equal(out, rout)
    c2 = pf.Context(0, sweep_wide=2)                  # every sweep launch in the throughput form
    n = cols * rows
    d = [{"L": c2.dev_alloc(n * 4), "R": c2.dev_alloc(n * 4), "b": c2.dev_alloc(n * 4), "o": c2.dev_alloc(n * 4), "f0": c2.dev_alloc(n * 8), "f1": c2.dev_alloc(n * 8)} for _ in range(3)]
    for k in d:
        c2.upload(k["L"], L); c2.upload(k["R"], R); c2.upload(k["b"], blend)
    c2.novel_view_batch_dev([k["L"] for k in d], [k["R"] for k in d], cols, rows, mp, [k["b"] for k in d], [k["o"] for k in d], [k["f0"] for k in d], [k["f1"] for k in d], in_flight=3)
    for k in d:
        assert np.array_equal(c2.download(np.empty((rows, cols, 2), np.float32), k["f0"]), rLR)
        assert np.array_equal(c2.download(np.empty((rows, cols, 2), np.float32), k["f1"]), rRL)
        assert np.array_equal(c2.download(np.empty((rows, cols, 4), np.uint8), k["o"]), rout)
    c2.close()


def test_blend_only(ctx, orc, pair512):
    L, R, blend = pair512
    r = np.random.default_rng(3)
    fLR = (r.standard_normal((512, 512, 2)) * 3).astype(np.float32); fRL = (r.standard_normal((512, 512, 2)) * 3).astype(np.float32)
    ref = orc.combine_novel_views(L, R, fLR, fRL, blend)
    got = ctx.blend(L, R, fLR, fRL, blend)
    assert np.array_equal(got, ref), "%d blended bytes differ" % int((got != ref).sum())
    # blend == 0 / 1 with zero flow reproduces L / R up to the reference's float->uchar truncation
    # (weights sum to 1-ulp), alpha 255 where both inputs are valid and (0,0,0,0) elsewhere
    z = np.zeros((512, 512, 2), np.float32)
    for b, src in ((0.0, L), (1.0, R)):
        bl = np.full((512, 512), b, np.float32)
        o = ctx.blend(L, R, z, z, bl)
        ro = orc.combine_novel_views(L, R, z, z, bl)
        assert np.array_equal(o, ro)
        m = (L[..., 3] > 0) & (R[..., 3] > 0)
        d = src[m][:, :3].astype(np.int32) - o[m][:, :3].astype(np.int32)
        assert d.min() >= 0 and d.max() <= 1 and (o[m][:, 3] == 255).all() and (o[~m] == 0).all()


def test_single_direction_flow_unpadded(ctx, orc, synth):
    L, R, _ = synth.make_pair_np(320, 256, 77)
    for hint, mp in ((0, 20), (1, 20), (3, 0)):
        ref = orc.compute_optical_flow(L, R, mp, hint)
        got = ctx.flow(L, R, mp, hint)
        assert np.array_equal(got, ref)


def test_constant_images(ctx, orc):
    """Known answer: with no image gradient only the regularisers act (d/df of 0.001*|blur-f| + 0.01*|f|/W
    by forward differences), so the flow drifts slightly negative and is the same in both directions."""
    img = np.full((256, 300, 4), 128, np.uint8); img[..., 3] = 255
    f0, f1 = ctx.flow_bidir(img, img, 0)
    r0, r1 = orc.flow_bidir(img, img, 0)
    assert np.array_equal(f0, r0) and np.array_equal(f1, r1) and np.array_equal(f0, f1)
    assert f0.max() <= 0 and np.abs(f0).max() < 0.1


def test_bad_arguments(ctx, pf):
    with pytest.raises(pf.PanoflowError):
        ctx.flow(np.zeros((1, 1, 4), np.uint8), np.zeros((1, 1, 4), np.uint8), 0, 0)
    with pytest.raises(pf.PanoflowError):
        ctx.flow(np.zeros((64, 64, 4), np.uint8), np.zeros((64, 64, 4), np.uint8), 0, 9)


def test_stitch_prepare_and_gather(ctx, orc, synth):
    import torch
    L, R = synth.make_canvas_pair(640, 420, 21)
    L, R = L.numpy(), R.numpy()
    rmap, rovl, rovr, rblend, rmd = orc.stitch_prepare(L, R, True)
    mp, ovl, ovr, bl, md = ctx.stitch_prepare(L, R)
    assert np.array_equal(mp, rmap) and np.array_equal(ovl, rovl) and np.array_equal(ovr, rovr)
    assert np.array_equal(md, rmd)
    assert np.array_equal(bl, rblend), "max|d| %g" % np.abs(bl - rblend).max()
    assert set(np.unique(rmap)) >= {0, 50, 100, 150}
    merged = np.where((rmap == 150)[..., None], ovl, 0).astype(np.uint8)
    merged[200:210, 300:330] = 0  # a hole in the merged middle -> 8-direction probe path
    assert np.array_equal(ctx.stitch_gather(L, R, merged, rmap), orc.stitch_gather(L, R, merged, rmap))


def test_cpp_dropin_headers_one_stitch_step(orc, synth, pf, tmp_path):
    """The reference's own call sequence (CPU/main.cpp:70-95) through the C++ drop-in headers
    (Stitchtools + NovelViewGeneratorAsymmetricFlow) vs the oracle running the same sequence."""
    import os, subprocess
    from conftest import PKG
    exe = os.path.join(PKG, "examples", "stitch_pair")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    cols, rows = 640, 420
    L, R = synth.make_canvas_pair(cols, rows, 21)
    L, R = L.numpy(), R.numpy()
    L.tofile(tmp_path / "L.bgra"); R.tofile(tmp_path / "R.bgra")
    subprocess.check_call([exe, str(cols), str(rows), str(tmp_path / "L.bgra"), str(tmp_path / "R.bgra"), "pixflow_search_20", str(tmp_path / "o")])
    rd = lambda n, dt, sh: np.fromfile(tmp_path / ("o." + n), dtype=dt).reshape(sh)
    mp, ovl, ovr, blend, _ = orc.stitch_prepare(L, R, True)
    fLR, fRL = orc.flow_bidir(ovl, ovr, 20)
    merged = orc.combine_novel_views(ovl, ovr, fLR, fRL, blend)
    final = orc.stitch_gather(L, R, merged, mp)
    assert np.array_equal(rd("map.u8", np.uint8, (rows, cols)), mp)
    assert np.array_equal(rd("blend.f32", np.float32, (rows, cols)), blend)
    assert np.array_equal(rd("flowLR.f32", np.float32, (rows, cols, 2)), fLR)
    assert np.array_equal(rd("flowRL.f32", np.float32, (rows, cols, 2)), fRL)
    gm = rd("merged.bgra", np.uint8, (rows, cols, 4)); gf = rd("final.bgra", np.uint8, (rows, cols, 4))
    assert np.array_equal(gm, merged) and np.array_equal(gf, final)
    # unknown algorithm name -> VrCamException -> exit code 1 (PixFlow.hpp:499)
    assert subprocess.call([exe, str(cols), str(rows), str(tmp_path / "L.bgra"), str(tmp_path / "R.bgra"), "nope", str(tmp_path / "x")]) == 1


@pytest.mark.parametrize("cols,rows", [(52, 52), (61, 53), (201, 157), (96, 300), (300, 96)])
def test_odd_and_minimal_sizes(ctx, orc, synth, cols, rows):
    """Smallest pyramid (one level: half-res 26x26), odd sizes, extreme aspect ratios (both band orientations)."""
    L, R, _ = synth.make_pair_np(cols, rows, 9 + cols)
    for mp in (0, 20):
        r0, r1 = orc.flow_bidir(L, R, mp)
        f0, f1 = ctx.flow_bidir(L, R, mp)
        assert np.array_equal(f0, r0) and np.array_equal(f1, r1)


def test_degenerate_alpha(ctx, orc, synth):
    """No pixel gated (alpha 0 everywhere in one image) and every pixel gated (alpha 255 everywhere)."""
    L, R, _ = synth.make_pair_np(160, 120, 3)
    L0 = L.copy(); L0[..., 3] = 0
    for a, b in ((L0, R), (L, L0)):
        r0, r1 = orc.flow_bidir(a, b, 20)
        f0, f1 = ctx.flow_bidir(a, b, 20)
        assert np.array_equal(f0, r0) and np.array_equal(f1, r1)
    Lf = L.copy(); Rf = R.copy(); Lf[..., 3] = 255; Rf[..., 3] = 255
    r0, r1 = orc.flow_bidir(Lf, Rf, 0)
    f0, f1 = ctx.flow_bidir(Lf, Rf, 0)
    assert np.array_equal(f0, r0) and np.array_equal(f1, r1)


def test_repeated_calls_and_size_changes_reuse_context(ctx, orc, synth):
    """The arena grows and is reused; results must not depend on what ran before (stale hand-off state etc.)."""
    for (cols, rows, seed) in [(240, 200, 1), (128, 96, 2), (240, 200, 1), (320, 256, 3)]:
        L, R, blend = synth.make_pair_np(cols, rows, seed)
        r0, r1 = orc.flow_bidir(L, R, 0)
        out, f0, f1 = ctx.novel_view(L, R, 0, blend)
        assert np.array_equal(f0, r0) and np.array_equal(f1, r1)


def test_fused_stitch_step_equals_object_sequence(ctx, synth):
    """pf_stitch_step (device-resident iteration, chained R kept in HBM) runs the same kernels as
    stitch_prepare + novel_view + stitch_gather: identical bytes, step after step."""
    cols, rows = 480, 320
    top, imgs = synth.make_stitch_set(cols, rows, 5, 5)
    top = top.numpy(); imgs = [im.numpy() for im in imgs[:3]]
    R = top
    for i, L in enumerate(imgs):
        mp, ovl, ovr, blend, _ = ctx.stitch_prepare(L, R)
        merged, _, _ = ctx.novel_view(ovl, ovr, 20, blend)
        R = ctx.stitch_gather(L, R, merged, mp)
        fused = ctx.stitch_step(L, top if i == 0 else None, 20)
        assert np.array_equal(fused, R), "step %d" % (i + 1)


def test_flow_bidir_fuzz_alpha_shapes(ctx, orc, synth):
    """6 seeded random pairs (odd sizes, both algorithms) whose alpha channels keep only a few random rectangles: every pyramid
    level then has its own, sometimes empty, window of updated pixels.  Flows bit-identical to the oracle."""
    r = np.random.default_rng(4242)
    for case in range(6):
        cols, rows = int(r.integers(60, 420)), int(r.integers(60, 420))
        L, R, _ = synth.make_pair_np(cols, rows, 500 + case)
        L = L.copy(); R = R.copy()
        for img in (L, R):
            keep = np.zeros((rows, cols), bool)
            for _ in range(int(r.integers(1, 4))):
                x0, y0 = int(r.integers(0, cols)), int(r.integers(0, rows))
                keep[y0:int(r.integers(y0, rows)) + 1, x0:int(r.integers(x0, cols)) + 1] = True
            if case == 3:
                keep[:] = False                      # one image without any valid pixel
            img[..., 3] = np.where(keep, img[..., 3], 0)
        max_pct = 20 if case % 2 else 0
        r0, r1 = orc.flow_bidir(L, R, max_pct)
        g0, g1 = ctx.flow_bidir(L, R, max_pct)
        assert np.array_equal(g0.view(np.uint32), r0.view(np.uint32)) and np.array_equal(g1.view(np.uint32), r1.view(np.uint32)), \
            "case %d (%dx%d, max_pct %d): %d / %d mismatches" % (case, cols, rows, max_pct, (g0 != r0).sum(), (g1 != r1).sum())


def test_stitch_prefetch_is_result_neutral(ctx, synth):
    """pf_stitch_prefetch only moves the next image's upload under the current step's compute: same composites, also when
    the announced buffer is then NOT the one passed (the hint is ignored)."""
    cols, rows = 480, 320
    top, imgs = synth.make_stitch_set(cols, rows, 12, 4)
    top = top.numpy(); imgs = [np.ascontiguousarray(im.numpy()) for im in imgs]

    def chain(mode):
        outs = []
        for i, im in enumerate(imgs):
            if mode == "prefetch":
                ctx.stitch_prefetch(imgs[i + 1] if i + 1 < len(imgs) else None)
            elif mode == "wrong":
                ctx.stitch_prefetch(imgs[0])        # never the next image (except by accident at i = -1)
            outs.append(ctx.stitch_step(im, top if i == 0 else None, 20, want_out=True))
        ctx.stitch_prefetch(None)
        return outs

    ref = chain("plain")
    for mode in ("prefetch", "wrong"):
        got = chain(mode)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref)), mode
