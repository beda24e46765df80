"""GPU parity, stage by stage: every HIP kernel family vs the oracle's restatement of the same
reference step on the same seeded inputs, through the C ABI.  Bar: bit-exact (np.array_equal treats
+0 == -0) for everything except the blend's libm transcendentals."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(11)


@pytest.fixture(scope="module", params=["latency", "wide", "throughput"])
def ctx(pf, request):
    """Every stage test runs with every form of the sweep (pf_config::sweep_wide): the latency form a lone pair uses (8 lanes per pixel),
    the same step with two compute waves per SIMD ("wide"), and the throughput form of the batch mode (2 lanes per pixel, bands of 32
    rows, non-speculative two-round step, skewed gather window)."""
    form = {"latency": 0, "wide": 1, "throughput": 2}[request.param]
    c = pf.Context(0, exp=form == 1, sweep_wide=form)   # form 1 (measured and rejected) only exists in the lab build
    yield c
    c.close()


def _planes(synth, orc, cols=300, rows=260, seed=5):
    L, R, _ = synth.make_pair_np(cols, rows, seed)
    I0, A0 = orc.preprocess(L)
    I1, A1 = orc.preprocess(R)
    return L, R, I0, A0, I1, A1


@pytest.mark.parametrize("cols,rows,pad", [(300, 260, 0), (301, 203, 15), (512, 512, 25)])
def test_preprocess(ctx, orc, synth, cols, rows, pad):
    L, _, _ = synth.make_pair_np(cols, rows, 3)
    if pad:
        Lp = np.concatenate([L[:, cols - pad:], L, L[:, :pad]], axis=1)
    else:
        Lp = L
    I, A = orc.preprocess(Lp)
    g, a = ctx.stage_preprocess(L, pad)
    assert np.array_equal(a, A)
    assert np.array_equal(g, I)


def test_pyr_down(ctx, orc):
    src = rng.random((256, 281)).astype(np.float32)
    for (w, h) in orc.pyramid_sizes(281, 256)[1:4]:
        ref = orc.pyr_down(src, w, h)
        got = ctx.stage_pyr_down(src, w, h)
        assert np.array_equal(got, ref)
        src = ref


def test_gradients(ctx, orc):
    img = rng.random((97, 131)).astype(np.float32)
    ix, iy = orc.gradients(img)
    g = ctx.stage_gradients(img)
    assert np.array_equal(g[..., 0], ix) and np.array_equal(g[..., 1], iy)


@pytest.mark.parametrize("ksize,sigma,cn,shape", [(5, 0.25, 1, (60, 77)), (3, 1.0, 2, (50, 64)), (15, 8.0, 2, (90, 120)), (15, 8.0, 2, (26, 29))])
def test_gauss(ctx, orc, ksize, sigma, cn, shape):
    img = rng.standard_normal(shape + ((cn,) if cn > 1 else ())).astype(np.float32)
    assert np.array_equal(ctx.stage_gauss(img, ksize, sigma), orc.gaussian_blur(img, ksize, sigma))


def test_median5(ctx, orc):
    f = rng.standard_normal((70, 93, 2)).astype(np.float32)
    f[10:20, 10:30] = 0.0
    assert np.array_equal(ctx.stage_median5(f), orc.median5(f))


def test_upsample_cubic(ctx, orc):
    f = rng.standard_normal((45, 25, 2)).astype(np.float32)
    ref = orc.resize_cubic_f32(f, 28, 50) * np.float32(1.0 / np.float32(0.9)) + np.float32(0)
    got = ctx.stage_upsample_cubic(f, 28, 50, float(np.float32(1.0) / np.float32(0.9)))
    assert np.array_equal(got, ref)


def test_final(ctx, orc):
    f = rng.standard_normal((100, 110, 2)).astype(np.float32)
    pad_cols, rows, pad = 220, 200, 10
    up = orc.resize_linear_f32(f, pad_cols, rows) * np.float32(2.0) + np.float32(0)
    ref = orc.gaussian_blur(up, 3, 1.0)[:, pad:pad_cols - pad]
    got = ctx.stage_final(f, pad_cols, rows, pad, 2.0)
    assert np.array_equal(got, ref)


def test_diffusion(ctx, orc):
    f = rng.standard_normal((64, 80, 2)).astype(np.float32)
    a0 = rng.random((64, 80)).astype(np.float32); a1 = rng.random((64, 80)).astype(np.float32)
    a0[:, :20] = 0; a1[30:, :] = 1
    assert np.array_equal(ctx.stage_diffusion(a0, a1, f), orc.diffusion(a0, a1, f))


@pytest.mark.parametrize("hint", [1, 2, 3, 4])
def test_adjust_initial_flow(ctx, orc, synth, hint):
    _, _, I0, A0, I1, A1 = _planes(synth, orc)
    sizes = orc.pyramid_sizes(I0.shape[1], I0.shape[0])
    for (w, h) in sizes[1:]:
        I0, I1, A0, A1 = (orc.pyr_down(p, w, h) for p in (I0, I1, A0, A1))
    ref = orc.adjust_initial_flow(I0, I1, A0, A1, hint, 20)
    got = ctx.stage_adjust_initial_flow(I0, I1, A0, A1, hint, 20)
    assert np.array_equal(got, ref)
    assert np.abs(ref).max() > 0  # the search moved something


@pytest.mark.parametrize("w,h", [(90, 70), (150, 131), (64, 257)])
@pytest.mark.parametrize("forward", [1, 0])
def test_sweep_bit_exact(ctx, orc, w, h, forward):
    """The hard one: the GPU wavefront must reproduce the sequential raster sweep bit for bit,
    including hand-off between 64-row bands (h > 64) and gated-off pixels."""
    r = np.random.default_rng(100 + w + h + forward)
    img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 2, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32)
    a0[h // 3: h // 3 + 9, w // 4: w // 2] = 0.5   # gated-off hole
    a1[:, :3] = 0.0
    ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, forward)
    got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, forward)
    assert np.array_equal(got, ref), "max |d| = %g, mismatches = %d" % (np.abs(got - ref).max(), (got != ref).sum())
    assert not np.array_equal(ref, flow)


@pytest.mark.parametrize("max_pct,hint", [(0, 3), (20, 3), (20, 1)])
def test_level_coarsest(ctx, orc, synth, max_pct, hint):
    _, _, I0, A0, I1, A1 = _planes(synth, orc, 400, 300, 9)
    sizes = orc.pyramid_sizes(I0.shape[1], I0.shape[0])
    for (w, h) in sizes[1:]:
        I0, I1, A0, A1 = (orc.pyr_down(p, w, h) for p in (I0, I1, A0, A1))
    ref = orc.level(I0, I1, A0, A1, None, hint, max_pct)
    got = ctx.stage_level(I0, I1, A0, A1, None, hint, max_pct)
    assert np.array_equal(got, ref)


def test_level_mid_with_incoming_flow(ctx, orc, synth):
    _, _, I0, A0, I1, A1 = _planes(synth, orc, 400, 300, 9)
    h, w = I0.shape
    fin = (rng.standard_normal((h, w, 2)) * 0.7).astype(np.float32)
    ref = orc.level(I0, I1, A0, A1, fin, 3, 0)
    got = ctx.stage_level(I0, I1, A0, A1, fin, 3, 0)
    assert np.array_equal(got, ref), "max |d| = %g" % np.abs(got - ref).max()


def test_sweep_v1_kernel_cross_check(pf, orc):
    """Two independent GPU implementations of the sweep (v1: 64 rows/wave, 5 dependent evaluations;
    v2: prepass + 8 lanes/pixel + I/O waves) must both equal the sequential oracle."""
    import os
    r = np.random.default_rng(77)
    h, w = 203, 171
    img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 1, axis=0) + 0.03 * r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 2.0).astype(np.float32)
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    a = np.ones((h, w), np.float32); a[50:60, 30:90] = 0.2
    c1 = pf.Context(0, exp=True, sweep_impl=1)     # the v1 kernel only exists in the lab build (libpanoflow_exp.so)
    c2 = pf.Context(0)
    c3 = pf.Context(0, exp=True)                   # the lab build's v2 sweep: -DPF_SAFE_PK, the packed chains as compiler-scheduled code
    for fwd in (1, 0):
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a, a, flow, fwd)
        assert np.array_equal(c1.stage_sweep(g0, g1, blurred, a, a, flow, fwd), ref)
        assert np.array_equal(c2.stage_sweep(g0, g1, blurred, a, a, flow, fwd), ref)
        assert np.array_equal(c3.stage_sweep(g0, g1, blurred, a, a, flow, fwd), ref)
    c1.close(); c2.close(); c3.close()


def test_asm_block_packed_chains_equal_the_safe_build(pf, synth):
    """The product's sweep issues its serial packed-fp32 chains as asm blocks without the compiler's wait states (csrc/exact_forms.hpp);
    the lab build (-DPF_SAFE_PK) compiles the same arithmetic as plain packed-vector C++.  A whole bidirectional solve must give the
    same bits from both, and the load-time probe (run by pf_create for every device, here once more) must report no mismatch."""
    L, R, blend = synth.make_pair_np(700, 520, 31)
    prod = pf.Context(0); safe = pf.Context(0, exp=True)
    assert prod.selftest_packed_chains() == 0 and safe.selftest_packed_chains() == 0
    o0, a0, b0 = prod.novel_view(L, R, 20, blend)
    o1, a1, b1 = safe.novel_view(L, R, 20, blend)
    assert np.array_equal(a0, a1) and np.array_equal(b0, b1) and np.array_equal(o0, o1)
    prod.close(); safe.close()


@pytest.mark.parametrize("w,h", [(140, 100), (100, 140)])
def test_sweep_large_flows_use_global_fallback(ctx, orc, w, h):
    """Proposals pointing further than the LDS gather window (+-7 texels) must take the HBM fallback and still be
    bit-exact; also covers both band orientations (W>H normal, W<H transposed)."""
    r = np.random.default_rng(5 + w)
    img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 12.0).astype(np.float32)      # many |flow| > 7, some far outside the image
    flow[::7, ::5] *= 20.0
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    a = np.ones((h, w), np.float32)
    for fwd in (1, 0):
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a, a, flow, fwd)
        got = ctx.stage_sweep(g0, g1, blurred, a, a, flow, fwd)
        assert np.array_equal(got, ref), "mismatches %d" % (got != ref).sum()


@pytest.mark.parametrize("w,h,amp", [(420, 150, 30.0), (150, 420, 30.0), (700, 260, 60.0), (300, 330, 12.0)])
def test_sweep_large_smooth_flows_window_follows(ctx, orc, w, h, amp):
    """Round 5: the sweeps' LDS gather window follows the flow (centred per chunk on pixel + the rounded blurred flow).  A LARGE, SMOOTH flow
    field -- tens of pixels, varying along and across the bands so that the window's offsets drift in both axes, running past the image
    borders (clamped centres), with holes in the alpha (pixels that are not updated) and a discontinuity (the offset may only move one texel
    per chunk: a stretch of chunks must fall back to HBM) -- in every sweep form and both band orientations, forward and backward:
    bit-identical to the oracle.  (Small flows and wild random ones are the neighbouring tests.)"""
    r = np.random.default_rng(int(amp) + w)
    img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    flow = np.stack([amp * np.sin(2.3 * np.pi * y / h) + 0.4 * amp * np.cos(3.1 * np.pi * x / w), 0.35 * amp * np.sin(2.7 * np.pi * x / w + 0.6) - 0.2 * amp * y / h], -1).astype(np.float32)
    flow[h // 3:h // 3 + 9, w // 2:] += np.float32(0.6 * amp)                     # a discontinuity across a few rows
    flow += (r.standard_normal((h, w, 2)) * 0.7).astype(np.float32)              # proposals differ from pixel to pixel
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32)
    a0[h // 5:h // 5 + 23, w // 4:w // 4 + 61] = 0.3; a1[2 * h // 3:, :w // 6] = 0.5                     # pixels that are not updated
    for fwd in (1, 0):
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, fwd)
        got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, fwd)
        assert np.array_equal(got, ref), "forward %d: mismatches %d" % (fwd, (got != ref).sum())


@pytest.mark.parametrize("w,h", [(400, 96), (96, 400)])
def test_sweep_flows_converging_on_the_far_border(ctx, orc, w, h):
    """The window offset along the step axis is cut back at the image's far border (every window centre must lie inside the image): there it
    falls by up to 8 texels per chunk while the offset across the band may drift at the same time.  Flows that point at the far border column /
    row from tens of pixels away (so that the samples stay inside the cut-back window) with a steady drift across and proposals scattered
    over the whole window, both band orientations, forward and backward, every form.  (In the throughput form the window's lower edge then
    moves BACK by a slot per chunk and the loader fills in what recently entered rows lack: those are the window's one-texel margin, only
    reachable through a rounding corner case -- tests/test_follow_window_model.py checks that bookkeeping on the CPU.)"""
    r = np.random.default_rng(w)
    img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    y, x = np.mgrid[0:h, 0:w].astype(np.float32)
    for sgn in (1.0, -1.0):
        if w > h:   # bands along x: converge on the right (sgn > 0) / left border, drift in y
            fx = np.minimum(40.0, (w - 1 - x) if sgn > 0 else x) * sgn
            fy = -sgn * np.clip(((x - (w - 120)) if sgn > 0 else (120 - x)) / 8.0, 0.0, 15.0)
        else:       # bands along y
            fy = np.minimum(40.0, (h - 1 - y) if sgn > 0 else y) * sgn
            fx = -sgn * np.clip(((y - (h - 120)) if sgn > 0 else (120 - y)) / 8.0, 0.0, 15.0)
        # proposals scattered over the whole window (+-4.5 around the smooth field): its edge rows / columns are sampled too
        flow = (np.stack([fx, fy], -1) + r.uniform(-4.5, 4.5, (h, w, 2))).astype(np.float32)
        blurred = orc.gaussian_blur(np.stack([fx, fy], -1).astype(np.float32), 15, 8.0)
        a = np.ones((h, w), np.float32)
        for fwd in (1, 0):
            ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a, a, flow, fwd)
            got = ctx.stage_sweep(g0, g1, blurred, a, a, flow, fwd)
            assert np.array_equal(got, ref), "sign %g forward %d: mismatches %d" % (sgn, fwd, (got != ref).sum())


@pytest.mark.parametrize("w,h", [(96, 40), (40, 96)])
def test_sweep_operands_outside_fast_math_range(ctx, orc, w, h):
    """The sweep kernel's cheap exact sqrt/division are only valid for operands that are 0 or in [2^-95, 2^100]; any
    step that sees something else must be redone with the IEEE sequence.  Plant tiny (1e-33 .. 1e-40, incl. denormal)
    and huge (1e31) flow components and flow/blurred differences and demand bit-equality with the oracle."""
    r = np.random.default_rng(11 + w)
    img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    flow[3::11, 2::7, 1] = np.float32(1e-33)            # regulariser operand 0.01*|f.y| ~ 1e-35 < 2^-95
    flow[5::13, 1::9, 0] = np.float32(-3e-39)           # denormal
    flow[7::17, 4::10, :] = np.float32(1e31)            # > 2^100
    blurred[2::9, 3::8, :] = flow[2::9, 3::8, :] + np.float32(1e-20)   # smoothness operand ~ 1e-40
    g0[1::6, 5::12, :] = g1[1::6, 5::12, :] + np.float32(1e-25)        # data-term operand ~ 1e-50 where the flow is ~0
    flow[1::6, 5::12, :] = 0.0
    a = np.ones((h, w), np.float32)
    for fwd in (1, 0):
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a, a, flow, fwd)
        got = ctx.stage_sweep(g0, g1, blurred, a, a, flow, fwd)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "mismatches %d" % (got.view(np.uint32) != ref.view(np.uint32)).sum()


@pytest.mark.parametrize("w,h,box", [(150, 90, (37, 21, 118, 70)), (90, 150, (5, 40, 60, 149)), (130, 130, (0, 0, 50, 130)), (100, 64, (99, 63, 100, 64)),
                                     (120, 80, (17, 9, 111, 15)), (96, 72, None), (100, 100, (40, 0, 60, 100)), (200, 300, (50, 10, 150, 290)),
                                     (300, 200, (10, 50, 290, 150))])
def test_sweep_only_covers_the_window_of_gated_pixels(ctx, orc, w, h, box):
    """The v2 sweep only processes the bounding box of the gated pixels (alpha0, alpha1 > 0.9); everything outside must keep
    its flow and still act as 'previous pixel' / 'row above' proposal for the window's first column and row.  Windows inside
    the image, touching borders, one pixel, a thin stripe, and no gated pixel at all (identity); plus holes inside the window."""
    r = np.random.default_rng(3 * w + h)
    img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 2.0).astype(np.float32)
    blurred = orc.gaussian_blur(flow, 15, 8.0)
    a0 = np.zeros((h, w), np.float32); a1 = np.ones((h, w), np.float32)
    if box is not None:
        x0, y0, x1, y1 = box
        a0[y0:y1, x0:x1] = 1.0
        a0[y0 + (y1 - y0) // 3:y0 + (y1 - y0) // 2, x0 + (x1 - x0) // 4:x0 + (x1 - x0) // 2] = 0.5     # a hole (not updated) inside
        a0[y0, x0] = 1.0; a0[y1 - 1, x1 - 1] = 1.0                                                   # keep the corners gated
    for fwd in (1, 0):
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, fwd)
        got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, fwd)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "fwd=%d mismatches %d" % (fwd, (got.view(np.uint32) != ref.view(np.uint32)).sum())
        if box is None:
            assert np.array_equal(got, flow)


def test_sweep_fuzz_sizes_gates_and_flows(ctx, orc):
    """40 seeded random cases: image size, gate pattern (random rectangles of valid alpha with holes, sometimes nothing),
    flow magnitude (small ... far outside the LDS window) and direction; every one bit-identical to the oracle."""
    r = np.random.default_rng(20260928)
    for case in range(40):
        w, h = int(r.integers(9, 230)), int(r.integers(9, 230))
        img0 = r.random((h, w)).astype(np.float32); img1 = r.random((h, w)).astype(np.float32)
        g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
        flow = (r.standard_normal((h, w, 2)) * float(r.choice([0.3, 2.0, 9.0]))).astype(np.float32)
        blurred = orc.gaussian_blur(flow, 15, 8.0)
        a0 = np.zeros((h, w), np.float32); a1 = np.ones((h, w), np.float32)
        for _ in range(int(r.integers(0, 4))):
            x0, y0 = int(r.integers(0, w)), int(r.integers(0, h))
            x1, y1 = int(r.integers(x0, w)) + 1, int(r.integers(y0, h)) + 1
            a0[y0:y1, x0:x1] = 1.0
        if r.random() < 0.5:
            x0, y0 = int(r.integers(0, w)), int(r.integers(0, h))
            a1[y0:y0 + int(r.integers(1, 20)), x0:x0 + int(r.integers(1, 20))] = 0.3
        fwd = int(r.integers(0, 2))
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, fwd)
        got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, fwd)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "case %d (%dx%d fwd=%d): %d mismatches" % (
            case, w, h, fwd, (got.view(np.uint32) != ref.view(np.uint32)).sum())


def test_level_fuzz_alpha_patterns(ctx, orc):
    """12 seeded random whole-level cases (blurred flow, both sweeps, medians, diffusion) with alpha planes made of random
    rectangles -- including levels whose gate is empty (sweeps are the identity) or a thin stripe -- bit-identical to the oracle."""
    r = np.random.default_rng(777)
    for case in range(12):
        w, h = int(r.integers(26, 180)), int(r.integers(26, 180))
        I0 = r.random((h, w)).astype(np.float32); I1 = np.roll(I0, int(r.integers(-3, 4)), axis=1) + 0.03 * r.random((h, w)).astype(np.float32)
        A0 = np.zeros((h, w), np.float32); A1 = np.zeros((h, w), np.float32)
        for A in (A0, A1):
            for _ in range(int(r.integers(0, 3)) + (case % 4 != 0)):
                x0, y0 = int(r.integers(0, w)), int(r.integers(0, h))
                A[y0:int(r.integers(y0, h)) + 1, x0:int(r.integers(x0, w)) + 1] = float(r.choice([1.0, 0.95, 0.6]))
        fin = (r.standard_normal((h, w, 2)) * 1.2).astype(np.float32)
        ref = orc.level(I0, I1, A0, A1, fin, 0, 0)
        got = ctx.stage_level(I0, I1, A0, A1, fin, 0, 0)
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), "case %d (%dx%d): %d mismatches" % (case, w, h, (got.view(np.uint32) != ref.view(np.uint32)).sum())


@pytest.mark.parametrize("mode", ["1", "2", "relax", "fuse"])
def test_sweep_record_experiment_paths_are_bit_identical(pf, orc, synth, mode):
    """Rejected-on-measurement alternatives must stay exact.  They live in the lab build (libpanoflow_exp.so, -DPF_EXPERIMENTS), not
    in the product library.  Mode relax = pf_config::sweep_impl 3, the event-driven relaxation sweep on LDS-resident tiles
    (kernels_relax.inl: same fixed point reached in any evaluation order; slower than the wavefront because the longest dependency
    chain, not the anti-diagonal count, still sets its time: profiles/r02_relaxation_sweep.txt).  record_path 1 (loader waves compute
    the records) and 2 (prepass blocks inside the sweep launch, G16/R1 hand-off) are the two record-path experiments.  Mode fuse = the
    throughput mode's launch fusion (upsample inside the next level's Gaussian, second median inside the diffusion kernel) forced on
    for every level -- product code, product library."""
    if mode == "relax":
        c = pf.Context(0, exp=True, sweep_impl=3)
    elif mode == "fuse":
        c = pf.Context(0, fuse_small_level_px=100000000)
    else:
        c = pf.Context(0, exp=True, record_path=int(mode))
    for (w, h, fwd) in [(150, 131, 1), (64, 257, 0), (300, 90, 1)]:
        r = np.random.default_rng(7 + w + fwd)
        img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 2, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
        g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
        flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
        bl = orc.gaussian_blur(flow, 15, 8.0)
        a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32); a0[h // 3: h // 3 + 9, w // 4: w // 2] = 0.5; a1[:, :3] = 0.0
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], bl, a0, a1, flow, fwd)
        got = c.stage_sweep(g0, g1, bl, a0, a1, flow, fwd)
        assert np.array_equal(got, ref), (w, h, fwd)
    L, R, blend = synth.make_pair_np(320, 256, 4321)
    f0, f1 = c.flow_bidir(L, R, 20)
    r0, r1 = orc.flow_bidir(L, R, 20)
    assert np.array_equal(f0, r0) and np.array_equal(f1, r1)
    c.close()


def test_product_library_ships_one_sweep(pf):
    """libpanoflow.so contains the wavefront sweep only: the cross-check implementations are refused, and it reads no PANOFLOW_*
    environment switch (none of the names is in the binary)."""
    with pytest.raises(pf.PanoflowError, match="PF_EXPERIMENTS"):
        pf.Context(0, sweep_impl=1)
    with pytest.raises(pf.PanoflowError, match="PF_EXPERIMENTS"):
        pf.Context(0, record_path=2)
    with pytest.raises(pf.PanoflowError, match="PF_EXPERIMENTS"):
        pf.Context(0, sweep_wide=1)   # the rejected wide shape of round 4
    for form in (3, 4):   # the throughput form's rejected record paths of rounds 4 / 5 no longer exist in any build
        with pytest.raises(pf.PanoflowError):
            pf.Context(0, exp=True, sweep_wide=form)
    blob = open(pf.SO_PATH, "rb").read()
    assert b"PANOFLOW_" not in blob and b"k_sweep_relax" not in blob
    assert b"k_sweep_relax" in open(pf.SO_PATH_EXP, "rb").read()
