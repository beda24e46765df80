"""Boundary / oracle hygiene: the raw countblend ratio, median selection on ties and signed zeros
(and what NaN does), and the three latent hazards of the reference that both sides DEFINE instead of reproducing
(SURVEY.md section 5): out-of-bounds probes in Gather, the single wrap in generateNovelViewPoint, step == 0 in countblend."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
rng = np.random.default_rng(23)


@pytest.fixture(scope="module")
def ctx(pf):
    c = pf.Context(0)
    yield c
    c.close()


def test_countblend_raw_ratio_and_merged_dis(ctx, orc, synth):
    """Stitchtools::countblend returns minLdis/(minRdis+minLdis) BEFORE any smoothing and writes MergedDis
    (CPU/StitchTool.cpp:185-190): pf_stitch_raw_blend against the oracle's GenerateBlend with the smoothing switched off."""
    L, R = synth.make_canvas_pair(640, 420, 9)
    L = L.numpy(); R = R.numpy()
    raw, md = ctx.stitch_raw_blend(L, R)
    mp, _, _, ref_raw, ref_md = orc.stitch_prepare(L, R, False)
    assert (mp == 150).mean() > 0.05
    assert np.array_equal(raw, ref_raw) and np.array_equal(md, ref_md)
    _, _, _, smooth, _ = ctx.stitch_prepare(L, R)
    assert not np.array_equal(raw, smooth)               # the smoothed ramp is a different thing


def test_median5_ties_and_signed_zeros(ctx, orc):
    """medianBlur(5) is pure selection: with heavy ties and both zeros present the selected VALUE equals the oracle's
    (np.array_equal: +0 == -0; which zero is picked is unspecified on both sides -- OpenCV's sorting network and
    nth_element disagree on that themselves)."""
    f = rng.integers(-2, 3, size=(61, 47, 2)).astype(np.float32)        # five distinct values: ties everywhere
    f[rng.random(f.shape) < 0.2] = -0.0
    got = ctx.stage_median5(f)
    assert np.array_equal(got, orc.median5(f))
    flat = np.full((33, 40, 2), 7.25, np.float32)
    assert np.array_equal(ctx.stage_median5(flat), flat)


def test_median5_nan_is_dropped_not_propagated(ctx):
    """NaN has no defined rank (the reference's result depends on OpenCV's SIMD vs scalar path).  Defined here: the
    min/max exchange network treats a NaN as absent (fminf/fmaxf return the other operand), so a window with a few NaNs
    yields a finite value inside the range of its finite members, and windows without NaN are unaffected."""
    f = rng.standard_normal((40, 52, 2)).astype(np.float32)
    clean = ctx.stage_median5(f)
    g = f.copy(); g[20, 30, 0] = np.nan; g[5, 5, 1] = np.nan
    got = ctx.stage_median5(g)
    assert np.isfinite(got).all()
    far = np.ones(f.shape[:2], bool); far[18:23, 28:33] = False; far[3:8, 3:8] = False
    assert np.array_equal(got[far], clean[far])
    win = f[18:23, 28:33, 0]
    assert win.min() <= got[20, 30, 0] <= win.max()


def test_gather_out_of_bounds_probes_are_no_match(ctx, orc):
    """Gather's 8-direction probes (StitchTool.cpp:70-90) index the map without bounds checks; defined as 'no match'.
    An overlap pixel (code 150, merged alpha 0) in a corner whose only L-only neighbour lies to the right."""
    cols, rows = 64, 48
    L = np.zeros((rows, cols, 4), np.uint8); R = np.zeros_like(L); merged = np.zeros_like(L)
    mp = np.zeros((rows, cols), np.uint8)
    mp[0:3, 0:3] = 150                      # overlap pixels in the top-left corner, nothing merged there
    mp[0:3, 3:10] = 100                     # L-only to the right
    mp[40:48, 60:64] = 150                  # bottom-right corner, nothing within reach -> stays (0,0,0,255)
    L[...] = (10, 20, 30, 255); R[...] = (200, 210, 220, 255)
    got = ctx.stitch_gather(L, R, merged, mp)
    assert np.array_equal(got, orc.stitch_gather(L, R, merged, mp))
    assert tuple(got[0, 0]) == (10, 20, 30, 255)           # found L at distance 3 to the right; probes at x-3 < 0 ignored
    assert tuple(got[47, 63]) == (0, 0, 0, 255)


def test_novel_view_point_wraps_with_true_modulo(ctx, orc, synth):
    """generateNovelViewPoint wraps x once (OpticalFlow.cpp:17-20): |flow * t| > cols would read out of bounds.
    Defined as a true modulo on both sides."""
    cols, rows = 96, 64
    L, R, blend = synth.make_pair_np(cols, rows, 3)
    L[..., 3] = 255; R[..., 3] = 255
    f0 = np.zeros((rows, cols, 2), np.float32); f1 = np.zeros_like(f0)
    f0[..., 0] = 2.5 * cols; f1[..., 0] = -3.25 * cols; f0[::2, :, 1] = 5 * rows; f1[1::2, :, 1] = -7 * rows   # y clamps, x wraps
    got = ctx.blend(L, R, f0, f1, blend)
    ref = orc.combine_novel_views(L, R, f0, f1, blend)
    assert np.array_equal(got, ref), "%d bytes differ" % int((got != ref).sum())


@pytest.mark.parametrize("cols,rows", [(150, 120), (120, 260)])
def test_countblend_step_zero_is_defined(ctx, orc, synth, cols, rows):
    """min(cols, rows) < 200 makes the reference's search stride 0 (an endless loop, StitchTool.cpp:153-158) and
    rows < 130 a zero-size blur kernel (an OpenCV assertion): defined as stride 1 / 'skip the blur' on both sides."""
    L, R = synth.make_canvas_pair(cols, rows, 4)
    L = L.numpy(); R = R.numpy()
    mp, ovl, ovr, bl, md = ctx.stitch_prepare(L, R)
    rmp, rovl, rovr, rbl, rmd = orc.stitch_prepare(L, R, True)
    assert (mp == 150).any()
    assert np.array_equal(mp, rmp) and np.array_equal(bl, rbl) and np.array_equal(md, rmd)


def test_stitch_prefetch_is_one_shot_and_result_neutral(pf, synth):
    """pf_stitch_prefetch is a pure optimisation with one-shot records: a prefetched copy is consumed by the very next step or
    dropped -- a later step that happens to pass the same host address (allocator reuse) gets a fresh upload, and the library
    never reads a hint's host buffer after the step it was announced for."""
    cols, rows = 480, 320
    top, imgs = synth.make_stitch_set(cols, rows, 91, 4)
    top = top.numpy(); imgs = [im.numpy() for im in imgs]

    def chain(c, seq, prefetch):
        outs = []
        for i, im in enumerate(seq):
            if prefetch and i + 1 < len(seq):
                c.stitch_prefetch(seq[i + 1])
            outs.append(c.stitch_step(im, top if i == 0 else None, 20, want_out=True))
        return outs

    c = pf.Context(0)
    ref = chain(c, imgs, False)
    got = chain(c, imgs, True)                     # every step after the first consumes a prefetched copy
    assert all(np.array_equal(a, b) for a, b in zip(ref, got))
    # stale-address scenario: announce `buf`, let the next step upload it, then skip it, then reuse the address for another image
    ref2 = chain(c, [imgs[0], imgs[2], imgs[3]], False)
    buf = imgs[1].copy()
    c.stitch_prefetch(buf)
    o0 = c.stitch_step(imgs[0], top, 20)           # uploads buf (= image 1) behind its own kernels
    o1 = c.stitch_step(imgs[2], None, 20)          # not the announced image: the prefetched copy is dropped
    buf[...] = imgs[3]                             # same address, new content
    o2 = c.stitch_step(buf, None, 20)              # must NOT match the stale record
    assert np.array_equal(o0, ref2[0]) and np.array_equal(o1, ref2[1]) and np.array_equal(o2, ref2[2])
    # the allocator-reuse case of the round-3 advice: the announced buffer is overwritten with ANOTHER image (same address, size and
    # step) between the step that uploaded it and the step that passes it -- the content signature taken at upload time no longer
    # matches, so the step uploads what the buffer holds now instead of using the stale device copy
    ref3 = chain(c, [imgs[0], imgs[3]], False)
    buf2 = imgs[1].copy()
    c.stitch_prefetch(buf2)
    p0 = c.stitch_step(imgs[0], top, 20)           # uploads buf2 (= image 1) behind its own kernels
    buf2[...] = imgs[3]                            # "freed and re-read": same address, another image
    p1 = c.stitch_step(buf2, None, 20)
    assert np.array_equal(p0, ref3[0]) and np.array_equal(p1, ref3[1])
    c.close()


def test_device_checksum(pf):
    import torch
    c = pf.Context(0)
    a = torch.randint(0, 256, ((1 << 20) + 13,), dtype=torch.uint8, device="cuda")
    b = a.clone()
    torch.cuda.synchronize()
    n = a.numel()
    h = c.checksum_dev(a.data_ptr(), n)
    assert h == c.checksum_dev(b.data_ptr(), n) and h != 0
    b[n - 1] ^= 1                                  # the last (odd-tail) byte
    b[12345] ^= 4
    torch.cuda.synchronize()
    assert c.checksum_dev(b.data_ptr(), n) != h
    b[12345] ^= 4
    torch.cuda.synchronize()
    assert c.checksum_dev(b.data_ptr(), n) != h and c.checksum_dev(b.data_ptr(), n - 1) == c.checksum_dev(a.data_ptr(), n - 1)
    # position-sensitive: swapping two different words changes it
    w = a.clone()
    if not torch.equal(w[0:8], w[8:16]):
        t = w[0:8].clone(); w[0:8] = w[8:16]; w[8:16] = t
        torch.cuda.synchronize()
        assert c.checksum_dev(w.data_ptr(), n) != h
    c.close()


def test_generate_blend_reads_the_current_map(ctx, orc, synth):
    """Stitchtools::GenerateBlend (StitchTool.cpp:98-146) works on the public Map member: pf_stitch_match / pf_stitch_generate_blend are
    the two halves of prepare(), and the ramp of an EDITED map equals the oracle's ramp for images that have exactly that map."""
    cols, rows = 420, 300
    L, R = synth.make_canvas_pair(cols, rows, 5)
    L = L.numpy(); R = R.numpy()
    mp, ovl, ovr = ctx.stitch_match(L, R)
    rmp, rovl, rovr, rbl, rmd = orc.stitch_prepare(L, R, True)
    assert np.array_equal(mp, rmp) and np.array_equal(ovl, rovl) and np.array_equal(ovr, rovr)
    bl, md = ctx.stitch_generate_blend(mp)
    assert np.array_equal(bl, rbl) and np.array_equal(md, rmd)
    # edit the map: carve a block of the overlap out for the left image, and drop a stripe entirely
    ed = mp.copy()
    ys, xs = np.nonzero(mp == 150)
    y0, x0 = int(ys.mean()), int(xs.mean())
    ed[y0 - 20:y0 + 20, x0 - 15:x0 + 15] = 100
    ed[5:9, :] = 0
    L2 = L.copy(); R2 = R.copy()
    L2[..., 3] = np.where(ed >= 100, 255, 0); R2[..., 3] = np.where((ed == 50) | (ed == 150), 255, 0)
    emp, _, _, ebl, emd = orc.stitch_prepare(L2, R2, True)
    assert np.array_equal(emp, ed)
    bl2, md2 = ctx.stitch_generate_blend(ed)
    assert np.array_equal(bl2, ebl) and np.array_equal(md2, emd) and not np.array_equal(bl2, bl)


def test_missing_hardware_queues_raise_a_warning():
    """The throughput mode needs GPU_MAX_HW_QUEUES >= 3 x lanes + 2 in the environment before the first HIP call (include/panoflow.h);
    a caller that forgets gets correct results from serialised streams.  That must not be silent (round-4 review, weak #7): a fresh
    process WITHOUT the variable runs a batch (32 in flight = two lanes = 8 streams > the runtime's default 4 queues), gets the same
    strips as one-at-a-time calls, and finds the condition in pf_last_warning / pf_warning_count -- raised ONCE per condition, however
    many calls meet it, and never as an entry of the kernel profile list; the same process with the variable set high enough raises nothing."""
    import os, subprocess, sys, textwrap
    code = textwrap.dedent('''
        import importlib.util, os, sys, numpy as np, torch
        def load_pkg_module(name):   # (not through tests/conftest.py: importing it sets GPU_MAX_HW_QUEUES)
            spec = importlib.util.spec_from_file_location("pano_amd_" + name, os.path.join(%r, "panorama-opticalflow_amd", name + ".py"))
            mod = importlib.util.module_from_spec(spec); sys.modules["pano_amd_" + name] = mod; spec.loader.exec_module(mod)
            return mod
        pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
        dev = torch.device("cuda", 0)
        cols, rows, n = 96, 80, 18
        pairs = [synth.make_pair(cols, rows, 40 + i, dev)[:3] for i in range(n)]
        outs = [torch.zeros((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(n)]
        c = pf.Context(0)
        assert c.last_warning() == ("", 0)
        c.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0, [p[2].data_ptr() for p in pairs],
                               [o.data_ptr() for o in outs], None, None, in_flight=32)
        one = torch.zeros_like(outs[0])
        c.novel_view_dev(pairs[7][0].data_ptr(), pairs[7][1].data_ptr(), cols, rows, 0, pairs[7][2].data_ptr(), one.data_ptr())
        assert torch.equal(one, outs[7])
        c.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0, [p[2].data_ptr() for p in pairs],
                               [o.data_ptr() for o in outs], None, None, in_flight=32)   # the same condition again: no second warning
        msg, cnt = c.last_warning()
        print("WARN", cnt, msg)
        print("PROF", c.profile().get("warnings"))
    ''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    def run(env_queues):
        env = dict(os.environ); env.pop("GPU_MAX_HW_QUEUES", None)
        if env_queues: env["GPU_MAX_HW_QUEUES"] = env_queues
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        return r.stdout
    out = run(None)
    warn = [l for l in out.splitlines() if l.startswith("WARN")][0]
    assert warn.split()[1] == "1" and "GPU_MAX_HW_QUEUES" in warn and "8 HIP streams" in warn and ">= 8" in warn, warn
    assert "PROF None" in out, out
    out = run("16")
    assert [l for l in out.splitlines() if l.startswith("WARN")][0].split()[1] == "0", out


_POISON_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, %(root)r + "/oracle")
from tests.conftest import load_pkg_module
import orc
orc.build()
pf = load_pkg_module("pyabi")
bad = 0
for form in (0, 2):
    ctx = pf.Context(0, exp=True, sweep_wide=form)
    for (w, h) in ((260, 150), (150, 260)):
        r = np.random.default_rng(77 + w + form)
        img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 3, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
        g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
        yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
        flow = np.stack([9.0 * np.sin(xx / 40.0) + 0.5 * r.standard_normal((h, w)), 6.0 * np.cos(yy / 30.0) + 0.5 * r.standard_normal((h, w))], -1).astype(np.float32)
        a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32)
        # pixels that are NOT updated, next to pixels that are, carrying flows no window holds: huge, infinite, NaN
        a0[h // 4: h // 4 + 11, w // 5: w // 2] = 0.3
        a1[2 * h // 3:, ::3] = 0.0
        off = (a0 <= 0.9) | (a1 <= 0.9)
        wild = np.array([1.0e30, -3.0e38, np.inf, -np.inf, np.nan, 250.0, -777.0, 0.0], np.float32)
        idx = np.argwhere(off)
        flow[idx[:, 0], idx[:, 1], 0] = wild[(idx[:, 0] + 3 * idx[:, 1]) %% 8]
        flow[idx[:, 0], idx[:, 1], 1] = wild[(5 * idx[:, 0] + idx[:, 1]) %% 8]
        blurred = orc.gaussian_blur(np.nan_to_num(flow, nan=0.0, posinf=0.0, neginf=0.0).clip(-50, 50), 15, 8.0)
        for fwd in (1, 0):
            ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, fwd)
            got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, fwd)
            same = (got.view(np.uint32) == ref.view(np.uint32)) | ((got == 0) & (ref == 0))
            n = int((~same).sum())
            print("form", form, w, h, "forward", fwd, "differing", n, "updated pixels changed", int(((ref != flow) & ~np.isnan(ref)).any(-1).sum()))
            bad += n
    ctx.close()
print("POISON_RESULT", bad)
"""


def test_sweep_garbage_in_the_gather_window_never_reaches_a_result():
    """Pixels that are not updated evaluate their (discarded) proposals on whatever their LDS window slot holds -- with the flow-following window
    possibly a slot no loader wrote since the kernel started -- and pixels that are updated must only ever read texels their loader brought.
    The lab build with PANOFLOW_POISON_LDS=1 fills every sweep workgroup's gather windows with NaN / inf / +-1e38 first; not-updated pixels
    carry huge / infinite / NaN flows next to updated ones.  Both product sweep forms (latency, throughput; the lab build compiles the same
    templates) must still equal the oracle bit for bit: garbage can neither beat the kept energy nor trigger a redo that changes a result."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PANOFLOW_POISON_LDS="1")
    r = subprocess.run([sys.executable, "-c", _POISON_SCRIPT % {"root": root}], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    print(r.stdout[-3000:]); print(r.stderr[-2000:])
    assert r.returncode == 0, r.stderr[-2000:]
    assert "POISON_RESULT 0" in r.stdout
