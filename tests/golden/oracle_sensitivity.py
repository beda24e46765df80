#!/usr/bin/env python
"""FP-sensitivity of the oracle (SURVEY.md section 7 / BASELINE.md section 4; round-4 review, next #4).  CPU only.

The oracle (oracle/pixflow_oracle.cpp) is a RESTATEMENT of the reference + the OpenCV-3.2 primitives it calls; the reference itself
cannot be built in this image, so parity is unpinned.  This script measures how far a real OpenCV / another compiler COULD land from the
restatement, by rebuilding the oracle with one plausible floating-point difference at a time and comparing end to end:
  fma        g++ -mfma -ffp-contract=fast           (a build that contracts a*b+c; the reference's own build line does not)
  gauss_symm -DORC_VAR_GAUSS_SYMM                   (15-tap Gaussian row pass as centre + symmetric pairs instead of left-to-right)
  box_float  -DORC_VAR_BOX_FLOAT                    (box-filter sliding sums in float instead of double: blend ramp only)
  libm_ulp   -DORC_VAR_LIBM_ULP                     (exp / tanhf one ulp up: novel-view blend only)
  all        every one of the above together
on the 512x512 plumbing pair and the 2000x4000 strip (seed 1234, both presets at 512^2), and -- for the box filter -- on a synthetic
stitch set.  Per variant: max |dflow| (px), the 99.9th percentile, the fraction of pixels whose flow differs at all and by more than
1e-2 px, blended bytes that differ, PSNR of the blended strip.  Usage: python tests/golden/oracle_sensitivity.py [--quick] > table
"""
import ctypes as C, json, os, subprocess, sys, tempfile, threading, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg_module
synth = load_pkg_module("synth")
SRC = os.path.join(ROOT, "oracle", "pixflow_oracle.cpp")
BASE = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-function"]
VARIANTS = [("baseline", ["-ffp-contract=off"]),
            ("fma", ["-mfma", "-ffp-contract=fast"]),
            ("gauss_symm", ["-ffp-contract=off", "-DORC_VAR_GAUSS_SYMM"]),
            ("box_float", ["-ffp-contract=off", "-DORC_VAR_BOX_FLOAT"]),
            ("libm_ulp", ["-ffp-contract=off", "-DORC_VAR_LIBM_ULP"]),
            ("all", ["-mfma", "-ffp-contract=fast", "-DORC_VAR_GAUSS_SYMM", "-DORC_VAR_BOX_FLOAT", "-DORC_VAR_LIBM_ULP"])]
p = lambda a: a.ctypes.data_as(C.c_void_p)


def build(tmp):
    libs = {}
    for name, fl in VARIANTS:
        so = os.path.join(tmp, "orc_%s.so" % name)
        subprocess.check_call(BASE + fl + ["-o", so, SRC])
        libs[name] = C.CDLL(so)
    return libs


def solve(lib, L, R, blend, max_pct):
    rows, cols, _ = L.shape
    fl = [np.empty((rows, cols, 2), np.float32) for _ in (0, 1)]
    th = [threading.Thread(target=lambda d=d: lib.orc_flow_one_dir(p(L), p(R), cols, rows, max_pct, d, p(fl[d]))) for d in (0, 1)]
    [t.start() for t in th]; [t.join() for t in th]
    out = np.empty((rows, cols, 4), np.uint8)
    lib.orc_combine_novel_views(p(L), p(R), p(fl[0]), p(fl[1]), p(np.ascontiguousarray(blend, np.float32)), cols, rows, p(out))
    return fl[0], fl[1], out


def epe_vs_analytic(f_l2r, cols, rows):
    """median / 99th-percentile end-point error of flow L->R against the synthetic scene's analytic displacement (a point of L at x shows T(x + d/2), R shows it at x + d: flow L->R = +d), valid-alpha region"""
    import torch
    ys = torch.arange(rows, dtype=torch.float64)[:, None].expand(rows, cols); xs = torch.arange(cols, dtype=torch.float64)[None, :].expand(rows, cols)
    dx, dy = synth.displacement(xs, ys, cols, rows, 1.0)
    a = synth.alpha_mask(xs, ys, cols, rows).numpy()
    e = np.sqrt((f_l2r[..., 0] - dx.numpy()) ** 2 + (f_l2r[..., 1] - dy.numpy()) ** 2)[a]
    return float(np.median(e)), float(np.quantile(e, 0.99))


def compare(ref, got):
    d = np.sqrt(((np.concatenate([ref[0], ref[1]]).astype(np.float64) - np.concatenate([got[0], got[1]])) ** 2).sum(-1))
    comp = np.abs(np.concatenate([ref[0], ref[1]]) - np.concatenate([got[0], got[1]]))
    nb = int((ref[2] != got[2]).sum())
    mse = float(((ref[2][..., :3].astype(np.float64) - got[2][..., :3]) ** 2).mean())
    rows, cols, _ = got[0].shape
    em, e99 = epe_vs_analytic(got[0], cols, rows)
    return {"epe_vs_analytic_median_px": em, "epe_vs_analytic_p99_px": e99, "median_dflow_px": float(np.median(d)), "mean_dflow_px": float(d.mean()),
            "max_abs_dflow_px": float(comp.max()), "p999_dflow_px": float(np.quantile(d, 0.999)), "frac_px_flow_differs": float((d > 0).mean()),
            "frac_px_dflow_gt_1e-2": float((comp.max(-1) > 1e-2).mean()), "blend_bytes_off": nb, "blend_bytes_total": int(ref[2].size),
            "blend_max_byte_delta": int(np.abs(ref[2].astype(np.int16) - got[2]).max()), "blend_psnr_db": (float("inf") if mse == 0 else float(10 * np.log10(255.0 ** 2 / mse)))}


def main():
    quick = "--quick" in sys.argv
    cases = [("512x512 pixflow_low", 512, 512, 0), ("512x512 pixflow_search_20", 512, 512, 20)]
    if not quick:
        cases.append(("2000x4000 pixflow_low", 2000, 4000, 0))
    res = {"variants": {n: " ".join(f) for n, f in VARIANTS}, "cases": {}}
    with tempfile.TemporaryDirectory() as tmp:
        libs = build(tmp)
        for title, cols, rows, mp in cases:
            L, R, blend, _ = synth.make_pair(cols, rows, 1234, "cpu")
            L, R, blend = L.numpy(), R.numpy(), blend.numpy()
            t0 = time.time(); ref = solve(libs["baseline"], L, R, blend, mp)
            sys.stderr.write("%s baseline %.1f s\n" % (title, time.time() - t0))
            em, e99 = epe_vs_analytic(ref[0], cols, rows)
            res["cases"][title] = {"baseline": {"epe_vs_analytic_median_px": em, "epe_vs_analytic_p99_px": e99}}
            for name, _ in VARIANTS[1:]:
                got = solve(libs[name], L, R, blend, mp)
                res["cases"][title][name] = compare(ref, got)
                sys.stderr.write("  %s %s\n" % (name, json.dumps(res["cases"][title][name])))
        # the box filter only enters through Stitchtools::GenerateBlend's ramp smoothing (CPU/StitchTool.cpp:130-143)
        cc, cr = (1800, 800) if quick else (3600, 1600)
        top, imgs = synth.make_stitch_set(cc, cr, 1234, 2, "cpu")
        A, B = imgs[0].numpy(), imgs[1].numpy()
        def ramp(lib):
            mp_ = np.empty((cr, cc), np.uint8); ovL = np.empty_like(A); ovR = np.empty_like(A); bl = np.empty((cr, cc), np.float32); md = np.empty((cr, cc), np.float32)
            lib.orc_stitch_prepare(p(A), p(B), cc, cr, 1, p(mp_), p(ovL), p(ovR), p(bl), p(md))
            return bl
        r0 = ramp(libs["baseline"])
        res["cases"]["stitch ramp %dx%d (Stitchtools::prepare blend)" % (cc, cr)] = {
            n: {"max_abs_dblend": float(np.abs(r0 - ramp(libs[n])).max()), "frac_px_differs": float((r0 != ramp(libs[n])).mean())} for n in ("fma", "box_float", "all")}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
