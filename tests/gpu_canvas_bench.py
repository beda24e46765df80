"""Diagnostic (not a test): one stitch step on a 9000x4000 full-canvas pair (sparse overlap), timings."""
import sys, os, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(__file__))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9000, 4000)
top, imgs = synth.make_stitch_set(cols, rows, 1234, 5, "cuda")
ctx = pf.Context(0)
L = imgs[0].cpu().numpy(); R = top.cpu().numpy()
for rep in range(2):
    t0 = time.time(); mp, ovl, ovr, blend, md = ctx.stitch_prepare(L, R); t1 = time.time()
    ctx.profile_reset(); ctx.profile_enable(1)
    out, f0, f1 = ctx.novel_view(ovl, ovr, 20, blend, want_flows=False); t2 = time.time()
    ctx.profile_enable(0); pr = ctx.profile()
    fin = ctx.stitch_gather(L, R, out, mp); t3 = time.time()
    print("   sweep %.1f ms (both dirs summed), other kernels %.1f ms" % (pr["sweep"][0], sum(v[0] for k, v in pr.items() if k != "sweep")))
    print("prepare %.3f s  novel_view %.3f s  gather %.3f s  (host buffers, overlap frac %.3f)" % (t1 - t0, t2 - t1, t3 - t2, (mp == 150).mean()), flush=True)
