"""Diagnostic (not a test): BASELINE config 4 -- the 5+top stitch chain at 9000x4000, pixflow_search_20, end to end
from host images to the final composite on the host (fused device-resident steps)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9000, 4000)
top, imgs = synth.make_stitch_set(cols, rows, 1234, 5, "cuda")
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
ctx = pf.Context(0)
for rep in range(2):
    t0 = time.time(); ts = []
    for i, L in enumerate(imgs):
        t1 = time.time()
        out = ctx.stitch_step(L, top if i == 0 else None, 20, want_out=(i == len(imgs) - 1))
        ts.append(time.time() - t1)
    print("5+top stitch %dx%d: total %.3f s, steps %s" % (cols, rows, time.time() - t0, " ".join("%.3f" % t for t in ts)), flush=True)
print("final alpha coverage %.3f" % (out[..., 3] > 0).mean())
if os.environ.get("PROFILE"):
    ctx.profile_enable(1); ctx.profile_reset()
    t1 = time.time(); ctx.stitch_step(imgs[2], None, 20, want_out=False); dt = time.time() - t1
    prof = ctx.profile()
    print("one profiled step %.3f s; kernel families (ms): %s" % (dt, ", ".join("%s %.2f/%d" % (k, v[0], v[1]) for k, v in sorted(prof.items(), key=lambda kv: -kv[1][0]))))
