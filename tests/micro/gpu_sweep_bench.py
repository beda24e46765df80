"""Diagnostic (not a test): per-step time of the sweep kernel for several (W,H) shapes."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi")
_form = int(os.environ.get("SW_WIDE", "0"))
ctx = pf.Context(0, exp=_form == 1, sweep_wide=_form)   # form 1 lives in the lab build only
ctx.profile_enable(True)
r = np.random.default_rng(0)
shapes = [(4000, 8), (4000, 32), (4000, 64), (1100, 2000), (2000, 1100), (300, 500)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for (w, h) in shapes:
    g0 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
    g1 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
    flow = r.standard_normal((h, w, 2)).astype(np.float32)
    bl = flow * 0.9
    a = np.ones((h, w), np.float32)
    best = 1e9
    for rep in range(6):
        ctx.profile_reset()
        ctx.stage_sweep(g0, g1, bl, a, a, flow, 1)
        best = min(best, ctx.profile()["sweep"][0])
    ms = best
    steps = w + h - 1
    print("W=%5d H=%5d  sweep(prep+main) %8.3f ms  = %7.3f us/step (steps=%d)" % (w, h, ms, 1000 * ms / steps, steps), flush=True)
