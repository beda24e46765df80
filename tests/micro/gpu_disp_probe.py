"""Diagnostic (not a test): sweep time vs displacement magnitude (how often proposals leave the +-7 texel LDS window)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
ctx = pf.Context(0)
ctx.profile_enable(2)
for scale in (1.0, 2.0, 3.0, 5.0):
    L, R, blend = synth.make_pair_np(2000, 4000, 1234, scale)
    best = None
    for rep in range(2):
        ctx.profile_reset(); f0, f1 = ctx.flow_bidir(L, R, 0)
        ms, n = ctx.profile()["sweep"]
        best = ms if best is None or ms < best else best
    print("disp_scale %.1f: max |flow| %.1f px (full res), sweeps %.2f ms per pair (both directions)" % (scale, float(np.abs(f0).max()), best), flush=True)
