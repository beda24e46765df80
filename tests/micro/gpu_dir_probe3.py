"""Diagnostic: sweep time of ONE direction alone (pf_flow on the wrap-padded pair = the geometry each direction of the
bidirectional solve has) vs per direction inside pf_flow_bidir (2000x4000 strip)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
L, R, blend = synth.make_pair_np(2000, 4000, 1234)
pad = 100
wrap = lambda im: np.ascontiguousarray(np.concatenate([im[:, 2000 - pad:], im, im[:, :pad]], axis=1))
Lp, Rp = wrap(L), wrap(R)
ctx = pf.Context(0)
ctx.profile_enable(2)
for name, fn in (("one direction alone (pf_flow on the padded pair)", lambda: ctx.flow(Lp, Rp, 0, 3)), ("both directions (pf_flow_bidir)", lambda: ctx.flow_bidir(L, R, 0))):
    best = None
    for rep in range(4):
        ctx.profile_reset(); t = time.perf_counter(); fn(); dt = time.perf_counter() - t
        ms, n = ctx.profile()["sweep"]
        best = (ms, n, dt) if best is None or ms < best[0] else best
    print("%-55s sweep launches %4d  total %8.2f ms  per direction %7.2f ms" % (name, best[1], best[0], best[0] / (best[1] / 74.0)), flush=True)
