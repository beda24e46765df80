"""Diagnostic (not a test): sweep time of one flow direction alone vs both directions concurrently (2000x4000 strip)."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
L, R, blend = synth.make_pair_np(2000, 4000, 1234)
ctx = pf.Context(0)
ctx.profile_enable(2)
for name, fn in (("one direction (pf_flow on the padded pair is what each stream runs)", lambda: ctx.flow(L, R, 0, 3)),
                 ("both directions (pf_flow_bidir)", lambda: ctx.flow_bidir(L, R, 0))):
    best = None
    for rep in range(3):
        ctx.profile_reset(); fn()
        ms, n = ctx.profile()["sweep"]
        best = (ms, n) if best is None or ms < best[0] else best
    print("%-70s sweep launches %4d  total %8.2f ms  avg %7.1f us" % (name, best[1], best[0], 1000 * best[0] / best[1]), flush=True)
