"""Diagnostic (not a test): per-family kernel time of one flow direction alone vs both directions concurrently."""
import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
L, R, blend = synth.make_pair_np(2000, 4000, 1234)
ctx = pf.Context(0)
ctx.profile_enable(1)
for name, fn in (("one direction", lambda: ctx.flow(L, R, 0, 3)), ("both directions", lambda: ctx.flow_bidir(L, R, 0))):
    for rep in range(3):
        ctx.profile_reset(); fn()
    p = ctx.profile()
    print(name, {k: round(v[0], 2) for k, v in sorted(p.items(), key=lambda kv: -kv[1][0])}, flush=True)
