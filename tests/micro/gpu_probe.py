"""Diagnostic (not a test): prints per-stage GPU-vs-oracle deltas and timings on the GPU box."""
import sys, time, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
import orc
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
orc.build()
ctx = pf.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
L, R, blend = synth.make_pair_np(n, n, 1234)
t = time.time(); rLR, rRL = orc.flow_bidir(L, R, 0); tc = time.time() - t
ctx.novel_view(L, R, 0, blend)
ctx.profile_enable(True)
t = time.time(); out, fLR, fRL = ctx.novel_view(L, R, 0, blend); tg = time.time() - t
print("cpu %.3fs gpu(host buffers) %.3fs" % (tc, tg))
print("max|dflow|", np.abs(fLR - rLR).max(), np.abs(fRL - rRL).max(), "mismatch", (fLR != rLR).sum(), (fRL != rRL).sum())
for k, (ms, cnt) in sorted(ctx.profile().items(), key=lambda kv: -kv[1][0]):
    print("%-24s %9.3f ms %5d launches %8.1f us/launch" % (k, ms, cnt, 1000 * ms / max(cnt, 1)))
