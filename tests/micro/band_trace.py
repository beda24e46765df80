"""Diagnostic (not a test): per-band progress of ONE latency-form sweep launch -- when each band of 8 rows passed every 512th step.
Needs var_libs/lib_trace.so (kernels_sweep2.hip built with -DPF_SWEEP_STATS -DPF_SWEEP_TRACE) copied over libpanoflow.so by the caller
(tests/micro/band_trace.sh).  Usage: band_trace.py WxH [out.npy]"""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi")
ctx = pf.Context(0, sweep_wide=0)
ctx.profile_enable(True)
w, h = [int(v) for v in sys.argv[1].split("x")]
r = np.random.default_rng(0)
g0 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
g1 = r.standard_normal((h, w, 2)).astype(np.float32) * 0.1
flow = r.standard_normal((h, w, 2)).astype(np.float32)
bl = flow * 0.9
a = np.ones((h, w), np.float32)
best = 1e9
for rep in range(4):
    ctx.profile_reset()
    ctx.stage_sweep(g0, g1, bl, a, a, flow, 1)
    best = min(best, ctx.profile()["sweep"][0])
lib = pf.lib()
nb = 1024
out = np.zeros((nb, 40), np.int64)
rc = lib.pf_debug_band_trace(out.ctypes.data_as(ctypes.c_void_p), nb)
assert rc == 0
steps = w + h - 1
print("W=%d H=%d sweep(prep+main) %.3f ms = %.4f us/step (instrumented build)" % (w, h, best, 1000 * best / steps))
nbands = (h + 7) // 8   # (wider than tall: bands of 8 rows)
t = out[:nbands].astype(np.float64) / 100.0   # us
t0 = t[0, 4]
ncp = int(np.count_nonzero(out[0, 4:]))
print("bands %d, checkpoints %d (every 512 steps)" % (nbands, ncp))
print("band  edgeW spins | lag to the band before at each checkpoint (us) ... | at the end | own us/step between checkpoints 1 and last")
for b in range(nbands):
    if not (b < 12 or b % 16 < 5 or b >= nbands - 2): continue
    cps = t[b, 4:4 + ncp]
    lag = (cps - t[b - 1, 4:4 + ncp]) if b else cps - t0
    endlag = t[b, 3] - (t[b - 1, 3] if b else t0)
    rate = (cps[-1] - cps[1]) / (512.0 * (ncp - 2)) if ncp > 2 else 0   # (checkpoint 0 is the band's first chunk START, before its first wait for the band above)
    print("%4d %6d %6d | %s | %8.2f | %.4f" % (b, out[b, 0], out[b, 1], " ".join("%7.2f" % v for v in lag), endlag, rate))
np.save(sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/band_trace_%dx%d.npy" % (w, h), out[:nbands])
