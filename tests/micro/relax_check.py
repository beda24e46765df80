"""PANOFLOW_SWEEP=3 (event-driven relaxation sweep, kernels_relax.inl) against the oracle and against the default kernel: parity, then time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "oracle"))
from conftest import load_pkg_module
import orc
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
orc.build()
os.environ["PANOFLOW_SWEEP"] = "3"; c3 = pf.Context(0)
os.environ["PANOFLOW_SWEEP"] = "2"; c2 = pf.Context(0)
bad = 0
for (w, h, fwd) in [(40, 30, 1), (48, 48, 1), (49, 50, 0), (150, 131, 1), (64, 257, 0), (300, 90, 1), (500, 400, 0), (500, 400, 1)]:
    r = np.random.default_rng(7 + w + fwd)
    img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 2, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
    g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
    flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
    bl = orc.gaussian_blur(flow, 15, 8.0)
    a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32); a0[h // 3: h // 3 + 9, w // 4: w // 2] = 0.5; a1[:, :3] = 0.0
    ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], bl, a0, a1, flow, fwd)
    try:
        got = c3.stage_sweep(g0, g1, bl, a0, a1, flow, fwd)
    except Exception as e:
        print("stage %dx%d fwd=%d: ERROR %s" % (w, h, fwd, e)); bad += 1; continue
    nm = int((got.view(np.uint32) != ref.view(np.uint32)).any(-1).sum())
    print("stage %dx%d fwd=%d: %d mismatching pixels" % (w, h, fwd, nm), flush=True)
    bad += nm != 0
L, R, blend = synth.make_pair_np(320, 256, 4321)
f0, f1 = c3.flow_bidir(L, R, 20); r0, r1 = c2.flow_bidir(L, R, 20)
print("bidir 320x256 identical:", np.array_equal(f0, r0) and np.array_equal(f1, r1), flush=True)
if bad or os.environ.get("RX_QUICK"): sys.exit(1 if bad else 0)
L, R, blend = synth.make_pair_np(2000, 4000, 1234)
for name, c in (("wavefront", c2), ("relax", c3)):
    fa = c.flow_bidir(L, R, 0)
    ts = []
    for _ in range(5):
        t = time.perf_counter(); fb = c.flow_bidir(L, R, 0); ts.append(time.perf_counter() - t)
    print("%s 2000x4000 bidir: min %.2f ms median %.2f ms (host buffers in/out)" % (name, min(ts) * 1e3, sorted(ts)[2] * 1e3), flush=True)
    if name == "wavefront": ref = fa
    else: print("strip identical:", np.array_equal(fa[0], ref[0]) and np.array_equal(fa[1], ref[1]))
