import sys, os, numpy as np
sys.path.insert(0, os.path.dirname(__file__))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi")
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(__file__)), "oracle")); import orc
ctx = pf.Context(0)
w, h, forward = 90, 70, 1
r = np.random.default_rng(100 + w + h + forward)
img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 2, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
blurred = orc.gaussian_blur(flow, 15, 8.0)
for cut in (0, 1, 2, 3, 9):
    a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32)
    a1[:, :cut] = 0.0
    ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, forward)
    got = ctx.stage_sweep(g0, g1, blurred, a0, a1, flow, forward)
    bad = np.argwhere((got != ref).any(-1))
    print("cut", cut, "mismatching pixels", len(bad), "first", bad[:5].tolist(), "cols", sorted(set(bad[:, 1].tolist()))[:12])
