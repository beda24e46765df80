"""debug helper (GPU box): which pixels of a stage sweep differ from the oracle, per number of relaxation rounds"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
import orc
pf = load_pkg_module("pyabi")
for (w, h) in [(64, 257), (257, 64), (40, 96)]:
    for forward in (1, 0):
        r = np.random.default_rng(100 + w + h + forward)
        img0 = r.random((h, w)).astype(np.float32); img1 = np.roll(img0, 2, axis=1) + 0.05 * r.random((h, w)).astype(np.float32)
        g0 = np.stack(orc.gradients(img0), -1); g1 = np.stack(orc.gradients(img1), -1)
        flow = (r.standard_normal((h, w, 2)) * 1.5).astype(np.float32)
        blurred = orc.gaussian_blur(flow, 15, 8.0)
        a0 = np.ones((h, w), np.float32); a1 = np.ones((h, w), np.float32)
        a0[h // 3: h // 3 + 9, w // 4: w // 2] = 0.5
        a1[:, :3] = 0.0
        ref = orc.sweep(g0[..., 0], g0[..., 1], g1[..., 0], g1[..., 1], blurred, a0, a1, flow, forward)
        for K in (0, 1, 2, 16):
            os.environ["PANOFLOW_SPEC_ROUNDS"] = str(K)
            c = pf.Context(0)
            got = c.stage_sweep(g0, g1, blurred, a0, a1, flow, forward)
            c.close()
            bad = np.argwhere((got != ref).any(-1))
            print("w %d h %d fwd %d K %d: %d pixels differ%s" % (w, h, forward, K, len(bad), "" if len(bad) == 0 else "; first (y,x): %s, last %s, max|d| %g" % (bad[0], bad[-1], np.abs(got - ref).max())), flush=True)
