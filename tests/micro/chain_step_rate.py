"""GPU box: effective sweep time per swept step on the config-4 chain (pixflow_search_20, sparse overlaps) vs the dense pair."""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cc, cr = 9000, 4000
top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, torch.device("cuda", 0))
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
c = pf.Context(0, cc, cr)
for rep in range(2):
    for i, im in enumerate(imgs):
        c.profile_reset(); c.profile_enable(1)
        c.stitch_step(im, top if i == 0 else None, 20, want_out=(i == 4))
        c.profile_enable(0)
        p = c.profile(); sw = p["sweep"][0]; n = c.last_swept_steps()
        if rep: print("chain step %d: sweep kernels %.2f ms (both directions), swept steps/direction %d -> %.3f us per step" % (i, sw, n, 1000 * sw / 2 / max(n, 1)))
L, R, b, _ = synth.make_pair(cc, cr, 1234, torch.device("cuda", 0)); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device="cuda"); torch.cuda.synchronize()
for rep in range(2):
    c.profile_reset(); c.profile_enable(1)
    c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr())
    c.profile_enable(0)
    p = c.profile(); sw = p["sweep"][0]; n = c.last_swept_steps()
    if rep: print("dense pair (pixflow_low): sweep kernels %.2f ms, swept steps/direction %d -> %.3f us per step" % (sw, n, 1000 * sw / 2 / max(n, 1)))
