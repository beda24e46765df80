"""GPU box: how much do the two directions of a lone dense pair cost each other?  One direction alone (pf_flow on the wrap-padded 9900x4000
images: the same 4950x2000 pyramid) against the bidirectional solve: per-family HIP-event totals (pf_profile_*)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
cc, cr = 9000, 4000; pad = cc // 20
L, R, b, _ = synth.make_pair(cc, cr, 1234, dev)
Lh = L.cpu().numpy(); Rh = R.cpu().numpy(); bh = b.cpu().numpy()
Lp = np.ascontiguousarray(np.concatenate([Lh[:, -pad:], Lh, Lh[:, :pad]], axis=1)); Rp = np.ascontiguousarray(np.concatenate([Rh[:, -pad:], Rh, Rh[:, :pad]], axis=1))
c = pf.Context(0, cc + 2 * pad, cr)
fams = ["sweep", "median5", "gauss15_blurredFlow", "gauss15_diffusion", "upsample_cubic"]
def show(tag, p, wall):
    print("%-28s wall %.1f ms | %s | swept steps %d" % (tag, wall, "  ".join("%s %.2f" % (k, p[k][0]) for k in fams if k in p), c.last_swept_steps()), flush=True)
for rep in range(2):
    c.flow(Lp, Rp, 0, 0)
    c.profile_reset(); c.profile_enable(1); t = time.perf_counter(); c.flow(Lp, Rp, 0, 0); w = 1000 * (time.perf_counter() - t); c.profile_enable(0)
    show("one direction (host bufs)", c.profile(), w)
    c.novel_view(Lh, Rh, 0, bh, want_flows=False)
    c.profile_reset(); c.profile_enable(1); t = time.perf_counter(); c.novel_view(Lh, Rh, 0, bh, want_flows=False); w = 1000 * (time.perf_counter() - t); c.profile_enable(0)
    show("both directions (host bufs)", c.profile(), w)
