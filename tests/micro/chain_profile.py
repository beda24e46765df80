"""GPU box: where the config-4 chain's time goes (per step wall time, per-family HIP-event totals)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cc, cr = 9000, 4000
top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, torch.device("cuda", 0))
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
c = pf.Context(0, cc, cr)
def chain(prefetch):
    ts = []
    for i, im in enumerate(imgs):
        t = time.perf_counter()
        if prefetch: c.stitch_prefetch(None if i == 4 else imgs[i + 1])
        o = c.stitch_step(im, top if i == 0 else None, 20, want_out=(i == 4))
        ts.append(1000 * (time.perf_counter() - t))
    return ts
chain(True)
for pfm in (False, True, False, True):
    ts = chain(pfm); print("prefetch %d: steps ms %s  total %.1f ms  swept steps/dir (last) %d" % (pfm, ["%.1f" % t for t in ts], sum(ts), c.last_swept_steps()))
c.profile_reset(); c.profile_enable(1); chain(True); c.profile_enable(0)
for k, v in sorted(c.profile().items(), key=lambda kv: -kv[1][0]): print("   %-22s %8.2f ms  %d launches" % (k, v[0], v[1]))
# raw copy rates
n = cc * cr * 4
d = c.dev_alloc(n); t = time.perf_counter(); c.upload(d, imgs[0]); t1 = time.perf_counter() - t
buf = np.empty((cr, cc, 4), np.uint8); t = time.perf_counter(); c.download(buf, d); t2 = time.perf_counter() - t
print("pageable H2D 144 MB: %.1f ms, D2H: %.1f ms" % (1000 * t1, 1000 * t2))
