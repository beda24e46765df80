"""GPU box: compact timing of one library build -- dense 9000x4000 pair, 2000x4000 strip, config-4 chain, pf_stitch_prepare,
and the per-family HIP-event totals of one chain (the same-box A/B unit of tests/micro/ab3.sh)."""
import os, sys, time, statistics
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
tag = sys.argv[1] if len(sys.argv) > 1 else ""
fams = sys.argv[2].split(",") if len(sys.argv) > 2 else ["median5", "tile_blur", "box_blur", "countblend", "sweep", "adjust_initial_flow"]
def med(f, n=5):
    f(); return statistics.median([f() for _ in range(n)])
res = {}
for (cc, cr, key) in ((9000, 4000, "pair"), (2000, 4000, "strip")):
    c = pf.Context(0, cc, cr)
    L, R, b, _ = synth.make_pair(cc, cr, 1234, dev); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
    def one():
        t = time.perf_counter(); c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr()); return 1000 * (time.perf_counter() - t)
    res[key] = med(one, 7)
    if key == "pair":
        c.profile_reset(); c.profile_enable(1); one(); c.profile_enable(0); pp = c.profile()
    c.close(); del L, R, b, o
cc, cr = 9000, 4000
top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, dev)
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
torch.cuda.empty_cache()
c = pf.Context(0, cc, cr)
final = np.zeros((cr, cc, 4), np.uint8)
def chain():
    t = time.perf_counter()
    for i, im in enumerate(imgs):
        c.stitch_prefetch(None if i == 4 else imgs[i + 1])
        c.stitch_step(im, top if i == 0 else None, 20, want_out=(i == 4), out=final if i == 4 else None)
    return 1000 * (time.perf_counter() - t)
res["chain"] = med(chain, 5)
c.profile_reset(); c.profile_enable(1); chain(); c.profile_enable(0); pc = c.profile()
def prep():
    t = time.perf_counter(); c.stitch_prepare(imgs[1], imgs[0]); return 1000 * (time.perf_counter() - t)
res["prepare_host"] = med(prep, 3)
print("%-28s pair %.2f  strip %.2f  chain %.1f  prepare(host bufs) %.1f | pair: %s | chain: %s" % (
    tag, res["pair"], res["strip"], res["chain"], res["prepare_host"],
    " ".join("%s %.2f" % (k, pp[k][0]) for k in fams if k in pp), " ".join("%s %.2f" % (k, pc[k][0]) for k in fams if k in pc)), flush=True)
