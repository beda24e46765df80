"""GPU box: config-4 chain time under the variations bench.py introduces (pinned result buffer, a context that solved a dense
pair before, other contexts alive)."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cc, cr = 9000, 4000
dev = torch.device("cuda", 0)
top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, dev)
top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
torch.cuda.empty_cache()

def chain(c, final):
    t1 = time.perf_counter()
    for i, im in enumerate(imgs):
        last = i == 4
        c.stitch_prefetch(None if last else imgs[i + 1])
        c.stitch_step(im, top if i == 0 else None, 20, want_out=last, out=final if last else None)
    return 1000 * (time.perf_counter() - t1)

def run(tag, c, final):
    chain(c, final)
    ts = sorted(chain(c, final) for _ in range(5))
    print("%-55s median %.1f ms  (min %.1f max %.1f)" % (tag, ts[2], ts[0], ts[-1]), flush=True)

c = pf.Context(0, cc, cr)
run("fresh context, pageable result buffer", c, np.zeros((cr, cc, 4), np.uint8))
run("fresh context, pinned result buffer", c, c.host_array((cr, cc, 4)))
L, R, blend, _ = synth.make_pair(cc, cr, 1234, dev)
out = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev)
for _ in range(3):
    c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, blend.data_ptr(), out.data_ptr())
run("same context after dense pairs, pageable", c, np.zeros((cr, cc, 4), np.uint8))
c2 = pf.Context(0, cc, cr)
run("second context (first one alive), pageable", c2, np.zeros((cr, cc, 4), np.uint8))
c.close()
run("second context (first closed), pageable", c2, np.zeros((cr, cc, 4), np.uint8))
