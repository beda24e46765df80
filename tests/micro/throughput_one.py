"""GPU box: Mpix/s of pf_novel_view_batch_dev on 2000x4000 strips for ONE lane count (environment decides queues / CU partitions):
throughput_one.py <in_flight> [cols rows]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
infl = int(sys.argv[1])
cols, rows = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2000, 4000)
dev = torch.device("cuda", 0)
nb = int(os.environ.get("TP_PAIRS", "24"))
pairs = [synth.make_pair(cols, rows, 7000 + i, dev) for i in range(nb)]
outs = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
torch.cuda.synchronize()
bp = int(os.environ.get("TP_BATCH", "-1"))
knobs = {"batch_pairs": bp}
if os.environ.get("TP_STAGGER"): knobs["stagger_levels"] = int(os.environ["TP_STAGGER"])
if os.environ.get("TP_FUSE"): knobs["fuse_small_level_px"] = int(os.environ["TP_FUSE"])
if os.environ.get("TP_WIDE"): knobs["sweep_wide"] = int(os.environ["TP_WIDE"])
if os.environ.get("TP_WIDE_THR"): knobs["sweep_wide_threshold"] = int(os.environ["TP_WIDE_THR"])
if os.environ.get("TP_WIDE_TR"): knobs["sweep_throughput_transposed"] = int(os.environ["TP_WIDE_TR"])
if os.environ.get("TP_GRAD_FULL"): knobs["full_width_batch_gradients"] = int(os.environ["TP_GRAD_FULL"])
c = pf.Context(0, cols, rows, exp=knobs.get("sweep_wide") == 1, **knobs)   # form 1 lives in the lab build only
call = lambda: c.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0,
                                      [p[2].data_ptr() for p in pairs], [o.data_ptr() for o in outs], None, None, in_flight=infl)
call()
best = 1e9
for _ in range(int(os.environ.get("TP_LOOPS", "2"))):
    t = time.perf_counter(); call(); best = min(best, time.perf_counter() - t)
ref = pf.Context(0, cols, rows) if os.environ.get("TP_CHECK") else None
print("queues %s knobs %s in_flight %2d: %.1f Mpix/s (%.2f ms per pair)" % (os.environ.get("GPU_MAX_HW_QUEUES"), knobs, infl,
                                                                             nb * cols * rows / 1e6 / best, 1000 * best / nb), flush=True)
