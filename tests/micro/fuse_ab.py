"""GPU box: lone pair (9000x4000 dense, 2000x4000 strip) against pf_config::fuse_small_level_px (levels up to this many pixels fold the upsample
into the first Gaussian and the second median into the diffusion Gaussian: two launches fewer per level, longer launches)."""
import os, sys, time, statistics
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
for (cc, cr) in ((9000, 4000), (2000, 4000)):
    L, R, b, _ = synth.make_pair(cc, cr, 1234, dev); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
    for rep in range(2):
        for k in (-1, 0, 3000, 10000, 30000, 100000, 300000, 1000000):
            c = pf.Context(0, cc, cr, fuse_small_level_px=k)
            def one():
                t = time.perf_counter(); c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr()); return 1000 * (time.perf_counter() - t)
            one(); one()
            print("%dx%d fuse_small_level_px %8d: %.3f ms" % (cc, cr, k, statistics.median([one() for _ in range(9)])), flush=True)
            c.close()
