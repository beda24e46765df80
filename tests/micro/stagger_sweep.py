"""Diagnostic (not a test): the lone dense pair against pf_config::stagger_levels (how many coarse levels direction 1 runs behind direction 0).
Usage: stagger_sweep.py [cols rows] -> ms per pair (median of 9) per setting; -1 = the library's own choice."""
import sys, os, time, statistics, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
cc, cr = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9000, 4000)
L, R, b, _ = synth.make_pair(cc, cr, 1234, dev); o = torch.empty((cr, cc, 4), dtype=torch.uint8, device=dev); torch.cuda.synchronize()
for rep in range(2):
    for st in (-1, 0, 1, 2, 3, 4, 5, 6, 8):
        c = pf.Context(0, cc, cr, stagger_levels=st)
        ts = []
        for i in range(11):
            torch.cuda.synchronize(); t = time.perf_counter()
            c.novel_view_dev(L.data_ptr(), R.data_ptr(), cc, cr, 0, b.data_ptr(), o.data_ptr()); torch.cuda.synchronize()
            ts.append(1000 * (time.perf_counter() - t))
        print("%dx%d stagger_levels %2d: %.3f ms (min %.3f)" % (cc, cr, st, statistics.median(ts[2:]), min(ts[2:])), flush=True)
        c.close()
