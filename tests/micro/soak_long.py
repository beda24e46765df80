"""Long soak (diagnostic, not collected): the same inputs N times through pf_novel_view_dev / the batch entry point; every result must
hash to the first one.  soak_long.py [n_strip] [n_canvas]"""
import hashlib, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
n_strip = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n_canvas = int(sys.argv[2]) if len(sys.argv) > 2 else 30
dev = torch.device("cuda", 0)
def run(cols, rows, n, in_flight):
    L, R, B, _ = synth.make_pair(cols, rows, 4242, dev)
    torch.cuda.synchronize()   # the library's streams do not wait for torch's: inputs must be complete before the first call
    c = pf.Context(0, cols, rows)
    nb = max(in_flight, 1) * 2
    outs = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
    f0 = [torch.empty((rows, cols, 2), dtype=torch.float32, device=dev) for _ in range(nb)]
    f1 = [torch.empty((rows, cols, 2), dtype=torch.float32, device=dev) for _ in range(nb)]
    ref = None; bad = 0; t = time.perf_counter()
    for it in range(n // nb):
        c.novel_view_batch_dev([L.data_ptr()] * nb, [R.data_ptr()] * nb, cols, rows, 0, [B.data_ptr()] * nb, [o.data_ptr() for o in outs],
                               [f.data_ptr() for f in f0], [f.data_ptr() for f in f1], in_flight=in_flight)
        for k in range(nb):
            h = hashlib.sha256(outs[k].cpu().numpy().tobytes() + f0[k].cpu().numpy().tobytes() + f1[k].cpu().numpy().tobytes()).hexdigest()
            if ref is None: ref = h
            bad += h != ref
    print("%dx%d, %d in flight: %d solves, %d differ from the first (%.1f s)" % (cols, rows, in_flight, (n // nb) * nb, bad, time.perf_counter() - t), flush=True)
    c.close()
    return bad
bad = run(2000, 4000, n_strip, 1) + run(2000, 4000, n_strip, 4) + run(9000, 4000, n_canvas, 1) + run(9000, 4000, n_canvas, 2)
sys.exit(1 if bad else 0)
