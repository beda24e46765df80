"""Determinism check (diagnostic): N solves of the same pair, hash of each result plane.  det_check.py cols rows n"""
import hashlib, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
cols, rows, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
L, R, B, _ = synth.make_pair(cols, rows, 4242, dev)
torch.cuda.synchronize()   # the library's streams do not wait for torch's: inputs must be complete before the first call
c = pf.Context(0, cols, rows)
out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev)
f0 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev); f1 = torch.empty_like(f0)
hs = []
for it in range(n):
    c.novel_view_dev(L.data_ptr(), R.data_ptr(), cols, rows, 0, B.data_ptr(), out.data_ptr(), f0.data_ptr(), f1.data_ptr())
    h = tuple(hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()[:8] for t in (out, f0, f1))
    hs.append(h)
    if it == 0: ref = [t.clone() for t in (out, f0, f1)]
    else:
        d0 = int((f0 != ref[1]).any(-1).sum()); d1 = int((f1 != ref[2]).any(-1).sum())
        if d0 or d1:
            ys, xs = torch.nonzero((f0 != ref[1]).any(-1), as_tuple=True)
            print("   run %d: f0 differs at %d px (bbox x %s..%s y %s..%s), f1 at %d px" % (it, d0, xs.min().item() if d0 else -1, xs.max().item() if d0 else -1, ys.min().item() if d0 else -1, ys.max().item() if d0 else -1, d1))
print(os.environ.get("TAG", ""), "distinct results:", len(set(hs)), "of", n, hs[:3])
