"""Soak (not a test): many repetitions of the strip solve, alone and with a second context hammering the GPU, must be
bit-identical every time (the sweep's hand-off protocols are timing dependent; a rare race would show up here)."""
import sys, os, threading, time, hashlib, numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch
sys.path.insert(0, os.path.dirname(__file__))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda", 0)
cols, rows = 2000, 4000
L, R, blend, _ = synth.make_pair(cols, rows, 1234, dev)
out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev); f0 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev); f1 = torch.empty_like(f0)
ctx = pf.Context(0)
def run():
    ctx.novel_view_dev(L.data_ptr(), R.data_ptr(), cols, rows, 0, blend.data_ptr(), out.data_ptr(), f0.data_ptr(), f1.data_ptr())
    return (int(f0.view(torch.int32).sum(dtype=torch.int64)), int(f1.view(torch.int32).sum(dtype=torch.int64)), int(out.sum(dtype=torch.int64)))
ref = run(); bad = 0; t0 = time.time()
for i in range(n):
    bad += run() != ref
print("alone: %d runs, %d mismatches, %.1f ms each" % (n, bad, 1000 * (time.time() - t0) / n), flush=True)
# second context on another pair (different seed, search_20) in a thread
L2, R2, b2, _ = synth.make_pair(1500, 3000, 77, dev)
o2 = torch.empty((3000, 1500, 4), dtype=torch.uint8, device=dev); g0 = torch.empty((3000, 1500, 2), dtype=torch.float32, device=dev); g1 = torch.empty_like(g0)
c2 = pf.Context(0); stop = False; other = []
def hammer():
    first = None
    while not stop:
        c2.novel_view_dev(L2.data_ptr(), R2.data_ptr(), 1500, 3000, 20, b2.data_ptr(), o2.data_ptr(), g0.data_ptr(), g1.data_ptr())
        h = (int(g0.view(torch.int32).sum(dtype=torch.int64)), int(o2.sum(dtype=torch.int64)))
        if first is None: first = h
        other.append(h != first)
t = threading.Thread(target=hammer); t.start(); bad = 0; t0 = time.time()
for i in range(n):
    bad += run() != ref
stop = True; t.join()
print("with a second context: %d runs, %d mismatches, %.1f ms each; other context %d runs, %d mismatches" % (n, bad, 1000 * (time.time() - t0) / n, len(other), sum(other)), flush=True)
