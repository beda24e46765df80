"""Per-tile timeline of ONE relaxation-sweep launch (needs var_libs/lib_rxstats.so): rx_trace.py <level width> <fwd 0|1> [cols rows]"""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["PANOFLOW_SWEEP"] = "3"
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
pf.SO_PATH = os.path.join(ROOT, "var_libs", "lib_rxstats.so")
Wsel, fwd = int(sys.argv[1]), int(sys.argv[2])
cols, rows = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (2000, 4000)
ctx = pf.Context(0)
lib = ctx.l
L, R, blend = synth.make_pair_np(cols, rows, 1234)
ctx.flow(L, R, 0, 0)
assert lib.pf_debug_rx_select(Wsel, fwd) == 0
ctx.flow(L, R, 0, 0)
buf = np.zeros((4096, 16), np.int64)
assert lib.pf_debug_rx_dump(buf.ctypes.data_as(C.c_void_p), 4096) == 0
ntx = int(buf[0, 15]); used = buf[buf[:, 4] > 0]
t0 = used[:, 0].min()
print("level W=%d fwd=%d: %d tiles (%d per row); kernel span %.1f us" % (Wsel, fwd, len(used), ntx, (used[:, 4].max() - t0) / 100.0))
print("   t   i   j |  start  initDone lastBusy   finSeen      end (us) | rounds busy evals | dense n/us sparse n/us evals")
step = max(1, len(used) // 60)
for t in range(0, len(used), step):
    d = buf[t]
    us = lambda v: (v - t0) / 100.0 if v else -1
    print("%4d %3d %3d | %7.1f %8.1f %8.1f %8.1f %8.1f | %5d %4d %6d | %2d %6.1f %3d %6.1f %5d" % (t, t % ntx, t // ntx, us(d[0]), us(d[1]), us(d[2]), us(d[3]), us(d[4]), d[5], d[6], d[7], d[10], d[11] / 100.0, d[12], d[13] / 100.0, d[14]))
b = used
print("sums over tiles (us): init %.0f  dense %.0f  sparse %.0f  resident %.0f" % (((b[:, 1] - b[:, 0]).sum()) / 100.0, b[:, 11].sum() / 100.0, b[:, 13].sum() / 100.0, (b[:, 4] - b[:, 0]).sum() / 100.0))
col0 = buf[0:len(used):ntx]
print("left column of tiles (i = 0): busy rounds sum %d, evaluated pixels sum %d; end of tile (0,j) minus end of (0,j-1), us:" % (col0[:, 6].sum(), col0[:, 7].sum()))
print("   " + " ".join("%.0f" % ((col0[j, 4] - col0[j - 1, 4]) / 100.0) for j in range(1, len(col0))))
print("   busy rounds per tile: " + " ".join("%d" % v for v in col0[:, 6]))
