"""GPU box: what LARGE displacements cost on the current code (round-4 review, missing #4 / next #2).
The dense 9000x4000 pair of the bench with synth.make_pair(disp_scale = 1, 4, 8): lone pair (median of 5) and 16 in flight (one batch);
with DISP_STATS=1 (and var_libs/lib_stats.so copied over libpanoflow.so by the caller: a -DPF_SWEEP_STATS build) one solve per scale that
reports the wave-steps whose gather left the LDS window.  disp_probe.py [scales...]"""
import ctypes, os, statistics, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from conftest import load_pkg_module
pf = load_pkg_module("pyabi"); synth = load_pkg_module("synth")
dev = torch.device("cuda", 0)
cols, rows = int(os.environ.get("DISP_COLS", "9000")), int(os.environ.get("DISP_ROWS", "4000"))
scales = [float(v) for v in sys.argv[1:]] or [1.0, 4.0, 8.0]
stats = os.environ.get("DISP_STATS") == "1"
nfl = int(os.environ.get("DISP_INFLIGHT", "16"))
mpix = cols * rows / 1e6
ctx = pf.Context(0, cols, rows)
out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev)
f0 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev); f1 = torch.empty_like(f0)
for sc in scales:
    L, R, blend, _ = synth.make_pair(cols, rows, 1234, dev, disp_scale=sc)
    torch.cuda.synchronize()
    call = lambda: ctx.novel_view_dev(L.data_ptr(), R.data_ptr(), cols, rows, 0, blend.data_ptr(), out.data_ptr(), f0.data_ptr(), f1.data_ptr())
    call()
    if stats:
        l = pf.lib()
        z = (ctypes.c_ulonglong * 4)()
        l.pf_debug_sweep_stats(z, 1)
        call()
        l.pf_debug_sweep_stats(z, 1)
        print("disp_scale %g lone pair: latency-form wave-steps %d, gather rounds that left the LDS window %d (%.2f %%)" % (sc, z[0], z[1], 100.0 * z[1] / max(1, z[0])), flush=True)
        nb = 8
        pairs = [synth.make_pair(cols, rows, 6000 + i, dev, disp_scale=sc)[:3] for i in range(nb)]
        outs = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(nb)]
        cb = pf.Context(0)
        callb = lambda: cb.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0,
                                                [p[2].data_ptr() for p in pairs], [o.data_ptr() for o in outs], None, None, in_flight=nb)
        callb(); l.pf_debug_sweep_stats(z, 1); callb(); l.pf_debug_sweep_stats(z, 1)
        print("disp_scale %g batch of %d: latency form %d wave-steps / %d rounds outside (%.2f %%); throughput form %d wave-steps (two gather rounds each) / %d rounds outside (%.2f %% of the rounds)" %
              (sc, nb, z[0], z[1], 100.0 * z[1] / max(1, z[0]), z[2], z[3], 50.0 * z[3] / max(1, z[2])), flush=True)
        cb.close(); del pairs, outs
        continue
    ts = []
    for _ in range(5):
        t = time.perf_counter(); call(); ts.append(time.perf_counter() - t)
    tm = statistics.median(ts)
    # end-point error against the analytic displacement (flow L->R = +d inside the valid region: L shows T(x + d/2) at x, R shows it at x + d): the solve still finds the field
    ys = torch.arange(rows, dtype=torch.float64, device=dev)[:, None].expand(rows, cols); xs = torch.arange(cols, dtype=torch.float64, device=dev)[None, :].expand(rows, cols)
    dx, dy = synth.displacement(xs, ys, cols, rows, sc)
    a = synth.alpha_mask(xs, ys, cols, rows)
    epe = torch.sqrt((f0[..., 0].double() - dx) ** 2 + (f0[..., 1].double() - dy) ** 2)[a]
    fmax = float(f0.abs().max())
    del xs, ys, dx, dy, a
    print("disp_scale %g lone pair: %.2f ms (%.1f Mpix/s); max |flow| %.1f px full-res = %.1f at level 0; EPE vs analytic median %.3f / p99 %.3f px" %
          (sc, 1000 * tm, mpix / tm, fmax, fmax / 2, float(epe.median()), float(epe.kthvalue(int(0.99 * epe.numel())).values)), flush=True)
    del epe
    if nfl > 0:
        pairs = [synth.make_pair(cols, rows, 6000 + i, dev, disp_scale=sc)[:3] for i in range(nfl)]
        outs = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(nfl)]
        torch.cuda.synchronize()
        cb = pf.Context(0)
        callb = lambda: cb.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0,
                                                [p[2].data_ptr() for p in pairs], [o.data_ptr() for o in outs], None, None, in_flight=nfl)
        callb()
        tb = []
        for _ in range(3):
            t = time.perf_counter(); callb(); tb.append(time.perf_counter() - t)
        tbm = statistics.median(tb)
        print("disp_scale %g %d in flight: %.2f ms per pair (%.1f Mpix/s)" % (sc, nfl, 1000 * tbm / nfl, nfl * mpix / tbm), flush=True)
        cb.close(); del pairs, outs
        torch.cuda.empty_cache()
