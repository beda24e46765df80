#!/usr/bin/env python
"""Benchmark of the hot path: Mpix/s of bidirectional PixFlow + novel-view blend on one overlap strip
per GPU (BASELINE.json configs[1]: 2000x4000, pixflow_low), inputs resident in HBM when the clock starts.

One process per GPU (torch.distributed / RCCL for the barrier, the max-over-ranks time and the final
gather of the blended strips to rank 0); each rank owns one independent pair -> weak scaling.
Prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "panorama-opticalflow_amd")


def _load(name):
    modname = "pano_amd_" + name
    if modname in sys.modules:
        return sys.modules[modname]
    spec = importlib.util.spec_from_file_location(modname, os.path.join(PKG, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def cpu_baseline(L, R, blend, max_pct):
    """The oracle (a port: the reference's CPU/ cannot be compiled here) on the GPU box's host cores, on the
    SAME pair: both flow directions on 2 threads (the only result-preserving parallelism the algorithm has,
    OpticalFlow.cpp:130-139) + the blend."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    res = [None, None]

    def run(d):
        res[d] = orc.flow_one_dir(L, R, max_pct, d)

    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    out = orc.combine_novel_views(L, R, res[0], res[1], blend)
    dt = time.perf_counter() - t0
    return dt, res[0], res[1], out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cols", type=int, default=2000)
    ap.add_argument("--rows", type=int, default=4000)
    ap.add_argument("--alg", default="pixflow_low")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--concurrent", type=int, default=1, help="independent pairs in flight per GPU (one context + host thread each); 1 = the BASELINE config")
    args = ap.parse_args()

    # a context drives 4 HIP streams (front end, two flow directions, blend ramp) next to torch's and RCCL's: with the
    # runtime's default of 4 hardware queues two of them could share a queue and serialise
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(min(24, max(8, 4 * args.concurrent))))
    import numpy as np
    import torch  # first: the HIP runtime it loads is the one libpanoflow.so then binds to
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("PANOFLOW_FORCE_DIST") == "1"   # exercise the RCCL path on a single GPU
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    pf = _load("pyabi"); synth = _load("synth"); shard = _load("shard")
    cols, rows = args.cols, args.rows
    max_pct = pf.max_percentage_by_name(args.alg)
    ctx = pf.Context(local_rank)

    # one independent synthetic pair per rank (seed 1234 + rank), generated straight into HBM
    L, R, blend, _ = synth.make_pair(cols, rows, 1234 + rank, dev)
    out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev)
    f0 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev)
    f1 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev)
    my_pairs = shard.pairs_for_rank(world, rank, world)   # one pair per GPU: pair i on rank i
    assert my_pairs == [rank]
    torch.cuda.synchronize()

    # optional throughput mode: more independent pairs in flight on the same GPU (a sweep only occupies ~35 of 256 CUs)
    extra = []
    for j in range(1, max(1, args.concurrent)):
        Lj, Rj, bj, _ = synth.make_pair(cols, rows, 1234 + rank + 1000 * j, dev)
        extra.append((pf.Context(local_rank), Lj, Rj, bj, torch.empty_like(out), torch.empty_like(f0), torch.empty_like(f1)))
    torch.cuda.synchronize()

    def one(cx, Lx, Rx, bx, ox, fx0, fx1):
        cx.novel_view_dev(Lx.data_ptr(), Rx.data_ptr(), cols, rows, max_pct, bx.data_ptr(), ox.data_ptr(), fx0.data_ptr(), fx1.data_ptr())

    # the only exchange of the path: final gather of the blended strips to rank 0 over RCCL/xGMI.  It overlaps the next
    # pair's compute (two result buffers, one gather in flight); the fence waits for the last one.
    og = shard.OverlappedGather(out, world, rank) if (world > 1 or force_dist) else None

    def step():
        # flows + blended strip end up resident in HBM; the call is synchronous on return
        ths = [threading.Thread(target=one, args=e) for e in extra]
        for t in ths:
            t.start()
        o = og.out_buffer() if og else out
        ctx.novel_view_dev(L.data_ptr(), R.data_ptr(), cols, rows, max_pct, blend.data_ptr(), o.data_ptr(), f0.data_ptr(), f1.data_ptr())
        for t in ths:
            t.join()
        if og:
            og.submit()

    def fence():
        if og:
            og.wait()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    ctx.profile_enable(0 if args.no_profile else 2)   # timed region: HIP events around the dominant kernel (the sweeps) only
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    dt = shard.max_over_ranks(dt, dev)

    prof = ctx.profile()
    # per-family breakdown from ONE extra, untimed step with every family instrumented
    ctx.profile_reset(); ctx.profile_enable(1); step(); ctx.profile_enable(0)
    prof_all = ctx.profile()
    if rank == 0:
        mpix = cols * rows / 1e6
        value = world * max(1, args.concurrent) * mpix * args.steps / dt
        P, nlev, sweep_steps = pf.level_pixels(cols, rows)
        b_alg = pf.algorithmic_bytes(cols, rows)
        res = {
            "metric": "Mpix/s bidirectional optical flow (overlap strip) at 1/2/4/8 GPU", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d overlap strip, %s, flow L->R + R->L + novel-view blend, %d pair(s) in flight per GPU" % (cols, rows, args.alg, max(1, args.concurrent)),
                       "levels": nlev, "level_pixels": P, "sweep_steps_per_direction": sweep_steps, "final_gather": "rccl, overlapped with the next pair" if world > 1 else "none"},
        }
        # roofline of the dominant kernel (the exact wavefront sweep): algorithmic bytes per launch =
        # 48 B per level-pixel (alpha/grad0 16 + blurred 8 + flow r/w 16 + grad1 gather 8)
        # x the level's pixels; 2 sweeps x 2 directions x all levels = 4*48*P bytes per step.
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the
        # figure comes from the committed rocprofv3 --pmc pass of this same command (profiles/, see its note); it only
        # applies to the default workload.
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r01_pmc_bench.json")
        if (cols, rows, args.alg) == (2000, 4000, "pixflow_low") and os.path.exists(pmc_path):
            try:
                pl = json.load(open(pmc_path))["sweep_per_launch"]
                traffic = round(0.5 * (pl["traffic_bytes_lo"] + pl["traffic_bytes_hi"]))
                traffic_src = "profiles/r01_pmc_bench.json (FETCH_SIZE+WRITE_SIZE per sweep launch; read side bracketed [raw,2x raw], midpoint reported)"
            except Exception:
                pass
        if "sweep" in prof and prof["sweep"][1] > 0:
            ms, n = prof["sweep"]
            bytes_total = 48.0 * P * 4 * args.steps
            ach = bytes_total / (ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": "k_sweep_prep+k_sweep2", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 6),
                               "traffic": traffic, "traffic_source": traffic_src, "launches": n, "avg_launch_us": round(1000 * ms / n, 2),
                               "note": "exact Gauss-Seidel sweep is dependency-latency bound (critical path %d wavefront steps/direction), not HBM bound" % sweep_steps}
        else:
            res["roofline"] = {"bound": "hbm", "kernel": "k_sweep", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}
        ach_path = b_alg * args.steps / dt / 1e9
        res["roofline_path"] = {"bound": "hbm", "achieved": round(ach_path, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach_path / 8000.0, 6),
                                "algorithmic_bytes_per_pair": b_alg}
        res["kernels_ms_per_step"] = {k: round(v[0], 3) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
        if world == 1 and not args.no_cpu_baseline:
            Lh, Rh, bh = L.cpu().numpy(), R.cpu().numpy(), blend.cpu().numpy()
            tcpu, r0, r1, rout = cpu_baseline(Lh, Rh, bh, max_pct)
            g0, g1, gout = f0.cpu().numpy(), f1.cpu().numpy(), (og.bufs[(og.k - 1) % 2] if og else out).cpu().numpy()
            off = np.abs(gout.astype(np.int32) - rout.astype(np.int32))
            res["cpu_baseline"] = {"value": round(mpix / tcpu, 4), "unit": "Mpix/s", "cores": 2, "kind": "port",
                                   "sample": "the same %dx%d pair, whole path once: 2 flow directions on 2 threads + blend (%.1f s)" % (cols, rows, tcpu)}
            res["parity_vs_cpu"] = {"max_abs_dflow_px": float(max(np.abs(g0 - r0).max(), np.abs(g1 - r1).max())),
                                    "blend_pixels_off": int((off > 0).sum()), "blend_max_lsb": int(off.max())}
        line = json.dumps(res)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    if rank == 0:
        # RCCL prints a version banner through C stdio, which a pipe only delivers at exit: flush it first so that the
        # JSON line is the last thing on stdout
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(line, flush=True)


if __name__ == "__main__":
    main()
