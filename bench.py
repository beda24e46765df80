#!/usr/bin/env python
"""Benchmark of the hot path: Mpix/s of bidirectional PixFlow + novel-view blend, inputs resident in HBM
when the clock starts.

ONE per-GPU workload for every --gpus N, so that the driver's N = 1, 2, 4, 8 values divide into a scaling curve:
  north_star / BASELINE.json configs[4] -- one DENSE 9000x4000 overlap pair per GPU (seed 1234 + rank), pixflow_low, flows L->R and
  R->L + novel-view blend through pf_novel_view_dev; weak scaling, no collective on the data path.  With N > 1 (or
  PANOFLOW_FORCE_DIST=1 on one GPU) the blended strips are additionally gathered to rank 0 over RCCL inside libpanoflow.so
  (pf_dist_*), overlapped with the next pair.  `value` = N x 36 Mpix x steps / max-over-ranks time.
At N = 1 the same JSON line carries, as extra keys timed in-process (never `value`):
  `strip_2000x4000`  BASELINE configs[1] (one 2000x4000 strip, what rounds 1-2 quoted `value` on),
  `config4_chain`    BASELINE configs[3] (5+top stitch chain at 9000x4000, pixflow_search_20, host images -> host composite),
  `throughput_mode`  several independent strips side by side on the GPU (pf_novel_view_batch_dev),
  `roofline.latency_bound` (dependency-chain bound of the exact sweeps), `cpu_baseline` + `parity_vs_cpu`.
(--cols/--rows override the workload.)

One process per GPU (torch.distributed / RCCL for the barrier, the max-over-ranks time and the final gather);
prints ONE JSON line on rank 0.
"""
import argparse
import importlib.util
import json
import os
import statistics
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


def cpu_baseline(L, R, blend, max_pct, more_pairs=()):
    """The oracle (a port: the reference's CPU/ cannot be compiled here) on the GPU box's host cores, on the SAME pair.  Three legs,
    ONE AFTER THE OTHER:
    (ii) 2 threads: one per flow direction, the only result-preserving parallelism the algorithm has (OpticalFlow.cpp:130-139);
    (iii) config 5: one pair per two cores -- `more_pairs` further (L, R) sub-strips solved at the same time as this one, two
          threads each (pairs are independent: CPU/main.cpp:70,82);
    (i) 1 thread: both directions one after the other (the reference has no threading of its own)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import orc
    orc.build()
    res = [None, None]

    def timed(fn, *a):
        t0 = time.perf_counter()
        r = fn(*a)
        return time.perf_counter() - t0, r

    def two_threads(Lx, Rx, keep):
        th = [threading.Thread(target=lambda d=d: keep.__setitem__(d, orc.flow_one_dir(Lx, Rx, max_pct, d))) for d in (0, 1)]
        [t.start() for t in th]; [t.join() for t in th]

    t_two, _ = timed(two_threads, L, R, res)
    t_many = None
    if more_pairs:
        sinks = [[None, None] for _ in more_pairs]
        def many():
            th = [threading.Thread(target=two_threads, args=(L, R, [None, None]))]
            th += [threading.Thread(target=two_threads, args=(Lp, Rp, sk)) for (Lp, Rp), sk in zip(more_pairs, sinks)]
            [t.start() for t in th]; [t.join() for t in th]
        t_many, _ = timed(many)
    t_one, _ = timed(lambda: [orc.flow_one_dir(L, R, max_pct, d) for d in (0, 1)])
    t_blend, out = timed(orc.combine_novel_views, L, R, res[0], res[1], blend)
    return t_one + t_blend, t_two + t_blend, (t_many + t_blend) if t_many else None, res[0], res[1], out


def measure_t_step(pf, ctx, np):
    """Step time of ONE lone band of the sweep kernel (8 rows x 4096 columns, dense random data): the machine's floor for
    this instruction sequence, with no band-to-band skew.  HIP events around the kernel (the 'sweep' family)."""
    rng = np.random.default_rng(5)
    h, w = 8, 4096
    g0 = rng.standard_normal((h, w, 2)).astype(np.float32) * 0.05
    g1 = rng.standard_normal((h, w, 2)).astype(np.float32) * 0.05
    bl = rng.standard_normal((h, w, 2)).astype(np.float32)
    fl = rng.standard_normal((h, w, 2)).astype(np.float32)
    a = np.ones((h, w), np.float32)
    ctx.stage_sweep(g0, g1, bl, a, a, fl, True)          # warm-up (allocations)
    ctx.profile_reset(); ctx.profile_enable(1)
    for _ in range(3):
        ctx.stage_sweep(g0, g1, bl, a, a, fl, True)
    ctx.profile_enable(0)
    ms, n = ctx.profile()["sweep"]
    ctx.profile_reset()
    return 1000.0 * ms / n / (w + h - 1)


def copy_rates(torch, pf, ctx, np, L, R, out, f0, f1):
    """H2D / D2H of the `value` workload's buffers, reported separately and never part of `value` (the timed region
    starts with the images resident in HBM and ends with flows + strip resident in HBM): the two input images up, both flows and
    the blended strip down, through the library's own copy entry points, from pageable memory and from pf_host_alloc (page-locked)
    memory; plus the device's measured copy bandwidth (a 2 GiB device-to-device copy, read + write counted)."""
    def timed(fn):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); return time.perf_counter() - t0
    ups = [(L, L.numel()), (R, R.numel())]
    downs = [(f0, f0.numel() * 4), (f1, f1.numel() * 4), (out, out.numel())]
    res = {}
    for kind in ("pageable", "pinned"):
        hb = {}
        for t, nbytes in ups + downs:
            hb[id(t)] = np.empty(nbytes, np.uint8) if kind == "pageable" else ctx.host_array((nbytes,), np.uint8)
            hb[id(t)][:] = 1   # touch the pages
        up = lambda: [ctx.upload(t.data_ptr(), hb[id(t)]) for t, _ in ups]
        down = lambda: [ctx.download(hb[id(t)], t.data_ptr()) for t, _ in downs]
        saved = [t.clone() for t, _ in ups]
        down(); tu = min(timed(up) for _ in range(2)); td = min(timed(down) for _ in range(2))
        for (t, _), sv in zip(ups, saved):
            t.copy_(sv)       # (the uploads overwrote the inputs with the staging buffers' content)
        nu = sum(n for _, n in ups); nd = sum(n for _, n in downs)
        res[kind] = {"h2d_ms": round(1000 * tu, 2), "h2d_GBps": round(nu / tu / 1e9, 2), "d2h_ms": round(1000 * td, 2), "d2h_GBps": round(nd / td / 1e9, 2)}
        del hb
    res["h2d_bytes"] = sum(n for _, n in ups); res["d2h_bytes"] = sum(n for _, n in downs)
    a = torch.empty(1 << 31, dtype=torch.uint8, device=L.device); b = torch.empty_like(a)
    b.copy_(a)
    td2d = min(timed(lambda: b.copy_(a)) for _ in range(3))
    res["hbm_copy_measured_GBps"] = round(2 * a.numel() / td2d / 1e9, 1)
    res["note"] = "copies of the `value` workload's buffers through pf_upload / pf_download; never part of `value`; hbm_copy = 2 GiB device-to-device, read + write"
    del a, b
    torch.cuda.empty_cache()
    return res


def fixture_verdict(np, cols, rows, alg, seed, L, R, blend, f0, f1, strip):
    """This rank's pair against the oracle fixture of ITS seed (tests/golden/dense_<size>[_s<seed>].npz: SHA-256 of the inputs and of
    the oracle's two flows and blended strip, computed in the build container): the TIMED pair's own outputs, no extra pass."""
    import hashlib
    fx = os.path.join(ROOT, "tests", "golden", "dense_%dx%d%s.npz" % (cols, rows, "" if seed == 1234 else "_s%d" % seed))
    if alg != "pixflow_low" or not os.path.exists(fx):
        return None, "no fixture for %dx%d seed %d %s" % (cols, rows, seed, alg)
    g = np.load(fx)
    sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()
    if [sha(L), sha(R), sha(blend)] != [str(v) for v in g["sha_inputs"]]:
        return False, "the generated inputs are not the fixture's (synth.make_pair must give the host's bytes on every device)"
    got = [sha(f0), sha(f1), sha(strip)]
    want = [str(v) for v in g["sha_outputs"]]
    return got == want, {"flow_l2r_bit_identical": got[0] == want[0], "flow_r2l_bit_identical": got[1] == want[1], "blend_byte_identical": got[2] == want[2],
                         "strip_sha256": want[2]}


def config5_strong(torch, np, pf, synth, shard, dist, pfd, local_rank, dev, rank, world, cols, rows, args, pair0):
    """BASELINE configs[4] as STRONG scaling (an extra key, never `value`): `--pairs-total` (8) fixture pairs -- seeds 1234 ... 1241, the
    pairs tests/golden/dense_9000x4000[_s<seed>].npz hold the oracle's SHA-256 for -- sharded round-robin over the ranks (pair i on rank
    i mod N: CPU/main.cpp:70,82, pairs are independent), each rank's share as ONE batch in flight (pf_novel_view_batch_dev), every strip
    then gathered into rank 0's HBM.  Timed: barrier -> batch call -> gathers -> wait -> barrier, max over ranks, median of 3 after a
    warm-up.  Off the clock every rank checks the flows + strip of its own pairs against their fixtures (verdicts all-reduced) and rank 0
    checks every gathered strip against the SHA-256 its producer's fixture holds.  At N = 1 this is the `pairs_9000x4000` figure on the
    fixture seeds, validated."""
    import hashlib
    P = args.pairs_total
    mine = shard.pairs_for_rank(P, rank, world)
    per_rank = (P + world - 1) // world
    pairs = [pair0 if i == rank else synth.make_pair(cols, rows, 1234 + i, dev)[:3] for i in mine]   # (this rank's timed pair IS pair `rank`)
    strips = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in mine]
    fl0 = [torch.empty((rows, cols, 2), dtype=torch.float32, device=dev) for _ in mine]
    fl1 = [torch.empty((rows, cols, 2), dtype=torch.float32, device=dev) for _ in mine]
    # rank 0: one receive area per round of the round-robin (slot r of round j = pair j * world + r)
    recv = [torch.empty((world, rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(per_rank)] if (pfd and rank == 0) else []
    dummy = torch.zeros((rows, cols, 4), dtype=torch.uint8, device=dev) if (pfd and len(mine) < per_rank) else None
    cb = pf.Context(local_rank)
    torch.cuda.synchronize()
    nbytes = rows * cols * 4

    def fence():
        if pfd:
            pfd.wait(); pfd.barrier()
        elif dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def once():
        if mine:
            cb.novel_view_batch_dev([p[0].data_ptr() for p in pairs], [p[1].data_ptr() for p in pairs], cols, rows, 0, [p[2].data_ptr() for p in pairs],
                                    [o.data_ptr() for o in strips], [f.data_ptr() for f in fl0], [f.data_ptr() for f in fl1], in_flight=len(mine))
        if pfd:   # every rank takes part in every round (a rank without a pair in the last round sends a dummy strip)
            for j in range(per_rank):
                src = strips[j] if j < len(mine) else dummy
                pfd.gather_async(src.data_ptr(), recv[j].data_ptr() if rank == 0 else 0, nbytes)

    once(); fence()
    ts = []
    for _ in range(3):
        fence(); t0 = time.perf_counter(); once(); fence(); dt = time.perf_counter() - t0
        ts.append(pfd.max(dt) if pfd else (shard.max_over_ranks(dt, dev) if dist is not None else dt))
    t = statistics.median(ts)
    # ---- validation, off the clock ----
    ok = True; have = True
    for k, i in enumerate(mine):
        v, _ = fixture_verdict(np, cols, rows, args.alg, 1234 + i, pairs[k][0], pairs[k][1], pairs[k][2], fl0[k], fl1[k], strips[k])
        have = have and v is not None
        ok = ok and bool(v)
    if dist is not None:
        code = torch.tensor([2 if not have else (1 if ok else 0)], device=dev)
        dist.all_reduce(code, op=dist.ReduceOp.MIN)
        have = int(code.item()) != 2; ok = int(code.item()) == 1
    slots_ok = None
    if pfd and rank == 0 and have:
        slots_ok = True
        for i in range(P):
            fxr = os.path.join(ROOT, "tests", "golden", "dense_%dx%d%s.npz" % (cols, rows, "" if i == 0 else "_s%d" % (1234 + i)))
            want = str(np.load(fxr)["sha_outputs"][2])
            slots_ok = slots_ok and hashlib.sha256(recv[i // world][i % world].cpu().numpy().tobytes()).hexdigest() == want
    cb.close()
    del pairs, strips, fl0, fl1, recv
    torch.cuda.empty_cache()
    return {"value": round(P * cols * rows / 1e6 / t, 3), "unit": "Mpix/s", "scaling": "strong", "pairs_total": P, "ranks": world, "in_flight_per_rank": per_rank,
            "seconds": round(t, 4), "ms_per_pair": round(1000 * t / P, 3), "runs": 3, "warmup": 1, "statistic": "median of max-over-ranks",
            "fixture_ok": (ok if have else None), "gathered_strips_equal_fixtures": slots_ok,
            "gather": "pf_dist_gather_async, %d round(s), inside the timed region" % per_rank if pfd else "none (single rank)",
            "workload": "BASELINE configs[4] as strong scaling: the %d oracle-fixture pairs (seeds 1234..%d), pair i on rank i mod N, each rank's share as one batch in flight" % (P, 1233 + P)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--cols", type=int, default=9000, help="default 9000 (BASELINE configs[4] / north_star: one 9000x4000 pair per GPU, for EVERY --gpus N)")
    ap.add_argument("--rows", type=int, default=4000)
    ap.add_argument("--alg", default="pixflow_low")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the 2000x4000 strip / config-4 chain / throughput / lone-band step-time legs")
    ap.add_argument("--no-profile", action="store_true", help="do not record per-kernel HIP events in the timed region")
    ap.add_argument("--pairs-total", type=int, default=8, help="extra key config5_strong (never `value`): BASELINE configs[4] read as STRONG scaling -- this many fixture pairs "
                    "(seeds 1234...) in total, sharded round-robin over the ranks, each rank's share as ONE batch in flight, all strips gathered to rank 0; 0 = skip")
    ap.add_argument("--concurrent", type=int, default=1, help="independent pairs in flight per GPU (one context + host thread each); 1 = the BASELINE config")
    args = ap.parse_args()

    # a context drives 4 HIP streams (front end, two flow directions, blend ramp) next to torch's and RCCL's: with the
    # runtime's default of 4 hardware queues two of them could share a queue and serialise
    # (the config5_strong leg -- 8 pairs in flight on two batch lanes beside the main context -- needs them with or without the extras: with 8 queues
    # it ran at 1,650 instead of 2,454 Mpix/s in round 5's driver-command line, profiles/r05_bench_driver_cmd_line.json)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "24" if (not args.no_extras or args.pairs_total > 0) else str(min(24, max(8, 4 * args.concurrent))))
    import numpy as np
    import torch  # first: the HIP runtime it loads is the one libpanoflow.so then binds to
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0")); local_rank = int(os.environ.get("LOCAL_RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = os.environ.get("PANOFLOW_FORCE_DIST") == "1"   # exercise the RCCL path on a single GPU (same workload, same `value`)
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=dev)

    pf = _load("pyabi"); synth = _load("synth"); shard = _load("shard")
    cols, rows = args.cols, args.rows
    max_pct = pf.max_percentage_by_name(args.alg)
    ctx = pf.Context(local_rank, cols, rows)            # pre-sized: the first step pays no allocation

    # one independent synthetic pair per rank (seed 1234 + rank), generated straight into HBM
    L, R, blend, _ = synth.make_pair(cols, rows, 1234 + rank, dev)
    out = torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev)
    f0 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev)
    f1 = torch.empty((rows, cols, 2), dtype=torch.float32, device=dev)
    my_pairs = shard.pairs_for_rank(world, rank, world)   # one pair per GPU: pair i on rank i
    assert my_pairs == [rank]
    torch.cuda.synchronize()

    # optional: more independent pairs in flight on the same GPU (a sweep only occupies a fraction of the 256 CUs)
    extra = []
    for j in range(1, max(1, args.concurrent)):
        Lj, Rj, bj, _ = synth.make_pair(cols, rows, 1234 + rank + 1000 * j, dev)
        extra.append((pf.Context(local_rank, cols, rows), Lj, Rj, bj, torch.empty_like(out), torch.empty_like(f0), torch.empty_like(f1)))
    torch.cuda.synchronize()

    def one(cx, Lx, Rx, bx, ox, fx0, fx1):
        cx.novel_view_dev(Lx.data_ptr(), Rx.data_ptr(), cols, rows, max_pct, bx.data_ptr(), ox.data_ptr(), fx0.data_ptr(), fx1.data_ptr())

    # the only exchange of the path: final gather of the blended strips to rank 0 over RCCL/xGMI.  It overlaps the next
    # pair's compute (two result buffers, one gather in flight); the fence waits for the last one.
    og = None
    pfd = None
    gather_note = ""
    if world > 1 or force_dist:
        # control plane only: rank 0's ncclUniqueId reaches the other ranks through torch.distributed; the gather itself runs
        # inside libpanoflow.so (pf_dist_*: grouped ncclSend/ncclRecv on its own HIP stream)
        try:
            ids = [pf.dist_unique_id() if rank == 0 else None]
        except pf.PanoflowError as e:
            ids = [None]; gather_note = str(e)
        dist.broadcast_object_list(ids, src=0)
        try:
            pfd = pf.Dist(local_rank, ids[0], rank, world) if ids[0] is not None else None
        except pf.PanoflowError as e:
            pfd = None; gather_note = str(e)
        # every rank must take the same path: agree on whether the library's communicator came up everywhere
        okt = torch.tensor([1 if pfd is not None else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        if int(okt.item()) == 1:
            og = shard.RcclGather(out, pfd)
        else:
            # reported loudly in config.final_gather; the gather then goes through torch.distributed (still RCCL)
            if pfd is not None:
                pfd.close()
            pfd = None
            og = shard.OverlappedGather(out, world, rank)
            gather_note = "pf_dist_* unavailable (%s): torch.distributed gather used instead" % (gather_note or "another rank failed")
    step_ms = []

    def step():
        # flows + blended strip end up resident in HBM; the call is synchronous on return
        t_s = time.perf_counter()
        ths = [threading.Thread(target=one, args=e) for e in extra]
        for t in ths:
            t.start()
        o = og.out_buffer() if og else out
        ctx.novel_view_dev(L.data_ptr(), R.data_ptr(), cols, rows, max_pct, blend.data_ptr(), o.data_ptr(), f0.data_ptr(), f1.data_ptr())
        for t in ths:
            t.join()
        if og:
            og.submit()
        step_ms.append(1000 * (time.perf_counter() - t_s))

    def fence():
        if og:
            og.wait()
            if pfd:
                pfd.barrier()
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.profile_reset()
    ctx.profile_enable(0 if args.no_profile else 2)   # timed region: HIP events around the dominant kernel (the sweeps) only
    fence()
    del step_ms[:]
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    ctx.profile_enable(False)
    dt = pfd.max(dt) if pfd else (shard.max_over_ranks(dt, dev) if og else dt)
    gathered_ok = None
    if og and rank == 0:
        # the strip this rank produced last must be what arrived in its own slot of the receive area
        gathered_ok = bool(torch.equal(og.last()[0], og.bufs[(og.k - 1) % 2]))
    med_ms = statistics.median(step_ms)
    swept = ctx.last_swept_steps()

    # ---- self-validation, off the clock: EVERY rank holds the outputs of its own timed pair (seed 1234 + rank) against the oracle
    # fixture of that seed; the verdicts are all-reduced, and rank 0 additionally checks every gathered slot against the SHA-256 of
    # the strip its producer should have made -- an N-GPU run needs nothing else to be trusted ----
    last_out = og.bufs[(og.k - 1) % 2] if og else out
    fx_ok, fx_detail = fixture_verdict(np, cols, rows, args.alg, 1234 + rank, L, R, blend, f0, f1, last_out)
    fx_all = fx_ok
    if world > 1 or force_dist:
        code = torch.tensor([2 if fx_ok is None else (1 if fx_ok else 0)], device=dev)
        dist.all_reduce(code, op=dist.ReduceOp.MIN)
        fx_all = None if int(code.item()) == 2 else bool(int(code.item()))
    slots_ok = None
    if og and rank == 0 and fx_ok is not None:
        import hashlib
        slots_ok = True
        for r in range(world):
            fxr = os.path.join(ROOT, "tests", "golden", "dense_%dx%d%s.npz" % (cols, rows, "" if r == 0 else "_s%d" % (1234 + r)))
            if not os.path.exists(fxr):
                slots_ok = None; break
            want_sha = str(np.load(fxr)["sha_outputs"][2])
            slots_ok = slots_ok and hashlib.sha256(og.last()[r].cpu().numpy().tobytes()).hexdigest() == want_sha

    prof = ctx.profile()
    # per-family breakdown from ONE extra, untimed step with every family instrumented
    ctx.profile_reset(); ctx.profile_enable(1); step(); ctx.profile_enable(0)
    fence()
    prof_all = ctx.profile()
    ctx.profile_reset()
    strong = None
    if args.pairs_total > 0 and (cols, rows, args.alg) == (9000, 4000, "pixflow_low") and args.concurrent <= 1 and args.pairs_total <= 8:
        strong = config5_strong(torch, np, pf, synth, shard, dist if (world > 1 or force_dist) else None, pfd, local_rank, dev, rank, world, cols, rows, args, (L, R, blend))
    if rank == 0:
        mpix = cols * rows / 1e6
        npairs = world * max(1, args.concurrent)
        value = npairs * mpix * args.steps / dt
        P, nlev, sweep_steps = pf.level_pixels(cols, rows)
        b_alg = pf.algorithmic_bytes(cols, rows)
        which = "BASELINE configs[4] / north_star" if (cols, rows) == (9000, 4000) else ("BASELINE configs[1]" if (cols, rows) == (2000, 4000) else "custom size")
        res = {
            "metric": "Mpix/s bidirectional optical flow (overlap strip) at 1/2/4/8 GPU", "value": round(value, 3), "unit": "Mpix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1000 * dt / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: one dense %dx%d overlap pair per GPU (the same for every --gpus N), %s, flow L->R + R->L + novel-view blend, %d pair(s) in flight per GPU"
                                   % (which, cols, rows, args.alg, max(1, args.concurrent)),
                       "levels": nlev, "level_pixels": P, "sweep_steps_per_direction": sweep_steps, "swept_steps_per_direction_in_gated_window": swept,
                       "final_gather": ((gather_note or "rccl send/recv to rank 0 inside libpanoflow.so (pf_dist_*), overlapped with the next pair") + "; rank-0 slot verified: %s; every rank's slot equals its oracle fixture's strip: %s" % (gathered_ok, slots_ok)) if og else "none (single rank)"},
            "fixture_ok": fx_all,
            "ms_per_step_median": round(med_ms, 3), "value_at_median": round(npairs * mpix / (med_ms * 1e-3), 3),
        }
        # roofline of the dominant kernel (the exact wavefront sweep): algorithmic bytes per launch =
        # 48 B per level-pixel (alpha/grad0 16 + blurred 8 + flow r/w 16 + grad1 gather 8)
        # x the level's pixels; 2 sweeps x 2 directions x all levels = 4*48*P bytes per step.
        # HBM traffic per launch of the dominant kernel: PMC counters cannot be read from inside this process, so the figure is
        # READ FROM A COMMITTED FILE -- the rocprofv3 --pmc passes of this same command (profiles/, see its note) -- and labelled so.
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "r06_pmc_bench.json")
        if not os.path.exists(pmc_path):
            pmc_path = os.path.join(ROOT, "profiles", "r05_pmc_bench.json")
        if (cols, rows, args.alg) == (9000, 4000, "pixflow_low") and os.path.exists(pmc_path):
            try:
                pl = json.load(open(pmc_path))["sweep_per_launch"]
                traffic = round(0.5 * (pl["traffic_bytes_lo"] + pl["traffic_bytes_hi"]))
                traffic_src = "from_file: profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, per sweep launch; read side bracketed [raw, 2x raw], midpoint reported; NOT measured by this run)" % os.path.basename(pmc_path)
            except Exception:
                pass
        if "sweep" in prof and prof["sweep"][1] > 0:
            ms, n = prof["sweep"]
            bytes_total = 48.0 * P * 4 * args.steps
            ach = bytes_total / (ms * 1e-3) / 1e9
            res["roofline"] = {"bound": "hbm", "kernel": "k_sweep_prep+k_sweep2", "achieved": round(ach, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach / 8000.0, 6),
                               "traffic": traffic, "traffic_source": traffic_src, "launches": n, "avg_launch_us": round(1000 * ms / n, 2),
                               "algorithmic_bytes_per_launch": round(bytes_total / n),
                               "note": "exact Gauss-Seidel sweep is dependency-latency bound (see latency_bound), not HBM bound"}
            sweep_ms_per_dir = ms / args.steps / 2.0          # the two directions run concurrently on two streams
        else:
            res["roofline"] = {"bound": "hbm", "kernel": "k_sweep_prep+k_sweep2", "achieved": None, "peak": 8000.0, "unit": "GB/s", "frac": None, "traffic": None}
            sweep_ms_per_dir = None
        ach_path = b_alg * args.steps / dt / 1e9
        if strong is not None:
            res["config5_strong"] = strong
        res["roofline_path"] = {"bound": "hbm", "achieved": round(ach_path, 3), "peak": 8000.0, "unit": "GB/s", "frac": round(ach_path / 8000.0, 6),
                                "algorithmic_bytes_per_pair": b_alg}
        res["kernels_ms_per_step"] = {k: round(v[0], 3) for k, v in sorted(prof_all.items(), key=lambda kv: -kv[1][0])}
        res["parity_vs_cpu"] = {"full_pair_vs_oracle_fixture": dict(fx_detail, inputs="the timed pair itself (every rank its own: seed 1234 + rank; verdicts all-reduced into fixture_ok)",
                                                                    fixture="tests/golden/dense_%dx%d[_s<seed>].npz (SHA-256 of the oracle's outputs, computed in the build container)" % (cols, rows))
                                if isinstance(fx_detail, dict) else fx_detail}
        if not args.no_extras and args.concurrent <= 1:
            res["copies"] = copy_rates(torch, pf, ctx, np, L, R, last_out, f0, f1)

        if world == 1 and not args.no_extras and args.concurrent <= 1:
            # ---- the honest bound of the sweeps: a dependency chain of `swept` steps per direction x the time of one step
            # of a lone band (measured live); isa_model_us = the product's step as one wave in order, priced with the lone-wave slot
            # measurements (profiles/r06_sweep_step_isa.txt) ----
            t_step = measure_t_step(pf, ctx, np)
            if sweep_ms_per_dir:
                bound_ms = swept * t_step * 1e-3
                lb = {"swept_steps": swept, "t_step_us": round(t_step, 4), "bound_ms": round(bound_ms, 3),
                      "measured_sweep_ms_per_direction": round(sweep_ms_per_dir, 3), "frac_of_bound": round(bound_ms / sweep_ms_per_dir, 4),
                      "note": "bound = swept_steps x t_step of ONE lone band (8 rows x 4096, HIP events); the two directions run concurrently"}
                isa = os.path.join(ROOT, "profiles", "r06_sweep_step_isa.json")
                if os.path.exists(isa):
                    try:
                        hw = json.load(open(isa))
                        # the issue-slot model of ONE wave in order (tests/micro/isa_chain.py on the product's assembly, priced with the lone-wave slot
                        # measurements of profiles/r06_slot_model.txt): a floor of the step only while it stays below the measured t_step -- said so here
                        lb["isa_model_us"] = hw["isa_model_us"]
                        lb["isa_model_issue_slots_per_step"] = hw.get("issue_slots_per_step")
                        lb["isa_model_ms"] = round(swept * hw["isa_model_us"] * 1e-3, 3)
                        lb["t_step_vs_isa_model"] = round(t_step / hw["isa_model_us"], 4)
                        lb["isa_model_is_below_measured_step"] = bool(hw["isa_model_us"] <= t_step)
                        lb["frac_of_isa_model"] = round(swept * hw["isa_model_us"] * 1e-3 / sweep_ms_per_dir, 4)
                        lb["isa_model_source"] = "from_file: profiles/r06_sweep_step_isa.txt (one wave issuing the product's step in order, one slot per ~4.46 cycles: profiles/r06_slot_model.txt)"
                    except Exception:
                        pass
                res["roofline"]["latency_bound"] = lb
            del extra[:]
            # ---- BASELINE configs[1]: ONE 2000x4000 strip (what rounds 1-2 quoted `value` on) ----
            sc, sr = 2000, 4000
            Ls, Rs, bs, _ = synth.make_pair(sc, sr, 1234, dev)
            os_ = torch.empty((sr, sc, 4), dtype=torch.uint8, device=dev)
            cs = pf.Context(local_rank, sc, sr)
            torch.cuda.synchronize()
            ts = []
            for i in range(8):
                t1 = time.perf_counter()
                cs.novel_view_dev(Ls.data_ptr(), Rs.data_ptr(), sc, sr, max_pct, bs.data_ptr(), os_.data_ptr())
                ts.append(time.perf_counter() - t1)
            tm = statistics.median(ts[1:])
            res["strip_2000x4000"] = {"value": round(sc * sr / 1e6 / tm, 3), "unit": "Mpix/s", "ms_per_pair": round(1000 * tm, 3), "alg": args.alg,
                                      "workload": "BASELINE configs[1]: one 2000x4000 strip", "swept_steps_per_direction": cs.last_swept_steps(), "steps": 7, "warmup": 1,
                                      "roofline_path_frac": round(pf.algorithmic_bytes(sc, sr) / tm / 8e12, 6)}
            cs.close()
            del Ls, Rs, bs, os_
            # ---- BASELINE configs[3]: the full 5+top chain, 9000x4000, pixflow_search_20, host images -> host composite ----
            cc, cr = 9000, 4000
            top, imgs = synth.make_stitch_set(cc, cr, 1234, 5, dev)
            top = top.cpu().numpy(); imgs = [im.cpu().numpy() for im in imgs]
            torch.cuda.empty_cache()
            cx = ctx if (cols, rows) == (cc, cr) else pf.Context(local_rank, cc, cr)
            final = cx.host_array((cr, cc, 4))       # the caller's result buffer (page-locked, pf_host_alloc), reused across runs like the reference's Mat

            def chain():
                t1 = time.perf_counter()
                for i, im in enumerate(imgs):
                    last = i == len(imgs) - 1
                    cx.stitch_prefetch(None if last else imgs[i + 1])      # image i+1 goes up while step i computes
                    o = cx.stitch_step(im, top if i == 0 else None, 20, want_out=last, out=final if last else None)
                return time.perf_counter() - t1, o

            chain()
            tc = sorted(chain()[0] for _ in range(3))[1]
            res["config4_chain"] = {"seconds": round(tc, 4), "unit": "s", "workload": "5+top stitch chain, 9000x4000, pixflow_search_20, pf_stitch_step x5 (host images in, host composite out)",
                                    "Mpix/s_canvas": round(5 * cc * cr / 1e6 / tc, 2), "runs": 3, "warmup": 1, "statistic": "median"}
            del final
            if cx is not ctx:
                cx.close()
        if world == 1 and not args.no_cpu_baseline:
            # A 9000x4000 pair costs the oracle ~2 x 100 s; the bounded sample is a 2000-column sub-strip of the SAME pair (the
            # whole path on it: 2 flow directions + blend), timed on 1 and on 2 host threads, and compared with the GPU path
            # run on exactly that sub-strip.  The full pair is held to the oracle through the committed fixture's SHA-256.
            x0 = max(0, (cols - 2000) // 2); x1 = min(cols, x0 + 2000)
            Lh, Rh, bh = L[:, x0:x1].contiguous().cpu().numpy(), R[:, x0:x1].contiguous().cpu().numpy(), blend[:, x0:x1].contiguous().cpu().numpy()
            # leg (iii): the same sub-strip of config 5's seven other pairs (seeds 1235..1241), generated on the GPU (the host's bytes)
            more = []
            if not args.no_extras:
                for sd in range(1235, 1242):
                    Lp, Rp, _, _ = synth.make_pair(cols, rows, sd, dev)
                    more.append((Lp[:, x0:x1].contiguous().cpu().numpy(), Rp[:, x0:x1].contiguous().cpu().numpy()))
                    del Lp, Rp
            t1, t2, t3, r0, r1, rout = cpu_baseline(Lh, Rh, bh, max_pct, more)
            cg = pf.Context(local_rank)
            gout, g0, g1 = cg.novel_view(Lh, Rh, max_pct, bh)
            cg.close()
            smp = (x1 - x0) * rows / 1e6
            res["cpu_baseline"] = {"value": round(smp / t2, 4), "unit": "Mpix/s", "cores": 2, "kind": "port",
                                   "sample": "columns [%d, %d) of the same %dx%d pair (a %dx%d sub-strip), whole path once: 2 flow directions on 2 threads + blend (%.1f s); the three legs run one after the other" % (x0, x1, cols, rows, x1 - x0, rows, t2),
                                   "one_thread": {"value": round(smp / t1, 4), "unit": "Mpix/s", "cores": 1, "seconds": round(t1, 2)},
                                   "pairs_in_parallel": ({"value": round((1 + len(more)) * smp / t3, 4), "unit": "Mpix/s", "cores": 2 * (1 + len(more)), "pairs": 1 + len(more), "seconds": round(t3, 2),
                                                          "note": "SURVEY 8(d) leg (iii), config 5: the same sub-strip of the 8 pairs (seeds 1234..1241) at the same time, two threads each"} if t3 else None),
                                   "host_threads_available": os.cpu_count(),
                                   "which_is_the_configurations": "`value` here = the bounded SAMPLE (a 2000-column sub-strip of the configuration's pair, measured by this run); the figure for the "
                                                                  "configuration itself -- the whole 9000x4000 pair -- is `full_pair` (read from a committed file, measured once on a GPU box's host)"}
            fp = os.path.join(ROOT, "profiles", "r05_cpu_full_pair.json")
            if (cols, rows, args.alg) == (9000, 4000, "pixflow_low") and os.path.exists(fp):
                try:   # the WHOLE pair once on a GPU box's host (tests/micro/cpu_full_pair.py); read from the committed file, not run here (minutes)
                    fj = json.load(open(fp))
                    res["cpu_baseline"]["full_pair"] = {"from_file": "profiles/r05_cpu_full_pair.json", "two_threads": fj["two_threads"], "one_thread": fj["one_thread"],
                                                        "cpu_model": fj["cpu_model"], "nproc": fj["nproc"], "outputs_equal_committed_fixture_sha256": fj["outputs_equal_committed_fixture_sha256"]}
                except Exception:
                    pass
            res["parity_vs_cpu"].update({"sample_max_abs_dflow_px": float(max(np.abs(g0 - r0).max(), np.abs(g1 - r1).max())),
                                         "sample_blend_bytes_off": int((gout != rout).sum())})
        if world == 1 and not args.no_extras and args.concurrent <= 1:
            # the main context's five streams give way first: a lane drives three streams and the runtime maps streams to
            # GPU_MAX_HW_QUEUES hardware queues round-robin -- more live streams than queues and two busy ones share a queue
            ctx.close()
            sc, sr = 2000, 4000
            os_ = torch.empty((sr, sc, 4), dtype=torch.uint8, device=dev)
            # ---- throughput mode (never `value`): 16 independent strips in flight on this GPU (one batch: the 16 pairs share every kernel
            # launch), through the C ABI's batch entry ----
            nb, infl = 16, 16
            pairs_b = [synth.make_pair(sc, sr, 5000 + i, dev) for i in range(nb)]
            outs_b = [torch.empty_like(os_) for _ in range(nb)]
            torch.cuda.synchronize()
            ct = pf.Context(local_rank, sc, sr)          # its own context: the lanes it creates go away with it
            call_b = lambda: ct.novel_view_batch_dev([p[0].data_ptr() for p in pairs_b], [p[1].data_ptr() for p in pairs_b], sc, sr, max_pct,
                                                      [p[2].data_ptr() for p in pairs_b], [o.data_ptr() for o in outs_b], None, None, in_flight=infl)
            call_b()
            tbs = []
            for _ in range(3):
                t1 = time.perf_counter(); call_b(); tbs.append(time.perf_counter() - t1)
            tb = statistics.median(tbs)
            res["throughput_mode"] = {"value": round(nb * sc * sr / 1e6 / tb, 3), "unit": "Mpix/s", "pairs": nb, "in_flight": infl, "entry": "pf_novel_view_batch_dev",
                                      "workload": "16 independent 2000x4000 strips in flight: one batch, all 16 through each set of launches (blockIdx.z = pair)", "runs": 3, "warmup": 1, "statistic": "median",
                                      "note": "several independent pairs side by side on one GPU; an extra figure, not the per-GPU workload `value` is quoted on"}
            del pairs_b, outs_b, os_
            ct.close()
            torch.cuda.empty_cache()
            # ... and the same on the per-GPU workload's size: dense 9000x4000 pairs, 8 in flight (one batch: the round-3 figure's configuration),
            # 16 (one batch of 16) and 32 (two lanes x 16: where the chip saturates)
            pairs_c, outs_c = [], []
            ct = pf.Context(local_rank)
            for nfl in (8, 16, 32):
                pairs_c += [synth.make_pair(cols, rows, 6000 + i, dev)[:3] for i in range(len(pairs_c), nfl)]
                outs_c += [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(len(outs_c), nfl)]
                torch.cuda.synchronize()
                call_c = lambda: ct.novel_view_batch_dev([p[0].data_ptr() for p in pairs_c], [p[1].data_ptr() for p in pairs_c], cols, rows, max_pct,
                                                          [p[2].data_ptr() for p in pairs_c], [o.data_ptr() for o in outs_c], None, None, in_flight=nfl)
                call_c()
                tcs = []
                for _ in range(3):
                    t1 = time.perf_counter(); call_c(); tcs.append(time.perf_counter() - t1)
                tcm = statistics.median(tcs)
                key = "pairs_%dx%d" % (cols, rows) + ("" if nfl == 8 else "_%d_in_flight" % nfl)
                res["throughput_mode"][key] = {"value": round(nfl * mpix / tcm, 3), "unit": "Mpix/s", "pairs": nfl, "in_flight": nfl, "runs": 3, "warmup": 1,
                                               "statistic": "median", "roofline_path_frac": round(nfl * b_alg / tcm / 8e12, 6)}
            del pairs_c, outs_c
            # ---- large displacements (round-4 review, next #2): the same dense pair with synth's displacement field scaled x4 / x8 (up to 18 / 36 px
            # at the solver's half resolution; real rig parallax is why the reference has pixflow_search_20 at all): a lone pair and 16 in flight.
            # The sweeps' LDS gather window follows the flow since round 5; before, x8 cost +47 % (lone pair) / +45 % (batch). ----
            ld = {"x1": {"lone_pair_ms": round(med_ms, 3), "in_flight_16_ms_per_pair": round(1000 * mpix / res["throughput_mode"]["pairs_%dx%d_16_in_flight" % (cols, rows)]["value"], 3)}}
            cl = pf.Context(local_rank, cols, rows)
            for sc in (4, 8):
                Lx, Rx, bx, _ = synth.make_pair(cols, rows, 1234, dev, disp_scale=float(sc))
                torch.cuda.synchronize()
                one_l = lambda: cl.novel_view_dev(Lx.data_ptr(), Rx.data_ptr(), cols, rows, max_pct, bx.data_ptr(), out.data_ptr(), f0.data_ptr(), f1.data_ptr())
                one_l()
                tl = []
                for _ in range(5):
                    t1 = time.perf_counter(); one_l(); tl.append(time.perf_counter() - t1)
                fmax = float(f0.abs().max())
                del Lx, Rx, bx
                pairs_d = [synth.make_pair(cols, rows, 6000 + i, dev, disp_scale=float(sc))[:3] for i in range(16)]
                outs_d = [torch.empty((rows, cols, 4), dtype=torch.uint8, device=dev) for _ in range(16)]
                torch.cuda.synchronize()
                call_d = lambda: ct.novel_view_batch_dev([p[0].data_ptr() for p in pairs_d], [p[1].data_ptr() for p in pairs_d], cols, rows, max_pct,
                                                          [p[2].data_ptr() for p in pairs_d], [o.data_ptr() for o in outs_d], None, None, in_flight=16)
                call_d()
                td = []
                for _ in range(3):
                    t1 = time.perf_counter(); call_d(); td.append(time.perf_counter() - t1)
                ld["x%d" % sc] = {"lone_pair_ms": round(1000 * statistics.median(tl), 3), "in_flight_16_ms_per_pair": round(1000 * statistics.median(td) / 16, 3),
                                  "max_abs_flow_px_full_res": round(fmax, 1)}
                del pairs_d, outs_d
                torch.cuda.empty_cache()
            cl.close()
            for k in ("x4", "x8"):
                ld[k]["lone_pair_vs_x1"] = round(ld[k]["lone_pair_ms"] / ld["x1"]["lone_pair_ms"], 4)
                ld[k]["in_flight_16_vs_x1"] = round(ld[k]["in_flight_16_ms_per_pair"] / ld["x1"]["in_flight_16_ms_per_pair"], 4)
            ld["note"] = "synth.make_pair(disp_scale = 4, 8): the dense pair's analytic displacement field scaled; lone pair = median of 5 calls of pf_novel_view_dev, 16 in flight = median of 3 calls of pf_novel_view_batch_dev"
            res["large_displacement"] = ld
            ct.close()
            torch.cuda.empty_cache()
        line = json.dumps(res)
    if pfd:
        pfd.close()
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
