"""Throughput mode and its safety: several pairs in flight on one GPU must give the very
bits a lone pair gives, every time, with every wait inside the sweep kernels bounded by wall-clock time (oversubscription
must never surface as PF_ERR_TIMEOUT).  The former tests/gpu_soak.py script, as collected tests."""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _dev_pair(pf, c, synth, cols, rows, seed):
    L, R, blend = synth.make_pair_np(cols, rows, seed)
    n = cols * rows
    d = {"L": c.dev_alloc(n * 4), "R": c.dev_alloc(n * 4), "b": c.dev_alloc(n * 4), "o": c.dev_alloc(n * 4), "f0": c.dev_alloc(n * 8), "f1": c.dev_alloc(n * 8)}
    c.upload(d["L"], L); c.upload(d["R"], R); c.upload(d["b"], blend)
    return d


def _fetch(c, d, cols, rows):
    return (c.download(np.empty((rows, cols, 4), np.uint8), d["o"]), c.download(np.empty((rows, cols, 2), np.float32), d["f0"]),
            c.download(np.empty((rows, cols, 2), np.float32), d["f1"]))


@pytest.mark.parametrize("in_flight,batch_pairs,wide", [(4, -1, -1), (6, -1, -1), (6, 1, -1), (5, 5, -1), (7, 2, -1), (8, 8, -1), (8, 8, 1), (5, 5, 1), (6, 3, 1), (8, 8, 2), (5, 5, 2), (6, 3, 2), (12, 12, 2), (12, 12, -1)])
def test_batch_entry_point_equals_single_calls(pf, synth, in_flight, batch_pairs, wide):
    """pf_novel_view_batch_dev = lanes x batches: pairs of a batch share every kernel launch (blockIdx.z = pair, slab buffers, one
    sweep window = the union of the pairs' windows), lanes run side by side.  Whatever the split -- also with a ragged last batch,
    with flows the caller does not want, and with pairs whose gates differ -- the results are the bits of single calls.
    wide = 1 / 2: every sweep launch of the batch in the wide workgroup shape / in the throughput form (pf_config::sweep_wide; by default
    only launches that oversubscribe the chip take the throughput form), held against single calls in the latency form."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    cols, rows, n = 1000, 1400, (12 if in_flight > 8 else 7)
    c = pf.Context(0, exp=wide == 1, batch_pairs=batch_pairs, sweep_wide=wide)   # form 1 only exists in the lab build
    ref_ctx = pf.Context(0, sweep_wide=0)
    pairs = [_dev_pair(pf, c, synth, cols, rows, 100 + i) for i in range(n)]
    # pair 2: an extra hole in the alpha of both images -> a different gate / bounding box than its batch mates
    L2, R2, _ = synth.make_pair_np(cols, rows, 102)
    L2[:, :300, 3] = 0; R2[:, :300, 3] = 0; L2[900:, :, 3] = 0; R2[900:, :, 3] = 0
    c.upload(pairs[2]["L"], L2); c.upload(pairs[2]["R"], R2)
    ref = []
    for d in pairs:
        ref_ctx.novel_view_dev(d["L"], d["R"], cols, rows, 20, d["b"], d["o"], d["f0"], d["f1"])
        ref.append(_fetch(ref_ctx, d, cols, rows))
    ref_ctx.close()
    for rep in range(2):
        for d in pairs:   # scrub the outputs so a stale result cannot pass
            c.upload(d["o"], np.zeros((rows, cols, 4), np.uint8)); c.upload(d["f0"], np.zeros((rows, cols, 2), np.float32))
        want_flows = rep == 0
        c.novel_view_batch_dev([d["L"] for d in pairs], [d["R"] for d in pairs], cols, rows, 20, [d["b"] for d in pairs], [d["o"] for d in pairs],
                               [d["f0"] for d in pairs] if want_flows else None, [d["f1"] for d in pairs] if want_flows else None, in_flight=in_flight)
        for d, r in zip(pairs, ref):
            got = _fetch(c, d, cols, rows)
            assert np.array_equal(got[0], r[0])
            if want_flows:
                assert np.array_equal(got[1], r[1]) and np.array_equal(got[2], r[2])
    c.close()


def test_two_contexts_soak_bit_identical(pf, synth):
    """two host threads, two contexts, different sizes and algorithms, 60 solves each: every result equals the first"""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    specs = [(2000, 4000, 0, 1234), (1500, 3000, 20, 77)]
    bad = [0, 0]; runs = [0, 0]; errs = []

    def work(k):
        try:
            cols, rows, pct, seed = specs[k]
            c = pf.Context(0, cols, rows)
            d = _dev_pair(pf, c, synth, cols, rows, seed)
            first = None
            for _ in range(60):
                c.novel_view_dev(d["L"], d["R"], cols, rows, pct, d["b"], d["o"], d["f0"], d["f1"])
                got = _fetch(c, d, cols, rows)
                h = tuple(int(a.view(np.uint8).astype(np.uint64).sum()) for a in got) + (hash(got[1].tobytes()),)
                if first is None:
                    first = h
                bad[k] += h != first; runs[k] += 1
            c.close()
        except Exception as e:   # a PF_ERR_TIMEOUT would land here
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    assert runs == [60, 60] and bad == [0, 0]


def test_oversubscribed_gpu_does_not_time_out(pf, synth):
    """8 pairs in flight on one GPU (beyond what is co-resident at the large levels): slow is fine, PF_ERR_TIMEOUT is not"""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    cols, rows, n = 2000, 4000, 8
    c = pf.Context(0)
    d0 = _dev_pair(pf, c, synth, cols, rows, 1234)
    outs = [(c.dev_alloc(cols * rows * 4), c.dev_alloc(cols * rows * 8), c.dev_alloc(cols * rows * 8)) for _ in range(n)]
    c.novel_view_dev(d0["L"], d0["R"], cols, rows, 0, d0["b"], d0["o"], d0["f0"], d0["f1"])
    ref = _fetch(c, d0, cols, rows)
    c.novel_view_batch_dev([d0["L"]] * n, [d0["R"]] * n, cols, rows, 0, [d0["b"]] * n, [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], in_flight=8)
    for o in outs:
        got = _fetch(c, {"o": o[0], "f0": o[1], "f1": o[2]}, cols, rows)
        assert all(np.array_equal(a, b) for a, b in zip(got, ref))
    c.close()


def test_batches_of_changing_size_on_one_context(pf, synth):
    """the slabs of a batch are re-laid-out when the image size (or the batch size) changes: sizes A, B, A, then a larger batch, on one context,
    every result equal to a single call's"""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    c = pf.Context(0)
    ref_ctx = pf.Context(0)
    for (cols, rows, n, infl) in ((640, 480, 3, 3), (512, 704, 4, 4), (640, 480, 3, 3), (512, 704, 6, 6), (336, 272, 8, 8)):
        pairs = [_dev_pair(pf, c, synth, cols, rows, 300 + 7 * i + cols) for i in range(n)]
        ref = []
        for d in pairs:
            ref_ctx.novel_view_dev(d["L"], d["R"], cols, rows, 0, d["b"], d["o"], d["f0"], d["f1"])
            ref.append(_fetch(ref_ctx, d, cols, rows))
            c.upload(d["o"], np.zeros((rows, cols, 4), np.uint8))
        c.novel_view_batch_dev([d["L"] for d in pairs], [d["R"] for d in pairs], cols, rows, 0, [d["b"] for d in pairs], [d["o"] for d in pairs],
                               [d["f0"] for d in pairs], [d["f1"] for d in pairs], in_flight=infl)
        for d, r in zip(pairs, ref):
            got = _fetch(c, d, cols, rows)
            assert all(np.array_equal(a, b) for a, b in zip(got, r)), (cols, rows, n)
        for d in pairs:
            for k in d.values():
                c.dev_free(k)
    c.close(); ref_ctx.close()


def test_throughput_form_soak_is_deterministic(pf, synth):
    """The throughput form's compute waves request their records by LDS-DMA and count the completions by hand (kernels_sweep_t.inl):
    a miscounted wait would show up as a rare wrong record, not as a steady failure.  40 batches of 8 pairs with every sweep in that
    form (sweep_wide = 2), plus 6 batches of four 4000x2400 pairs (several workgroups per sweep, granule hand-offs): every output must
    equal the first batch's, which must equal single calls in the latency form."""
    import hashlib
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    for (cols, rows, n, reps) in ((1000, 1400, 8, 40), (4000, 2400, 4, 6)):
        c = pf.Context(0, sweep_wide=2)
        ref_ctx = pf.Context(0, sweep_wide=0)
        pairs = [_dev_pair(pf, c, synth, cols, rows, 500 + i) for i in range(n)]
        want = []
        for d in pairs:
            ref_ctx.novel_view_dev(d["L"], d["R"], cols, rows, 0, d["b"], d["o"], d["f0"], d["f1"])
            want.append(tuple(hashlib.sha256(a.tobytes()).hexdigest() for a in _fetch(ref_ctx, d, cols, rows)))
        ref_ctx.close()
        for rep in range(reps):
            c.novel_view_batch_dev([d["L"] for d in pairs], [d["R"] for d in pairs], cols, rows, 0, [d["b"] for d in pairs], [d["o"] for d in pairs],
                                   [d["f0"] for d in pairs], [d["f1"] for d in pairs], in_flight=n)
            got = [tuple(hashlib.sha256(a.tobytes()).hexdigest() for a in _fetch(c, d, cols, rows)) for d in pairs]
            assert got == want, "repetition %d of %dx%d differs" % (rep, cols, rows)
        for d in pairs:
            for k in d.values():
                c.dev_free(k)
        c.close()
