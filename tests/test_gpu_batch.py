"""Multi-GPU path on the GPU tier, run with one rank (the only size a 1-GPU box offers): the C++ batch driver end to end
(compute -> RCCL gather into rank 0 -> checksum of every gathered strip against its producer's) and the pf_dist_* ABI."""
import json
import os
import subprocess

import numpy as np
import pytest

from conftest import PKG

pytestmark = pytest.mark.gpu


def test_pano_batch_one_gpu_three_pairs():
    exe = os.path.join(PKG, "tools", "pano_batch")
    r = subprocess.run([exe, "-pairs", "3", "-size", "640x480", "-flow_alg", "pixflow_search_20", "-gpus", "1", "-verify", "1"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["gpus"] == 1 and res["pairs"] == 3 and res["verified_pairs"] == 3 and res["Mpix/s"] > 0


def test_pano_batch_in_flight_matches_one_at_a_time():
    """-in_flight K: a GPU solves K of its pairs through one set of launches per round and gathers them as one block; with a ragged
    last round.  The gathered strips verify against their producers' checksums, like the one-pair-per-round run."""
    exe = os.path.join(PKG, "tools", "pano_batch")
    out = {}
    for k in ("1", "3"):
        r = subprocess.run([exe, "-pairs", "5", "-size", "640x480", "-flow_alg", "pixflow_low", "-gpus", "1", "-verify", "1", "-in_flight", k], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        out[k] = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
        assert out[k]["pairs"] == 5 and out[k]["verified_pairs"] == 5 and out[k]["in_flight_per_gpu"] == int(k)


def test_pf_dist_self_gather_and_max(pf):
    c = pf.Context(0)
    d = pf.Dist(0, pf.dist_unique_id(), 0, 1)
    n = 1 << 20
    src = np.random.default_rng(3).integers(0, 255, n, dtype=np.uint8)
    a = c.dev_alloc(n); b = c.dev_alloc(n)
    c.upload(a, src)
    d.gather_async(a, b, n)
    d.wait()
    got = c.download(np.empty(n, np.uint8), b)
    assert np.array_equal(got, src)
    assert d.max(3.5) == 3.5
    d.barrier()
    d.close(); c.dev_free(a); c.dev_free(b); c.close()
