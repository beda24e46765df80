#!/usr/bin/env python
"""Golden fixture for BASELINE config 4: the full 5+top stitch chain at 9000x4000, pixflow_search_20
(CPU/main.cpp:60-105), computed by the CPU oracle in the build container.

The oracle needs minutes per step at this size, so the GPU tier cannot run it; this script runs it ONCE
here and stores what the `-m gpu` test (tests/test_gpu_fullsize.py::test_config4_chain_vs_oracle_fixture)
needs to hold the HIP path to it:

  * SHA-256 of every synthetic input canvas (the test regenerates them and refuses to compare on a mismatch),
  * step 1 (identical inputs on both sides): SHA-256 of map, blend ramp, MergedDis and of both flow fields --
    the HIP path must reproduce these BIT FOR BIT -- and of the step's composite,
  * every step: SHA-256 of the oracle composite (provenance) + a stride-8 subsample of it (what PSNR and the
    <=1-LSB bound are evaluated on; 562,500 sample pixels per step).

Run:  python tests/golden/make_chain_golden.py [cols rows]     (writes tests/golden/chain_<cols>x<rows>.npz)
"""
import hashlib
import importlib.util
import os
import sys
import threading
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import orc  # noqa: E402

STRIDE = 8
SEED = 1234
MAX_PCT = 20


def load_synth():
    spec = importlib.util.spec_from_file_location("pano_amd_synth", os.path.join(ROOT, "panorama-opticalflow_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9000, 4000)
    orc.build()
    synth = load_synth()
    t0 = time.time()
    top, imgs = synth.make_stitch_set(cols, rows, SEED, 5, "cpu")
    top = top.numpy(); imgs = [im.numpy() for im in imgs]
    print("inputs generated in %.0f s" % (time.time() - t0), flush=True)
    out = {"cols": cols, "rows": rows, "seed": SEED, "max_pct": MAX_PCT, "stride": STRIDE,
           "sha_inputs": np.array([sha(top)] + [sha(im) for im in imgs])}
    R = top
    sha_steps = []
    for i, L in enumerate(imgs):
        t1 = time.time()
        mp, ovl, ovr, blend, md = orc.stitch_prepare(L, R, True)
        res = [None, None]

        def run(d):
            res[d] = orc.flow_one_dir(ovl, ovr, MAX_PCT, d)

        th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]
        [t.start() for t in th]; [t.join() for t in th]
        merged = orc.combine_novel_views(ovl, ovr, res[0], res[1], blend)
        R = orc.stitch_gather(L, R, merged, mp)
        if i == 0:
            out["sha_step1"] = np.array([sha(mp), sha(blend), sha(md), sha(res[0]), sha(res[1]), sha(merged)])
            out["merged1_sub"] = merged[::STRIDE, ::STRIDE].copy()
        sha_steps.append(sha(R))
        out["final%d_sub" % (i + 1)] = R[::STRIDE, ::STRIDE].copy()
        print("step %d: %.0f s, sha %s" % (i + 1, time.time() - t1, sha_steps[-1][:16]), flush=True)
    out["sha_final"] = np.array(sha_steps)
    path = os.path.join(HERE, "chain_%dx%d.npz" % (cols, rows))
    np.savez_compressed(path, **out)
    print("wrote %s (%.1f MB) in %.0f s" % (path, os.path.getsize(path) / 1e6, time.time() - t0))


if __name__ == "__main__":
    main()
