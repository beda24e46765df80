#!/usr/bin/env python
"""Golden fixture for BASELINE config 5's per-GPU workload: ONE DENSE 9000x4000 overlap pair (seed 1234, pixflow_low,
the pair bench.py times on every GPU), solved by the CPU oracle in the build container.

The oracle needs minutes per direction at this size, so the GPU tier cannot run it; this script runs it ONCE here
and stores what tests/test_gpu_fullsize.py::test_dense_canvas_pair_vs_oracle_fixture needs:

  * SHA-256 of the synthetic inputs (L, R, blend) -- the test regenerates them and refuses to compare on a mismatch,
  * SHA-256 of both flow fields and of the blended strip (NovelViewGeneratorAsymmetricFlow::prepare +
    combineNovelViews, CPU/OpticalFlow.cpp:30-145) -- the HIP path must reproduce all three BIT FOR BIT,
  * a stride-16 subsample of the three outputs (to say WHERE a mismatch is, should there ever be one).

Run:  python tests/golden/make_dense_golden.py [cols rows [alg [seed]]]   (writes tests/golden/dense_<cols>x<rows>.npz)

With a seed other than 1234 (config 5's other pairs: rank r of an N-GPU run solves seed 1234 + r) the fixture holds the
SHA-256 strings only and is written to dense_<cols>x<rows>_s<seed>.npz (a few hundred bytes): enough for every rank of a
multi-GPU bench / pano_batch run to validate its own pair.
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

STRIDE = 16
SEED = 1234


def load_synth():
    spec = importlib.util.spec_from_file_location("pano_amd_synth", os.path.join(ROOT, "panorama-opticalflow_amd", "synth.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def write_sidecar(cols=9000, rows=4000):
    """dense_<cols>x<rows>.sha256.txt: the fixtures' hashes as plain text (one line per seed), for the C++ driver (pano_batch -golden)."""
    import glob
    rowsf = []
    for f in glob.glob(os.path.join(HERE, "dense_%dx%d*.npz" % (cols, rows))):
        g = np.load(f)
        rowsf.append((int(g["seed"]), [str(v) for v in g["sha_inputs"]] + [str(v) for v in g["sha_outputs"]]))
    with open(os.path.join(HERE, "dense_%dx%d.sha256.txt" % (cols, rows)), "w") as o:
        o.write("# BASELINE config 5 (8 dense 9000x4000 pairs, seeds 1234..1241, pixflow_low): SHA-256 of the synthetic inputs (L, R, blend ramp) and of the\n"
                "# oracle's outputs (flow L->R, flow R->L, blended strip), copied from dense_9000x4000[_s<seed>].npz by tests/golden/make_dense_golden.py --sidecar\n"
                "# (plain text so that the C++ driver can read it: pano_batch -golden <dir>).  seed  sha_L sha_R sha_blend  sha_flow_l2r sha_flow_r2l sha_strip\n")
        for seed, sh in sorted(rowsf):
            o.write("%d %s\n" % (seed, " ".join(sh)))


def main():
    if "--sidecar" in sys.argv:
        return write_sidecar()
    cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (9000, 4000)
    max_pct = {"pixflow_low": 0, "pixflow_search_20": 20}[sys.argv[3] if len(sys.argv) > 3 else "pixflow_low"]
    seed = int(sys.argv[4]) if len(sys.argv) > 4 else SEED
    orc.build()
    synth = load_synth()
    t0 = time.time()
    L, R, blend = synth.make_pair_np(cols, rows, seed)
    print("inputs generated in %.0f s" % (time.time() - t0), flush=True)
    res = [None, None]

    def run(d):
        t1 = time.time()
        res[d] = orc.flow_one_dir(L, R, max_pct, d)
        print("direction %d: %.0f s" % (d, time.time() - t1), flush=True)

    th = [threading.Thread(target=run, args=(d,)) for d in (0, 1)]
    [t.start() for t in th]; [t.join() for t in th]
    out = orc.combine_novel_views(L, R, res[0], res[1], blend)
    fix = {"cols": cols, "rows": rows, "seed": seed, "max_pct": max_pct, "stride": STRIDE,
           "sha_inputs": np.array([sha(L), sha(R), sha(blend)]),
           "sha_outputs": np.array([sha(res[0]), sha(res[1]), sha(out)])}
    if seed == SEED:
        fix.update({"flow_l2r_sub": res[0][::STRIDE, ::STRIDE].copy(), "flow_r2l_sub": res[1][::STRIDE, ::STRIDE].copy(),
                    "out_sub": out[::STRIDE, ::STRIDE].copy()})
    path = os.path.join(HERE, "dense_%dx%d%s.npz" % (cols, rows, "" if seed == SEED else "_s%d" % seed))
    np.savez_compressed(path, **fix)
    print("wrote %s (%.1f MB) in %.0f s; max |flow| = %.2f px" % (path, os.path.getsize(path) / 1e6, time.time() - t0,
                                                                  float(max(np.abs(res[0]).max(), np.abs(res[1]).max()))))


if __name__ == "__main__":
    main()
