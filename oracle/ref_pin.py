#!/usr/bin/env python
"""Pins the oracle to the REAL reference -- on a box where `make -C oracle ref OPENCV_PREFIX=...` could build it
(OpenCV 3.2 + gflags + glog; not this image, see oracle/Makefile).  TEST INFRASTRUCTURE ONLY.

Writes a synthetic Test_data directory (top.tif + 1..5.tif, the files CPU/main.cpp:60-105 reads), runs
oracle/_ref/pano_ref on it for both algorithms, runs the oracle on the same images and
  * reports per-step PSNR / max LSB difference between the two (the oracle's claim is: identical),
  * stores the reference's own FinalResult as tests/golden/ref_<alg>_<cols>x<rows>.npz, i.e. fixtures produced by the
    reference itself, which tests/test_golden.py then holds the oracle (and through it the HIP path) to.
Until this has run somewhere, DESIGN.md / the oracle header say "parity unpinned".
"""
import importlib.util
import os
import subprocess
import sys
import tempfile

import numpy as np
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
import orc  # noqa: E402

EXE = os.path.join(HERE, "_ref", "pano_ref")


def main():
    cols, rows = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1800, 800)
    if not os.path.exists(EXE):
        raise SystemExit("oracle/_ref/pano_ref is not built: the reference needs OpenCV 3.2 + gflags + glog (make -C oracle ref OPENCV_PREFIX=...)")
    spec = importlib.util.spec_from_file_location("pano_amd_synth", os.path.join(ROOT, "panorama-opticalflow_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
    orc.build()
    top, imgs = synth.make_stitch_set(cols, rows, 1234, 5, "cpu")
    top = top.numpy(); imgs = [im.numpy() for im in imgs]
    for alg, pct in (("pixflow_low", 0), ("pixflow_search_20", 20)):
        with tempfile.TemporaryDirectory() as d:
            Image.fromarray(top[..., [2, 1, 0, 3]], "RGBA").save(os.path.join(d, "top.tif"))
            for i, im in enumerate(imgs):
                Image.fromarray(im[..., [2, 1, 0, 3]], "RGBA").save(os.path.join(d, "%d.tif" % (i + 1)))
            subprocess.check_call([EXE, "-test_dir", d, "-top_img", "top.tif", "-flow_alg", alg])
            ref = np.array(Image.open(os.path.join(d, "FinalResult.png")))[..., [2, 1, 0, 3]]
        R = top
        for L in imgs:
            mp, ovl, ovr, blend, _ = orc.stitch_prepare(L, R, True)
            f0, f1 = orc.flow_bidir(ovl, ovr, pct)
            R = orc.stitch_gather(L, R, orc.combine_novel_views(ovl, ovr, f0, f1, blend), mp)
        dlt = np.abs(ref.astype(np.int32) - R.astype(np.int32))
        mse = float(np.mean(dlt.astype(np.float64) ** 2))
        print("%s: reference vs oracle after 5 steps: max |d| = %d LSB, pixels differing = %d, PSNR = %s dB" %
              (alg, dlt.max(), int((dlt > 0).sum()), "inf" if mse == 0 else "%.1f" % (10 * np.log10(255.0 ** 2 / mse))))
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", "ref_%s_%dx%d.npz" % (alg, cols, rows)), final=ref, cols=cols, rows=rows, seed=1234)


if __name__ == "__main__":
    main()
