"""Regenerates tests/golden/*.npz: small seeded inputs and the ORACLE's outputs for them.

The reference ships no golden vectors (its Test_data is absent) and cannot be built here (OpenCV 3.2,
gflags, glog missing), so these fixtures pin the oracle against drift -- they are produced by
oracle/pixflow_oracle.cpp, not by the reference.  Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from conftest import load_pkg_module  # noqa: E402
import orc  # noqa: E402

synth = load_pkg_module("synth")
orc.build()


def main():
    # 1. flow + blend, both presets
    cols, rows = 160, 128
    L, R, blend = synth.make_pair_np(cols, rows, 1234)
    out = {"L": L, "R": R, "blend": blend}
    for name, mp in (("low", 0), ("s20", 20)):
        f0, f1 = orc.flow_bidir(L, R, mp)
        out["flowLR_" + name] = f0; out["flowRL_" + name] = f1
        out["merged_" + name] = orc.combine_novel_views(L, R, f0, f1, blend)
    np.savez_compressed(os.path.join(HERE, "flow_160x128.npz"), **out)
    # 2. StitchTool: map / masks / ramp / composite
    cols, rows = 240, 200
    Lc, Rc = synth.make_canvas_pair(cols, rows, 21)
    Lc, Rc = Lc.numpy(), Rc.numpy()
    mp_, ovl, ovr, bl, md = orc.stitch_prepare(Lc, Rc, True)
    merged = np.where((mp_ == 150)[..., None], ovl, 0).astype(np.uint8)
    merged[100:106, 110:130] = 0
    final = orc.stitch_gather(Lc, Rc, merged, mp_)
    np.savez_compressed(os.path.join(HERE, "stitch_240x200.npz"), L=Lc, R=Rc, map=mp_, blend=bl, mergedDis=md, merged=merged, final=final)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
