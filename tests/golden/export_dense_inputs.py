#!/usr/bin/env python
"""Writes the synthetic inputs of config 5's pairs as raw files for the C++ driver (pano_batch -golden <dir>), which cannot call
synth.py: <dir>/pair_<seed>_L.bgra, pair_<seed>_R.bgra (cols x rows x 4 bytes) and blend.f32 (cols x rows floats, the same ramp
for every pair), plus a copy of dense_<cols>x<rows>.sha256.txt.  Generated on the GPU when there is one (the host's bytes on any
device, synth.py).   export_dense_inputs.py <dir> [n_pairs [cols rows]]"""
import importlib.util
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    out = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    cols, rows = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (9000, 4000)
    spec = importlib.util.spec_from_file_location("pano_amd_synth", os.path.join(ROOT, "panorama-opticalflow_amd", "synth.py"))
    synth = importlib.util.module_from_spec(spec); spec.loader.exec_module(synth)
    import torch
    dev = "cuda" if torch.cuda.is_available() else "cpu"
    os.makedirs(out, exist_ok=True)
    for p in range(n):
        L, R, blend, _ = synth.make_pair(cols, rows, 1234 + p, dev)
        L.cpu().numpy().tofile(os.path.join(out, "pair_%d_L.bgra" % (1234 + p)))
        R.cpu().numpy().tofile(os.path.join(out, "pair_%d_R.bgra" % (1234 + p)))
        if p == 0:
            blend.cpu().numpy().tofile(os.path.join(out, "blend.f32"))
    side = os.path.join(HERE, "dense_%dx%d.sha256.txt" % (cols, rows))
    if os.path.exists(side):
        shutil.copy(side, out)


if __name__ == "__main__":
    main()
