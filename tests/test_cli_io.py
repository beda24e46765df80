"""The drop-in CLI: TIFF/PNG I/O on the CPU, and the whole -test_dir/-top_img/-flow_alg chain
(CPU/main.cpp:47-110) on the GPU against the oracle running the same chain."""
import os
import subprocess

import numpy as np
import pytest
from PIL import Image

from conftest import PKG, ROOT

EXE = os.path.join(PKG, "tools", "pano_stitch")


def _bgra_to_rgba(a):
    return a[..., [2, 1, 0, 3]]


def _oracle_chain(orc, top, imgs, alg_pct):
    R = top; steps = []
    for L in imgs:
        mp, ovl, ovr, blend, _ = orc.stitch_prepare(L, R, True)
        f0, f1 = orc.flow_bidir(ovl, ovr, alg_pct)
        merged = orc.combine_novel_views(ovl, ovr, f0, f1, blend)
        R = orc.stitch_gather(L, R, merged, mp)
        steps.append(R)
    return steps


def _psnr(a, b):
    mse = np.mean((a.astype(np.float64) - b.astype(np.float64)) ** 2)
    return 99.0 if mse == 0 else 10 * np.log10(255.0 ** 2 / mse)


def test_cli_rejects_missing_flags_and_bad_files(tmp_path):
    if not os.path.exists(EXE):
        pytest.skip("tools/pano_stitch not built")
    r = subprocess.run([EXE, "-test_dir", str(tmp_path)], capture_output=True, text=True)
    assert r.returncode != 0 and "missing required command line argument" in r.stderr   # CPU/util.hpp:45-49
    r = subprocess.run([EXE, "-test_dir", str(tmp_path), "-top_img", "nope.tif", "-flow_alg", "pixflow_low"], capture_output=True, text=True)
    assert r.returncode != 0 and "failed to load image" in r.stderr                       # CPU/util.cpp:22


@pytest.mark.gpu
@pytest.mark.parametrize("compression,fused", [(None, "1"), ("tiff_lzw", "0"), ("tiff_adobe_deflate", "1")])
def test_cli_three_step_chain(tmp_path, orc, synth, compression, fused):
    cols, rows, n = 480, 320, 3
    top, imgs = synth.make_stitch_set(cols, rows, 77, 5)
    top = top.numpy(); imgs = [im.numpy() for im in imgs[:n]]
    kw = {} if compression is None else {"compression": compression}
    Image.fromarray(_bgra_to_rgba(top), "RGBA").save(tmp_path / "top.tif", **kw)
    for i, im in enumerate(imgs):
        Image.fromarray(_bgra_to_rgba(im), "RGBA").save(tmp_path / ("%d.tif" % (i + 1)), **kw)
    out = subprocess.run([EXE, "-test_dir", str(tmp_path), "-top_img", "top.tif", "--flow_alg=pixflow_search_20", "-steps", str(n), "-fused", fused],
                         capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    assert "Part1 Finished!RUNTIME (sec) = " in out.stdout and "TotalRunTime (sec) = " in out.stdout
    rd = lambda name: np.array(Image.open(tmp_path / name))[..., [2, 1, 0, 3]]
    ref = _oracle_chain(orc, top, imgs, 20)
    # every step of the chain is byte-identical to the oracle chain: the flows were bit-exact all along, and the blend now
    # evaluates tanhf / exp with the host libm's roundings (csrc/libm_exact.hpp), so no LSB is left for the next step to amplify
    for name, r in zip(("ProcessResult1.png", "ProcessResult2.png", "FinalResult.png"), ref):
        got = rd(name)
        assert np.array_equal(got, r), "%s: %d bytes differ, PSNR %.1f dB" % (name, int((got != r).sum()), _psnr(got, r))


@pytest.mark.gpu
def test_cli_four_input_mode(tmp_path, orc, synth):
    """CPU_4Input/main.cpp:54-113: centre-row alpha crop, 1+3 -> L, 2+4 -> R (saturating), one stitch step."""
    cols, rows = 480, 320
    _, imgs = synth.make_stitch_set(cols, rows, 31, 4)
    imgs = [im.numpy().copy() for im in imgs]
    imgs[0][: rows // 2 + 5, 40:60, 3] = 255; imgs[0][rows // 2 + 5:, 40:60] = 0   # columns the centre-row crop must clear
    for i, im in enumerate(imgs):
        Image.fromarray(_bgra_to_rgba(im), "RGBA").save(tmp_path / ("%d.tif" % (i + 1)))
    out = subprocess.run([EXE, "-inputs", "4", "-test_dir", str(tmp_path), "-flow_alg", "pixflow_low"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    cropped = []
    for im in imgs:
        c = im.copy(); c[:, im[rows // 2, :, 3] == 0] = 0; cropped.append(c.astype(np.int32))
    L = np.clip(cropped[0] + cropped[2], 0, 255).astype(np.uint8); R = np.clip(cropped[1] + cropped[3], 0, 255).astype(np.uint8)
    ref = _oracle_chain(orc, R, [L], 0)[0]
    got = np.array(Image.open(tmp_path / "FinalResult.png"))[..., [2, 1, 0, 3]]
    assert np.array_equal(got, ref), "%d bytes differ" % int((got != ref).sum())
