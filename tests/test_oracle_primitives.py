"""Pins the oracle's restated OpenCV-3.2 primitives against INDEPENDENT implementations available in
this image (scipy.ndimage, torch.nn.functional.interpolate).  The reference has no tests or golden
vectors and OpenCV is not installed, so this is the strongest pin available ("parity unpinned")."""
import numpy as np
import pytest
import scipy.ndimage as ndi
import torch
import torch.nn.functional as F

rng = np.random.default_rng(7)


def test_gaussian_kernel_closed_form(orc):
    for n, s in [(5, 0.25), (3, 0.5), (3, 1.0), (15, 8.0)]:
        k = orc.gaussian_kernel(n, s)
        x = np.arange(n) - (n - 1) / 2
        ref = np.exp(-0.5 * x * x / (s * s)); ref /= ref.sum()
        assert np.allclose(k, ref, rtol=0, atol=2e-7)
        assert abs(float(k.astype(np.float64).sum()) - 1.0) < 3e-7


@pytest.mark.parametrize("ksize,sigma,cn", [(5, 0.25, 1), (3, 0.5, 1), (3, 1.0, 2), (15, 8.0, 2)])
def test_gaussian_blur_vs_scipy_mirror(orc, ksize, sigma, cn):
    img = rng.standard_normal((37, 53, cn)).astype(np.float32)
    got = orc.gaussian_blur(img, ksize, sigma)
    k = orc.gaussian_kernel(ksize, sigma).astype(np.float64)
    ref = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(got - ref).max() < 2e-6


def test_gaussian_blur_tiny_image_reflect(orc):
    img = rng.standard_normal((5, 4, 1)).astype(np.float32)  # narrower than the 15-tap kernel
    got = orc.gaussian_blur(img, 15, 8.0)
    k = orc.gaussian_kernel(15, 8.0).astype(np.float64)
    ref = ndi.correlate1d(ndi.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0, mode="mirror")
    assert np.abs(got - ref).max() < 2e-6


def test_sobel_replicate(orc):
    img = rng.standard_normal((20, 31)).astype(np.float32)
    gx = orc.sobel1(img, 1, 0); gy = orc.sobel1(img, 0, 1)
    assert np.array_equal(gx, ndi.correlate1d(img, np.array([-1, 0, 1], np.float32), axis=1, mode="nearest"))
    assert np.array_equal(gy, ndi.correlate1d(img, np.array([-1, 0, 1], np.float32), axis=0, mode="nearest"))


def test_median5_exact(orc):
    img = rng.standard_normal((23, 29, 2)).astype(np.float32)
    img[3:6, 4:9] = 0.0  # ties
    got = orc.median5(img)
    for c in range(2):
        assert np.array_equal(got[..., c], ndi.median_filter(img[..., c], size=5, mode="nearest"))


@pytest.mark.parametrize("k", [3, 4, 10])
def test_box_blur_vs_uniform_filter(orc, k):
    img = rng.random((40, 50)).astype(np.float32)
    got = orc.box_blur_roi(img, 0, 0, 50, 40, k)
    # window [x-k/2, x-k/2+k-1]; scipy centres at floor(k/2) with origin=0 for odd, shift for even
    origin = 0 if k % 2 else 0  # scipy's even-size window is [x-k/2, x+k/2-1], the same anchor as OpenCV's k/2
    ref = ndi.uniform_filter(img.astype(np.float64), size=k, mode="mirror", origin=origin)
    assert np.abs(got - ref).max() < 1e-6


def test_box_blur_roi_reads_parent(orc):
    img = rng.random((30, 30)).astype(np.float32)
    got = orc.box_blur_roi(img, 10, 12, 4, 4, 3)
    ref = ndi.uniform_filter(img.astype(np.float64), size=3, mode="mirror")
    assert np.abs(got[12:16, 10:14] - ref[12:16, 10:14]).max() < 1e-6
    mask = np.ones_like(img, bool); mask[12:16, 10:14] = False
    assert np.array_equal(got[mask], img[mask])


@pytest.mark.parametrize("sw,sh,dw,dh", [(40, 30, 45, 34), (45, 34, 40, 30), (25, 45, 28, 50)])
def test_resize_cubic_f32_vs_torch(orc, sw, sh, dw, dh):
    img = rng.standard_normal((sh, sw, 2)).astype(np.float32)
    got = orc.resize_cubic_f32(img, dw, dh)
    t = torch.from_numpy(img).permute(2, 0, 1)[None].double()
    ref = F.interpolate(t, size=(dh, dw), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    assert np.abs(got - ref).max() < 2e-5


@pytest.mark.parametrize("sw,sh,dw,dh", [(281, 256, 253, 230), (40, 30, 36, 27), (36, 27, 80, 60)])
def test_resize_linear_f32_vs_torch(orc, sw, sh, dw, dh):
    img = rng.standard_normal((sh, sw)).astype(np.float32)
    got = orc.resize_linear_f32(img, dw, dh)
    t = torch.from_numpy(img)[None, None].double()
    ref = F.interpolate(t, size=(dh, dw), mode="bilinear", align_corners=False)[0, 0].numpy()
    # OpenCV keeps the source coordinate in float32 (ulp(280) = 3e-5), torch in double
    assert np.abs(got - ref).max() < 1e-4


def test_resize_cubic_u8_half_vs_float(orc):
    img = rng.integers(0, 256, (64, 90, 4), dtype=np.uint8)
    got = orc.resize_cubic_u8(img, 45, 32)
    t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
    ref = F.interpolate(t, size=(32, 45), mode="bicubic", align_corners=False)[0].permute(1, 2, 0).numpy()
    ref = np.clip(np.round(ref), 0, 255)
    assert np.abs(got.astype(np.int32) - ref).max() <= 1
    assert (got.astype(np.int32) != ref).mean() < 0.02
    # exactly 1/2 scale => taps (-192, 1216, 1216, -192)/2048 in both passes
    flat = np.full((8, 8, 4), 77, np.uint8)
    assert np.array_equal(orc.resize_cubic_u8(flat, 4, 4), np.full((4, 4, 4), 77, np.uint8))


def test_pyramid_sizes_match_survey(orc):
    # levels and coarsest sizes for the three configs
    for (w0, h0, n, last) in [(281, 256, 23, (29, 26)), (1100, 2000, 37, (25, 45)), (4950, 2000, 42, (67, 27))]:
        s = orc.pyramid_sizes(w0, h0)
        assert len(s) == n and s[-1] == last
    assert sum(w * h for w, h in orc.pyramid_sizes(1100, 2000)) == 11582676


def test_gradients_compose(orc):
    img = rng.random((30, 41)).astype(np.float32)
    ix, iy = orc.gradients(img)
    assert np.array_equal(ix, orc.gaussian_blur(orc.sobel1(img, 1, 0), 3, 0.5))
    assert np.array_equal(iy, orc.gaussian_blur(orc.sobel1(img, 0, 1), 3, 0.5))
