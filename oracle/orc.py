"""ctypes binding of the CPU oracle (oracle/pixflow_oracle.cpp).

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by the product package.  PARITY UNPINNED (see the .cpp header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liborc.so")

HINT_UNKNOWN, HINT_RIGHT, HINT_DOWN, HINT_LEFT, HINT_UP = 0, 1, 2, 3, 4


def build(force=False):
    src = os.path.join(_HERE, "pixflow_oracle.cpp")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liborc.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.orc_version.restype = C.c_char_p
        _lib.orc_pyramid_sizes.restype = C.c_int
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def resize_cubic_u8(src, dw, dh):
    src = _u8(src); sh, sw, cn = src.shape
    dst = np.empty((dh, dw, cn), np.uint8)
    lib().orc_resize_cubic_u8(_p(src), sw, sh, cn, _p(dst), dw, dh)
    return dst


def _as3(a):
    a = _f32(a)
    return a[:, :, None] if a.ndim == 2 else a


def resize_linear_f32(src, dw, dh):
    s = _as3(src); sh, sw, cn = s.shape
    dst = np.empty((dh, dw, cn), np.float32)
    lib().orc_resize_linear_f32(_p(s), sw, sh, cn, _p(dst), dw, dh)
    return dst[:, :, 0] if np.ndim(src) == 2 else dst


def resize_cubic_f32(src, dw, dh):
    s = _as3(src); sh, sw, cn = s.shape
    dst = np.empty((dh, dw, cn), np.float32)
    lib().orc_resize_cubic_f32(_p(s), sw, sh, cn, _p(dst), dw, dh)
    return dst[:, :, 0] if np.ndim(src) == 2 else dst


def gaussian_kernel(n, sigma):
    out = np.empty(n, np.float32)
    lib().orc_gaussian_kernel(n, C.c_double(sigma), _p(out))
    return out


def gaussian_blur(src, ksize, sigma):
    s = _as3(src); h, w, cn = s.shape
    dst = np.empty_like(s)
    lib().orc_gaussian_blur_f32(_p(s), w, h, cn, ksize, C.c_double(sigma), _p(dst))
    return dst[:, :, 0] if np.ndim(src) == 2 else dst


def sobel1(src, dx, dy):
    s = _f32(src); h, w = s.shape
    dst = np.empty_like(s)
    lib().orc_sobel1(_p(s), w, h, dx, dy, _p(dst))
    return dst


def median5(src):
    s = _as3(src); h, w, cn = s.shape
    dst = np.empty_like(s)
    lib().orc_median5(_p(s), w, h, cn, _p(dst))
    return dst[:, :, 0] if np.ndim(src) == 2 else dst


def box_blur_roi(img, x0, y0, rw, rh, k):
    s = _f32(img).copy(); h, w = s.shape
    lib().orc_box_blur_roi(_p(s), w, h, x0, y0, rw, rh, k)
    return s


def gradients(I):
    s = _f32(I); h, w = s.shape
    ix = np.empty_like(s); iy = np.empty_like(s)
    lib().orc_gradients(_p(s), w, h, _p(ix), _p(iy))
    return ix, iy


def pyramid_sizes(w0, h0):
    ws = (C.c_int * 128)(); hs = (C.c_int * 128)()
    n = lib().orc_pyramid_sizes(w0, h0, ws, hs, 128)
    return [(ws[i], hs[i]) for i in range(n)]


def preprocess(bgra):
    s = _u8(bgra); rows, cols, _ = s.shape
    dw, dh = int(np.float32(cols) * np.float32(0.5)), int(np.float32(rows) * np.float32(0.5))
    I = np.empty((dh, dw), np.float32); A = np.empty((dh, dw), np.float32)
    lib().orc_preprocess(_p(s), cols, rows, _p(I), _p(A))
    return I, A


def set_params(pyr_scale=0.9, smoothness=0.001, vreg=0.01, hreg=0.01, step=0.5):
    """PixFlow's constructor arguments (CPU/PixFlow.hpp:46-68) for every later solver call of this process; reset_params() = the factory's presets."""
    l = lib()
    l.orc_set_params.argtypes = [C.c_float] * 5
    l.orc_set_params(pyr_scale, smoothness, vreg, hreg, step)


def reset_params():
    lib().orc_reset_params()


def pyr_down(src, dw, dh):
    return resize_linear_f32(src, dw, dh)


def sweep(I0x, I0y, I1x, I1y, blurred, a0, a1, flow, forward):
    f = _f32(flow).copy(); h, w, _ = f.shape
    lib().orc_sweep(_p(_f32(I0x)), _p(_f32(I0y)), _p(_f32(I1x)), _p(_f32(I1y)), _p(_f32(blurred)), _p(_f32(a0)), _p(_f32(a1)),
                    _p(f), w, h, int(forward))
    return f


def error_function(I0x, I0y, I1x, I1y, blurred, cand):
    c = _f32(cand); h, w, _ = c.shape
    err = np.empty((h, w), np.float32)
    lib().orc_error_function(_p(_f32(I0x)), _p(_f32(I0y)), _p(_f32(I1x)), _p(_f32(I1y)), _p(_f32(blurred)), w, h, _p(c), _p(err))
    return err


def adjust_initial_flow(I0, I1, a0, a1, hint, max_pct):
    I0 = _f32(I0); h, w = I0.shape
    flow = np.zeros((h, w, 2), np.float32)
    lib().orc_adjust_initial_flow(_p(I0), _p(_f32(I1)), _p(_f32(a0)), _p(_f32(a1)), w, h, hint, max_pct, _p(flow))
    return flow


def diffusion(a0, a1, flow):
    f = _f32(flow).copy(); h, w, _ = f.shape
    lib().orc_diffusion(_p(_f32(a0)), _p(_f32(a1)), _p(f), w, h)
    return f


def level(I0, I1, a0, a1, flow_in, hint, max_pct, stage_mask=0x1F):
    I0 = _f32(I0); h, w = I0.shape
    out = np.empty((h, w, 2), np.float32)
    fin = None if flow_in is None else _f32(flow_in)
    lib().orc_level(_p(I0), _p(_f32(I1)), _p(_f32(a0)), _p(_f32(a1)), w, h, None if fin is None else _p(fin), hint, max_pct,
                    stage_mask, _p(out))
    return out


def compute_optical_flow(bgra0, bgra1, max_pct, hint):
    a = _u8(bgra0); b = _u8(bgra1); rows, cols, _ = a.shape
    flow = np.empty((rows, cols, 2), np.float32)
    lib().orc_compute_optical_flow(_p(a), _p(b), cols, rows, max_pct, hint, _p(flow))
    return flow


def flow_one_dir(L, R, max_pct, direction):
    a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
    flow = np.empty((rows, cols, 2), np.float32)
    lib().orc_flow_one_dir(_p(a), _p(b), cols, rows, max_pct, direction, _p(flow))
    return flow


def flow_bidir(L, R, max_pct):
    a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
    f0 = np.empty((rows, cols, 2), np.float32); f1 = np.empty((rows, cols, 2), np.float32)
    lib().orc_flow_bidir(_p(a), _p(b), cols, rows, max_pct, _p(f0), _p(f1))
    return f0, f1


def combine_novel_views(L, R, flowLtoR, flowRtoL, blend):
    a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
    out = np.empty((rows, cols, 4), np.uint8)
    lib().orc_combine_novel_views(_p(a), _p(b), _p(_f32(flowLtoR)), _p(_f32(flowRtoL)), _p(_f32(blend)), cols, rows, _p(out))
    return out


def stitch_prepare(L, R, smooth=True):
    a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
    mp = np.empty((rows, cols), np.uint8); ovL = np.empty_like(a); ovR = np.empty_like(a)
    blend = np.empty((rows, cols), np.float32); md = np.empty((rows, cols), np.float32)
    lib().orc_stitch_prepare(_p(a), _p(b), cols, rows, int(smooth), _p(mp), _p(ovL), _p(ovR), _p(blend), _p(md))
    return mp, ovL, ovR, blend, md


def blend_smooth(blend, merged_dis):
    b = _f32(blend).copy(); rows, cols = b.shape
    lib().orc_blend_smooth(_p(b), _p(_f32(merged_dis)), cols, rows)
    return b


def stitch_gather(L, R, merged, mp):
    a = _u8(L); rows, cols, _ = a.shape
    out = np.empty((rows, cols, 4), np.uint8)
    lib().orc_stitch_gather(_p(a), _p(_u8(R)), _p(_u8(merged)), _p(_u8(mp)), cols, rows, _p(out))
    return out
