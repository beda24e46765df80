"""ctypes binding of the product's C ABI (include/panoflow.h -> libpanoflow.so).

Used by tests/, bench.py and __graft_entry__.py.  It is plumbing only: every function forwards to the
HIP library and raises if the library or the device is missing -- there is no CPU fallback.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libpanoflow.so")
# the lab build (-DPF_EXPERIMENTS): product sources + the cross-check sweep implementations; loaded by tests and diagnostics only
SO_PATH_EXP = os.path.join(_HERE, "libpanoflow_exp.so")

HINT_UNKNOWN, HINT_RIGHT, HINT_DOWN, HINT_LEFT, HINT_UP = 0, 1, 2, 3, 4


class PanoflowError(RuntimeError):
    pass


def build(force=False):
    """Compile the HIP library for gfx950 in-tree (hipcc cross-compiles without a GPU)."""
    cmd = ["make", "-C", _HERE, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return SO_PATH


_libs = {}


class Config(C.Structure):
    """pf_config (include/panoflow.h)"""
    _fields_ = [("struct_size", C.c_int), ("device", C.c_int), ("max_cols", C.c_int), ("max_rows", C.c_int), ("stagger_levels", C.c_int),
                ("fuse_small_level_px", C.c_int64), ("fine_gradient_blocks", C.c_int), ("pyramid_chaining", C.c_int), ("sweep_window", C.c_int),
                ("sparse_sweep", C.c_int), ("batch_pairs", C.c_int), ("sweep_wide", C.c_int), ("sweep_wide_threshold", C.c_int), ("sweep_throughput_transposed", C.c_int),
                ("full_width_batch_gradients", C.c_int), ("sweep_impl", C.c_int), ("record_path", C.c_int)]


class SolverParams(C.Structure):
    """pf_solver_params (include/panoflow.h): the constructor arguments of the reference's PixFlow<P> (CPU/PixFlow.hpp:46-68)"""
    _fields_ = [("pyr_scale_factor", C.c_float), ("smoothness_coef", C.c_float), ("vertical_regularization_coef", C.c_float),
                ("horizontal_regularization_coef", C.c_float), ("gradient_step_size", C.c_float), ("downscale_factor", C.c_float),
                ("directional_regularization_coef", C.c_float)]


def lib(exp=False):
    """exp=False: the product library.  exp=True: the lab build with the cross-check sweeps (tests / diagnostics only)."""
    _lib = _libs.get(bool(exp))
    if _lib is None:
        path = SO_PATH_EXP if exp else SO_PATH
        if not os.path.exists(path):
            raise PanoflowError("%s is not built (run __graft_entry__.build()); there is no fallback path" % os.path.basename(path))
        _lib = C.CDLL(path)
        _libs[bool(exp)] = _lib
        _lib.pf_create_cfg.restype = C.c_void_p
        _lib.pf_create_cfg.argtypes = [C.POINTER(Config)]
        _lib.pf_config_init.argtypes = [C.POINTER(Config)]
        _lib.pf_create.restype = C.c_void_p
        _lib.pf_create.argtypes = [C.c_int, C.c_int, C.c_int]
        _lib.pf_destroy.argtypes = [C.c_void_p]
        _lib.pf_last_error.restype = C.c_char_p
        _lib.pf_last_error.argtypes = [C.c_void_p]
        _lib.pf_version.restype = C.c_char_p
        _lib.pf_last_warning.restype = C.c_char_p
        _lib.pf_last_warning.argtypes = [C.c_void_p]
        _lib.pf_warning_count.argtypes = [C.c_void_p]
        _lib.pf_dev_alloc.restype = C.c_void_p
        _lib.pf_dev_alloc.argtypes = [C.c_void_p, C.c_size_t]
        _lib.pf_dev_free.argtypes = [C.c_void_p, C.c_void_p]
        _lib.pf_host_alloc.restype = C.c_void_p
        _lib.pf_host_alloc.argtypes = [C.c_void_p, C.c_size_t]
        _lib.pf_host_free.argtypes = [C.c_void_p, C.c_void_p]
        _lib.pf_algorithmic_bytes.restype = C.c_double
        _lib.pf_level_pixels.restype = C.c_longlong
        _lib.pf_dist_init.restype = C.c_void_p
        _lib.pf_dist_init.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int]
        _lib.pf_dist_destroy.argtypes = [C.c_void_p]
        _lib.pf_dist_last_error.restype = C.c_char_p
        _lib.pf_dist_last_error.argtypes = [C.c_void_p]
        _lib.pf_dist_gather_async.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        _lib.pf_dist_wait.argtypes = [C.c_void_p]
        _lib.pf_dist_max.argtypes = [C.c_void_p, C.c_void_p]
        _lib.pf_dist_barrier.argtypes = [C.c_void_p]
        _lib.pf_last_swept_steps.restype = C.c_longlong
        _lib.pf_last_swept_steps.argtypes = [C.c_void_p]
    return _lib


EXPORTS = [
    "pf_device_count", "pf_create", "pf_config_init", "pf_create_cfg", "pf_destroy", "pf_last_error", "pf_last_warning", "pf_warning_count", "pf_version", "pf_max_percentage_by_name",
    "pf_solver_params_init", "pf_set_solver_params", "pf_get_solver_params",
    "pf_flow", "pf_flow_bidir", "pf_blend", "pf_novel_view", "pf_stitch_prepare", "pf_stitch_match", "pf_stitch_generate_blend", "pf_stitch_raw_blend", "pf_stitch_gather", "pf_stitch_step", "pf_stitch_prefetch",
    "pf_dev_alloc", "pf_dev_free", "pf_host_alloc", "pf_host_free", "pf_upload", "pf_download", "pf_sync", "pf_checksum_dev", "pf_selftest_packed_chains",
    "pf_flow_bidir_dev", "pf_blend_dev", "pf_novel_view_dev", "pf_novel_view_batch_dev",
    "pf_stage_preprocess", "pf_stage_pyr_down", "pf_stage_gradients", "pf_stage_gauss", "pf_stage_median5", "pf_stage_sweep",
    "pf_stage_diffusion", "pf_stage_upsample_cubic", "pf_stage_final", "pf_stage_adjust_initial_flow", "pf_stage_level",
    "pf_stage_blend_smooth",
    "pf_profile_enable", "pf_profile_reset", "pf_profile_count", "pf_profile_get", "pf_algorithmic_bytes", "pf_level_pixels", "pf_last_swept_steps",
    "pf_dist_unique_id", "pf_dist_init", "pf_dist_destroy", "pf_dist_last_error", "pf_dist_gather_async", "pf_dist_wait", "pf_dist_max", "pf_dist_barrier",
]


def _p(a):
    # data_as keeps a reference to the array inside the returned ctypes object, so a converted temporary
    # (_p(_f32(x)) where x needed a copy) stays alive for the duration of the C call
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _u8(a):
    return np.ascontiguousarray(a, dtype=np.uint8)


def max_percentage_by_name(name):
    """makeOpticalFlowByName (CPU/PixFlow.hpp:459-500): raises on an unknown algorithm name."""
    v = lib().pf_max_percentage_by_name(name.encode())
    if v < 0:
        raise PanoflowError(lib().pf_last_error(None).decode())
    return v


def algorithmic_bytes(cols, rows):
    return float(lib().pf_algorithmic_bytes(cols, rows))


def level_pixels(cols, rows):
    n = C.c_int(0); s = C.c_longlong(0)
    p = lib().pf_level_pixels(cols, rows, C.byref(n), C.byref(s))
    return int(p), n.value, int(s.value)


def dist_unique_id():
    """rank 0: the 128-byte ncclUniqueId every rank passes to Dist()"""
    buf = C.create_string_buffer(128)
    if lib().pf_dist_unique_id(buf) != 0:
        raise PanoflowError("pf_dist_unique_id: " + lib().pf_dist_last_error(None).decode())
    return buf.raw


class Dist:
    """The path's only exchange: gather of per-pair results into rank 0's HBM over RCCL (pf_dist_*), asynchronous."""

    def __init__(self, device, unique_id, rank, world):
        self.l = lib()
        h = self.l.pf_dist_init(device, C.c_char_p(unique_id), rank, world)
        if not h:
            raise PanoflowError("pf_dist_init failed: " + self.l.pf_dist_last_error(None).decode())
        self.h = C.c_void_p(h); self.rank = rank; self.world = world

    def _chk(self, rc):
        if rc != 0:
            raise PanoflowError("pf_dist error %d: %s" % (rc, self.l.pf_dist_last_error(self.h).decode()))

    def gather_async(self, d_send, d_recv_all, nbytes):
        self._chk(self.l.pf_dist_gather_async(self.h, C.c_void_p(d_send), C.c_void_p(d_recv_all) if d_recv_all else None, C.c_size_t(nbytes)))

    def wait(self):
        self._chk(self.l.pf_dist_wait(self.h))

    def max(self, value):
        v = C.c_double(value)
        self._chk(self.l.pf_dist_max(self.h, C.byref(v)))
        return v.value

    def barrier(self):
        self._chk(self.l.pf_dist_barrier(self.h))

    def close(self):
        if self.h:
            self.l.pf_dist_destroy(self.h)
            self.h = None


class Context:
    def __init__(self, device=0, max_cols=0, max_rows=0, exp=False, **knobs):
        """knobs: fields of pf_config (stagger_levels, fuse_small_level_px, sweep_window, sparse_sweep, sweep_impl, record_path, ...);
        exp=True loads the lab build, the only one that accepts sweep_impl / record_path other than the defaults."""
        self.l = lib(exp)
        if knobs:
            cfg = Config()
            self.l.pf_config_init(C.byref(cfg))
            cfg.device, cfg.max_cols, cfg.max_rows = device, max_cols, max_rows
            for k, v in knobs.items():
                if not hasattr(cfg, k):
                    raise TypeError("unknown pf_config field %r" % k)
                setattr(cfg, k, v)
            h = self.l.pf_create_cfg(C.byref(cfg))
        else:
            h = self.l.pf_create(device, max_cols, max_rows)
        if not h:
            raise PanoflowError("pf_create failed: " + self.l.pf_last_error(None).decode())
        self.h = C.c_void_p(h)  # keep it a c_void_p: a bare int would be passed as a 32-bit C int

    def close(self):
        if self.h:
            for p in getattr(self, "_pinned", []):
                self.l.pf_host_free(self.h, C.c_void_p(p))
            self._pinned = []
            self.l.pf_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise PanoflowError("panoflow error %d: %s" % (rc, self.l.pf_last_error(self.h).decode()))

    # ---- host-buffer entry points ----
    def flow(self, i0, i1, max_pct, hint):
        a = _u8(i0); b = _u8(i1); rows, cols, _ = a.shape
        out = np.empty((rows, cols, 2), np.float32)
        self._chk(self.l.pf_flow(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), max_pct, hint, _p(out), C.c_size_t(cols * 8)))
        return out

    def flow_bidir(self, L, R, max_pct):
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        f0 = np.empty((rows, cols, 2), np.float32); f1 = np.empty((rows, cols, 2), np.float32)
        self._chk(self.l.pf_flow_bidir(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), max_pct, _p(f0), _p(f1), C.c_size_t(cols * 8)))
        return f0, f1

    def blend(self, L, R, f_l2r, f_r2l, blend):
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        out = np.empty((rows, cols, 4), np.uint8)
        self._chk(self.l.pf_blend(self.h, _p(a), _p(b), C.c_size_t(cols * 4), _p(_f32(f_l2r)), _p(_f32(f_r2l)), C.c_size_t(cols * 8), _p(_f32(blend)),
                                  C.c_size_t(cols * 4), cols, rows, _p(out), C.c_size_t(cols * 4)))
        return out

    def novel_view(self, L, R, max_pct, blend, want_flows=True):
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        out = np.empty((rows, cols, 4), np.uint8)
        f0 = np.empty((rows, cols, 2), np.float32) if want_flows else None
        f1 = np.empty((rows, cols, 2), np.float32) if want_flows else None
        self._chk(self.l.pf_novel_view(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), max_pct, _p(_f32(blend)), C.c_size_t(cols * 4), _p(out),
                                       C.c_size_t(cols * 4), _p(f0) if want_flows else None, _p(f1) if want_flows else None, C.c_size_t(cols * 8)))
        return out, f0, f1

    def stitch_prepare(self, L, R):
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        mp = np.empty((rows, cols), np.uint8); ovl = np.empty_like(a); ovr = np.empty_like(a)
        bl = np.empty((rows, cols), np.float32); md = np.empty((rows, cols), np.float32)
        self._chk(self.l.pf_stitch_prepare(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), _p(mp), C.c_size_t(cols), _p(ovl), _p(ovr), _p(bl),
                                           C.c_size_t(cols * 4), _p(md)))
        return mp, ovl, ovr, bl, md

    def stitch_match(self, L, R):
        """MatchImages + overlap masking only"""
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        mp = np.empty((rows, cols), np.uint8); ovl = np.empty_like(a); ovr = np.empty_like(a)
        self._chk(self.l.pf_stitch_match(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), _p(mp), C.c_size_t(cols), _p(ovl), _p(ovr)))
        return mp, ovl, ovr

    def stitch_generate_blend(self, mp):
        """GenerateBlend from a given map (the reference reads its public Map member)"""
        m = _u8(mp); rows, cols = m.shape
        bl = np.empty((rows, cols), np.float32); md = np.empty((rows, cols), np.float32)
        self._chk(self.l.pf_stitch_generate_blend(self.h, _p(m), C.c_size_t(cols), cols, rows, _p(bl), C.c_size_t(cols * 4), _p(md)))
        return bl, md

    def stitch_raw_blend(self, L, R):
        """GenerateBlend before its smoothing (what countblend returns in the overlap) + MergedDis."""
        a = _u8(L); b = _u8(R); rows, cols, _ = a.shape
        bl = np.empty((rows, cols), np.float32); md = np.empty((rows, cols), np.float32)
        self._chk(self.l.pf_stitch_raw_blend(self.h, _p(a), _p(b), cols, rows, C.c_size_t(cols * 4), _p(bl), C.c_size_t(cols * 4), _p(md)))
        return bl, md

    def stitch_gather(self, L, R, merged, mp):
        a = _u8(L); b = _u8(R); g = _u8(merged); m = _u8(mp); rows, cols, _ = a.shape
        out = np.empty((rows, cols, 4), np.uint8)
        self._chk(self.l.pf_stitch_gather(self.h, _p(a), _p(b), _p(g), C.c_size_t(cols * 4), _p(m), C.c_size_t(cols), cols, rows,
                                          _p(out), C.c_size_t(cols * 4)))
        return out

    def stitch_prefetch(self, next_L):
        """announce the NEXT stitch_step's left image (must be a contiguous uint8 array kept alive and unchanged until then)"""
        if next_L is None:
            self._chk(self.l.pf_stitch_prefetch(self.h, None, 0, 0, C.c_size_t(0)))
            return
        assert next_L.dtype == np.uint8 and next_L.flags["C_CONTIGUOUS"]
        rows, cols, _ = next_L.shape
        self._chk(self.l.pf_stitch_prefetch(self.h, _p(next_L), cols, rows, C.c_size_t(cols * 4)))

    def stitch_step(self, L, R, max_pct, want_out=True, out=None):
        """One iteration of main.cpp's loop on the device; R=None chains on the previous result kept in HBM.
        out: optional preallocated (rows, cols, 4) uint8 array for the composite (a caller that reuses its buffer, like
        the reference's Mat, does not pay a fresh 144 MB allocation + first-touch page faults per call)."""
        a = _u8(L); rows, cols, _ = a.shape
        r = None if R is None else _u8(R)     # named, so that a converted copy outlives the call
        if out is not None:
            assert out.dtype == np.uint8 and out.shape == (rows, cols, 4) and out.flags["C_CONTIGUOUS"]
        elif want_out:
            out = np.empty((rows, cols, 4), np.uint8)
        self._chk(self.l.pf_stitch_step(self.h, _p(a), None if r is None else _p(r), cols, rows, C.c_size_t(cols * 4), max_pct,
                                        None if out is None else _p(out), C.c_size_t(cols * 4)))
        return out

    # ---- device-resident entry points (raw device pointers as ints) ----
    def dev_alloc(self, nbytes):
        p = self.l.pf_dev_alloc(self.h, C.c_size_t(nbytes))
        if not p:
            raise PanoflowError(self.l.pf_last_error(self.h).decode())
        return p

    def selftest_packed_chains(self):
        """0 = the sweep's asm-block packed-fp32 chains give the compiler-scheduled forms' bits on this device"""
        self.l.pf_selftest_packed_chains.argtypes = [C.c_void_p]
        r = self.l.pf_selftest_packed_chains(self.h)
        if r < 0:
            self._chk(r)
        return r

    def dev_free(self, p):
        self.l.pf_dev_free(self.h, C.c_void_p(p))

    def host_array(self, shape, dtype=np.uint8):
        """numpy array over page-locked host memory (pf_host_alloc); freed with the context (keep the context alive while it is used)"""
        nbytes = int(np.prod(shape)) * np.dtype(dtype).itemsize
        p = self.l.pf_host_alloc(self.h, C.c_size_t(nbytes))
        if not p:
            raise PanoflowError(self.l.pf_last_error(self.h).decode())
        self._pinned = getattr(self, "_pinned", []) + [p]
        buf = (C.c_uint8 * nbytes).from_address(p)
        return np.frombuffer(buf, dtype=dtype).reshape(shape)

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.l.pf_upload(self.h, C.c_void_p(dptr), _p(arr), C.c_size_t(arr.nbytes)))

    def checksum_dev(self, dptr, nbytes):
        h = C.c_uint64(0)
        self._chk(self.l.pf_checksum_dev(self.h, C.c_void_p(dptr), C.c_size_t(nbytes), C.byref(h)))
        return h.value

    def download(self, arr, dptr):
        self._chk(self.l.pf_download(self.h, _p(arr), C.c_void_p(dptr), C.c_size_t(arr.nbytes)))
        return arr

    def novel_view_dev(self, d_l, d_r, cols, rows, max_pct, d_blend, d_out, d_f0=None, d_f1=None):
        self._chk(self.l.pf_novel_view_dev(self.h, C.c_void_p(d_l), C.c_void_p(d_r), cols, rows, max_pct, C.c_void_p(d_blend), C.c_void_p(d_out),
                                           C.c_void_p(d_f0) if d_f0 else None, C.c_void_p(d_f1) if d_f1 else None))

    def novel_view_batch_dev(self, d_l, d_r, cols, rows, max_pct, d_blend, d_out, d_f0=None, d_f1=None, in_flight=4):
        """throughput mode: lists of device pointers, `in_flight` pairs side by side on this GPU"""
        n = len(d_l)
        arr = lambda v: (C.c_void_p * n)(*[C.c_void_p(int(x)) if x else None for x in v])
        self._chk(self.l.pf_novel_view_batch_dev(self.h, n, arr(d_l), arr(d_r), cols, rows, max_pct, arr(d_blend), arr(d_out),
                                                 arr(d_f0) if d_f0 else None, arr(d_f1) if d_f1 else None, in_flight))

    def flow_bidir_dev(self, d_l, d_r, cols, rows, max_pct, d_f0, d_f1):
        self._chk(self.l.pf_flow_bidir_dev(self.h, C.c_void_p(d_l), C.c_void_p(d_r), cols, rows, max_pct, C.c_void_p(d_f0), C.c_void_p(d_f1)))

    def blend_dev(self, d_l, d_r, d_f0, d_f1, d_blend, cols, rows, d_out):
        self._chk(self.l.pf_blend_dev(self.h, C.c_void_p(d_l), C.c_void_p(d_r), C.c_void_p(d_f0), C.c_void_p(d_f1), C.c_void_p(d_blend), cols, rows,
                                      C.c_void_p(d_out)))

    # ---- stage-level entry points ----
    def stage_preprocess(self, bgra, pad=0):
        a = _u8(bgra); rows, cols, _ = a.shape
        dw = int(np.float32(cols + 2 * pad) * np.float32(0.5)); dh = int(np.float32(rows) * np.float32(0.5))
        g = np.empty((dh, dw), np.float32); al = np.empty((dh, dw), np.float32)
        self._chk(self.l.pf_stage_preprocess(self.h, _p(a), cols, rows, pad, _p(g), _p(al)))
        return g, al

    def stage_pyr_down(self, src, dw, dh):
        s = _f32(src); sh, sw = s.shape
        d = np.empty((dh, dw), np.float32)
        self._chk(self.l.pf_stage_pyr_down(self.h, _p(s), sw, sh, _p(d), dw, dh))
        return d

    def stage_gradients(self, img):
        s = _f32(img); h, w = s.shape
        g = np.empty((h, w, 2), np.float32)
        self._chk(self.l.pf_stage_gradients(self.h, _p(s), w, h, _p(g)))
        return g

    def stage_gauss(self, src, ksize, sigma):
        s = _f32(src)
        h, w = s.shape[:2]; cn = 1 if s.ndim == 2 else s.shape[2]
        d = np.empty_like(s)
        self._chk(self.l.pf_stage_gauss(self.h, _p(s), w, h, cn, ksize, C.c_double(sigma), _p(d)))
        return d

    def stage_median5(self, flow):
        s = _f32(flow); h, w, _ = s.shape
        d = np.empty_like(s)
        self._chk(self.l.pf_stage_median5(self.h, _p(s), w, h, _p(d)))
        return d

    def stage_sweep(self, g0, g1, blurred, a0, a1, flow, forward):
        f = _f32(flow).copy(); h, w, _ = f.shape
        self._chk(self.l.pf_stage_sweep(self.h, _p(_f32(g0)), _p(_f32(g1)), _p(_f32(blurred)), _p(_f32(a0)), _p(_f32(a1)), _p(f), w, h, int(forward)))
        return f

    def stage_diffusion(self, a0, a1, flow):
        f = _f32(flow).copy(); h, w, _ = f.shape
        self._chk(self.l.pf_stage_diffusion(self.h, _p(_f32(a0)), _p(_f32(a1)), _p(f), w, h))
        return f

    def stage_upsample_cubic(self, flow, dw, dh, scale):
        s = _f32(flow); sh, sw, _ = s.shape
        d = np.empty((dh, dw, 2), np.float32)
        self._chk(self.l.pf_stage_upsample_cubic(self.h, _p(s), sw, sh, _p(d), dw, dh, C.c_float(scale)))
        return d

    def stage_final(self, flow, pad_cols, rows, pad, scale):
        s = _f32(flow); sh, sw, _ = s.shape
        d = np.empty((rows, pad_cols - 2 * pad, 2), np.float32)
        self._chk(self.l.pf_stage_final(self.h, _p(s), sw, sh, pad_cols, rows, pad, C.c_float(scale), _p(d)))
        return d

    def stage_adjust_initial_flow(self, i0, i1, a0, a1, hint, max_pct):
        s = _f32(i0); h, w = s.shape
        f = np.empty((h, w, 2), np.float32)
        self._chk(self.l.pf_stage_adjust_initial_flow(self.h, _p(s), _p(_f32(i1)), _p(_f32(a0)), _p(_f32(a1)), w, h, hint, max_pct, _p(f)))
        return f

    def stage_level(self, i0, i1, a0, a1, flow_in, hint, max_pct):
        s = _f32(i0); h, w = s.shape
        f = np.empty((h, w, 2), np.float32)
        fin = None if flow_in is None else _f32(flow_in)
        self._chk(self.l.pf_stage_level(self.h, _p(s), _p(_f32(i1)), _p(_f32(a0)), _p(_f32(a1)), w, h, None if fin is None else _p(fin), hint, max_pct, _p(f)))
        return f

    def stage_blend_smooth(self, blend, md):
        b = _f32(blend).copy(); rows, cols = b.shape
        self._chk(self.l.pf_stage_blend_smooth(self.h, _p(b), _p(_f32(md)), cols, rows))
        return b

    def set_solver_params(self, **kw):
        """PixFlow's constructor arguments for every later solve on this context (pf_set_solver_params); no arguments = the factory's presets.
        Keys: pyr_scale_factor, smoothness_coef, vertical_regularization_coef, horizontal_regularization_coef, gradient_step_size, downscale_factor."""
        p = SolverParams()
        self.l.pf_solver_params_init(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise PanoflowError("unknown solver parameter %r" % k)
            setattr(p, k, v)
        self._chk(self.l.pf_set_solver_params(self.h, C.byref(p)))

    def solver_params(self):
        p = SolverParams()
        self._chk(self.l.pf_get_solver_params(self.h, C.byref(p)))
        return {k: getattr(p, k) for k, _ in SolverParams._fields_}

    def last_swept_steps(self):
        """dependent wavefront steps of one direction of the last solve (windows of gated pixels)"""
        return int(self.l.pf_last_swept_steps(self.h))

    def last_warning(self):
        """(text of the last performance warning raised on this context or "", how many were raised)  -- include/panoflow.h"""
        return self.l.pf_last_warning(self.h).decode(), int(self.l.pf_warning_count(self.h))

    # ---- profiling ----
    def profile_enable(self, on=True):
        """0/False off, 1/True every kernel family, 2 only the sweep kernels."""
        self.l.pf_profile_enable(self.h, int(on))

    def profile_reset(self):
        self.l.pf_profile_reset(self.h)

    def profile(self):
        out = {}
        for i in range(self.l.pf_profile_count(self.h)):
            name = C.create_string_buffer(64); ms = C.c_double(0); n = C.c_int(0)
            self.l.pf_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(n))
            out[name.value.decode()] = (ms.value, n.value)
        return out
