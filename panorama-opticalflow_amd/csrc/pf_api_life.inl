// Part of pf_api.hip (one translation unit, split along its seams in round 5): draining and error collection, creation / destruction of contexts, memory helpers, checksum.

// after the streams have drained: did any sweep band give up?  (the word lives in mapped pinned host memory and was
// written by k_collect_status at the end of each direction's stream: no copy, no further sync)
int check_sweeps(pf_ctx* c) {
  const int st = __atomic_load_n(c->h_status, __ATOMIC_ACQUIRE);
  if (st) { *c->h_status = 0; return fail(c, PF_ERR_TIMEOUT, "an in-kernel wait timed out (bits 0/1: sweep band of direction 0/1, bit 2: blend-ramp grid barrier; mask %d)", st); }
  return 0;
}

int finish(pf_ctx* c) {
  HIPCHK(c, hipStreamSynchronize(c->s_main));
  HIPCHK(c, hipStreamSynchronize(c->s_dir[0]));
  HIPCHK(c, hipStreamSynchronize(c->s_dir[1]));
  if (c->s_aux) HIPCHK(c, hipStreamSynchronize(c->s_aux));
  if (c->s_copy) HIPCHK(c, hipStreamSynchronize(c->s_copy));
  c->drained = true;
  if (c->prof) prof_collect(c);
  return 0;
}

// Every entry point that enqueues work owns one of these: whichever way the call returns (also on an early error,
// with copies from the caller's buffers or kernels still in flight), all four streams are idle afterwards, so the
// caller may free or reuse its buffers and the next call starts from a clean pipeline.
struct CallGuard {
  pf_ctx* c;
  explicit CallGuard(pf_ctx* c_) : c(c_) { if (c) c->drained = false; }
  ~CallGuard() {
    if (!c || c->drained) return;   // the normal exit went through finish(): nothing is in flight
    c->drained = true;
    hipStreamSynchronize(c->s_main); hipStreamSynchronize(c->s_dir[0]); hipStreamSynchronize(c->s_dir[1]);
    if (c->s_aux) hipStreamSynchronize(c->s_aux);
    if (c->s_copy) hipStreamSynchronize(c->s_copy);
  }
};

int use(pf_ctx* c) {
  if (!c) return fail(nullptr, PF_ERR_ARG, "null context");
  HIPCHK(c, hipSetDevice(c->device));
  return 0;
}

// packed rows on both sides (the usual case): one linear copy -- the DMA engines at the link rate when the host side is pinned
int up2d(pf_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, int rows) {
  if (dpitch == width_bytes && spitch == width_bytes) HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * size_t(rows), hipMemcpyHostToDevice, c->s_main));
  else HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyHostToDevice, c->s_main));
  return 0;
}
int down2d(pf_ctx* c, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width_bytes, int rows) {
  if (dpitch == width_bytes && spitch == width_bytes) HIPCHK(c, hipMemcpyAsync(dst, src, width_bytes * size_t(rows), hipMemcpyDeviceToHost, c->s_main));
  else HIPCHK(c, hipMemcpy2DAsync(dst, dpitch, src, spitch, width_bytes, rows, hipMemcpyDeviceToHost, c->s_main));
  return 0;
}

}  // namespace

// =================================================================================================
extern "C" {

int pf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

#ifdef PF_EXPERIMENTS
const char* pf_version(void) { return "panoflow-mi355x r3 (gfx950, lab build with the cross-check sweeps)"; }
#else
const char* pf_version(void) { return "panoflow-mi355x r3 (gfx950)"; }
#endif

void pf_config_init(pf_config* cfg) {
  if (!cfg) return;
  memset(cfg, 0, sizeof *cfg);
  cfg->struct_size = (int)sizeof *cfg;
  cfg->stagger_levels = -1; cfg->fuse_small_level_px = -1; cfg->fine_gradient_blocks = 64; cfg->pyramid_chaining = 1;
  cfg->sweep_window = 1; cfg->sparse_sweep = -1; cfg->sweep_impl = 2; cfg->record_path = 0; cfg->batch_pairs = -1;
  cfg->sweep_wide = -1; cfg->sweep_wide_threshold = 512; cfg->sweep_throughput_transposed = 1; cfg->full_width_batch_gradients = 1;
}

pf_ctx* pf_create(int device, int max_cols, int max_rows) {
  pf_config cfg; pf_config_init(&cfg);
  cfg.device = device; cfg.max_cols = max_cols; cfg.max_rows = max_rows;
  return pf_create_cfg(&cfg);
}

}  // extern "C"

namespace {
// lane = one of the extra stream / buffer sets of the throughput mode: it only ever runs pf_novel_view_dev, so it is pre-sized
// for a solve and the two internal flow planes, not for the stitch chain and the host-staging buffers (~93 B/px it would never use)
pf_ctx* create_ctx(const pf_config& cfg, bool lane) {
  const int device = cfg.device, max_cols = cfg.max_cols, max_rows = cfg.max_rows;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) { fail(nullptr, PF_ERR_DEVICE, "no HIP device available (this library has no CPU fallback)"); return nullptr; }
  if (device < 0 || device >= n) { fail(nullptr, PF_ERR_ARG, "device %d out of range (0..%d)", device, n - 1); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "hipSetDevice(%d) failed", device); return nullptr; }
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "hipGetDeviceProperties failed"); return nullptr; }
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) { fail(nullptr, PF_ERR_DEVICE, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName); return nullptr; }
  pf_ctx* c = new pf_ctx();
  c->device = device;
  bool ok = hipStreamCreateWithFlags(&c->s_main, hipStreamNonBlocking) == hipSuccess;
  for (int d = 0; d < 2 && ok; ++d) ok = hipStreamCreateWithFlags(&c->s_dir[d], hipStreamNonBlocking) == hipSuccess;
  // The runtime hands hardware queues to streams round-robin in creation order.  A context's five streams are created together so
  // that they land on five DIFFERENT queues: created on first use (after other contexts' streams), the blend-ramp stream ended
  // up sharing a queue with one of the flow directions and a 9000x4000 stitch step took 7 ms longer.  Lanes of the throughput
  // mode never stitch: three streams each, so that six lanes fit GPU_MAX_HW_QUEUES = 24.
  if (!lane) ok = ok && hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking) == hipSuccess && hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_alpha, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_gate, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_aux_go, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_aux_done, hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&c->ev_pre, hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&c->ev_fine, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&c->ev_fine2, hipEventDisableTiming) == hipSuccess &&
       hipEventCreateWithFlags(&c->ev_stagger, hipEventDisableTiming) == hipSuccess;
  for (int d = 0; d < 2 && ok; ++d) ok = hipEventCreateWithFlags(&c->ev_dir[d], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_status, 64, hipHostMallocMapped) == hipSuccess && hipHostGetDevicePointer((void**)&c->d_status, c->h_status, 0) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&c->h_gate, kMaxBatch * kGateWords * sizeof(int), hipHostMallocMapped) == hipSuccess &&   // one area per pair of a batch
       hipHostGetDevicePointer((void**)&c->d_gate, c->h_gate, 0) == hipSuccess;
  if (ok) { *c->h_status = 0; memset(c->h_gate, 0, kMaxBatch * kGateWords * sizeof(int)); }
  if (!ok) { fail(nullptr, PF_ERR_DEVICE, "stream/event creation failed"); delete c; return nullptr; }
  c->cfg = cfg;
  {
    // Once per device and process, under one mutex: (i) the sweep's asm-block packed chains (csrc/exact_forms.hpp) against the
    // compiler-scheduled forms of the same arithmetic, on THIS device -- a mismatch means the hardware assumption behind them does not
    // hold here: refuse, rather than compute wrong flows (a -DPF_SAFE_PK build has no such blocks and passes trivially); (ii) the
    // blend's two small tables (kernels_misc.hip: device globals, the same values for every context -- written once, so that no later
    // context rewrites them under a blend another context has in flight).  Only SUCCESS is cached: a probe that could not run
    // (a transient allocation / launch failure) is tried again by the next pf_create.
    // (round 6: the tables are written again by EVERY pf_create, under the mutex -- 2 x 256 floats, the same values each time, so a rewrite
    // under another context's blend in flight changes no bit -- instead of once per process: device globals do not survive a hipDeviceReset,
    // and a cached "done" would then leave later contexts with zeroed tables.  The failure paths leave the lock before they destroy the context.)
    static std::mutex init_mu;
    static std::map<int, bool> probed;
    bool failed = false;
    {
      std::lock_guard<std::mutex> lk(init_mu);
      if (!probed[device]) {
        unsigned* scratch = nullptr;
        int r = -1;
        if (hipMalloc((void**)&scratch, 256) == hipSuccess) { r = sweep_pk_probe(c->s_main, scratch); hipFree(scratch); }
        if (r != 0) {
          fail(nullptr, PF_ERR_DEVICE, r < 0 ? "the packed-fp32 probe could not run on device %d"
                                               : "device %d: the sweep's asm-block packed-fp32 chains do not reproduce the compiler-scheduled forms (%d threads differ); rebuild with -DPF_SAFE_PK",
               device, r);
          failed = true;
        } else probed[device] = true;
      }
      if (!failed) {
        launch_blend_tables(c->s_main);
        if (hipStreamSynchronize(c->s_main) != hipSuccess) { fail(nullptr, PF_ERR_DEVICE, "device %d: the blend tables could not be initialised", device); failed = true; }
      }
    }
    if (failed) { pf_destroy(c); return nullptr; }
  }
  c->g5 = make_gauss(5, 0.25); c->g3_05 = make_gauss(3, 0.5); c->g3_1 = make_gauss(3, 1.0); c->g15 = make_gauss(15, 8.0);
  // Pre-sizing: every buffer a bidirectional solve / a stitch step on max_cols x max_rows needs is
  // allocated now, so that the first call does not pay ~40 hipMallocs.  0 x 0 = allocate lazily (the arena only grows).
  if (max_cols > 0 && max_rows > 0) {
    const int pad = max_cols / 20;
    bool ok2 = check_dims(c, max_cols, max_rows, pad) == 0;
    if (ok2) { SolveBufs sb; ok2 = alloc_solve(c, make_geometry(max_cols, max_rows, pad), 2, sb) == 0; }
    const size_t n = size_t(max_cols) * max_rows;
    const struct { const char* name; size_t bytes; } io[] = {
        {"nv_flow_l2r", n * 8}, {"nv_flow_r2l", n * 8}, {"h_img0", n * 4}, {"h_img1", n * 4}, {"h_flow0", n * 8}, {"h_flow1", n * 8}, {"h_blend", n * 4}, {"h_out", n * 4},
        {"ch_l", n * 4}, {"ch_r", n * 4}, {"ch_final", n * 4}, {"st_map", n}, {"st_ovl", n * 4}, {"st_ovr", n * 4}, {"st_blend", n * 4}, {"st_md", n * 4},
        {"st_merged", n * 4}, {"st_rowsum", n * 8}, {"st_blur_tmp", n * 4}};
    for (const auto& e : io) if (ok2 && (!lane || strncmp(e.name, "nv_", 3) == 0)) ok2 = ensure(c, e.name, e.bytes) != nullptr;
    if (!ok2) { g_err = c->err; pf_destroy(c); return nullptr; }
  }
  return c;
}
}  // namespace

extern "C" {

pf_ctx* pf_create_cfg(const pf_config* user) {
  if (!user || user->struct_size != (int)sizeof(pf_config)) { fail(nullptr, PF_ERR_ARG, "pf_create_cfg: struct_size does not match this library's pf_config"); return nullptr; }
  pf_config cfg = *user;
#ifdef PF_EXPERIMENTS
  // lab build only: the diagnostics under tests/micro select variants per process through the environment
  auto env_int = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
  env_int("PANOFLOW_SWEEP", cfg.sweep_impl); env_int("PANOFLOW_PREP", cfg.record_path); env_int("PANOFLOW_STAGGER", cfg.stagger_levels);
  env_int("PANOFLOW_PYR_CHAIN", cfg.pyramid_chaining); env_int("PANOFLOW_FINE_GRAD_BLOCKS", cfg.fine_gradient_blocks);
  env_int("PANOFLOW_SPARSE", cfg.sparse_sweep); env_int("PANOFLOW_WIDE", cfg.sweep_wide); env_int("PANOFLOW_WIDE_THRESHOLD", cfg.sweep_wide_threshold);
  env_int("PANOFLOW_BATCH_GRAD_FULL", cfg.full_width_batch_gradients);
  if (getenv("PANOFLOW_NO_WINDOW")) cfg.sweep_window = 0;
  if (const char* e = getenv("PANOFLOW_FUSE_UPS_PX")) cfg.fuse_small_level_px = atol(e);
  if (cfg.sweep_impl != 1 && cfg.sweep_impl != 3) cfg.sweep_impl = 2;
  if (cfg.record_path < 0 || cfg.record_path > 2) cfg.record_path = 0;
#else
  if (cfg.sweep_impl != 2 || cfg.record_path != 0 || cfg.sweep_wide == 1) {
    fail(nullptr, PF_ERR_ARG, "sweep_impl / record_path / sweep_wide 1 select cross-check implementations that only the -DPF_EXPERIMENTS build (libpanoflow_exp.so) contains");
    return nullptr;
  }
#endif
  if (cfg.batch_pairs == 0 || cfg.batch_pairs < -1 || cfg.batch_pairs > kMaxBatch) { fail(nullptr, PF_ERR_ARG, "pf_create_cfg: batch_pairs must be -1 or 1..%d", kMaxBatch); return nullptr; }
  if (cfg.fine_gradient_blocks < 1 || cfg.stagger_levels < -1 || cfg.fuse_small_level_px < -1 || cfg.sparse_sweep < -1 || cfg.sparse_sweep > 1 ||
      cfg.sweep_wide < -1 || cfg.sweep_wide > 2 || cfg.sweep_wide_threshold < 0) {
    fail(nullptr, PF_ERR_ARG, "pf_create_cfg: knob out of range");
    return nullptr;
  }
  return create_ctx(cfg, false);
}

void pf_destroy(pf_ctx* c) {
  if (!c) return;
  for (pf_ctx* l : c->lanes) pf_destroy(l);
  c->lanes.clear();
  hipSetDevice(c->device);
  hipDeviceSynchronize();
  for (auto& kv : c->bufs) if (kv.second.p) hipFree(kv.second.p);
  for (auto e : c->ev_pool) hipEventDestroy(e);
  for (auto& p : c->prof_pending) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
  if (c->ev_pre) hipEventDestroy(c->ev_pre);
  if (c->ev_fine) hipEventDestroy(c->ev_fine);
  if (c->ev_fine2) hipEventDestroy(c->ev_fine2);
  if (c->ev_stagger) hipEventDestroy(c->ev_stagger);
  for (int d = 0; d < 2; ++d) { if (c->ev_dir[d]) hipEventDestroy(c->ev_dir[d]); if (c->s_dir[d]) hipStreamDestroy(c->s_dir[d]); }
  if (c->ev_aux_go) hipEventDestroy(c->ev_aux_go);
  if (c->ev_aux_done) hipEventDestroy(c->ev_aux_done);
  if (c->s_aux) hipStreamDestroy(c->s_aux);
  if (c->ev_alpha) hipEventDestroy(c->ev_alpha);
  if (c->ev_gate) hipEventDestroy(c->ev_gate);
  if (c->s_copy) hipStreamDestroy(c->s_copy);
  if (c->s_main) hipStreamDestroy(c->s_main);
  if (c->h_status) hipHostFree(c->h_status);
  if (c->h_gate) hipHostFree(c->h_gate);
  delete c;
}

const char* pf_last_error(const pf_ctx* c) { return c ? c->err.c_str() : g_err.c_str(); }
const char* pf_last_warning(const pf_ctx* c) { return c ? c->warn.c_str() : ""; }
int pf_warning_count(const pf_ctx* c) { return c ? c->warn_count : 0; }

void pf_solver_params_init(pf_solver_params* p) {
  if (!p) return;
  p->pyr_scale_factor = kPyrScaleFactor; p->smoothness_coef = kSmoothnessCoef; p->vertical_regularization_coef = kVerticalRegularizationCoef;
  p->horizontal_regularization_coef = kHorizontalRegularizationCoef; p->gradient_step_size = kGradientStepSize; p->downscale_factor = kDownscaleFactor;
  p->directional_regularization_coef = 0.0f;
}
// PixFlow's constructor arguments (CPU/PixFlow.hpp:46-68) as context state; see include/panoflow.h for the accepted ranges
int pf_set_solver_params(pf_ctx* c, const pf_solver_params* p) {
  if (!c) return fail(nullptr, PF_ERR_ARG, "null context");
  pf_solver_params q;
  if (p) q = *p; else pf_solver_params_init(&q);
  auto coef_ok = [](float v) { return std::isfinite(v) && v >= 0.0f; };
  if (!(q.pyr_scale_factor >= 0.25f && q.pyr_scale_factor <= 0.98f)) return fail(c, PF_ERR_ARG, "pyrScaleFactor %g outside [0.25, 0.98]", (double)q.pyr_scale_factor);
  if (!coef_ok(q.smoothness_coef) || !coef_ok(q.vertical_regularization_coef) || !coef_ok(q.horizontal_regularization_coef))
    return fail(c, PF_ERR_ARG, "smoothnessCoef / verticalRegularizationCoef / horizontalRegularizationCoef must be finite and >= 0");
  if (!coef_ok(q.gradient_step_size)) return fail(c, PF_ERR_ARG, "gradientStepSize must be finite and >= 0");
  if (q.downscale_factor != kDownscaleFactor) return fail(c, PF_ERR_ARG, "downscaleFactor %g: only 0.5 is supported (the 8-bit half-resolution path)", (double)q.downscale_factor);
#ifdef PF_EXPERIMENTS
  if (c->cfg.sweep_impl == 3 && p && memcmp(&q, &c->sp, sizeof q) != 0) {   // the relaxation experiment (kernels_relax.inl) is compiled for the presets
    pf_solver_params d; pf_solver_params_init(&d);
    if (memcmp(&q, &d, sizeof q) != 0) return fail(c, PF_ERR_ARG, "sweep_impl 3 (lab build) supports the preset parameters only");
  }
#endif
  c->sp = q;
  SolverCoef cf;
  cf.smooth = q.smoothness_coef; cf.vreg = q.vertical_regularization_coef; cf.hreg = q.horizontal_regularization_coef; cf.step = q.gradient_step_size;
  // the fast step's single fused multiply-add is the reference's multiply + subtract only for a power-of-two step size (pf_common.hpp: SolverCoef)
  int ex = 0;
  const bool pow2 = q.gradient_step_size > 0.0f && std::frexp(q.gradient_step_size, &ex) == 0.5f && ex - 1 >= -16 && ex - 1 <= 16;
  cf.guard_min = pow2 ? -94 : 0x7fffffff;
  c->cf = cf;
  for (pf_ctx* l : c->lanes) { l->sp = c->sp; l->cf = c->cf; }
  return 0;
}
int pf_get_solver_params(const pf_ctx* c, pf_solver_params* out) {
  if (!c || !out) return PF_ERR_ARG;
  *out = c->sp;
  return 0;
}

int pf_max_percentage_by_name(const char* name) {
  if (name && strcmp(name, "pixflow_low") == 0) return 0;
  if (name && strcmp(name, "pixflow_search_20") == 0) return 20;
  return fail(nullptr, PF_ERR_ARG, "unrecognized flow algorithm name: %s", name ? name : "(null)");
}

// ---- device memory helpers ----
void* pf_dev_alloc(pf_ctx* c, size_t bytes) {
  if (use(c)) return nullptr;
  void* p = nullptr;
  if (hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) { fail(c, PF_ERR_NOMEM, "hipMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void pf_dev_free(pf_ctx* c, void* p) { if (!use(c) && p) hipFree(p); }
// page-locked host memory for the caller's images: copies to and from it run at the link's DMA rate (a pageable destination is
// staged through the runtime's bounce buffers: 17.5 GB/s instead of ~55 GB/s for the 144 MB composite of a 9000x4000 step)
void* pf_host_alloc(pf_ctx* c, size_t bytes) {
  if (use(c)) return nullptr;
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { fail(c, PF_ERR_NOMEM, "hipHostMalloc(%zu) failed", bytes); return nullptr; }
  return p;
}
void pf_host_free(pf_ctx* c, void* p) { if (!use(c) && p) hipHostFree(p); }
int pf_upload(pf_ctx* c, void* dst, const void* src, size_t bytes) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
  return 0;
}
int pf_download(pf_ctx* c, void* dst, const void* src, size_t bytes) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  HIPCHK(c, hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
  return 0;
}
int pf_sync(pf_ctx* c) { if (int e = use(c)) return e; return finish(c); }
int pf_selftest_packed_chains(pf_ctx* c) {
  if (int e = use(c)) return e;
  unsigned* scratch = (unsigned*)ensure(c, "pk_probe", 256);
  if (!scratch) return PF_ERR_NOMEM;
  const int r = sweep_pk_probe(c->s_main, scratch);
  return r < 0 ? fail(c, PF_ERR_DEVICE, "the packed-fp32 probe could not run") : r;
}
// 64-bit content checksum of `bytes` bytes at d_ptr (8-byte aligned), computed on the device: results that live in HBM -- on this
// GPU or gathered from others -- are compared without a trip through the host.  ~25 us per 144 MB strip.
int pf_checksum_dev(pf_ctx* c, const void* d_ptr, size_t bytes, uint64_t* out) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_ptr || !out || (reinterpret_cast<uintptr_t>(d_ptr) & 7)) return fail(c, PF_ERR_ARG, "pf_checksum_dev: null or misaligned pointer");
  unsigned long long* acc = (unsigned long long*)ensure(c, "checksum_acc", 256);
  if (!acc) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(acc, 0, 8, c->s_main));
  launch_checksum64(c->s_main, d_ptr, bytes, acc);
  unsigned long long h = 0;
  HIPCHK(c, hipMemcpyAsync(&h, acc, 8, hipMemcpyDeviceToHost, c->s_main));
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  *out = h;
  return 0;
}

