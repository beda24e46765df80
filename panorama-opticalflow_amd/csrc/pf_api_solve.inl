// Part of pf_api.hip (one translation unit, split along its seams in round 5): one level, the buffers of a solve (arena or batch slabs), the orchestration of a solve / a batched solve on the HIP streams.
// diffusion): 0 for a lone pair, set by the throughput mode for its lanes; pf_config::fuse_small_level_px overrides both.
int64_t fuse_small_px(const pf_ctx* c) { return c->cfg.fuse_small_level_px >= 0 ? c->cfg.fuse_small_level_px : c->fuse_ups_px; }

// the product library ships ONE sweep (k_sweep_prep + k_sweep2); the lab build (-DPF_EXPERIMENTS, libpanoflow_exp.so) adds the
// cross-check implementations the test-suite holds it against
inline bool launch_sweep_any(hipStream_t st, const SweepArgs& a, float* rec, bool relax) {
#ifdef PF_EXPERIMENTS
  if (relax) return launch_sweep_relax(st, a);
#endif
  (void)relax;
  return launch_sweep2(st, a, rec);
}

// One level of one direction (PixFlow.hpp:272-340, gradients excluded: they are precomputed for all levels).
// flow_a holds the incoming flow and receives the level's result (flow_b, blurred, tmp are scratch).
struct LevelBufs { float *flow_a, *flow_b, *blurred, *tmp, *rec; };
// box = bounding box (min x, min y, max x, max y) of the gated pixels of this level, or nullptr for "everything"
void run_level(pf_ctx* c, hipStream_t st, const float* g0, const float* g1, const float* a0, const float* a1, const uint8_t* gate, int w, int h, int sparse,
               const int* box, const LevelBufs& b, unsigned long long* bnd_fwd, unsigned long long* bnd_bwd, int* ctrl_fwd, int* ctrl_bwd, float** result,
               int* pc_fwd = nullptr, int* pc_bwd = nullptr, const float* ups_src = nullptr, int ups_w = 0, int ups_h = 0, Batch bt = Batch()) {
  // ups_src: flow_a does not hold this level's incoming flow yet -- it is the upsample of the coarser level's result (ups_w x ups_h),
  // computed by the Gaussian's tile loader on the way (small levels: one launch instead of two)
  if (ups_src) { PROF(c, st, "gauss15_blurredFlow"); launch_gauss15_upsample(st, ups_src, ups_w, ups_h, 1.0f / c->sp.pyr_scale_factor, b.flow_a, b.blurred, w, h, c->g15, bt); }
  else { PROF(c, st, "gauss15_blurredFlow"); launch_gauss15(st, b.flow_a, b.tmp, b.blurred, w, h, c->g15, bt); }
  SweepArgs sa;
  sa.bt = bt; sa.cf = c->cf;
  sa.g0 = reinterpret_cast<const float2*>(g0); sa.g1 = reinterpret_cast<const float2*>(g1);
  sa.blurred = reinterpret_cast<const float2*>(b.blurred); sa.gate = gate; sa.W = w; sa.H = h; sa.sparse = sparse;
  if (box) { sa.ax0 = box[0]; sa.ay0 = box[1]; sa.ax1 = box[2] + 1; sa.ay1 = box[3] + 1; }   // empty (max < min): the sweeps are the identity
  // workgroup shape (pf_config::sweep_wide): a lone pair never oversubscribes the chip (126 workgroups at 9000x4000) and keeps the latency form
  sa.wide = c->cfg.sweep_wide < 0 ? (bt.n > 1 ? -1 : 0) : c->cfg.sweep_wide;
  sa.wide_threshold_wgs = c->cfg.sweep_wide_threshold;
  sa.wide_tr = c->cfg.sweep_throughput_transposed;
  sa.concurrent_sweeps = 2 * bt.n * c->lanes_running;   // both directions of every pair of every lane's batch sweep at the same time
  // Timing a sweep (profile mode 1 or 2) attaches the two events to the launches themselves (hipExtLaunchKernel) instead of
  // recording markers around them.  Same-box A/B, ms per step: no timing 27.38, markers 27.65, attached events 27.60 -- bench.py's
  // roofline needs per-launch HIP events inside its timed region, so ~0.2 ms of every timed step is the measurement itself.
  auto sweep = [&](SweepArgs& a) {
#ifdef PF_EXPERIMENTS
    if (c->cfg.sweep_impl == 1) { PROF(c, st, "sweep"); launch_sweep(st, a); return; }
    const bool relax = c->cfg.sweep_impl == 3;
    a.prep_mode = c->cfg.record_path;
#else
    const bool relax = false;
#endif
    if (!c->prof) { launch_sweep_any(st, a, b.rec, relax); return; }
    ProfPending p;
    { std::lock_guard<std::mutex> lk(c->prof_mu); p.id = prof_id(c, "sweep"); p.a = prof_event(c); p.b = prof_event(c); }
    a.ev_start = p.a; a.ev_stop = p.b;
    const bool launched = launch_sweep_any(st, a, b.rec, relax);
    a.ev_start = nullptr; a.ev_stop = nullptr;
    std::lock_guard<std::mutex> lk(c->prof_mu);
    if (launched) c->prof_pending.push_back(p); else { c->ev_pool.push_back(p.a); c->ev_pool.push_back(p.b); }
  };
  { sa.flow = reinterpret_cast<float2*>(b.flow_a); sa.boundary = bnd_fwd; sa.ctrl = ctrl_fwd; sa.prepcnt = pc_fwd; sa.forward = 1; sweep(sa); }
  { PROF(c, st, "median5"); launch_median5(st, b.flow_a, b.flow_b, w, h, bt); }
  { sa.flow = reinterpret_cast<float2*>(b.flow_b); sa.boundary = bnd_bwd; sa.ctrl = ctrl_bwd; sa.prepcnt = pc_bwd; sa.forward = 0; sweep(sa); }
  if ((long)w * h <= fuse_small_px(c)) {
    // throughput mode, small levels: the second median rides in the diffusion's tile loader (one launch fewer; result in b.tmp,
    // which nothing else uses: it must not be flow_a, the plane the next level's incoming flow is written to)
    PROF(c, st, "gauss15_diffusion"); launch_median_gauss15_mix(st, b.flow_b, a0, a1, w, h, c->g15, b.tmp, bt);
    *result = b.tmp;
    return;
  }
  { PROF(c, st, "median5"); launch_median5(st, b.flow_b, b.flow_a, w, h, bt); }
  { PROF(c, st, "gauss15_diffusion"); launch_gauss15_mix(st, b.flow_a, b.tmp, a0, a1, w, h, c->g15, b.flow_b, bt); }
  *result = b.flow_b;
}

// bounding boxes of the gated pixels of the levels described by t (device gate plane) -> host; one stream sync
int gate_boxes_to_host(pf_ctx* c, hipStream_t st, const uint8_t* gate, const LevelTable& t, size_t total, std::vector<int>& box) {
  box.assign(size_t(t.n) * 4, 0);
  for (int l = 0; l < t.n; ++l) { box[4 * l] = 0x7fffffff; box[4 * l + 1] = 0x7fffffff; box[4 * l + 2] = -1; box[4 * l + 3] = -1; }
  int* d_box = (int*)ensure(c, "gate_box", size_t(kLevelTableMax) * 4 * sizeof(int));
  if (!d_box) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemcpyAsync(d_box, box.data(), box.size() * sizeof(int), hipMemcpyHostToDevice, st));
  launch_gate_bbox(st, gate, t, total, d_box);
  HIPCHK(c, hipMemcpyAsync(box.data(), d_box, box.size() * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(c, hipStreamSynchronize(st));
  return 0;
}

// Everything one solve keeps in HBM (named grow-only arena): shared pyramids / gradients / gate, per-direction flow
// planes, record buffers, hand-off granules and control words.  Also used by pf_create's pre-sizing.
struct SolveBufs {
  float* pyrI[2]; float* pyrA[2]; float* grad[2];
  uint8_t* gate; float* half_tmp;
  std::vector<size_t> bnd_off; size_t bnd_total;
  LevelBufs lb[2]; unsigned long long* bnd[2]; int* ctrl[2]; float* ratio[2];
  int* prepcnt[2]; std::vector<size_t> pc_off; size_t pc_total;   // per sweep launch: one "records ready" counter per sweep workgroup
  int* gate_work = nullptr;     // batch slabs only: this pair's work area of k_gate_bbox_all (a lone solve uses the context's "gate_work")
  float* nv_flow[2] = {nullptr, nullptr};   // batch slabs only: internal flow planes for pairs whose caller does not want the flows
};
// Where a solve's buffers come from: the context's named grow-only arena (a lone solve), or -- for a batch of pairs solved by the same
// launches -- one slab per pair, all with the same layout and `stride` bytes apart, so that a kernel reaches pair z's copy of any buffer
// by adding z * stride to pair 0's pointer (pf_common.hpp: Batch).  base == nullptr is the sizing pass.
struct Carver {
  pf_ctx* c; bool slab; char* base; size_t off;
  void* get(const char* name, size_t bytes) {
    if (!slab) return ensure(c, name, bytes);
    const size_t o = off;
    off += (bytes + 255) & ~size_t(255);
    return base ? static_cast<void*>(base + o) : reinterpret_cast<void*>(size_t(256));   // sizing pass: any non-null value
  }
};
int alloc_solve(Carver& cv, const Geometry& g, int ndirs, SolveBufs& b) {
  pf_ctx* c = cv.c;
  const size_t n0 = size_t(g.w0) * g.h0;
  const char* nI[2] = {"pyrI0", "pyrI1"}; const char* nA[2] = {"pyrA0", "pyrA1"}; const char* nG[2] = {"grad0", "grad1"};
  for (int i = 0; i < 2; ++i) {
    b.pyrI[i] = (float*)cv.get(nI[i], g.P * 4); b.pyrA[i] = (float*)cv.get(nA[i], g.P * 4); b.grad[i] = (float*)cv.get(nG[i], g.P * 8);
    if (!b.pyrI[i] || !b.pyrA[i] || !b.grad[i]) return PF_ERR_NOMEM;
  }
  b.gate = (uint8_t*)cv.get("gate", g.P);
  b.half_tmp = (float*)cv.get("half_tmp", n0 * 4);
  if (!b.gate || !b.half_tmp) return PF_ERR_NOMEM;
  // hand-off rows + control words of every sweep launch of this solve
  b.bnd_off.assign(g.n, 0);
  b.bnd_total = 0;
  for (int l = 0; l < g.n; ++l) { b.bnd_off[l] = b.bnd_total; b.bnd_total += sweep_boundary_elems(g.ws[l], g.hs[l]); }
  b.pc_off.assign(g.n, 0);
  b.pc_total = 0;
  for (int l = 0; l < g.n; ++l) { b.pc_off[l] = b.pc_total; b.pc_total += 2 * size_t(sweep2_num_wgs_max(g.ws[l], g.hs[l])); }   // forward + backward sweep
  const char* nb[2][8] = {{"d0_flow_a", "d0_flow_b", "d0_blurred", "d0_tmp", "d0_bnd", "d0_ctrl", "d0_ratio", "d0_rec"},
                          {"d1_flow_a", "d1_flow_b", "d1_blurred", "d1_tmp", "d1_bnd", "d1_ctrl", "d1_ratio", "d1_rec"}};
  for (int d = 0; d < ndirs; ++d) {
    b.lb[d].flow_a = (float*)cv.get(nb[d][0], n0 * 8); b.lb[d].flow_b = (float*)cv.get(nb[d][1], n0 * 8);
    b.lb[d].blurred = (float*)cv.get(nb[d][2], n0 * 8); b.lb[d].tmp = (float*)cv.get(nb[d][3], n0 * 8);
    b.bnd[d] = (unsigned long long*)cv.get(nb[d][4], b.bnd_total * 2 * 8);
    b.ctrl[d] = (int*)cv.get(nb[d][5], size_t(g.n) * 2 * 2 * sizeof(int));
    b.ratio[d] = (float*)cv.get(nb[d][6], 256);
    b.prepcnt[d] = (int*)cv.get(d == 0 ? "d0_prepcnt" : "d1_prepcnt", b.pc_total * sizeof(int));
    if (!b.prepcnt[d]) return PF_ERR_NOMEM;
    b.lb[d].rec = (float*)cv.get(nb[d][7], sweep2_rec_bytes(g.w0, g.h0));
    if (!b.lb[d].rec) return PF_ERR_NOMEM;
    if (!b.lb[d].flow_a || !b.lb[d].flow_b || !b.lb[d].blurred || !b.lb[d].tmp || !b.bnd[d] || !b.ctrl[d] || !b.ratio[d]) return PF_ERR_NOMEM;
  }
  if (cv.slab) {
    b.gate_work = (int*)cv.get("gate_work", (4 * kLevelTableMax + 2) * sizeof(int));
    for (int d = 0; d < 2; ++d) b.nv_flow[d] = (float*)cv.get(d ? "nv_flow_r2l" : "nv_flow_l2r", size_t(g.cols) * g.rows * 8);
  } else if (!ensure(c, "gate_box", size_t(kLevelTableMax) * 4 * sizeof(int)) || !ensure(c, "gate_count", 256)) return PF_ERR_NOMEM;
  (void)c;
  return 0;
}
int alloc_solve(pf_ctx* c, const Geometry& g, int ndirs, SolveBufs& b) { Carver cv{c, false, nullptr, 0}; return alloc_solve(cv, g, ndirs, b); }
// slabs of a batch of nb pairs: returns pair 0's buffers and the slab stride
int alloc_solve_batch(pf_ctx* c, const Geometry& g, int nb, SolveBufs& b, size_t& stride) {
  Carver sizing{c, true, nullptr, 0};
  if (int e = alloc_solve(sizing, g, 2, b)) return e;
  stride = (sizing.off + 4095) & ~size_t(4095);
  const bool fresh = c->bufs.find("batch_slab") == c->bufs.end() || c->bufs["batch_slab"].cap < stride * size_t(nb);
  char* base = (char*)ensure(c, "batch_slab", stride * size_t(nb));
  if (!base) return PF_ERR_NOMEM;
  Carver cv{c, true, base, 0};
  if (int e = alloc_solve(cv, g, 2, b)) return e;
  const size_t work_off = size_t(reinterpret_cast<char*>(b.gate_work) - base);
  if (fresh || c->slab_stride != stride || c->slab_work_off != work_off || c->slab_pairs < nb) {   // new memory or a new layout: (re)initialise the self-resetting work areas
    std::vector<int> init(4 * kLevelTableMax + 2, 0);
    for (int l = 0; l < kLevelTableMax; ++l) { init[4 * l] = 0x7fffffff; init[4 * l + 1] = 0x7fffffff; init[4 * l + 2] = -1; init[4 * l + 3] = -1; }
    for (int p = 0; p < nb; ++p)
      if (hipMemcpy(reinterpret_cast<char*>(b.gate_work) + size_t(p) * stride, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
        return fail(c, PF_ERR_DEVICE, "initialising the batch slabs failed");
    c->slab_stride = stride; c->slab_work_off = work_off; c->slab_pairs = nb;
  }
  return 0;
}

// device work area of k_gate_bbox_all (self-resetting: initialised once)
int* gate_work(pf_ctx* c) {
  const bool fresh = c->bufs.find("gate_work") == c->bufs.end() || !c->bufs["gate_work"].p;
  int* w = (int*)ensure(c, "gate_work", (4 * kLevelTableMax + 2) * sizeof(int));
  if (w && fresh) {
    std::vector<int> init(4 * kLevelTableMax + 2, 0);
    for (int l = 0; l < kLevelTableMax; ++l) { init[4 * l] = 0x7fffffff; init[4 * l + 1] = 0x7fffffff; init[4 * l + 2] = -1; init[4 * l + 3] = -1; }
    if (hipMemcpy(w, init.data(), init.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess) return nullptr;
  }
  return w;
}
// Host side of k_gate_bbox_all: poll the epoch flag in mapped pinned memory (microseconds) instead of synchronising the
// stream; boxes (4 ints per level) and the level-0 count are then already in host memory.
int wait_gate_boxes(pf_ctx* c, hipStream_t st, int epoch, int nlevels, std::vector<int>& box, unsigned& count0, int pair = 0) {
  const int* hg = c->h_gate + size_t(pair) * kGateWords;
  volatile const int* flag = hg + 4 * kLevelTableMax + 1;
  const auto t0 = std::chrono::steady_clock::now();
  long spins = 0;
  while (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != epoch) {
    // the wait is microseconds long: stay on the core, but leave the pipeline to its sibling thread
#if defined(__x86_64__) || defined(__i386__)
    __builtin_ia32_pause();
#elif defined(__aarch64__)
    asm volatile("yield");
#endif
    if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
      HIPCHK(c, hipStreamSynchronize(st));   // surfaces a launch failure, if that is what happened
      if (__atomic_load_n(flag, __ATOMIC_ACQUIRE) != epoch) return fail(c, PF_ERR_DEVICE, "gate bounding boxes never arrived");
    }
  }
  box.assign(hg, hg + size_t(nlevels) * 4);
  count0 = (unsigned)hg[4 * kLevelTableMax];
  return 0;
}

// The whole solver for 1 or 2 directions on device-resident packed BGRA images, for nb same-size pairs at once.
// dir 0: I0 = img0, I1 = img1, hint0;  dir 1: I0 = img1, I1 = img0, hint1.  out[p * 2 + d]: cols x rows float2 (pad cropped).
// nb == 1: the context's arena.  nb > 1 (throughput mode): the pairs' buffers are slabs of one layout, every kernel covers all pairs
// (blockIdx.z = pair) -- ONE kernel boundary per step of the algorithm for nb pairs; the sweeps of a level share one window, the
// union of the pairs' bounding boxes (a sweep over a larger window gives the same result: pixels outside a pair's own box are not
// gated and keep their flow).
int solve_n(pf_ctx* c, int nb, const uint8_t* const* d_img0, const uint8_t* const* d_img1, int cols, int rows, int pad, int max_pct, int ndirs, const int* hints,
            float* const* d_out, float** used_out = nullptr /* [nb * 2]: where each flow went (a NULL d_out entry of a batch = a plane inside the pair's slab) */) {
  if (int e = check_dims(c, cols, rows, pad)) return e;
  if (max_pct < 0 || max_pct > 100) return fail(c, PF_ERR_ARG, "max_percentage %d out of range", max_pct);
  if (nb < 1 || nb > kMaxBatch) return fail(c, PF_ERR_ARG, "batch of %d pairs (1..%d)", nb, kMaxBatch);
  const Geometry g = make_geometry(cols, rows, pad, c->sp.pyr_scale_factor);
  SolveBufs sb;
  Batch bt;
  if (g.n > kLevelTableMax && c->sp.pyr_scale_factor != kPyrScaleFactor) return fail(c, PF_ERR_ARG, "pyrScaleFactor %g gives %d pyramid levels (at most %d)", (double)c->sp.pyr_scale_factor, g.n, kLevelTableMax);
  if (nb == 1) { if (int e = alloc_solve(c, g, ndirs, sb)) return e; }
  else {
    if (g.n > kLevelTableMax || g.P >= (size_t(1) << 31)) return fail(c, PF_ERR_ARG, "image too large for a batched solve");
    size_t stride = 0;
    if (int e = alloc_solve_batch(c, g, nb, sb, stride)) return e;
    bt.n = nb; bt.stride = stride;
  }
  float** pyrI = sb.pyrI; float** pyrA = sb.pyrA; float** grad = sb.grad;
  uint8_t* gate = sb.gate; float* half_tmp = sb.half_tmp;
  const std::vector<size_t>& bnd_off = sb.bnd_off; const size_t bnd_total = sb.bnd_total;
  LevelBufs* lb = sb.lb; unsigned long long** bnd = sb.bnd; int** ctrl = sb.ctrl; float** ratio = sb.ratio;
  *c->h_status = 0;
  hipStream_t sm = c->s_main;
  // --- shared front end on the main stream: half-res planes, pyramids, gradients + gate of ALL levels.
  // (Measured and rejected: the alpha path on a second stream -- alpha pyramids, gate, boxes beside the grey path: +0.3 ms
  // per pair with 72 instead of 36 small pyramid launches in front of the boxes; profiles/r02_frontend_ab.txt.) ---
  hipStream_t sg = sm;
  for (int i = 0; i < 2; ++i) {
    ExtPtrs imgs{};
    for (int p = 0; p < nb; ++p) imgs.p[p] = i ? d_img1[p] : d_img0[p];
    { PROF(c, sm, "downscale_gray"); launch_downscale_gray(sm, nullptr, cols, rows, pad, half_tmp, pyrA[i], g.w0, g.h0, bt, &imgs); }
    { PROF(c, sm, "preblur5"); launch_gauss_small(sm, half_tmp, pyrI[i], g.w0, g.h0, 1, c->g5, bt); }
  }
  // pyramids: one launch per level while the levels are large, then two and three levels per launch (the chain of dependent
  // ~5 us launches is otherwise ~0.2 ms in front of everything; kernels_pre.hip: k_pyr_chain)
  const int chainMode = c->cfg.pyramid_chaining;
  for (int l = 1; l < g.n;) {
    PROF(c, sm, "pyr_down");
    const size_t px = size_t(g.ws[l]) * g.hs[l];
    int k = 1;
    if (chainMode) { if (px <= 40000 && l + 2 < g.n) k = 3; else if (px <= 160000 && l + 1 < g.n) k = 2; }
    if (k == 1)
      launch_pyr_down4(sm, pyrI[0] + g.off[l - 1], pyrI[1] + g.off[l - 1], pyrA[0] + g.off[l - 1], pyrA[1] + g.off[l - 1], g.ws[l - 1], g.hs[l - 1],
                       pyrI[0] + g.off[l], pyrI[1] + g.off[l], pyrA[0] + g.off[l], pyrA[1] + g.off[l], g.ws[l], g.hs[l], bt);
    else
      launch_pyr_chain4(sm, pyrI[0], pyrI[1], pyrA[0], pyrA[1], g.ws.data(), g.hs.data(), g.off.data(), l - 1, k, bt);
    l += k;
  }
  // The host needs the per-level bounding boxes of the gate (they size the sweep launches) and the level-0 gate count (dense
  // or sparse sweep variant; full-canvas inputs, CPU/StitchTool.cpp:17-33): one fused kernel computes gate, boxes and count
  // and publishes them into mapped pinned memory; the host polls its epoch flag (microseconds, no blocking sync, no pageable
  // copies) while the gradients of all levels and the hand-off initialisation are still running behind it.
  bool have_table = false; LevelTable table;
  const int split = g.n > 10 ? 8 : 0;   // levels [0, split) are "fine": 80 % of the pixels of a 0.9x pyramid
  unsigned h_cnt = 0;
  std::vector<int> boxes;
  int epoch = 0;
  if (g.n <= kLevelTableMax && g.P < (size_t(1) << 31)) {
    LevelTable t; t.n = g.n;
    for (int l = 0; l < g.n; ++l) { t.w[l] = g.ws[l]; t.h[l] = g.hs[l]; t.off[l] = (unsigned)g.off[l]; }
    int* work = nb == 1 ? gate_work(c) : sb.gate_work;
    if (!work) return PF_ERR_NOMEM;
    epoch = ++c->gate_epoch;
    { PROF(c, sg, "gate"); launch_gate_bbox_all(sg, pyrA[0], pyrA[1], gate, t, g.P, work, c->d_gate, epoch, bt, kGateWords * sizeof(int)); }
    // gradients: the coarse levels first (a few percent of the pixels) -- the directions start on those -- the fine levels in a
    // second launch that runs while the coarse levels are already being solved (ev_fine, waited for at level split - 1)
    { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], t, g.off[split], g.P, c->g3_05, 0, bt); }
    have_table = true; table = t;
  } else {
    for (int l = 0; l < g.n; ++l) {
      PROF(c, sm, "gradients");
      launch_gradients(sm, pyrI[0] + g.off[l], g.ws[l], g.hs[l], grad[0] + 2 * g.off[l], c->g3_05);
      launch_gradients(sm, pyrI[1] + g.off[l], g.ws[l], g.hs[l], grad[1] + 2 * g.off[l], c->g3_05);
      launch_gate(sg, pyrA[0] + g.off[l], pyrA[1] + g.off[l], g.ws[l] * g.hs[l], gate + g.off[l]);
    }
  }
  for (int d = 0; d < ndirs; ++d) {
    PROF(c, sm, "init_handoff");
    launch_fill_u64(sm, bnd[d], bnd_total * 2, kNotReady, bt);
    launch_fill_u32(sm, reinterpret_cast<unsigned*>(ctrl[d]), size_t(g.n) * 2 * 2, 0u, bt);
    launch_fill_u32(sm, reinterpret_cast<unsigned*>(sb.prepcnt[d]), sb.pc_total, 0u, bt);
  }
  HIPCHK(c, hipEventRecord(c->ev_pre, sm));
  // Fine levels in two launches behind the coarse ones: levels [split2, split) (needed first, a quarter of the fine pixels), then the
  // finest levels [0, split2).  For a lone pair BOTH are NARROW launches: they run beside the sweeps of ~30 coarser levels and are not
  // needed for milliseconds, while at full width they take every wave slot of the chip -- and a sweep workgroup needs 11 free wave
  // slots and 115 KB of LDS on ONE CU: the first sweep of the first direction used to wait ~150 us for the full-width launch of
  // [split2, split) to drain (kernel timeline, tests/micro/pair_timeline.py), and the late direction, which starts k levels behind the
  // first and ends the call, with it (dense pair 46.23 -> 46.11 ms, profiles/r04_sweep_helpers_ab.txt 8).
  // (a batch keeps every CU busy anyway -- there is nothing to hide a narrow launch behind, and at 64 blocks per image it would run
  // for the whole solve: full width, pf_config::full_width_batch_gradients)
  const int fineBlocks = (nb > 1 && c->cfg.full_width_batch_gradients) ? 0 : c->cfg.fine_gradient_blocks;
  const int split2 = split > 4 ? 4 : 0;
  if (have_table && split > 0) { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], table, g.off[split2], g.off[split], c->g3_05, fineBlocks, bt); }
  HIPCHK(c, hipEventRecord(c->ev_fine, sm));
  if (have_table && split2 > 0) { PROF(c, sm, "gradients"); launch_gradients_all(sm, pyrI[0], pyrI[1], grad[0], grad[1], table, 0, g.off[split2], c->g3_05, fineBlocks, bt); }
  HIPCHK(c, hipEventRecord(c->ev_fine2, sm));
  double area0 = (double)g.ws[0] * g.hs[0];   // the sweeps only cover the window of gated pixels: density inside that window is what counts
  if (have_table) {
    // one set of boxes per pair; a batch sweeps the union (a superset of each pair's own window: same results)
    for (int p = 0; p < nb; ++p) {
      std::vector<int> bp; unsigned cnt = 0;
      if (int e = wait_gate_boxes(c, sg, epoch, g.n, bp, cnt, p)) return e;
      h_cnt += cnt;
      if (p == 0) boxes = bp;
      else for (int l = 0; l < g.n; ++l) {
        if (bp[4 * l + 2] < bp[4 * l] || bp[4 * l + 3] < bp[4 * l + 1]) continue;                          // this pair gates nothing at level l
        if (boxes[4 * l + 2] < boxes[4 * l] || boxes[4 * l + 3] < boxes[4 * l + 1]) { for (int k = 0; k < 4; ++k) boxes[4 * l + k] = bp[4 * l + k]; continue; }
        boxes[4 * l] = std::min(boxes[4 * l], bp[4 * l]); boxes[4 * l + 1] = std::min(boxes[4 * l + 1], bp[4 * l + 1]);
        boxes[4 * l + 2] = std::max(boxes[4 * l + 2], bp[4 * l + 2]); boxes[4 * l + 3] = std::max(boxes[4 * l + 3], bp[4 * l + 3]);
      }
    }
    if (!c->cfg.sweep_window) boxes.clear();
  } else {
    unsigned* d_cnt = (unsigned*)ensure(c, "gate_count", 256);
    if (!d_cnt) return PF_ERR_NOMEM;
    HIPCHK(c, hipMemsetAsync(d_cnt, 0, 4, sg));
    launch_count_gate(sg, gate, g.ws[0] * g.hs[0], d_cnt);
    HIPCHK(c, hipMemcpyAsync(&h_cnt, d_cnt, 4, hipMemcpyDeviceToHost, sg));
    HIPCHK(c, hipStreamSynchronize(sg));
  }
  if (!boxes.empty() && boxes[2] >= boxes[0] && boxes[3] >= boxes[1]) area0 = double(boxes[2] - boxes[0] + 1) * double(boxes[3] - boxes[1] + 1);
  int sparse = (double)h_cnt < 0.5 * area0 * nb ? 1 : 0;
  if (c->cfg.sparse_sweep >= 0) sparse = c->cfg.sparse_sweep ? 1 : 0;   // forced variant: results are identical either way
  // critical path of the exact sweeps given the windows: (w + h - 1) anti-diagonals per sweep, two sweeps per level
  c->last_swept_steps = 0;
  for (int l = 0; l < g.n; ++l) {
    int bw = g.ws[l], bh = g.hs[l];
    if (!boxes.empty()) { bw = boxes[4 * l + 2] - boxes[4 * l] + 1; bh = boxes[4 * l + 3] - boxes[4 * l + 1] + 1; }
    if (bw > 0 && bh > 0) c->last_swept_steps += 2 * (long long)(bw + bh - 1);
  }

  // --- the two directions are independent (OpticalFlow.cpp:130-139): one stream each.  The host enqueues them level by
  // level in turn (a direction's ~430 launches take the host >1 ms: enqueued one after the other, the second
  // direction's stream would sit idle that long) ---
  for (int d = 0; d < ndirs; ++d) HIPCHK(c, hipStreamWaitEvent(c->s_dir[d], c->ev_pre, 0));
  // Fewer launches or shorter launches?  Alone, a pair is faster with the separate upsample kernel (strip 27.36 vs 27.44 ms); with
  // several pairs in flight the time between a stream's kernels dominates and one launch fewer per level wins (+3 %): the
  // throughput mode turns the fusion on for its lanes (pf_novel_view_batch_dev).  pf_config::fuse_small_level_px overrides both.
  const long fuseUpsPx = fuse_small_px(c);
  auto fuse_ups = [&](int level) { return (long)g.ws[level] * g.hs[level] <= fuseUpsPx; };   // level whose incoming flow is upsampled inside its Gaussian
  float* prev_res[2] = {nullptr, nullptr};
  auto enqueue_level = [&](int d, int level) {
    hipStream_t st = c->s_dir[d];
    if (level == split - 1) hipStreamWaitEvent(st, c->ev_fine, 0);   // first level whose gradients come from the second launch
    if (split2 > 0 && level == split2 - 1) hipStreamWaitEvent(st, c->ev_fine2, 0);   // ... from the third (narrow) launch
    const int i0 = d, i1 = 1 - d;
    LevelBufs& b = lb[d];
    const int w = g.ws[level], h = g.hs[level];
    const size_t o = g.off[level];
    if (level == g.n - 1) {
      launch_fill_u32(st, reinterpret_cast<unsigned*>(b.flow_a), size_t(w) * h * 2, 0u, bt);  // PixFlow.hpp:298
      if (max_pct > 0 && hints[d] != PF_HINT_UNKNOWN) {
        PROF(c, st, "adjust_initial_flow");
        launch_adjust_initial_flow(st, pyrI[i0] + o, pyrI[i1] + o, pyrA[i0] + o, pyrA[i1] + o, w, h, hints[d], max_pct, ratio[d], b.flow_a, bt);
      }
    }
    float* res = nullptr;
    // small levels: the upsample of the previous (coarser) level's result rides in this level's first Gaussian
    const bool upsHere = level < g.n - 1 && fuse_ups(level);
    run_level(c, st, grad[i0] + 2 * o, grad[i1] + 2 * o, pyrA[i0] + o, pyrA[i1] + o, gate + o, w, h, sparse, boxes.empty() ? nullptr : &boxes[4 * level], b,
              bnd[d] + bnd_off[level],
              bnd[d] + bnd_total + bnd_off[level], ctrl[d] + level * 4, ctrl[d] + level * 4 + 2, &res,
              sb.prepcnt[d] + sb.pc_off[level], sb.prepcnt[d] + sb.pc_off[level] + sweep2_num_wgs_max(w, h),
              upsHere ? prev_res[d] : nullptr, upsHere ? g.ws[level + 1] : 0, upsHere ? g.hs[level + 1] : 0, bt);
    prev_res[d] = res;
    if (level > 0) {
      if (!fuse_ups(level - 1)) {
        PROF(c, st, "upsample_cubic");
        launch_upsample_cubic(st, res, w, h, b.flow_a, g.ws[level - 1], g.hs[level - 1], 1.0f / c->sp.pyr_scale_factor, bt);
      }
    } else {
      PROF(c, st, "final_flow");
      ExtPtrs outs{};
      for (int p = 0; p < nb; ++p) {
        float* o = d_out[p * 2 + d];
        if (!o && nb > 1) o = reinterpret_cast<float*>(reinterpret_cast<char*>(sb.nv_flow[d]) + size_t(p) * bt.stride);
        outs.p[p] = o;
        if (used_out) used_out[p * 2 + d] = o;
      }
      launch_final_flow(st, res, w, h, g.ce, rows, pad, 1.0f / kDownscaleFactor, c->g3_1, nullptr, bt, &outs);
    }
  };
  // (Measured and rejected: one host thread per direction -- +0.1 ms per pair; the GPU, not the host, paces the launches.)
  // Direction 1 starts when direction 0 has finished its k coarsest levels.  Started together, the two directions stay in lockstep:
  // their throughput kernels (Gaussians, medians, prepass) run beside each other, each at half speed, and their sweeps -- which
  // leave most CUs idle -- run beside each other too.  A small offset puts one direction's throughput kernels beside the other's
  // sweeps.  The late direction finishes k coarse levels later, which is what limits k: measured (profiles/r02_frontend_ab.txt)
  // strip 27.36 -> 27.18 ms at k = 2, 9000x4000 pair 59.1 -> 57.7 ms at k = 4-6.  Not for the lanes of the throughput mode (they are
  // out of phase with each other anyway: -1 %).  Re-measured in round 5 (tests/micro/stagger_ab.py): dense pair k = 0 / 2 / 3 / 4 / 6:
  // 48.30 / 47.46 / 47.31 / 47.45 / 48.10 ms, strip k = 0 / 2 / 3 / 4: 22.37 / 22.28 / 22.45 / 22.65.  pf_config::stagger_levels overrides.
  const int stagger = c->cfg.stagger_levels >= 0 ? c->cfg.stagger_levels : (c->is_lane ? 0 : (size_t(g.w0) * g.h0 >= 5000000 ? 3 : 2));
  if (stagger > 0 && ndirs == 2 && g.n > 1) {
    const int k = stagger < g.n ? stagger : g.n - 1;
    for (int t = 0; t < g.n + k; ++t) {
      const int l0 = g.n - 1 - t, l1 = g.n - 1 - (t - k);
      if (l0 >= 0) {
        enqueue_level(0, l0);
        if (t == k - 1) HIPCHK(c, hipEventRecord(c->ev_stagger, c->s_dir[0]));
      }
      if (t >= k && l1 >= 0) {
        if (t == k) HIPCHK(c, hipStreamWaitEvent(c->s_dir[1], c->ev_stagger, 0));
        enqueue_level(1, l1);
      }
    }
  } else {
    for (int level = g.n - 1; level >= 0; --level)
      for (int d = 0; d < ndirs; ++d) enqueue_level(d, level);
  }
  for (int d = 0; d < ndirs; ++d) {
    launch_collect_status(c->s_dir[d], ctrl[d], g.n * 4, c->d_status, 1 << d, bt);
    HIPCHK(c, hipEventRecord(c->ev_dir[d], c->s_dir[d]));
    HIPCHK(c, hipStreamWaitEvent(sm, c->ev_dir[d], 0));
  }
  HIPCHK(c, hipGetLastError());
  return 0;
}
int solve(pf_ctx* c, const uint8_t* d_img0, const uint8_t* d_img1, int cols, int rows, int pad, int max_pct, int ndirs, const int* hints,
          float* const* d_out) {
  float* outs[2] = {d_out[0], ndirs > 1 ? d_out[1] : nullptr};
  return solve_n(c, 1, &d_img0, &d_img1, cols, rows, pad, max_pct, ndirs, hints, outs);
}
