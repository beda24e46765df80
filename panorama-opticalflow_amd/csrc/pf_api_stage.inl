// Part of pf_api.hip (one translation unit, split along its seams in round 5): the stage-level entry points the parity tests call, profiling queries, geometry queries.
// ---- stage-level entry points (tests) ----
#define STAGE_BEGIN(c) if (int e_ = use(c)) return e_; CallGuard guard_(c); hipStream_t sm = c->s_main; (void)sm
static void* stage_up(pf_ctx* c, const char* name, const void* host, size_t bytes) {
  void* d = ensure(c, name, bytes);
  if (d && host) hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, c->s_main);
  return d;
}
static int stage_down(pf_ctx* c, void* host, const void* dev, size_t bytes) {
  HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->s_main));
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_stage_preprocess(pf_ctx* c, const uint8_t* bgra, int cols, int rows, int pad, float* gray_half, float* alpha_half) {
  STAGE_BEGIN(c);
  if (int e = check_dims(c, cols, rows, pad)) return e;
  const int dw = int((cols + 2 * pad) * kDownscaleFactor), dh = int(rows * kDownscaleFactor);
  uint8_t* d = (uint8_t*)stage_up(c, "sg_a", bgra, size_t(cols) * rows * 4);
  float* t = (float*)ensure(c, "sg_b", size_t(dw) * dh * 4); float* g = (float*)ensure(c, "sg_c", size_t(dw) * dh * 4); float* a = (float*)ensure(c, "sg_d", size_t(dw) * dh * 4);
  if (!d || !t || !g || !a) return PF_ERR_NOMEM;
  launch_downscale_gray(sm, d, cols, rows, pad, t, a, dw, dh);
  launch_gauss_small(sm, t, g, dw, dh, 1, c->g5);
  HIPCHK(c, hipMemcpyAsync(gray_half, g, size_t(dw) * dh * 4, hipMemcpyDeviceToHost, sm));
  return stage_down(c, alpha_half, a, size_t(dw) * dh * 4);
}
int pf_stage_pyr_down(pf_ctx* c, const float* src, int sw, int sh, float* dst, int dw, int dh) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", src, size_t(sw) * sh * 4); float* d = (float*)ensure(c, "sg_b", size_t(dw) * dh * 4);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_resize_linear(sm, s, sw, sh, d, dw, dh, 1, 1.f, false);
  return stage_down(c, dst, d, size_t(dw) * dh * 4);
}
int pf_stage_gradients(pf_ctx* c, const float* img, int w, int h, float* gxy) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", img, size_t(w) * h * 4); float* d = (float*)ensure(c, "sg_b", size_t(w) * h * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_gradients(sm, s, w, h, d, c->g3_05);
  return stage_down(c, gxy, d, size_t(w) * h * 8);
}
int pf_stage_gauss(pf_ctx* c, const float* src, int w, int h, int cn, int ksize, double sigma, float* dst) {
  STAGE_BEGIN(c);
  if (!((ksize == 3 || ksize == 5) && (cn == 1 || cn == 2)) && !(ksize == 15 && cn == 2)) return fail(c, PF_ERR_ARG, "unsupported gaussian %d/%d", ksize, cn);
  const size_t nb = size_t(w) * h * cn * 4;
  float* s = (float*)stage_up(c, "sg_a", src, nb); float* d = (float*)ensure(c, "sg_b", nb); float* t = (float*)ensure(c, "sg_c", nb);
  if (!s || !d || !t) return PF_ERR_NOMEM;
  const Gauss g = make_gauss(ksize, sigma);
  if (ksize == 15) launch_gauss15(sm, s, t, d, w, h, g); else launch_gauss_small(sm, s, d, w, h, cn, g);
  return stage_down(c, dst, d, nb);
}
int pf_stage_median5(pf_ctx* c, const float* flow, int w, int h, float* out) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(w) * h * 8); float* d = (float*)ensure(c, "sg_b", size_t(w) * h * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  // the stage entry runs BOTH forms of the kernel (direct and LDS-tiled; the solver picks by level size) and requires identical bits
  float* d2 = (float*)ensure(c, "sg_c", size_t(w) * h * 8); int* neq = (int*)ensure(c, "sg_d", 256);
  if (!d2 || !neq) return PF_ERR_NOMEM;
  launch_median5_form(sm, s, d, w, h, false);
  launch_median5_form(sm, s, d2, w, h, true);
  HIPCHK(c, hipMemsetAsync(neq, 0, 4, sm));
  launch_count_diff_u32(sm, reinterpret_cast<const uint32_t*>(d), reinterpret_cast<const uint32_t*>(d2), size_t(w) * h * 2, neq);
  int hneq = 0;
  HIPCHK(c, hipMemcpyAsync(&hneq, neq, 4, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, out, d, size_t(w) * h * 8)) return e;
  if (hneq) return fail(c, PF_ERR_DEVICE, "median5: the direct and the LDS-tiled kernel disagree in %d words", hneq);
  return 0;
}
int pf_stage_sweep(pf_ctx* c, const float* g0, const float* g1, const float* blurred, const float* a0, const float* a1, float* flow, int w, int h, int forward) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* dg0 = (float*)stage_up(c, "sg_a", g0, n * 8); float* dg1 = (float*)stage_up(c, "sg_b", g1, n * 8); float* dbl = (float*)stage_up(c, "sg_c", blurred, n * 8);
  float* da0 = (float*)stage_up(c, "sg_d", a0, n * 4); float* da1 = (float*)stage_up(c, "sg_e", a1, n * 4); float* df = (float*)stage_up(c, "sg_f", flow, n * 8);
  uint8_t* gate = (uint8_t*)ensure(c, "sg_g", n);
  const size_t nb = sweep_boundary_elems(w, h);
  unsigned long long* bnd = (unsigned long long*)ensure(c, "sg_h", nb * 8); int* ctrl = (int*)ensure(c, "sg_i", 16);
  if (!dg0 || !dg1 || !dbl || !da0 || !da1 || !df || !gate || !bnd || !ctrl) return PF_ERR_NOMEM;
  launch_gate(sm, da0, da1, (int)n, gate);
  launch_fill_u64(sm, bnd, nb, kNotReady);
  HIPCHK(c, hipMemsetAsync(ctrl, 0, 16, sm));
  int* pcnt = (int*)ensure(c, "sg_pc", 2 * size_t(sweep2_num_wgs_max(w, h)) * sizeof(int));
  if (!pcnt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(pcnt, 0, 2 * size_t(sweep2_num_wgs_max(w, h)) * sizeof(int), sm));
  SweepArgs sa; sa.cf = c->cf; sa.prepcnt = pcnt; sa.g0 = (const float2*)dg0; sa.g1 = (const float2*)dg1; sa.blurred = (const float2*)dbl; sa.gate = gate; sa.flow = (float2*)df;
  sa.boundary = bnd; sa.ctrl = ctrl; sa.W = w; sa.H = h; sa.forward = forward; sa.sparse = (w * h) % 2;   // stage test: exercise both variants
  sa.wide = c->cfg.sweep_wide > 0 ? c->cfg.sweep_wide : 0;   // the sweep form the context was created for (auto = latency form: one pair)
  if (sa.wide == 2) sa.sparse = 0;            // (the throughput form has no sparse variant)
  {
    std::vector<int> box; LevelTable t; t.n = 1; t.w[0] = w; t.h[0] = h; t.off[0] = 0;
    if (int e = gate_boxes_to_host(c, sm, gate, t, n, box)) return e;
    sa.ax0 = box[0]; sa.ay0 = box[1]; sa.ax1 = box[2] + 1; sa.ay1 = box[3] + 1;
  }
  float* rec = (float*)ensure(c, "sg_rec", sweep2_rec_bytes(w, h));
  if (!rec) return PF_ERR_NOMEM;
#ifdef PF_EXPERIMENTS
  sa.prep_mode = c->cfg.record_path;
  if (c->cfg.sweep_impl == 1) { PROF(c, sm, "sweep"); launch_sweep(sm, sa); } else
#endif
  { PROF(c, sm, "sweep"); (void)launch_sweep_any(sm, sa, rec, c->cfg.sweep_impl == 3); }
  int hc[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(hc, ctrl, 16, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, flow, df, n * 8)) return e;
  if (hc[1]) return fail(c, PF_ERR_TIMEOUT, "sweep band timed out");
#ifdef PF_SWEEP_STATS_PRINT
  fprintf(stderr, "[panoflow] sweep %dx%d: edge waits %d, spin iterations %d\n", w, h, hc[2], hc[3]);
#endif
  return 0;
}
int pf_stage_diffusion(pf_ctx* c, const float* a0, const float* a1, float* flow, int w, int h) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* da0 = (float*)stage_up(c, "sg_a", a0, n * 4); float* da1 = (float*)stage_up(c, "sg_b", a1, n * 4); float* df = (float*)stage_up(c, "sg_c", flow, n * 8);
  float* t = (float*)ensure(c, "sg_d", n * 8); float* o = (float*)ensure(c, "sg_e", n * 8);
  if (!da0 || !da1 || !df || !t || !o) return PF_ERR_NOMEM;
  launch_gauss15_mix(sm, df, t, da0, da1, w, h, c->g15, o);
  return stage_down(c, flow, o, n * 8);
}
int pf_stage_upsample_cubic(pf_ctx* c, const float* flow, int sw, int sh, float* out, int dw, int dh, float scale) {
  STAGE_BEGIN(c);
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(sw) * sh * 8); float* d = (float*)ensure(c, "sg_b", size_t(dw) * dh * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_upsample_cubic(sm, s, sw, sh, d, dw, dh, scale);
  return stage_down(c, out, d, size_t(dw) * dh * 8);
}
int pf_stage_final(pf_ctx* c, const float* flow, int sw, int sh, int pad_cols, int rows, int pad, float scale, float* out) {
  STAGE_BEGIN(c);
  const int cols = pad_cols - 2 * pad;
  float* s = (float*)stage_up(c, "sg_a", flow, size_t(sw) * sh * 8); float* d = (float*)ensure(c, "sg_b", size_t(cols) * rows * 8);
  if (!s || !d) return PF_ERR_NOMEM;
  launch_final_flow(sm, s, sw, sh, pad_cols, rows, pad, scale, c->g3_1, d);
  return stage_down(c, out, d, size_t(cols) * rows * 8);
}
int pf_stage_adjust_initial_flow(pf_ctx* c, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, int hint, int max_pct, float* flow_out) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* d0 = (float*)stage_up(c, "sg_a", i0, n * 4); float* d1 = (float*)stage_up(c, "sg_b", i1, n * 4); float* da0 = (float*)stage_up(c, "sg_c", a0, n * 4);
  float* da1 = (float*)stage_up(c, "sg_d", a1, n * 4); float* df = (float*)ensure(c, "sg_e", n * 8); float* rt = (float*)ensure(c, "sg_f", 256);
  if (!d0 || !d1 || !da0 || !da1 || !df || !rt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(df, 0, n * 8, sm));
  if (max_pct > 0) launch_adjust_initial_flow(sm, d0, d1, da0, da1, w, h, hint, max_pct, rt, df);
  return stage_down(c, flow_out, df, n * 8);
}
int pf_stage_level(pf_ctx* c, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h, const float* flow_in, int hint, int max_pct,
                   float* flow_out) {
  STAGE_BEGIN(c);
  const size_t n = size_t(w) * h;
  float* d0 = (float*)stage_up(c, "sg_a", i0, n * 4); float* d1 = (float*)stage_up(c, "sg_b", i1, n * 4); float* da0 = (float*)stage_up(c, "sg_c", a0, n * 4);
  float* da1 = (float*)stage_up(c, "sg_d", a1, n * 4);
  float* g0 = (float*)ensure(c, "sg_e", n * 8); float* g1 = (float*)ensure(c, "sg_f", n * 8); uint8_t* gate = (uint8_t*)ensure(c, "sg_g", n);
  LevelBufs b; b.rec = (float*)ensure(c, "sg_rec", sweep2_rec_bytes(w, h)); if (!b.rec) return PF_ERR_NOMEM;
  b.flow_a = (float*)ensure(c, "sg_h", n * 8); b.flow_b = (float*)ensure(c, "sg_i", n * 8); b.blurred = (float*)ensure(c, "sg_j", n * 8); b.tmp = (float*)ensure(c, "sg_k", n * 8);
  const size_t nb = sweep_boundary_elems(w, h);
  unsigned long long* bnd = (unsigned long long*)ensure(c, "sg_l", nb * 16); int* ctrl = (int*)ensure(c, "sg_m", 16); float* rt = (float*)ensure(c, "sg_n", 256);
  if (!d0 || !d1 || !da0 || !da1 || !g0 || !g1 || !gate || !b.flow_a || !b.flow_b || !b.blurred || !b.tmp || !bnd || !ctrl || !rt) return PF_ERR_NOMEM;
  launch_gradients(sm, d0, w, h, g0, c->g3_05);
  launch_gradients(sm, d1, w, h, g1, c->g3_05);
  launch_gate(sm, da0, da1, (int)n, gate);
  launch_fill_u64(sm, bnd, nb * 2, kNotReady);
  HIPCHK(c, hipMemsetAsync(ctrl, 0, 16, sm));
  if (flow_in) HIPCHK(c, hipMemcpyAsync(b.flow_a, flow_in, n * 8, hipMemcpyHostToDevice, sm));
  else {
    HIPCHK(c, hipMemsetAsync(b.flow_a, 0, n * 8, sm));
    if (max_pct > 0 && hint != PF_HINT_UNKNOWN) launch_adjust_initial_flow(sm, d0, d1, da0, da1, w, h, hint, max_pct, rt, b.flow_a);
  }
  float* res = nullptr;
  std::vector<int> box;
  { LevelTable t; t.n = 1; t.w[0] = w; t.h[0] = h; t.off[0] = 0; if (int e = gate_boxes_to_host(c, sm, gate, t, n, box)) return e; }
  const size_t npc = size_t(sweep2_num_wgs_max(w, h));
  int* pcnt = (int*)ensure(c, "sg_pc", 2 * npc * sizeof(int));
  if (!pcnt) return PF_ERR_NOMEM;
  HIPCHK(c, hipMemsetAsync(pcnt, 0, 2 * npc * sizeof(int), sm));
  run_level(c, sm, g0, g1, da0, da1, gate, w, h, (w + h) % 2, box.data(), b, bnd, bnd + nb, ctrl, ctrl + 2, &res, pcnt, pcnt + npc);
  int hc[4] = {0, 0, 0, 0};
  HIPCHK(c, hipMemcpyAsync(hc, ctrl, 16, hipMemcpyDeviceToHost, sm));
  if (int e = stage_down(c, flow_out, res, n * 8)) return e;
  if (hc[1] || hc[3]) return fail(c, PF_ERR_TIMEOUT, "sweep band timed out");
  return 0;
}
int pf_stage_blend_smooth(pf_ctx* c, float* blend, const float* md, int cols, int rows) {
  STAGE_BEGIN(c);
  const size_t n = size_t(cols) * rows;
  float* db = (float*)stage_up(c, "st_blend", blend, n * 4); float* dmd = (float*)stage_up(c, "st_md", md, n * 4);
  if (!db || !dmd) return PF_ERR_NOMEM;
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (int e = stage_down(c, blend, db, n * 4)) return e;
  return check_sweeps(c);
}

// ---- profiling ----
int pf_profile_enable(pf_ctx* c, int on) { if (!c) return PF_ERR_ARG; c->prof = on < 0 ? 0 : (on > 2 ? 1 : on); return 0; }
int pf_profile_reset(pf_ctx* c) { if (!c) return PF_ERR_ARG; for (auto& t : c->prof_tot) t = ProfEntry(); return 0; }
// (kernel families only: warnings are reported by pf_last_warning / pf_warning_count, not as a pseudo-entry of this list -- round 5 had one)
int pf_profile_count(pf_ctx* c) { return c ? (int)c->prof_names.size() : 0; }
int pf_profile_get(pf_ctx* c, int idx, char* name, int cap, double* ms, int* launches) {
  if (!c || idx < 0 || idx >= (int)c->prof_names.size()) return PF_ERR_ARG;
  if (name && cap > 0) { strncpy(name, c->prof_names[idx].c_str(), cap - 1); name[cap - 1] = 0; }
  if (ms) *ms = c->prof_tot[idx].ms;
  if (launches) *launches = c->prof_tot[idx].n;
  return 0;
}

long long pf_last_swept_steps(pf_ctx* c) { return c ? c->last_swept_steps : 0; }

long long pf_level_pixels(int cols, int rows, int* n_levels, long long* sweep_steps) {
  const Geometry g = make_geometry(cols, rows, cols / 20);
  long long steps = 0;
  for (int l = 0; l < g.n; ++l) steps += g.ws[l] + g.hs[l] - 1;
  if (n_levels) *n_levels = g.n;
  if (sweep_steps) *sweep_steps = 2 * steps;
  return (long long)g.Pexact;
}
double pf_algorithmic_bytes(int cols, int rows) {  // B_alg = 472.75*P + 102.4*C*R
  return 472.75 * (double)pf_level_pixels(cols, rows, nullptr, nullptr) + 102.4 * (double)cols * rows;
}

