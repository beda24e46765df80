// Part of pf_api.hip (one translation unit, split along its seams in round 5): the flow / blend / novel-view entry points: device-resident, throughput mode (batches x lanes), host buffers.
// ---- device-resident entry points ----
int pf_flow_bidir_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_pct, float* d_l2r, float* d_r2l) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_l2r || !d_r2l) return fail(c, PF_ERR_ARG, "null device pointer");
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};  // OpticalFlow.cpp:134,139
  float* outs[2] = {d_l2r, d_r2l};
  const int pad = cols / 20;                            // OpticalFlow.cpp:113
  if (int e = solve(c, d_l, d_r, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_blend_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, const float* d_l2r, const float* d_r2l, const float* d_blend, int cols,
                 int rows, uint8_t* d_out) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_l2r || !d_r2l || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, d_l, d_r, d_l2r, d_r2l, d_blend, cols, rows, d_out); }
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_novel_view_dev(pf_ctx* c, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_pct, const float* d_blend, uint8_t* d_out,
                      float* d_l2r, float* d_r2l) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!d_l || !d_r || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "null device pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  float* f0 = d_l2r ? d_l2r : (float*)ensure(c, "nv_flow_l2r", size_t(cols) * rows * 8);
  float* f1 = d_r2l ? d_r2l : (float*)ensure(c, "nv_flow_r2l", size_t(cols) * rows * 8);
  if (!f0 || !f1) return PF_ERR_NOMEM;
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};
  float* outs[2] = {f0, f1};
  const int pad = cols / 20;
  if (int e = solve(c, d_l, d_r, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, d_l, d_r, f0, f1, d_blend, cols, rows, d_out); }
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// ---- throughput mode ----
// One pair keeps ~70 workgroups of a sweep busy (two directions x ~35 bands-of-4): the exact sweeps are a dependency chain, so
// most of the 256 CUs idle.  When pairs are plentiful, `in_flight` of them are on the GPU at the same time, in two ways that combine:
//   * BATCHES (round 3): B pairs go through the SAME launches (blockIdx.z = pair, solve_n): one kernel boundary per step of the
//     algorithm for B pairs.  With several independent streams the kernels themselves barely slow down, but the gap between a
//     stream's dependent kernels grows with the number of busy hardware queues (2.7 -> 23 us per launch from 1 to 2 pairs in
//     flight, profiles/r02_throughput_mode.txt); a batch pays each gap once for B pairs;
//   * LANES: further stream / buffer sets on the same device ("lane", created on first use and kept), each driven by its own host
//     thread and each working through its own batches, out of phase with the others.
// in_flight = lanes x pairs per batch; pf_config::batch_pairs picks the split (-1: see batch_split()).  Results are identical to
// n_pairs calls of pf_novel_view_dev.  Set GPU_MAX_HW_QUEUES >= 3 * lanes + 2 before the first HIP call.
namespace {
void batch_split(const pf_ctx* c, int in_flight, int& lanes, int& per_batch) {
  // measured (24 strips of 2000x4000, Mpix/s, tests/micro/tp_batch_sweep.sh): 6 in flight as 6 lanes 838, 3 x 2 911, 2 x 3 974, one batch of 6 1071;
  // 8 in flight as 4 x 2 1032, 2 x 4 1131, one batch of 8 1237; 12 = 2 lanes x 6 1326; 16 = 2 x 8 1368: the fewest lanes win
  per_batch = c->cfg.batch_pairs > 0 ? c->cfg.batch_pairs : (in_flight <= kMaxBatch ? in_flight : (in_flight + 1) / 2);
  if (per_batch > kMaxBatch) per_batch = kMaxBatch;
  if (per_batch > in_flight) per_batch = in_flight;
  lanes = (in_flight + per_batch - 1) / per_batch;
}
// one batch: pairs [first, first + count) of the arrays through one set of launches on `lane`
int novel_view_group(pf_ctx* lane, int first, int count, const uint8_t* const* d_l, const uint8_t* const* d_r, int cols, int rows, int max_pct,
                     const float* const* d_blend, uint8_t* const* d_out, float* const* d_l2r, float* const* d_r2l) {
  if (count == 1)
    return pf_novel_view_dev(lane, d_l[first], d_r[first], cols, rows, max_pct, d_blend[first], d_out[first], d_l2r ? d_l2r[first] : nullptr, d_r2l ? d_r2l[first] : nullptr);
  if (int e = use(lane)) return e;
  CallGuard guard_(lane);
  if (int e = check_dims(lane, cols, rows, cols / 20)) return e;
  float* outs[2 * kMaxBatch]; float* used[2 * kMaxBatch];
  for (int p = 0; p < count; ++p) {
    if (!d_l[first + p] || !d_r[first + p] || !d_blend[first + p] || !d_out[first + p]) return fail(lane, PF_ERR_ARG, "null device pointer (pair %d)", first + p);
    outs[2 * p] = d_l2r ? d_l2r[first + p] : nullptr; outs[2 * p + 1] = d_r2l ? d_r2l[first + p] : nullptr;
  }
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT};
  if (int e = solve_n(lane, count, d_l + first, d_r + first, cols, rows, cols / 20, max_pct, 2, hints, outs, used)) return e;
  BlendPtrs bp{};
  for (int p = 0; p < count; ++p) { bp.L[p] = d_l[first + p]; bp.R[p] = d_r[first + p]; bp.fLR[p] = used[2 * p]; bp.fRL[p] = used[2 * p + 1]; bp.blend[p] = d_blend[first + p]; bp.out[p] = d_out[first + p]; }
  { PROF(lane, lane->s_main, "blend"); launch_blend_batch(lane->s_main, bp, count, cols, rows); }
  HIPCHK(lane, hipGetLastError());
  if (int e = finish(lane)) return e;
  return check_sweeps(lane);
}
}  // namespace
int pf_novel_view_batch_dev(pf_ctx* c, int n_pairs, const uint8_t* const* d_l, const uint8_t* const* d_r, int cols, int rows, int max_pct,
                            const float* const* d_blend, uint8_t* const* d_out, float* const* d_l2r, float* const* d_r2l, int in_flight) {
  if (int e = use(c)) return e;
  if (n_pairs < 0 || !d_l || !d_r || !d_blend || !d_out) return fail(c, PF_ERR_ARG, "bad argument");
  if (in_flight < 1) in_flight = 1;
  if (in_flight > 2 * kMaxBatch) in_flight = 2 * kMaxBatch;
  if (in_flight > n_pairs) in_flight = n_pairs > 0 ? n_pairs : 1;
  int nlanes = 1, per_batch = 1;
  batch_split(c, in_flight, nlanes, per_batch);
  check_hw_queues(c, 3 * nlanes + 2, "pf_novel_view_batch_dev");   // three streams per lane + this context's blend-ramp and copy streams
  while ((int)c->lanes.size() < nlanes - 1) {
    pf_config lc = c->cfg; lc.max_cols = per_batch > 1 ? 0 : cols; lc.max_rows = per_batch > 1 ? 0 : rows;   // a batching lane lives in its slabs: nothing to pre-size
    pf_ctx* l = create_ctx(lc, true);
    if (!l) return fail(c, PF_ERR_NOMEM, "cannot create lane %d: %s", (int)c->lanes.size() + 1, g_err.c_str());
    c->lanes.push_back(l);
  }
  for (pf_ctx* l : c->lanes) { l->prof = c->prof; l->sp = c->sp; l->cf = c->cf; }   // profiling covers every lane (collected into the lane's own totals); lanes solve with the owner's parameters
  const int ngroups = (n_pairs + per_batch - 1) / per_batch;
  std::vector<int> rc(nlanes, 0);
  std::vector<std::string> msg(nlanes);
  auto run = [&](int k) {
    pf_ctx* lane = k == 0 ? c : c->lanes[k - 1];
    struct Restore { pf_ctx* l; long v; bool b; ~Restore() { l->fuse_ups_px = v; l->is_lane = b; l->lanes_running = 1; } } restore{lane, lane->fuse_ups_px, lane->is_lane};
    lane->lanes_running = nlanes;
    // lanes side by side: launches count more than their length, the small levels fold two kernels into their neighbours (see solve_n()).
    // A batch pays every launch once for all its pairs, and there the separate (shorter) kernels win again: 8 in one batch 1374 vs 1347 Mpix/s.
    if (in_flight > 1) { lane->fuse_ups_px = per_batch > 1 ? 0 : 262144; lane->is_lane = true; }
    for (int gidx = k; gidx < ngroups; gidx += nlanes) {
      const int first = gidx * per_batch, count = std::min(per_batch, n_pairs - first);
      const int e = novel_view_group(lane, first, count, d_l, d_r, cols, rows, max_pct, d_blend, d_out, d_l2r, d_r2l);
      if (e) { rc[k] = e; msg[k] = lane->err; return; }
    }
  };
  std::vector<std::thread> th;
  for (int k = 1; k < nlanes; ++k) th.emplace_back(run, k);
  run(0);
  for (auto& t : th) t.join();
  for (int k = 0; k < nlanes; ++k) if (rc[k]) return fail(c, rc[k], "lane %d: %s", k, msg[k].c_str());
  if (c->prof)   // per-kernel-family times of the lanes are reported with the owning context's
    for (pf_ctx* l : c->lanes)
      for (size_t i = 0; i < l->prof_names.size(); ++i) {
        const int id = prof_id(c, l->prof_names[i].c_str());
        c->prof_tot[id].ms += l->prof_tot[i].ms; c->prof_tot[id].n += l->prof_tot[i].n;
        l->prof_tot[i] = ProfEntry();
      }
  return 0;
}

// ---- host-buffer entry points ----
int pf_flow(pf_ctx* c, const uint8_t* i0, const uint8_t* i1, int cols, int rows, size_t step, int max_pct, int hint, float* flow, size_t fstep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!i0 || !i1 || !flow) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, 0)) return e;
  if (step < size_t(cols) * 4 || fstep < size_t(cols) * 8) return fail(c, PF_ERR_ARG, "row step too small");
  if (hint < 0 || hint > 4) return fail(c, PF_ERR_ARG, "unexpected direction %d", hint);
  const size_t ib = size_t(cols) * rows * 4;
  uint8_t* d0 = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* d1 = (uint8_t*)ensure(c, "h_img1", ib);
  float* df = (float*)ensure(c, "h_flow0", size_t(cols) * rows * 8);
  if (!d0 || !d1 || !df) return PF_ERR_NOMEM;
  if (int e = up2d(c, d0, size_t(cols) * 4, i0, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, d1, size_t(cols) * 4, i1, step, size_t(cols) * 4, rows)) return e;
  const int hints[2] = {hint, hint}; float* outs[2] = {df, nullptr};
  if (int e = solve(c, d0, d1, cols, rows, 0, max_pct, 1, hints, outs)) return e;
  if (int e = down2d(c, flow, fstep, df, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_novel_view(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, const float* blend, size_t bstep,
                  uint8_t* out, size_t ostep, float* f_l2r, float* f_r2l, size_t fstep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  if (step < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  if (out && !blend) return fail(c, PF_ERR_ARG, "blend is required when out_bgra is given");
  const size_t ib = size_t(cols) * rows * 4, fb = size_t(cols) * rows * 8;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", ib);
  float* d0 = (float*)ensure(c, "h_flow0", fb); float* d1 = (float*)ensure(c, "h_flow1", fb);
  if (!dl || !dr || !d0 || !d1) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT}; float* outs[2] = {d0, d1};
  const int pad = cols / 20;
  if (int e = solve(c, dl, dr, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  if (out) {
    float* db = (float*)ensure(c, "h_blend", size_t(cols) * rows * 4); uint8_t* dout = (uint8_t*)ensure(c, "h_out", ib);
    if (!db || !dout) return PF_ERR_NOMEM;
    if (int e = up2d(c, db, size_t(cols) * 4, blend, bstep, size_t(cols) * 4, rows)) return e;
    { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, dl, dr, d0, d1, db, cols, rows, dout); }
    if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  }
  if (f_l2r) if (int e = down2d(c, f_l2r, fstep, d0, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (f_r2l) if (int e = down2d(c, f_r2l, fstep, d1, size_t(cols) * 8, size_t(cols) * 8, rows)) return e;
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

int pf_flow_bidir(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, float* f_l2r, float* f_r2l,
                  size_t fstep) {
  return pf_novel_view(c, l, r, cols, rows, step, max_pct, nullptr, 0, nullptr, 0, f_l2r, f_r2l, fstep);
}

int pf_blend(pf_ctx* c, const uint8_t* l, const uint8_t* r, size_t step, const float* f_l2r, const float* f_r2l, size_t fstep, const float* blend,
             size_t bstep, int cols, int rows, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !f_l2r || !f_r2l || !blend || !out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || ostep < size_t(cols) * 4 || bstep < size_t(cols) * 4 || fstep < size_t(cols) * 8) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t ib = size_t(cols) * rows * 4, fb = size_t(cols) * rows * 8;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", ib); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", ib); uint8_t* dout = (uint8_t*)ensure(c, "h_out", ib);
  float* d0 = (float*)ensure(c, "h_flow0", fb); float* d1 = (float*)ensure(c, "h_flow1", fb); float* db = (float*)ensure(c, "h_blend", ib);
  if (!dl || !dr || !dout || !d0 || !d1 || !db) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, d0, size_t(cols) * 8, f_l2r, fstep, size_t(cols) * 8, rows)) return e;
  if (int e = up2d(c, d1, size_t(cols) * 8, f_r2l, fstep, size_t(cols) * 8, rows)) return e;
  if (int e = up2d(c, db, size_t(cols) * 4, blend, bstep, size_t(cols) * 4, rows)) return e;
  { PROF(c, c->s_main, "blend"); launch_blend(c->s_main, dl, dr, d0, d1, db, cols, rows, dout); }
  if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}
