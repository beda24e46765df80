// Part of pf_api.hip (one translation unit, split along its seams in round 5): Stitchtools: prepare / match / blend ramp / gather, the device-resident chain step and its prefetch.

static int blend_smooth_dev(pf_ctx* c, float* d_blend, const float* d_md, int cols, int rows, hipStream_t sm = nullptr) {
  const int step = cols <= rows ? cols / 200 : rows / 200, k1 = rows / 130, k2 = rows / 400;
  if (!sm) sm = c->s_main;
  if (step > 0 && k1 > 0) {
    // the tile kernel keeps a (step+k1-1)^2 window and (step+k1-1) x step row sums in LDS: 160 KB per CU bound the canvas at ~15000 rows
    if (tile_blur_lds_bytes(step, k1) > 160 * 1024) return fail(c, PF_ERR_ARG, "canvas %dx%d too large for the blend-ramp tile smoothing (LDS)", cols, rows);
    void* work = ensure(c, "st_tile_work", tile_blur_work_bytes(cols, rows, step, k1) + 256);
    if (!work) return PF_ERR_NOMEM;
    { PROF(c, sm, "tile_blur"); launch_tile_blur(sm, d_blend, d_md, cols, rows, step, k1, work); }
    launch_collect_status(sm, static_cast<const int*>(work), 2, c->d_status, 4);   // word 1 = a grid barrier of the tile smoothing gave up
  }
  if (k2 > 0) {
    double* rs = (double*)ensure(c, "st_rowsum", size_t(cols) * rows * 8);
    float* tmp = (float*)ensure(c, "st_blur_tmp", size_t(cols) * rows * 4);
    if (!rs || !tmp) return PF_ERR_NOMEM;
    PROF(c, sm, "box_blur");
    launch_box_blur(sm, d_blend, tmp, rs, cols, rows, k2);
    HIPCHK(c, hipMemcpyAsync(d_blend, tmp, size_t(cols) * rows * 4, hipMemcpyDeviceToDevice, sm));
  }
  return 0;
}

int pf_stitch_prepare(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, uint8_t* map_out, size_t mstep, uint8_t* ovl,
                      uint8_t* ovr, float* blend_out, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || (map_out && mstep < size_t(cols)) || (blend_out && bstep < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dl || !dr || !dm || !dol || !dor || !db || !dmd) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  { PROF(c, sm, "countblend"); launch_countblend(sm, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (map_out) if (int e = down2d(c, map_out, mstep, dm, cols, cols, rows)) return e;
  if (ovl) if (int e = down2d(c, ovl, step, dol, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (ovr) if (int e = down2d(c, ovr, step, dor, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (blend_out) if (int e = down2d(c, blend_out, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// Stitchtools::MatchImages (StitchTool.cpp:38-50) + the overlap masking of prepare() (:17-33) alone: map and the two masked images.
int pf_stitch_match(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, uint8_t* map_out, size_t mstep, uint8_t* ovl, uint8_t* ovr) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || (map_out && mstep < size_t(cols))) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  if (!dl || !dr || !dm || !dol || !dor) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, c->s_main, "match_images"); launch_match_images(c->s_main, dl, dr, cols, rows, dm, dol, dor); }
  if (map_out) if (int e = down2d(c, map_out, mstep, dm, cols, cols, rows)) return e;
  if (ovl) if (int e = down2d(c, ovl, step, dol, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (ovr) if (int e = down2d(c, ovr, step, dor, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

// Stitchtools::GenerateBlend (StitchTool.cpp:98-146) from a GIVEN map -- the reference reads its public `Map` member there, so a
// caller that edits the map between MatchImages() and GenerateBlend() gets the ramp of the edited map.
int pf_stitch_generate_blend(pf_ctx* c, const uint8_t* map, size_t mstep, int cols, int rows, float* blend_out, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!map || !blend_out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (mstep < size_t(cols) || bstep < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dm || !db || !dmd) return PF_ERR_NOMEM;
  if (int e = up2d(c, dm, cols, map, mstep, cols, rows)) return e;
  { PROF(c, c->s_main, "countblend"); launch_countblend(c->s_main, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows)) return e;
  if (int e = down2d(c, blend_out, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  return check_sweeps(c);
}

// GenerateBlend's per-pixel part alone (StitchTool.cpp:113-125 with countblend :148-191): the ramp BEFORE the tile / global
// box smoothing, i.e. what Stitchtools::countblend(x, y) returns for overlap pixels, and MergedDis.
int pf_stitch_raw_blend(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, float* raw_blend, size_t bstep, float* merged_dis) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !raw_blend) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || bstep < size_t(cols) * 4) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4);
  if (!dl || !dr || !dm || !dol || !dor || !db || !dmd) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  { PROF(c, sm, "countblend"); launch_countblend(sm, dm, cols, rows, db, dmd); }
  if (int e = down2d(c, raw_blend, bstep, db, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  if (merged_dis) if (int e = down2d(c, merged_dis, size_t(cols) * 4, dmd, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}

int pf_stitch_gather(pf_ctx* c, const uint8_t* l, const uint8_t* r, const uint8_t* merged, size_t step, const uint8_t* map, size_t mstep, int cols,
                     int rows, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l || !r || !merged || !map || !out) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_image(c, cols, rows)) return e;
  if (step < size_t(cols) * 4 || ostep < size_t(cols) * 4 || mstep < size_t(cols)) return fail(c, PF_ERR_ARG, "row step too small");
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "h_img0", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "h_img1", n * 4); uint8_t* dg = (uint8_t*)ensure(c, "st_merged", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dout = (uint8_t*)ensure(c, "h_out", n * 4);
  if (!dl || !dr || !dg || !dm || !dout) return PF_ERR_NOMEM;
  if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dg, size_t(cols) * 4, merged, step, size_t(cols) * 4, rows)) return e;
  if (int e = up2d(c, dm, cols, map, mstep, cols, rows)) return e;
  { PROF(c, c->s_main, "gather"); launch_gather(c->s_main, dl, dr, dg, dm, cols, rows, dout); }
  if (int e = down2d(c, out, ostep, dout, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  HIPCHK(c, hipGetLastError());
  return finish(c);
}


// One whole iteration of the reference's stitch loop (CPU/main.cpp:70-95) without leaving the device:
// Stitchtools::prepare -> NovelViewGeneratorAsymmetricFlow::prepare/generateNovelView -> Gather.
// r_bgra == NULL chains on the previous call's result, which stays resident in HBM (main.cpp:64-65).
// Content signature of a host image: 16 evenly spaced rows, 8 bytes at a time (~0.1 ms at 9000x4000).  The prefetched device copy of
// an image is only used if the caller's buffer still carries the signature it had when it was uploaded: pointer, size and step alone
// cannot tell a buffer from another image that an allocator later placed at the same address (the intended use is one cv::Mat freed and
// re-read per image).
static uint64_t host_image_sig(const uint8_t* p, int cols, int rows, size_t step) {
  uint64_t h = 0x9E3779B97F4A7C15ull;
  const size_t rb = size_t(cols) * 4;
  for (int i = 0; i < 16; ++i) {
    const uint8_t* row = p + size_t((long long)(rows - 1) * i / 15) * step;
    for (size_t o = 0; o + 8 <= rb; o += 8) { uint64_t v; memcpy(&v, row + o, 8); h = (h ^ v) * 0xBF58476D1CE4E5B9ull; h ^= h >> 29; }
  }
  return h;
}

int pf_stitch_step(pf_ctx* c, const uint8_t* l, const uint8_t* r, int cols, int rows, size_t step, int max_pct, uint8_t* out, size_t ostep) {
  if (int e = use(c)) return e;
  CallGuard guard_(c);
  if (!l) return fail(c, PF_ERR_ARG, "null pointer");
  if (int e = check_dims(c, cols, rows, cols / 20)) return e;
  if (step < size_t(cols) * 4 || (out && ostep < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "row step too small");
  check_hw_queues(c, 5, "pf_stitch_step");   // front end, two flow directions, blend ramp, prefetch copy
  const size_t n = size_t(cols) * rows;
  uint8_t* dl = (uint8_t*)ensure(c, "ch_l", n * 4); uint8_t* dr = (uint8_t*)ensure(c, "ch_r", n * 4); uint8_t* dfin = (uint8_t*)ensure(c, "ch_final", n * 4);
  uint8_t* dm = (uint8_t*)ensure(c, "st_map", n); uint8_t* dol = (uint8_t*)ensure(c, "st_ovl", n * 4); uint8_t* dor = (uint8_t*)ensure(c, "st_ovr", n * 4);
  float* db = (float*)ensure(c, "st_blend", n * 4); float* dmd = (float*)ensure(c, "st_md", n * 4); uint8_t* dmerged = (uint8_t*)ensure(c, "st_merged", n * 4);
  float* f0 = (float*)ensure(c, "nv_flow_l2r", n * 8); float* f1 = (float*)ensure(c, "nv_flow_r2l", n * 8);
  if (!dl || !dr || !dfin || !dm || !dol || !dor || !db || !dmd || !dmerged || !f0 || !f1) return PF_ERR_NOMEM;
  hipStream_t sm = c->s_main;
  uint8_t* dnext = (uint8_t*)ensure(c, "ch_l_next", n * 4);
  if (!dnext) return PF_ERR_NOMEM;
  // both prefetch records are one-shot: latched and cleared here, whatever this step does with them
  const pf_ctx::HostImage ready = c->ready, hint = c->hint;
  c->ready = pf_ctx::HostImage(); c->hint = pf_ctx::HostImage();
  if (ready.src == l && ready.cols == cols && ready.rows == rows && ready.step == step && ready.sig == host_image_sig(l, cols, rows, step)) {
    // this step's left image was uploaded while the previous step computed: the two buffers trade places (no copy; the old
    // "ch_l" is free -- the previous call drained every stream -- and receives the next prefetch)
    std::swap(c->bufs["ch_l"], c->bufs["ch_l_next"]);
    std::swap(dl, dnext);
  } else {
    if (int e = up2d(c, dl, size_t(cols) * 4, l, step, size_t(cols) * 4, rows)) return e;
  }
  if (r) { if (int e = up2d(c, dr, size_t(cols) * 4, r, step, size_t(cols) * 4, rows)) return e; }
  else {
    if (c->chain_cols != cols || c->chain_rows != rows) return fail(c, PF_ERR_ARG, "pf_stitch_step: no previous result of this size to chain on");
    HIPCHK(c, hipMemcpyAsync(dr, dfin, n * 4, hipMemcpyDeviceToDevice, sm));
  }
  { PROF(c, sm, "match_images"); launch_match_images(sm, dl, dr, cols, rows, dm, dol, dor); }
  // The blend ramp (GenerateBlend + countblend + smoothing, StitchTool.cpp:98-191) only depends on the map and is only
  // needed by the final blend: it runs on its own stream beside the two flow solves.  Its launches (a dozen since the tile smoothing
  // became ONE persistent launch in round 3; ~850 before) are enqueued AFTER the solver's, so that the solver's first kernel is not
  // kept waiting by them.
  if (!c->s_aux) HIPCHK(c, hipStreamCreateWithFlags(&c->s_aux, hipStreamNonBlocking));
  if (!c->s_copy) HIPCHK(c, hipStreamCreateWithFlags(&c->s_copy, hipStreamNonBlocking));
  hipStream_t sa = c->s_aux;
  HIPCHK(c, hipEventRecord(c->ev_aux_go, sm));
  const int hints[2] = {PF_HINT_LEFT, PF_HINT_RIGHT}; float* outs[2] = {f0, f1};
  const int pad = cols / 20;
  if (int e = solve(c, dol, dor, cols, rows, pad, max_pct, 2, hints, outs)) return e;
  HIPCHK(c, hipStreamWaitEvent(sa, c->ev_aux_go, 0));
  { PROF(c, sa, "countblend"); launch_countblend(sa, dm, cols, rows, db, dmd); }
  if (int e = blend_smooth_dev(c, db, dmd, cols, rows, sa)) return e;
  HIPCHK(c, hipEventRecord(c->ev_aux_done, sa));
  HIPCHK(c, hipStreamWaitEvent(sm, c->ev_aux_done, 0));
  { PROF(c, sm, "blend"); launch_blend(sm, dol, dor, f0, f1, db, cols, rows, dmerged); }
  { PROF(c, sm, "gather"); launch_gather(sm, dl, dr, dmerged, dm, cols, rows, dfin); }
  if (out) if (int e = down2d(c, out, ostep, dfin, size_t(cols) * 4, size_t(cols) * 4, rows)) return e;
  // everything of this step is enqueued: upload the NEXT step's left image now (announced with pf_stitch_prefetch); the
  // host-side staging of a pageable source runs while the GPU computes
  if (hint.src && hint.src != l && hint.cols == cols && hint.rows == rows) {
    if (hint.step == size_t(cols) * 4) HIPCHK(c, hipMemcpyAsync(dnext, hint.src, n * 4, hipMemcpyHostToDevice, c->s_copy));
    else HIPCHK(c, hipMemcpy2DAsync(dnext, size_t(cols) * 4, hint.src, hint.step, size_t(cols) * 4, rows, hipMemcpyHostToDevice, c->s_copy));
    HIPCHK(c, hipStreamSynchronize(c->s_copy));
    c->ready = hint;
    c->ready.sig = host_image_sig(hint.src, cols, rows, hint.step);
  }
  HIPCHK(c, hipGetLastError());
  if (int e = finish(c)) return e;
  c->chain_cols = cols; c->chain_rows = rows;
  return check_sweeps(c);
}

// Announce the left image of the pf_stitch_step call AFTER the coming one: the coming step uploads it while its own kernels run
// (the copy is issued after they are enqueued).  One-shot: the hint is consumed by the coming step; the buffer must stay valid
// and unchanged until the step after it has returned, and that step must pass the same pointer / size / step -- anything else
// simply uploads as usual and the prefetched copy is dropped.  NULL cancels.
int pf_stitch_prefetch(pf_ctx* c, const uint8_t* next_l, int cols, int rows, size_t step) {
  if (!c) return fail(nullptr, PF_ERR_ARG, "null context");
  if (next_l && (cols <= 0 || rows <= 0 || step < size_t(cols) * 4)) return fail(c, PF_ERR_ARG, "bad argument");
  c->hint.src = next_l; c->hint.cols = cols; c->hint.rows = rows; c->hint.step = step;
  return 0;
}

