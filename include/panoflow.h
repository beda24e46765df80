/* =============================================================================================
 * panoflow.h -- C ABI of the MI355X-native bidirectional optical-flow blending path.
 *
 * Drop-in boundary for MungoMeng/Panorama-OpticalFlow (reference paths are relative to
 * /root/reference/).  Everything here is plain pointers + sizes; no C++/torch/OpenCV types.  The
 * C++ mirror of the reference's OpticalFlow.hpp / StitchTool.hpp / PixFlow.hpp classes
 * (panorama-opticalflow_amd/include/) is a thin layer over these entry points; INTEGRATION.md shows
 * the binding a maintainer of the reference would add.
 *
 * Conventions
 *   - return 0 on success, negative pf_status on failure; pf_last_error() gives the message.
 *     No exceptions cross this ABI (the C++ wrapper rethrows util::VrCamException).
 *   - images: 8-bit BGRA interleaved, row-major, `step` = bytes per row  (CV_8UC4)
 *     flow  : float (dx,dy) interleaved, pixels of the full-res image      (CV_32FC2)
 *     blend : float in [0,1] = weight of the RIGHT image                   (CV_32FC1)
 *     map   : uint8 region codes 0/50/100/150                              (CV_8UC1)
 *   - every call is synchronous on return; a context is owned by one host thread and one GPU.
 *   - there is NO CPU fallback: without a usable gfx950 device pf_create() fails.
 * ============================================================================================= */
#ifndef PANOFLOW_H_
#define PANOFLOW_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pf_ctx pf_ctx;

typedef enum pf_status {
  PF_OK = 0,
  PF_ERR_ARG = -1,      /* bad argument (null pointer, size out of range, unknown algorithm) */
  PF_ERR_DEVICE = -2,   /* HIP runtime / launch failure */
  PF_ERR_NOMEM = -3,    /* device allocation failed */
  PF_ERR_TIMEOUT = -4   /* a sweep band gave up waiting for its predecessor (never expected) */
} pf_status;

/* OpticalFlowInterface::DirectionHint, CPU/PixFlow.hpp:19 */
typedef enum pf_hint { PF_HINT_UNKNOWN = 0, PF_HINT_RIGHT = 1, PF_HINT_DOWN = 2, PF_HINT_LEFT = 3, PF_HINT_UP = 4 } pf_hint;

/* ---- lifetime -------------------------------------------------------------------------------
 * Replaces the reference's runtime device probe (GPU/OpticalFlow.cpp:132-189, GPU/StitchTool.cpp:33-60). */
int pf_device_count(void);
/* max_cols x max_rows > 0: every device buffer a bidirectional solve / a stitch step of that size needs is allocated
 * now (the first call then pays no allocation); 0 x 0: allocate lazily.  The arena only ever grows, so larger images
 * are still accepted later.  NULL on failure; pf_last_error(NULL) explains. */
pf_ctx* pf_create(int device, int max_cols, int max_rows);
/* The same with the scheduling knobs exposed.  None of them changes a result (tests hold every setting to the same bits);
 * they exist so that a deployment can tune for its image sizes without environment variables.  pf_config_init() fills
 * in the defaults (= what pf_create uses); set struct_size = sizeof(pf_config). */
typedef struct pf_config {
  int struct_size;
  int device, max_cols, max_rows;   /* as pf_create */
  int stagger_levels;       /* direction R->L starts this many coarse levels behind L->R (-1: 2, or 3 for >= 5 Mpix half-res) */
  int64_t fuse_small_level_px; /* levels up to this many pixels fold the upsample / second median into the neighbouring Gaussian
                               launches (-1: 0 for a lone pair, 262144 for the lanes of the throughput mode) */
  int fine_gradient_blocks; /* width of the launch that computes the gradients of the 4 finest levels beside the coarse sweeps (64) */
  int pyramid_chaining;     /* 1: small pyramid levels are built two or three per launch, 0: one launch per level (1) */
  int sweep_window;         /* 1: a sweep covers the bounding box of the gated pixels only, 0: the whole level (1) */
  int sparse_sweep;         /* -1: pick the sweep variant that skips ungated anti-diagonals from the gate density, 0 / 1: force (-1) */
  int batch_pairs;          /* throughput mode (pf_novel_view_batch_dev): pairs that go through ONE set of launches (1..16; in_flight =
                               lanes x batch_pairs, at most 32).  -1: in_flight itself up to 16 (one lane), half of it (two lanes) beyond */
  int sweep_wide;           /* form of the sweep launches.  0 = latency form: 8 lanes per pixel evaluate a step's six energies at once, bands of 8 rows,
                               ONE compute wave per SIMD -- the shortest step, what a lone pair wants.  2 = throughput form: 2 lanes per pixel, the
                               reference's own order without speculation (two gather rounds per step), bands of 32 rows -- less than half the VALU
                               instructions per pixel, what a batch that oversubscribes the chip wants.  -1 (default) = throughput form for the launches of a
                               batch that oversubscribe the chip (sweep_wide_threshold), latency form otherwise.  Same bits in every form.
                               (1 = the latency step with two compute waves per SIMD: measured and rejected, lab build only -- libpanoflow.so answers
                               PF_ERR_ARG.  The throughput form's loader-staged and fused-prepass record paths of rounds 4 / 5 are gone: profiles/.) */
  int sweep_wide_threshold; /* sweep_wide = -1: a launch takes the throughput form when (sweeps running at the same time: pairs of the batch x 2
                               directions x lanes) x (its latency-form workgroups) exceeds this (512: two rounds of the chip) */
  int sweep_throughput_transposed; /* sweep_wide = -1: sweeps whose bands step along y (windows taller than wide, e.g. 2000x4000 strips) may take the
                               throughput form too (1: +17 % on 16 strips in flight; 0 keeps them in the latency form) */
  int full_width_batch_gradients; /* 1: in a batched solve the finest levels' gradients are one full-width launch (nothing to hide them
                               behind: the batch keeps every CU busy anyway), 0: the narrow launch of a lone pair (1) */
  /* Cross-check implementations -- only in libpanoflow_exp.so (the -DPF_EXPERIMENTS build used by the test-suite);
   * libpanoflow.so rejects anything but the defaults with PF_ERR_ARG. */
  int sweep_impl;           /* 2: wavefront sweep (k_sweep_prep + k_sweep2); 1: the independent 64-rows-per-wave kernel; 3: LDS-tile relaxation */
  int record_path;          /* 0: k_sweep_prep in front of the sweep; 1: loader waves compute the records; 2: prepass blocks inside the sweep launch */
} pf_config;
void pf_config_init(pf_config* cfg);
/* Self-test that pf_create already ran once for the context's device: the sweep's asm-block packed-fp32 chains against the
 * compiler-scheduled forms of the same arithmetic, and (round 6) the step's partial DPP writes issued back to back against the same
 * sequence with wait states (the two hardware assumptions of the scheduled sweep TU).  0 = identical bits (or a
 * -DPF_SAFE_PK build, which has no such blocks); > 0 = threads whose results differed (pf_create would have refused the device);
 * < 0 = error code. */
int pf_selftest_packed_chains(pf_ctx* ctx);
pf_ctx* pf_create_cfg(const pf_config* cfg);
void pf_destroy(pf_ctx* ctx);
const char* pf_last_error(const pf_ctx* ctx);  /* ctx may be NULL (creation errors) */
/* Conditions that cost PERFORMANCE, never results: the last warning raised on this context ("" if none) and how many were raised.
 * Today there is one: a call that drives more HIP streams than the HIP runtime has hardware queues (pf_novel_view_batch_dev needs
 * 3 x lanes + 2, pf_stitch_step 5; the runtime sizes its pool from GPU_MAX_HW_QUEUES -- default 4 -- when the process makes its first
 * HIP call) runs correctly but with streams sharing queues.  The library reads GPU_MAX_HW_QUEUES only to report this; it reads no
 * other environment variable (once per process).  A context raises the warning once per condition, not once per call. */
const char* pf_last_warning(const pf_ctx* ctx);
int pf_warning_count(const pf_ctx* ctx);
const char* pf_version(void);

/* makeOpticalFlowByName, CPU/PixFlow.hpp:459-500: "pixflow_low" -> 0, "pixflow_search_20" -> 20,
 * anything else -> PF_ERR_ARG (the reference throws VrCamException). */
int pf_max_percentage_by_name(const char* flow_alg_name);

/* The constructor arguments of the reference's PixFlow<P> (CPU/PixFlow.hpp:46-68).  Its factory only ever passes the values
 * pf_solver_params_init() fills in (:459-497, both presets), but the class accepts any: so does this library (round 6).  The
 * parameters belong to the context and apply to every later solve on it (pf_flow*, pf_novel_view*, pf_stitch_step, the lanes of
 * pf_novel_view_batch_dev).  Results are bit-identical to the reference's arithmetic for every accepted set.
 *   pyr_scale_factor  in [0.25, 0.98]: level sizes int(w * s + 0.5f) while both > 24 (:137-151), inter-level flow scale 1.0f / s (:124);
 *                     at most 96 levels (PF_ERR_ARG from the solve otherwise);
 *   smoothness_coef, vertical_/horizontal_regularization_coef: finite and >= 0 (errorFunction :450-453; a negative coefficient could
 *                     make an energy negative, which the sweep's "keep" sentinel excludes);
 *   gradient_step_size: finite and >= 0 (:321,334).  A power of two in [2^-16, 2^16] (the presets' 0.5 is one) runs at full speed; any
 *                     other value is computed with the IEEE multiply + subtract in every step of the sweeps (same bits as the reference,
 *                     about 2-3x the sweep time);
 *   downscale_factor: must be 0.5 (the 8-bit half-resolution path, :81-83, is built for it): anything else is PF_ERR_ARG;
 *   directional_regularization_coef: stored and ignored, as in the reference (no code there reads it). */
typedef struct pf_solver_params {
  float pyr_scale_factor, smoothness_coef, vertical_regularization_coef, horizontal_regularization_coef, gradient_step_size, downscale_factor,
      directional_regularization_coef;
} pf_solver_params;
void pf_solver_params_init(pf_solver_params* p);   /* 0.9, 0.001, 0.01, 0.01, 0.5, 0.5, 0 */
int pf_set_solver_params(pf_ctx* ctx, const pf_solver_params* p);   /* p == NULL: back to the presets */
int pf_get_solver_params(const pf_ctx* ctx, pf_solver_params* out);

/* ---- host-buffer entry points (the drop-in boundary) --------------------------------------- */

/* PixFlow<P>::computeOpticalFlow, CPU/PixFlow.hpp:72-135.  flow = I0 -> I1, cols x rows. */
int pf_flow(pf_ctx* ctx, const uint8_t* i0_bgra, const uint8_t* i1_bgra, int cols, int rows, size_t step_bytes,
            int max_percentage, int hint, float* flow_xy, size_t flow_step_bytes);

/* NovelViewGeneratorAsymmetricFlow::prepare, CPU/OpticalFlow.cpp:102-145: wrap-pad by cols/20,
 * L->R solve (hint LEFT) and R->L solve (hint RIGHT) concurrently, crop.  Either output may be NULL. */
int pf_flow_bidir(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes,
                  int max_percentage, float* flow_l2r, float* flow_r2l, size_t flow_step_bytes);

/* NovelViewUtil::combineNovelViews, CPU/OpticalFlow.cpp:30-92 (+ generateNovelViewPoint :9-28). */
int pf_blend(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, size_t step_bytes, const float* flow_l2r,
             const float* flow_r2l, size_t flow_step_bytes, const float* blend, size_t blend_step_bytes, int cols, int rows,
             uint8_t* out_bgra, size_t out_step_bytes);

/* prepare() + setBlend() + generateNovelView() in one call with the flows kept in HBM
 * (CPU/main.cpp:82-90).  flow outputs may be NULL. */
int pf_novel_view(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes,
                  int max_percentage, const float* blend, size_t blend_step_bytes, uint8_t* out_bgra, size_t out_step_bytes,
                  float* flow_l2r, float* flow_r2l, size_t flow_step_bytes);

/* Size limit of the three stitch entry points that build the blend ramp (pf_stitch_prepare, pf_stitch_generate_blend, pf_stitch_step):
 * the tile smoothing of GenerateBlend (CPU/StitchTool.cpp:130-143) keeps one tile's window, (step + k - 1)^2 floats + (step + k - 1) x step
 * doubles with step = min(cols, rows) / 200 and k = rows / 130, in the 160 KB of LDS of a CU: canvases up to ~15,000 rows (a 30000x15000
 * equirectangular panorama) -- beyond that they return PF_ERR_ARG.  The solver and blend entry points have no such limit. */
/* Stitchtools::prepare, CPU/StitchTool.cpp:7-36 (MatchImages :38-50, GenerateBlend :98-146,
 * countblend :148-191).  merged_dis may be NULL.  All planes cols x rows, packed rows of `step`. */
int pf_stitch_prepare(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes,
                      uint8_t* map_out, size_t map_step_bytes, uint8_t* overlapped_l, uint8_t* overlapped_r,
                      float* blend_out, size_t blend_step_bytes, float* merged_dis /* packed, nullable */);

/* The two halves of prepare() as the reference exposes them: Stitchtools::MatchImages (CPU/StitchTool.cpp:38-50) with the overlap
 * masking of :17-33, and Stitchtools::GenerateBlend (:98-146) computed from a GIVEN map -- the reference reads its public `Map`
 * member there, so a caller that edits Map between the two calls gets the ramp of the edited map.  Outputs may be NULL. */
int pf_stitch_match(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes, uint8_t* map_out,
                    size_t map_step_bytes, uint8_t* overlapped_l, uint8_t* overlapped_r);
int pf_stitch_generate_blend(pf_ctx* ctx, const uint8_t* map, size_t map_step_bytes, int cols, int rows, float* blend_out,
                             size_t blend_step_bytes, float* merged_dis /* packed, nullable */);

/* GenerateBlend's per-pixel loop alone, CPU/StitchTool.cpp:113-125: 0 / 1 / 0.5 by map code and, in the overlap,
 * Stitchtools::countblend(x, y) (:148-191) = minLdis / (minRdis + minLdis) -- the ramp BEFORE the smoothing of
 * :130-143 -- plus MergedDis (:185-188, nullable, packed). */
int pf_stitch_raw_blend(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes,
                        float* raw_blend_out, size_t blend_step_bytes, float* merged_dis);

/* Stitchtools::Gather, CPU/StitchTool.cpp:52-96. */
int pf_stitch_gather(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, const uint8_t* merged_bgra, size_t step_bytes,
                     const uint8_t* map, size_t map_step_bytes, int cols, int rows, uint8_t* out_bgra, size_t out_step_bytes);

/* One whole iteration of the stitch loop of CPU/main.cpp:70-95 on the device: Stitchtools::prepare ->
 * NovelViewGeneratorAsymmetricFlow::prepare + generateNovelView -> Stitchtools::Gather.  Only the inputs go up
 * and only the composite comes down.  r_bgra == NULL chains on the previous call's result, which stays in HBM
 * (main.cpp:64-65: R_i = FinalResult_{i-1}).  out_bgra may be NULL (intermediate steps). */
int pf_stitch_step(pf_ctx* ctx, const uint8_t* l_bgra, const uint8_t* r_bgra, int cols, int rows, size_t step_bytes,
                   int max_percentage, uint8_t* out_bgra, size_t out_step_bytes);

/* Hint, given BEFORE step i: the l_bgra of step i+1.  Step i then issues that host->device upload after its own kernels are
 * enqueued, so it overlaps the compute (CPU/main.cpp:66-69 reads image i+1 only after step i).  One-shot: step i consumes the
 * hint, step i+1 consumes (or, if it is called with anything else, drops) the uploaded copy.  CONTRACT: the buffer must stay valid
 * and UNCHANGED until step i+1 has returned.  The library guards against the common slip (a buffer reused for another image at the
 * same address) with a content signature over 16 evenly spaced rows taken at upload time -- a sample, not a hash of the whole image:
 * a partial overwrite that misses those rows is not detected and the stale device copy would be used.  NULL cancels.  With the
 * contract kept it is purely an optimisation: results are identical. */
int pf_stitch_prefetch(pf_ctx* ctx, const uint8_t* next_l_bgra, int cols, int rows, size_t step_bytes);

/* ---- device-resident entry points (packed buffers already in this context's HBM) -----------
 * Same semantics as above; used by bench.py (inputs resident when the clock starts) and by the
 * multi-GPU driver.  Pointers are device pointers on the context's device.  The calls are synchronous on
 * return, but they run on the context's own non-blocking streams and do NOT wait for work the caller has
 * in flight on other streams: inputs produced there (e.g. by a framework's kernels) must be complete --
 * synchronise that stream or the device -- before the call. */
void* pf_dev_alloc(pf_ctx* ctx, size_t bytes);
void pf_dev_free(pf_ctx* ctx, void* dptr);
/* Page-locked host memory for images handed to / received from the host-buffer entry points (optional: any host memory works,
 * pinned memory is copied at the link's DMA rate instead of through the runtime's bounce buffers). */
void* pf_host_alloc(pf_ctx* ctx, size_t bytes);
void pf_host_free(pf_ctx* ctx, void* hptr);
int pf_upload(pf_ctx* ctx, void* dst_dev, const void* src_host, size_t bytes);
int pf_download(pf_ctx* ctx, void* dst_host, const void* src_dev, size_t bytes);
int pf_sync(pf_ctx* ctx);
/* 64-bit content checksum of a device buffer (8-byte aligned), computed on the device: compares or verifies results that stay in
 * HBM (e.g. a strip at its producer and after the gather to rank 0) without moving them through the host. */
int pf_checksum_dev(pf_ctx* ctx, const void* d_ptr, size_t bytes, uint64_t* checksum_out);

int pf_flow_bidir_dev(pf_ctx* ctx, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_percentage,
                      float* d_flow_l2r, float* d_flow_r2l);
int pf_blend_dev(pf_ctx* ctx, const uint8_t* d_l, const uint8_t* d_r, const float* d_flow_l2r, const float* d_flow_r2l,
                 const float* d_blend, int cols, int rows, uint8_t* d_out);
/* flow + blend; the per-pair unit the benchmark times.  d_flow_* may be NULL (flows stay internal). */
int pf_novel_view_dev(pf_ctx* ctx, const uint8_t* d_l, const uint8_t* d_r, int cols, int rows, int max_percentage,
                      const float* d_blend, uint8_t* d_out, float* d_flow_l2r, float* d_flow_r2l);

/* Throughput mode: n_pairs independent pairs of one size, `in_flight` (1..32) of them on this GPU at a time (the exact sweeps of one
 * pair are a dependency chain that occupies ~1/4 of the CUs): batches of pairs that share every kernel launch, on one or more
 * lanes of streams (pf_config::batch_pairs).  Arrays of n_pairs device pointers; d_flow_* may be NULL (or hold NULL entries).
 * Same results as n_pairs calls of pf_novel_view_dev.  Needs GPU_MAX_HW_QUEUES >= 3 * lanes + 2 in the environment before the
 * first HIP call, or the lanes' streams share hardware queues (then pf_last_warning says so). */
int pf_novel_view_batch_dev(pf_ctx* ctx, int n_pairs, const uint8_t* const* d_l, const uint8_t* const* d_r, int cols, int rows,
                            int max_percentage, const float* const* d_blend, uint8_t* const* d_out, float* const* d_flow_l2r,
                            float* const* d_flow_r2l, int in_flight);

/* ---- multi-GPU: the path's only exchange ---------------------------------
 * Overlap pairs are independent units (no state shared between the reference's Stitchtools / NovelViewGenerator objects,
 * CPU/main.cpp:70,82): one rank (process or host thread) + one pf_ctx per GPU, pair i on rank i % world, NO collective
 * on the data path.  What remains is the final gather of the results into rank 0's HBM: grouped ncclSend/ncclRecv over
 * RCCL/xGMI on a stream of its own, so that the gather of pair k overlaps the compute of pair k+1.  RCCL is bound at run
 * time (dlopen): single-GPU users carry no dependency on it.  The reference has no counterpart (it is single-device). */
typedef struct pf_dist pf_dist;
int pf_dist_unique_id(void* id128);                                   /* rank 0: 128-byte ncclUniqueId to hand to every rank */
pf_dist* pf_dist_init(int device, const void* id128, int rank, int world);   /* NULL on failure; pf_dist_last_error(NULL) */
void pf_dist_destroy(pf_dist* d);
const char* pf_dist_last_error(const pf_dist* d);
/* every rank sends `bytes` from d_send; rank 0 receives rank r's block at d_recv_all + r*bytes (d_recv_all ignored elsewhere).
 * Asynchronous; at most one gather in flight (a second call first waits for the previous one). */
int pf_dist_gather_async(pf_dist* d, const void* d_send, void* d_recv_all, size_t bytes);
int pf_dist_wait(pf_dist* d);                                         /* host-blocking: the last gather has completed */
int pf_dist_max(pf_dist* d, double* value_inout);                     /* max over ranks (the job's time is the slowest rank's) */
int pf_dist_barrier(pf_dist* d);

/* ---- stage-level entry points (host buffers, packed) ---------------------------------------
 * One per reference step, so that tests can check every HIP kernel family against the oracle in
 * isolation.  They run the very kernels the entry points above chain together. */
int pf_stage_preprocess(pf_ctx* ctx, const uint8_t* bgra, int cols, int rows, int pad,
                        float* gray_half, float* alpha_half);                         /* PixFlow.hpp:78-103 (+ OpticalFlow.cpp:113-126 when pad>0) */
int pf_stage_pyr_down(pf_ctx* ctx, const float* src, int sw, int sh, float* dst, int dw, int dh);   /* PixFlow.hpp:146-148 */
int pf_stage_gradients(pf_ctx* ctx, const float* img, int w, int h, float* gxy /* (Ix,Iy) interleaved */); /* PixFlow.hpp:281-294 */
int pf_stage_gauss(pf_ctx* ctx, const float* src, int w, int h, int cn, int ksize, double sigma, float* dst); /* GaussianBlur call sites */
int pf_stage_median5(pf_ctx* ctx, const float* flow, int w, int h, float* out);        /* PixFlow.hpp:325,338 */
int pf_stage_sweep(pf_ctx* ctx, const float* g0xy, const float* g1xy, const float* blurred, const float* alpha0,
                   const float* alpha1, float* flow_inout, int w, int h, int forward);  /* PixFlow.hpp:315-324 / :328-337 */
int pf_stage_diffusion(pf_ctx* ctx, const float* alpha0, const float* alpha1, float* flow_inout, int w, int h); /* PixFlow.hpp:388-405 */
int pf_stage_upsample_cubic(pf_ctx* ctx, const float* flow, int sw, int sh, float* out, int dw, int dh, float scale); /* PixFlow.hpp:122-125 */
int pf_stage_final(pf_ctx* ctx, const float* flow, int sw, int sh, int pad_cols, int rows, int pad, float scale,
                   float* out /* (pad_cols-2*pad) x rows */);                          /* PixFlow.hpp:128-134 + OpticalFlow.cpp:143-144 */
int pf_stage_adjust_initial_flow(pf_ctx* ctx, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h,
                                 int hint, int max_percentage, float* flow_out);       /* PixFlow.hpp:226-270 */
int pf_stage_level(pf_ctx* ctx, const float* i0, const float* i1, const float* a0, const float* a1, int w, int h,
                   const float* flow_in /* nullable */, int hint, int max_percentage, float* flow_out); /* PixFlow.hpp:272-340 */
int pf_stage_blend_smooth(pf_ctx* ctx, float* blend_inout, const float* merged_dis, int cols, int rows); /* StitchTool.cpp:130-143 */

/* ---- per-kernel-family timing (HIP events on the streams the kernels run on) ---------------- */
int pf_profile_enable(pf_ctx* ctx, int on);   /* 0 off, 1 all kernel families, 2 only the sweep kernels */
int pf_profile_reset(pf_ctx* ctx);
int pf_profile_count(pf_ctx* ctx);                                       /* number of kernel families seen */
int pf_profile_get(pf_ctx* ctx, int idx, char* name, int name_cap, double* total_ms, int* launches);
/* algorithmic HBM bytes of one pf_novel_view on cols x rows (SURVEY.md section 8(d) model) */
double pf_algorithmic_bytes(int cols, int rows);
long long pf_level_pixels(int cols, int rows, int* n_levels, long long* sweep_steps);
/* dependent wavefront steps of ONE direction of the last solve: sum over levels and both sweeps of (w + h - 1) of the
 * window of gated pixels -- the length of the exact Gauss-Seidel dependency chain (SURVEY.md 8(d) "honest bound") */
long long pf_last_swept_steps(pf_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* PANOFLOW_H_ */
