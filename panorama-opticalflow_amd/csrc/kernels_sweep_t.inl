// K6, THROUGHPUT form of the exact Gauss-Seidel sweep (CPU/PixFlow.hpp:315-337) -- included by kernels_sweep2.hip.
//
// The latency form (compute_band above) spends lanes to buy time: 8 lanes per pixel evaluate the six energies a step may need AT
// ONCE, speculatively, so that a step is one gather round.  That is right for ONE pair, whose sweeps are a dependency chain and leave
// most of the chip idle.  A BATCH of pairs asks for several times more sweep workgroups than the chip has CUs, and there the
// measured bound is the SIMDs' VALU issue (profiles/r04_wide_sweep.txt: one 8-lane compute wave keeps its SIMD's VALU pipe ~60 % busy;
// a second one on the same SIMD runs at 0.43 instead of 0.29 us per step) -- what counts is VALU instructions per PIXEL:
//   latency form:    ~118 instructions per step of a wave = 8 pixels  -> 14.7 per pixel (+ a prepass of three energies per pixel)
//   throughput form: ~210 instructions per step of a wave = 32 pixels ->  6.6 per pixel (+ a prepass of one energy per pixel)
// This form does what the reference does, in the reference's order, two lanes per pixel and without speculation:
//   round 1: lane a evaluates E(along proposal), lane b E(across proposal)          (proposeFlowUpdate x 2, PixFlow.hpp:342-362)
//   select : current, then L, then T, strict '<' -> the winner W and E(W)           (both lanes, same instructions)
//   round 2: lane a evaluates E(W + eps e_x), lane b E(W + eps e_y)                 (errorGradient, PixFlow.hpp:364-386)
//   update : W - 0.5 * ((E(W+dx), E(W+dy)) - E(W)) / eps                            (PixFlow.hpp:322-323)
// Two dependent gather rounds per step (~0.45 us instead of 0.29), but four times the rows per wave.  Same arithmetic (d_error_fast with
// its range guard and whole-wave IEEE redo, exact_forms.hpp), same operands, same order => the same bits as the latency form and the oracle.
//
// Geometry: a band = 32 rows, lane = 32 * role + row; the partner's value comes by v_permlane32_swap (one instruction hands both lanes
// both values), the row above's result by a wave_shr:1 DPP move.  A workgroup = 3 compute waves (96 rows; the CU's 160 KB of LDS hold
// three bands' rings and windows) + 3 loaders + publisher + poller + drainer = 576 threads; the helpers share the fourth SIMD and the
// compute waves' SIMDs.  Rings, counters, granules, tickets and deadlines work as in the latency form.  The gather window is SKEWED:
// ring slot = (u + window row) & 63, because the 32 pixels of a step lie on an anti-diagonal 32 columns wide -- in (u + row) they all sit
// within 31 slots of each other (d_error_fast<.., SKEW>).  Dense sweeps only (no variant that skips ungated anti-diagonals); sparse
// and small launches keep the latency form.
// Round 5: ONE record path (the prepass kernel's record stream fetched by LDS-DMA; the loader-staged ring and the fused prepass of rounds 4 / 5
// were measured, rejected and removed: profiles/r04_throughput_form.txt, r05_fused_prepass.txt), and the gather window FOLLOWS THE FLOW like
// the latency form's (kernels_sweep2.hip, loader of k_sweep2): a torus in (skew slot D = u + v, row v), centred per chunk on the pixels +
// the rounded blurred flow of the chunk's centre pixel.
namespace {
constexpr int tRows = 32;                       // rows per compute wave
// The records reach a compute wave by LDS-DMA: it requests its own records six steps ahead (global_load_lds_dwordx4: one instruction copies a
// step's 32 records = 1 KB straight into an 8-step LDS ring, no registers, no loader; its completion is counted by hand with s_waitcnt vmcnt --
// the compiler does not track it, and with a register destination its conservative vmcnt(0) at every loop / branch join drained the queue
// twice per chunk: measured, 0.54 instead of 0.47 us per step) -- ~40 KB per band: FOUR bands (128 rows) per workgroup, one compute wave on
// every SIMD, and a batch's level-0 launches (8 pairs x 2 directions x 16 workgroups = 256) fit the chip in ONE round.
constexpr int tWaves = 4;
constexpr int tThreads = 64 * (2 * tWaves + 3);
constexpr int tPre = 6;                         // steps a record is requested ahead of its use
constexpr int tRSG = 8;                         // record ring (steps): tPre in flight + the one being read + one being overwritten
constexpr int tOS = 16;                         // result ring (steps)
constexpr int tRect = tRows + 2 * kRadT;        // rows of a chunk's window rectangle: the band's 32 rows -+ 6 (44)
constexpr int tRV = 48;                         // ring rows of the torus window: the rectangle + the rows it may have moved by while a chunk is in use (3) + 1
constexpr int kWCPT = kWC + 2;                  // window row stride: ring columns 0 and 1 again behind column 63 (the skewed footprint reaches slot + 2)
constexpr int tAhead = 3;                       // chunks the window loader may run ahead of its band (see the loader)

struct SmemTF {
  float4 rec[tWaves][tRSG][tRows][2];   // the 32-byte records (I0x, I0y, blurred.x, blurred.y | E(C), C.x, C.y, Ea)
  float2 out[tWaves][tOS][tRows];
  float2 win[tWaves][tRV + 1][kWCPT];   // texel (u, v) at row v mod tRV (row 0 again behind row tRV - 1), column (u + v) & 63
  float4 woff[tWaves][4];               // per chunk (ring of 4): the window's offset (ox, oy) in image axes and the ring-row base of its rows (an int)
  unsigned long long bnd[kBS];
  int recHead[tWaves], outHead[tWaves], outTail[tWaves];   // recHead: steps whose window is in LDS
  int pubTail, bndHead, abort, wg;
  long long deadline;
};
// the LDS-DMA destination travels in M0, whose LDS-address field is 16 bits wide: the record rings must lie in the first 64 KB of the
// workgroup's LDS (they are the struct's first member: 32 KB)
static_assert(offsetof(SmemTF, rec) == 0 && sizeof(SmemTF::rec) <= 65536, "LDS-DMA rings must sit below 64 KB");
static_assert(sizeof(SmemTF) <= 160 * 1024, "one workgroup per CU: 160 KB of LDS");

// V_PERMLANE32_SWAP: lanes 32-63 of `a` trade places with lanes 0-31 of `b`.  With a == b == v: lo = role a's v in every lane, hi = role b's.
__device__ __forceinline__ void swap_roles(float v, float& lo, float& hi) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  lo = __uint_as_float(r[0]); hi = __uint_as_float(r[1]);
}

// One step for one wave.  FAST: the exact cheap forms + range guard (emin / vmax out); !FAST: the IEEE sequence (cold redo path).
// (ob = ring-row base of the chunk's window rows, wof = the window's offset in image axes, live = the pixel is updated: d_error_fast<FOLLOW = 2>)
template <bool FAST, bool TR, bool FWD>
__device__ __forceinline__ float2 t_step(const float2* __restrict__ g1, __attribute__((address_space(3))) const float2* win, int ob, int W, int H, float wm2, float hm2,
                                         float fW, float rW, float rEps, const SolverCoef& cf, f2p posv, float4 ra, float eC, float2 C, float eCL, bool okL, bool okT, float2 along,
                                         float2 across, int role, int& emin, float& vmax, f2p wof, bool live) {
  auto energy = [&](float2 f, int& em, float& vm) -> float {
    if (FAST) return d_error_fast<TR, FWD, tRV, kWCPT, true, 2>(g1, win, ob, W, H, wm2, hm2, fW, rW, cf, posv, ra.x, ra.y, ra.z, ra.w, f2p{f.x, f.y}, em, vm, wof, live);
    em = 0; vm = 0.f;
    return d_error2(g1, W, wm2, hm2, fW, cf, int(posv.x), int(posv.y), ra.x, ra.y, ra.z, ra.w, f.x, f.y);
  };
  // ---- round 1: the two proposals (reference order: previous column = L, then previous row = T; transposed sweeps step along y) ----
  const float2 p1 = role ? across : along;
  int em1; float vm1;
  const float e1 = energy(p1, em1, vm1);
  float eA, eX;
  swap_roles(e1, eA, eX);
  const float eL = TR ? eX : eA, eT = TR ? eA : eX;
  const float2 fL = TR ? across : along, fT = TR ? along : across;
  const bool pickL = okL && (eL < eCL);   // eCL = E(C), or below every energy where L does not exist (see the records)
  const float cur = pickL ? eL : eC;
  const bool pickT = okT && (eT < cur);
  float2 Wf; Wf.x = pickL ? fL.x : C.x; Wf.y = pickL ? fL.y : C.y;
  Wf.x = pickT ? fT.x : Wf.x; Wf.y = pickT ? fT.y : Wf.y;
  const float eW = pickT ? eT : cur;
  // ---- round 2: the winner's finite differences, one per lane ----
  const float2 p2 = make_float2(Wf.x + (role ? 0.0f : kGradEpsilon), Wf.y + (role ? kGradEpsilon : 0.0f));
  int em2; float vm2;
  const float e2 = energy(p2, em2, vm2);
  float g1e, g2e;
  swap_roles(e2, g1e, g2e);
  float2 res;
  if (FAST) {
    const f2p dg = f2p{g1e, g2e} - f2p{eW, eW};
    const float ax = fabsf(dg.x), ay = fabsf(dg.y);
    const f2p gq = div_core2(dg, kGradEpsilon, rEps);
    emin = min(min(em1, em2), min(__builtin_amdgcn_frexp_expf(ax), __builtin_amdgcn_frexp_expf(ay)));
    vmax = __builtin_fmaxf(__builtin_fmaxf(vm1, vm2), __builtin_fmaxf(ax, ay));
    const f2p r = __builtin_elementwise_fma(gq, f2p{-cf.step, -cf.step}, f2p{Wf.x, Wf.y});   // exact: see select_step
    res = make_float2(r.x, r.y);
  } else {
    const float gx = (g1e - eW) / kGradEpsilon, gy = (g2e - eW) / kGradEpsilon;
    res = make_float2(Wf.x - cf.step * gx, Wf.y - cf.step * gy);
    emin = 0; vmax = 0.f;
  }
  return res;
}

// One compute wave of the throughput form: a band of 32 rows.  TOP as in compute_band.
template <int TOP, bool TR, bool FWD>
__device__ __forceinline__ bool compute_band_t(SmemTF& sm, const float2* __restrict__ g1, const float4* __restrict__ recg, int W, int H, int nsteps, int w, int band,
                                               int nact, bool publishes, float rW, float rEps, const SolverCoef cf, int uLo, int LSv) {
  constexpr int transposed = TR ? 1 : 0, forward = FWD ? 1 : 0;
  constexpr bool RG = true;
  const int lane = threadIdx.x & 63;
  const int r = lane & 31, role = lane >> 5;
  const int ib = band * tRows + r;
  const int LS = transposed ? H : W;
  const bool hasCross = (TOP != 0) || ib > 0;
  typedef __attribute__((address_space(3))) const float2 lds_cf2;
  lds_cf2* win = (lds_cf2*)&sm.win[w][0][0];
  asm volatile("" : "+s"(win));
  int ob = 0;                       // ring-row base of the current chunk's window rows (from the loader, per chunk)
  f2p wof = f2p{0.f, 0.f};          // the current chunk's window offset (image axes)
  float4 woNext = make_float4(0.f, 0.f, 0.f, 0.f);
  const float wm2 = float(W) - 2.0f, hm2 = float(H) - 2.0f, fW = float(W), fLast = float(LS - 1);
  const bool lastPub = publishes && (w == tWaves - 1);
  const bool hasNext = (w + 1 < nact);
  const int wp = (w > 0) ? w - 1 : 0;
  const int* topHead = (TOP == 1) ? &sm.outHead[wp] : &sm.bndHead;
  constexpr int kBias = (TOP == 1) ? tRows - 1 : 0;   // TOP==1: column c is the producer's step c + 31
  auto top_slot = [&](int c) -> const unsigned long long* {
    return (TOP == 1) ? reinterpret_cast<const unsigned long long*>(&sm.out[wp][(c + tRows - 1) % tOS][tRows - 1]) : &sm.bnd[c & (kBS - 1)];
  };
  const bool isL32 = lane == 32;
  float2 prev = make_float2(0.f, 0.f);
  bool dead = false, waitTop = true;
  unsigned long long tv = 0;
  int fcRec = 0, fcTail = 0, fcPub = 0, fcNext = 0;
  float4 ra = make_float4(0.f, 0.f, 0.f, 0.f), rb = ra;
  // the pixel's image coordinates: the position across the bands is fixed, the position along the step axis advances by one per
  // step (sweep order; mirrored for the backward sweep); exact small integers in fp32
  const float acrossPos = forward ? float(ib) : float((transposed ? W : H) - 1 - ib);
  float alongU = float(uLo - r);   // sweep-order column of step 0
  // RG: LDS-DMA of one step's records (64 lanes x 16 bytes = the step's 32 records) into ring slot (step % tRSG).  M0 carries the LDS
  // destination and is compiler-reserved: saved and restored inside the statement (cdna_hip_programming.md, LDS-DMA recipe).
  const char* recBytes = reinterpret_cast<const char*>(recg) + lane * 16;
  const unsigned recLds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)&sm.rec[w][0][0][0]);
  auto dma_step = [&](int step) {
    const char* gsrc = recBytes + size_t(step) * (tRows * 32);
    const unsigned dst = recLds + unsigned(step % tRSG) * (tRows * 32);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
  };
  if (RG) {
#pragma unroll
    for (int d = 0; d < tPre; ++d) dma_step(d);
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(tPre - 1) : "memory");   // step 0's records have landed
    const float4* rp0 = &sm.rec[w][0][r][0];
    ra = rp0[0]; rb = rp0[1];
    asm volatile("" : "+v"(ra.x), "+v"(ra.y), "+v"(ra.z), "+v"(ra.w), "+v"(rb.x), "+v"(rb.y), "+v"(rb.z), "+v"(rb.w));
  }
  for (int s0 = 0; s0 < nsteps; s0 += kChunk) {
    if (dead) return false;
    {
      int spins = 0;
      for (;;) {
        const int rec = __builtin_amdgcn_readfirstlane(fcRec);
        int lim = __builtin_amdgcn_readfirstlane(fcTail) + tOS;
        if (lastPub) { const int c1 = __builtin_amdgcn_readfirstlane(fcPub) + tOS; lim = lim < c1 ? lim : c1; }
        if (hasNext) { const int c2 = __builtin_amdgcn_readfirstlane(fcNext) + tOS + tRows - 1; lim = lim < c2 ? lim : c2; }
        if (__builtin_expect(rec >= s0 + kChunk && lim >= s0 + kChunk, 1)) break;
        if (spins) __builtin_amdgcn_s_sleep(1);
        fcRec = ld_cnt(&sm.recHead[w]); fcTail = ld_cnt(&sm.outTail[w]);
        if (lastPub) fcPub = ld_cnt(&sm.pubTail);
        if (hasNext) fcNext = ld_cnt(&sm.outHead[w + 1]);
        if (spin_expired(spins, sm) || (((spins & 255) == 0) && ld_cnt(&sm.abort))) return false;
      }
      {   // the chunk's window: offset and ring-row base, published by the loader before the chunk's recHead (wave-uniform: scalar registers);
          // read ahead in the middle of the previous chunk, good if the chunk was already published then (no waiting just now)
        float4 wo = woNext;
        if (__builtin_expect(spins != 0 || s0 == 0, 0)) wo = sm.woff[w][(s0 / kChunk) & 3];
        wof = f2p{__int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wo.x))), __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(wo.y)))};
        ob = __builtin_amdgcn_readfirstlane(__float_as_int(wo.z));
      }
    }
    // counters for the NEXT chunk's check: read in the middle of this chunk (a 16-step ring cannot satisfy a check that is a whole chunk old)
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f2w __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const f4v lds_f4;
    typedef __attribute__((address_space(3))) f2w lds_wf2;
    typedef __attribute__((address_space(3))) const unsigned long long lds_u64;
    lds_f4* recChunk = (lds_f4*)&sm.rec[w][0][r][0];                    // the ring IS one chunk long
    lds_f4* recNext = recChunk;
    lds_wf2* outChunk = (lds_wf2*)&sm.out[w][s0 % tOS][r];   // (role a stores)
    lds_u64* topChunk = (lds_u64*)((TOP == 1) ? top_slot(s0 + 1) : &sm.bnd[s0 & (kBS - 1)] + 1);
    lds_u64* topNext = (lds_u64*)top_slot(s0 + kChunk);
    typedef __attribute__((address_space(3))) int lds_int;
    lds_int* cntp = (lds_int*)&sm.outHead[w];
#pragma unroll
    for (int j = 0; j < kChunk; ++j) {
      const int s = s0 + j;
      if (RG) dma_step(s + tPre);   // into the slot of step s - 2 (the stream is padded past its end)
      if (j == 5) {
        fcRec = ld_cnt(&sm.recHead[w]); fcTail = ld_cnt(&sm.outTail[w]);
        woNext = sm.woff[w][((s0 / kChunk) + 1) & 3];   // (after the counter: published before it)
        if (lastPub) fcPub = ld_cnt(&sm.pubTail);
        if (hasNext) fcNext = ld_cnt(&sm.outHead[w + 1]);
      }
      if (TOP != 0) {
        if (__builtin_expect(waitTop, 0)) {
          if (!dead && s < LSv) {
            const int need = (s + 1 + (TOP == 2 ? kMarginAcrossT : 0) < LSv) ? s + 1 + (TOP == 2 ? kMarginAcrossT : 0) : LSv;
            int spins = 0;
            for (;;) {
              const int avail = __builtin_amdgcn_readfirstlane(ld_cnt(topHead)) - kBias;
              if (avail >= need) break;
              if (spin_expired(spins, sm) || (((spins & 255) == 0) && ld_cnt(&sm.abort))) { dead = true; break; }
            }
            tv = __hip_atomic_load(top_slot(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            unsigned tlo = unsigned(tv), thi = unsigned(tv >> 32);
            asm volatile("" : "+v"(tlo), "+v"(thi));
            tv = (unsigned long long)tlo | ((unsigned long long)thi << 32);
          }
        }
      }
      // the across proposal = the previous result of the row above: one lane down (same role); row 0 takes the ring value (lane 0 has
      // no source lane and keeps `old`; lane 32 would read lane 31 -- role a's row 31 -- and is patched).  First band: no row above row 0,
      // its across proposal is masked out of the selection (hasCross) and only has to be finite.
      const float2 tvf = (TOP != 0) ? unpack2(tv) : prev;
      float2 across;
      across.x = dpp<0x138>(tvf.x, prev.x);   // wave_shr:1
      across.y = dpp<0x138>(tvf.y, prev.y);
      if (TOP != 0) { across.x = isL32 ? tvf.x : across.x; across.y = isL32 ? tvf.y : across.y; }
      const float alongPos = forward ? alongU : fLast - alongU;
      const f2p posv = transposed ? f2p{acrossPos, alongPos} : f2p{alongPos, acrossPos};
      const float fpos = alongPos;
      alongU += 1.0f;
      const float eC = rb.x, eCa = rb.w;
      const float2 C = make_float2(rb.y, rb.z);
      const bool gated = eC >= 0.0f;
      const bool hasAlong = transposed ? (forward ? (fpos > 0.0f) : (fpos < fLast)) : true;
      const float eCL = transposed ? eC : eCa;
      const bool okL = transposed ? hasCross : hasAlong, okT = transposed ? hasAlong : hasCross;
      int emin; float vmax;
      float2 fin = t_step<true, TR, FWD>(g1, win, ob, W, H, wm2, hm2, fW, rW, rEps, cf, posv, ra, eC, C, eCL, okL, okT, prev, across, role, emin, vmax, wof, gated);
      // next step's inputs (LDS), behind the second gather round
      float4 na, nb; int hN = 0; unsigned long long tvN = tv;
      {
        // RG: steps s+1 .. s+tPre are requested; all but the newest tPre - 1 have landed after this wait, i.e. step s+1 has
        if (RG) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(tPre - 1) : "memory");
        lds_f4* rpn = (j + 1 < kChunk) ? recChunk + (j + 1) * (tRows * 2) : recNext;
        const f4v q0 = rpn[0]; const f4v q1 = rpn[1];
        na = make_float4(q0.x, q0.y, q0.z, q0.w); nb = make_float4(q1.x, q1.y, q1.z, q1.w);
      }
      if (TOP != 0) {
        lds_u64* tpn = (j + 1 < kChunk) ? topChunk + j * ((TOP == 1) ? tRows : 1) : topNext;
        hN = ld_cnt(topHead); tvN = __hip_atomic_load(tpn, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      asm volatile("" : "+v"(fin.x), "+v"(fin.y));
      if (__builtin_expect(__any((emin < cf.guard_min || !(vmax <= 0x1p100f)) && gated), 0)) {
        // an operand left the range where the fast forms are exact: the whole wave redoes the step with IEEE sqrt and division
        fin = t_step<false, TR, FWD>(g1, win, ob, W, H, wm2, hm2, fW, rW, rEps, cf, posv, ra, eC, C, eCL, okL, okT, prev, across, role, emin, vmax, wof, gated);
      }
      fin.x = gated ? fin.x : C.x; fin.y = gated ? fin.y : C.y;   // a pixel that is not updated keeps its flow (PixFlow.hpp:317); slots without a pixel carry C = 0
      if (TOP != 0) {
        waitTop = __any(s + 1 + kBias >= hN);
        asm volatile("" : "+v"(tvN));
        tv = tvN;
      }
      prev = fin;
      if (role == 0) outChunk[j * tRows] = f2w{fin.x, fin.y};
      asm volatile("" ::: "memory");
      __hip_atomic_store(cntp, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      ra = na; rb = nb;
    }
  }
#ifdef PF_SWEEP_STATS
  if (lane == 0) atomicAdd(&g_sweep_stats[2], (unsigned long long)nsteps);   // ([3] is counted where it happens: d_error_fast)
#endif
  return !dead;
}
}  // namespace

// Waves: [0, 4) compute (one per SIMD), [4, 8) the bands' window loaders, then publisher, poller, drainer.
template <bool TR, bool FWD>
__global__ __launch_bounds__(tThreads) void k_sweep_t(const float4* __restrict__ rec, const float2* __restrict__ g1, float2* __restrict__ flow,
                                                      unsigned long long* __restrict__ boundary, int* __restrict__ ctrl, int W, int H, int nstepsPad, int nbands,
                                                      float rW, float rEps, int uLo, int LSv, int bandLo, long long budgetTicks, size_t bstride,
                                                      const float2* __restrict__ blurred, SolverCoef cf) {
  {
    const size_t bo = size_t(blockIdx.z) * bstride;
    PF_BOFF(rec, bo); PF_BOFF(g1, bo); PF_BOFF(flow, bo); PF_BOFF(boundary, bo); PF_BOFF(ctrl, bo); PF_BOFF(blurred, bo);
  }
  constexpr int transposed = TR ? 1 : 0, forward = FWD ? 1 : 0;
  __shared__ SmemTF sm;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lane = tid & 63;
  if (tid == 0) {
    sm.wg = atomicAdd(&ctrl[0], 1);
    sm.bndHead = 0; sm.abort = 0; sm.pubTail = 0;
    sm.deadline = (long long)wall_clock64() + budgetTicks;
  }
  if (tid < tWaves) { sm.recHead[tid] = 0; sm.outHead[tid] = 0; sm.outTail[tid] = 0; }
#ifdef PF_EXPERIMENTS
  poison_window(&sm.win[0][0][0], int(sizeof(sm.win) / sizeof(float2)));
#endif
  __syncthreads();
  const int wg = sm.wg;
  const int LS = transposed ? H : W, LB = transposed ? W : H;
  const int nsteps = nstepsPad;
  const int band0 = wg * tWaves;
  const int nact = (nbands - band0) < tWaves ? (nbands - band0) : tWaves;
  const bool publishes = band0 + tWaves < nbands;
  const bool staticTop = bandLo > 0;
  auto give_up = [&]() { sm.abort = 1; __hip_atomic_store(&ctrl[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };

  if (wave < tWaves) {
    // ======================= compute wave: band of 32 rows =======================
    if (wave >= nact) return;
    __builtin_amdgcn_s_setprio(3);
    const int top = (wave > 0) ? 1 : ((wg > 0 || staticTop) ? 2 : 0);
    const int band = bandLo + band0 + wave;
    bool ok;
    const float4* recg = rec + size_t(band0 + wave) * nstepsPad * (tRows * 2);
    if (top == 1) ok = compute_band_t<1, TR, FWD>(sm, g1, recg, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    else if (top == 2) ok = compute_band_t<2, TR, FWD>(sm, g1, recg, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    else ok = compute_band_t<0, TR, FWD>(sm, g1, recg, W, H, nsteps, wave, band, nact, publishes, rW, rEps, cf, uLo, LSv);
    if (!ok) give_up();
    return;
  }
  __builtin_amdgcn_s_setprio(1);
  if (wave < 2 * tWaves) {
    // ======================= window loader of band w: gather window HBM -> LDS =======================
    // The window FOLLOWS THE FLOW (round 5; the latency form's loader in kernels_sweep2.hip has the full story).  Here in SKEWED coordinates:
    // a pixel of chunk j (steps s = 8j .. 8j + 7; row r sits at sweep column uLo + s - r, row vb + r) needs, with the chunk's offset (ou, ov) in
    // sweep order, the texels (u', v') with |u' - u - ou| <= 6, |v' - v - ov| <= 6 -- in D = u' + v' that is [DB + 8j - 12, DB + 8j + 20),
    // DB = uLo + vb + ou + ov, for ALL 32 rows at once, rows [vb + ov - 6, vb + ov + 38).  The LDS window is a torus: row v' mod 48 (ring row 0
    // again behind row 47), column D & 63 (columns 0 and 1 again behind 63).  Per chunk the loader brings the 6 .. 10 new D-slots of the 44 rows
    // (8 + the change of ou + ov; an offset moves by at most one texel per chunk and axis) and, when ov moved, the one new row over the D-slots
    // already present.  It runs at most tAhead = 3 chunks ahead of its band: the D-slots in use then span <= 32 + 24 + 6 of the ring's 64 (what
    // a new slot overwrites, D - 64, has left every active chunk's range), the rows <= 44 + 3 of 48.  Window centres are kept inside the
    // image (the test is made on the flow relative to the offset, before the sample's clamp to the image: d_error_fast<FOLLOW = 2>).
    const int w = wave - tWaves;
    if (w >= nact) return;
    float2* winw = &sm.win[w][0][0];
    const int vb = (bandLo + band0 + w) * tRows;
    auto ring_row = [&](int v) { int a = v % tRV; return a < 0 ? a + tRV : a; };
    auto win_store = [&](int D, int v, float2 val) {
      const int rr = ring_row(v), cc = D & (kWC - 1);
      float2* q = winw + rr * kWCPT + cc;
      q[0] = val;
      if (cc < 2) q[kWC] = val;
      if (rr == 0) { q[tRV * kWCPT] = val; if (cc < 2) q[tRV * kWCPT + kWC] = val; }
    };
    // the same for a block: its rows' ring rows are one wave-uniform base + the texel's row (one conditional subtraction, no division), and
    // its duplicates sit behind wave-uniform guards (dupc: the block's D-slots contain ring column 0 or 1; dupr: its rows contain ring row 0)
    auto win_store_block = [&](int D, int r0m, int dr, float2 val, bool dupc, bool dupr) {
      int rr = r0m + dr; rr = rr >= tRV ? rr - tRV : rr;
      const int cc = D & (kWC - 1);
      float2* q = winw + rr * kWCPT + cc;
      q[0] = val;
      if (dupc) { if (cc < 2) q[kWC] = val; }
      if (dupr) { if (rr == 0) { q[tRV * kWCPT] = val; if (cc < 2) q[tRV * kWCPT + kWC] = val; } }
    };
    auto tex_ptr = [&](int D, int v) -> const float2* {   // texel (u = D - v, v) in sweep order; nullptr outside the image (never sampled)
      const int u = D - v;
      if (u < 0 || u >= LS || v < 0 || v >= LB) return nullptr;
      const int cxc = TR ? v : u, cyc = TR ? u : v;
      const int x = FWD ? cxc : W - 1 - cxc, y = FWD ? cyc : H - 1 - cyc;
      return g1 + (y * W + x);
    };
    // image index of texel (D, v) = iC + iA * (D - v) + iB * v: one wave-uniform base + a per-lane constant for a block inside the image
    const int iA = TR ? (FWD ? W : -W) : (FWD ? 1 : -1), iB = TR ? (FWD ? 1 : -1) : (FWD ? W : -W), iC = FWD ? 0 : W * H - 1;
    // a block = 8 D-slots x 44 rows = 352 texels: texel t = lane + 64 k (k < 6; k = 5: lanes 0-31 only), D-slot = t & 7 fastest (a wave's loads run
    // along image rows when the bands step along x).  A chunk whose offset did not move needs exactly one block (the usual case): loads and
    // stores are then UNCONDITIONAL (no exec-mask work per texel: this wave shares its SIMD's vector pipe with the band's compute wave); a chunk
    // that needs 6, 7, 9 or 10 D-slots takes the predicated form / one more (narrow) block.
    constexpr int kBlkD = 8, kKT = (kBlkD * tRect + 63) / 64;   // 6
    static_assert(kBlkD * tRect == 64 * (kKT - 1) + 32, "the last texel slot belongs to lanes 0-31");
    int tdD[kKT], tdr[kKT], tio[kKT];
#pragma unroll
    for (int k = 0; k < kKT; ++k) {
      const int t = lane + 64 * k;
      tdD[k] = t & (kBlkD - 1); tdr[k] = t >> 3;   // (k = 5, lanes 32-63: row >= 44, no texel)
      tio[k] = iA * (tdD[k] - tdr[k]) + iB * tdr[k];
    }
    const bool lastHalf = lane < 32;
    // rounded blurred flow at the centre pixel of chunk (obase + lane): one gather serves 64 chunks; integers (scalar-unit work below)
    int ocx = 0, ocy = 0; int obase = -(1 << 20);
    auto refill_offsets = [&](int jb) {
      obase = jb;
      int ia = uLo + 8 * (jb + lane) + 3 - tRows / 2;
      ia = ia < uLo ? uLo : ia; ia = ia > uLo + LSv - 1 ? uLo + LSv - 1 : ia; ia = ia > LS - 1 ? LS - 1 : ia; ia = ia < 0 ? 0 : ia;
      int ibc = vb + tRows / 2; ibc = ibc > LB - 1 ? LB - 1 : ibc;
      const int cxc = TR ? ibc : ia, cyc = TR ? ia : ibc;
      const int x = FWD ? cxc : W - 1 - cxc, y = FWD ? cyc : H - 1 - cyc;
      const float2 bl = blurred[y * W + x];
      const float rx = __builtin_rintf(bl.x), ry = __builtin_rintf(bl.y);
      ocx = (fabsf(rx) < 1.0e6f) ? int(rx) : 0;
      ocy = (fabsf(ry) < 1.0e6f) ? int(ry) : 0;
    };
    // image-axis offset -> sweep order, cut back so that every window centre of chunk j lies inside the image (pixels of the chunk: columns
    // [uLo + 8j - 31, uLo + 8j + 7] clipped to the image, rows [vb, vb + 31]), and back
    auto sweep_offsets = [&](int j, int& ox, int& oy, int& ou, int& ov) {
      ou = TR ? (FWD ? oy : -oy) : (FWD ? ox : -ox);
      ov = TR ? (FWD ? ox : -ox) : (FWD ? oy : -oy);
      int plo = uLo + 8 * j - (tRows - 1), phi = uLo + 8 * j + (kChunk - 1);
      plo = plo < 0 ? 0 : (plo > LS - 1 ? LS - 1 : plo); phi = phi < 0 ? 0 : (phi > LS - 1 ? LS - 1 : phi);
      const int vhi = vb + tRows - 1 > LB - 1 ? LB - 1 : vb + tRows - 1;
      ou = ou < -plo ? -plo : ou; ou = ou > LS - 1 - phi ? LS - 1 - phi : ou;
      ov = ov < -vb ? -vb : ov; ov = ov > LB - 1 - vhi ? LB - 1 - vhi : ov;
      const int sx = TR ? ov : ou, sy = TR ? ou : ov;
      ox = FWD ? sx : -sx; oy = FWD ? sy : -sy;
    };
    auto publish_offsets = [&](int j, int ox, int oy, int row0) {   // (lane 0) what the band reads at the start of chunk j
      const int k = row0 >= 0 ? row0 / tRV : -((-row0 + tRV - 1) / tRV);   // floor: the rows [row0, row0 + 44] then lie in [k * 48, k * 48 + 96)
      if (lane == 0) sm.woff[w][j & 3] = make_float4(float(ox), float(oy), __int_as_float(k * tRV), 0.f);
    };
    // one block of D-slots [D0, D0 + n) x rows [row0, row0 + 48): loads into registers (issue), then stores
    float2 bv[kKT]; bool bok[kKT];
    auto block_load = [&](int D0, int n, int row0) {
      const bool inside = row0 >= 0 && row0 + tRect <= LB && D0 - (row0 + tRect - 1) >= 0 && D0 + kBlkD - 1 - row0 < LS;   // (wave-uniform)
      const int ibase = iC + iA * D0 + (iB - iA) * row0;
      if (inside && n == kBlkD) {   // the usual block: whole, inside the image
#pragma unroll
        for (int k = 0; k < kKT - 1; ++k) { bok[k] = true; bv[k] = g1[ibase + tio[k]]; }
        bok[kKT - 1] = lastHalf; bv[kKT - 1] = make_float2(0.f, 0.f);
        if (lastHalf) bv[kKT - 1] = g1[ibase + tio[kKT - 1]];
        return true;
      }
#pragma unroll
      for (int k = 0; k < kKT; ++k) {
        bv[k] = make_float2(0.f, 0.f);
        bok[k] = tdD[k] < n && tdr[k] < tRect;
        if (inside) { if (bok[k]) bv[k] = g1[ibase + tio[k]]; }
        else { const float2* q = bok[k] ? tex_ptr(D0 + tdD[k], row0 + tdr[k]) : nullptr; bok[k] = q != nullptr; if (bok[k]) bv[k] = *q; }
      }
      return false;
    };
    auto block_store = [&](int D0, int row0, bool whole) {
      const int r0m = ring_row(row0), c0 = D0 & (kWC - 1);   // (wave-uniform)
      const bool dupc = c0 + kBlkD > kWC || c0 < 2, dupr = r0m == 0 || r0m + tRect > tRV;
      if (whole) {
#pragma unroll
        for (int k = 0; k < kKT - 1; ++k) win_store_block(D0 + tdD[k], r0m, tdr[k], bv[k], dupc, dupr);
        if (lastHalf) win_store_block(D0 + tdD[kKT - 1], r0m, tdr[kKT - 1], bv[kKT - 1], dupc, dupr);
        return;
      }
#pragma unroll
      for (int k = 0; k < kKT; ++k) if (bok[k]) win_store_block(D0 + tdD[k], r0m, tdr[k], bv[k], dupc, dupr);
    };
    int rh = 0, idle = 0;
    // ---- chunk 0's whole rectangle (32 D-slots x 44 rows: four blocks, one after the other, once per band) ----
    int pox, poy, frontD, pov;
    int haveLo;   // every row of the current rectangle holds the skew slots [haveLo, frontD): a row that entered later than a slot was loaded lacks it
    {
      refill_offsets(0);
      pox = __builtin_amdgcn_readfirstlane(ocx); poy = __builtin_amdgcn_readfirstlane(ocy);
      int ou, ov; sweep_offsets(0, pox, poy, ou, ov);
      const int DB = uLo + vb + ou + ov, row0 = vb - kRadT + ov;
#pragma unroll 1
      for (int b = 0; b < 4; ++b) { const bool whole = block_load(DB - 2 * kRadT + 8 * b, 8, row0); block_store(DB - 2 * kRadT + 8 * b, row0, whole); }
      frontD = DB + 8 + 2 * kRadT; pov = ov; haveLo = DB - 2 * kRadT;
    }
    bool first = true;
    for (;;) {
      const int oh = first ? 0 : ld_cnt(&sm.outHead[w]);
      const bool ld = rh < nsteps && (rh + kChunk - oh <= kChunk * (tAhead + 1));
      if (!ld) {   // ring full (the usual state): a short idle iteration
        __builtin_amdgcn_s_sleep(kLoaderIdleSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { give_up(); break; }
        continue;
      }
      first = false;
      const int j = rh / kChunk;
      if (j - obase >= 64 || j < obase) refill_offsets(j);
      int tx = __builtin_amdgcn_readlane(ocx, j - obase), ty = __builtin_amdgcn_readlane(ocy, j - obase);
      tx = tx < pox - 1 ? pox - 1 : (tx > pox + 1 ? pox + 1 : tx);
      ty = ty < poy - 1 ? poy - 1 : (ty > poy + 1 ? poy + 1 : ty);
      int ou, ov; sweep_offsets(j, tx, ty, ou, ov);
      const int DB = uLo + vb + ou + ov, row0 = vb - kRadT + ov;
      const int need_lo = DB + 8 * j - 2 * kRadT, need_front = DB + 8 * j + 8 + 2 * kRadT;
      int n = need_front - frontD; n = n < 0 ? 0 : (n > 2 * kBlkD ? 2 * kBlkD : n);   // 6 .. 10 (0 for chunk 0, whose rectangle is in place)
      const int n1 = n > kBlkD ? kBlkD : n;
      const bool whole = block_load(frontD, n1, row0);
      // the row that entered the rectangle, over the D-slots that are already there (wave-uniform: the offset across moved)
      float2 sv = make_float2(0.f, 0.f); bool sok = false; int sD = 0, sV = 0;
      if (ov != pov) {
        sD = need_lo + lane; sV = ov > pov ? row0 + tRect - 1 : row0;
        const float2* q = sD < frontD ? tex_ptr(sD, sV) : nullptr;
        sok = q != nullptr;
        if (sok) sv = *q;
      }
      block_store(frontD, row0, whole);
      if (__any(sok)) { if (sok) win_store(sD, sV, sv); }
      if (ov != pov && need_lo > haveLo) haveLo = need_lo;   // (the row that just entered starts at need_lo)
      if (need_lo < haveLo) {
        // the window's lower edge moved BACK (wave-uniform, rare: the offset along the step axis is being cut back at the image's far border by
        // up to 8 per chunk while the offset across drifts the same way -- the sum may fall by 9): rows that entered during the last chunks lack
        // these slots (tests/test_follow_window_model.py found this)
        const int nb = haveLo - need_lo > kBlkD ? kBlkD : haveLo - need_lo;
        block_load(haveLo - nb, nb, row0); block_store(haveLo - nb, row0, false);
        haveLo -= nb;
      }
      if (n > kBlkD) {   // the ninth / tenth D-slot (wave-uniform, rare: the offsets' sum grew): one more, narrow block
        block_load(frontD + kBlkD, n - kBlkD, row0); block_store(frontD + kBlkD, row0, false);
      }
      publish_offsets(j, tx, ty, row0);
      frontD = frontD + n; pov = ov; pox = tx; poy = ty;
      rh += kChunk;
      st_cnt(&sm.recHead[w], rh);
      idle = 0;
      if (rh >= nsteps) break;
    }
    return;
  }
  if (wave == 2 * tWaves + 2) {
    // ======================= drainer: results LDS ring -> flow plane =======================
    int idle = 0;
    for (;;) {
      bool progress = false, done = true;
#pragma unroll
      for (int w = 0; w < tWaves; ++w) {
        if (w < nact) {
          int ot = sm.outTail[w];
          const int oh = ld_cnt(&sm.outHead[w]);
          int n = oh - ot; n = n > 8 ? 8 : n;
          if (n == 8 || (n > 0 && oh >= nsteps)) {   // whole chunks: all 64 lanes of the four stores at work, an eighth of the passes
#pragma unroll
            for (int u = 0; u < 4; ++u) {
              const int i = lane + 64 * u, j = i & 7, r = i >> 3, t = ot + j;   // a store covers 8 rows x 8 consecutive columns (64-byte runs)
              if (j < n) {
                const float2 val = sm.out[w][t % tOS][r];
                const int ia = uLo + t - r, ib = (bandLo + band0 + w) * tRows + r;
                if (t - r >= 0 && t - r < LSv && ib < LB) {
                  const int cx = transposed ? ib : ia, cy = transposed ? ia : ib;
                  const int x = forward ? cx : W - 1 - cx, y = forward ? cy : H - 1 - cy;
                  flow[size_t(y) * W + x] = val;
                }
              }
            }
            ot += n;
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            st_cnt(&sm.outTail[w], ot);
            progress = true;
          }
          if (ot < nsteps) done = false;
        }
      }
      if (done) break;
      if (progress) idle = 0;
      else {
        __builtin_amdgcn_s_sleep(kDrainSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { give_up(); break; }
      }
    }
    return;
  }
  if (wave == 2 * tWaves) {
    // ======================= publisher: last row of the workgroup -> granules in HBM =======================
    if (!publishes) return;
    unsigned long long* bnd_out = boundary + size_t(wg + 1) * LSv;
    const int wl = tWaves - 1;
    int pt = 0, idle = 0;
    while (pt < nsteps) {
      const int ohl = ld_cnt(&sm.outHead[wl]);
      int n = ohl - pt; n = n > tOS ? tOS : n;
      if (n > 0) {
        const int t = pt + lane;
        if (lane < n) {
          const float2 val = sm.out[wl][t % tOS][tRows - 1];
          const int cx = t - (tRows - 1);
          if (cx >= 0 && cx < LSv) __hip_atomic_store(bnd_out + cx, pack2(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        pt += n;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        st_cnt(&sm.pubTail, pt);
        idle = 0;
      } else {
        __builtin_amdgcn_s_sleep(kPubSleep);
        if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { give_up(); break; }
      }
    }
    return;
  }
  // ======================= poller: previous workgroup's granules HBM -> LDS ring =======================
  {
    if ((wg == 0 && !staticTop) || wave != 2 * tWaves + 1) return;
    const unsigned long long* bnd_in = boundary + size_t(wg) * LSv;
    int bh = 0, idle = 0;
    while (bh < LSv) {
      const int oh0 = ld_cnt(&sm.outHead[0]);
      if (bh + 64 - oh0 <= kBS) {
        unsigned long long g = kNotReady;
        if (bh + lane < LSv) {
          g = __hip_atomic_load(bnd_in + bh + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const bool ready = (g != kNotReady) || (bh + lane >= LSv);
        const unsigned long long m = __ballot(ready);
        const int n = (m == ~0ull) ? 64 : __builtin_ctzll(~m);
        if (n > 0) {
          if (lane < n && bh + lane < LSv) sm.bnd[(bh + lane) % kBS] = g;
          bh += n;
          st_cnt(&sm.bndHead, bh);
          idle = 0;
          continue;
        }
        __builtin_amdgcn_s_sleep(kPollSleep);
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
      if (spin_expired(idle, sm) || ld_cnt(&sm.abort)) { give_up(); break; }
    }
  }
}

// host side of the throughput form
static bool launch_sweep_t(hipStream_t st, const SweepArgs& a, float* rec) {
  const SweepWindow win = make_sweep_window(a.W, a.H, a.forward, a.ax0, a.ay0, a.ax1, a.ay1, tRows, tWaves, kChunk);
  if (win.empty) return false;
  const int tr = win.tr, uLo = win.uLo, uHi = win.uHi, LSv = win.LSv, bandLo = win.bandLo, nbands = win.nbands;
  const int nwg = win.nwg, nbandsPad = nwg * tWaves, nstepsPad = win.nstepsPad;
  const float rW = (float)(1.0 / (double)(float)a.W), rEps = (float)(1.0 / (double)kGradEpsilon);
  hipExtLaunchKernelGGL((k_sweep_prep<tRows, false>), dim3((unsigned)((nstepsPad + 256 / tRows - 1) / (256 / tRows)), (unsigned)nbandsPad, a.bt.n), dim3(256), 0, st, a.ev_start, nullptr, 0, a.g0, a.g1, a.blurred, a.gate,
                        a.flow, a.W, a.H, a.forward, tr, nstepsPad, nbandsPad, rW, reinterpret_cast<float4*>(rec), uLo, uHi, bandLo,
                        bandLo > 0 ? a.boundary : (unsigned long long*)nullptr, a.bt.stride, a.cf);
  const long long budget = 200000000ll + 1000ll * 50ll * (long long)(nstepsPad + 40 * nbands);
  const dim3 grid(nwg, 1, a.bt.n), block(tThreads);
  const float4* r4 = reinterpret_cast<const float4*>(rec);
#define PF_LAUNCH_SWEEP_T(TRV, FWV) hipExtLaunchKernelGGL((k_sweep_t<TRV, FWV>), grid, block, 0, st, nullptr, a.ev_stop, 0, r4, a.g1, a.flow, a.boundary, a.ctrl, a.W, a.H, nstepsPad, nbands, rW, rEps, uLo, LSv, bandLo, budget, a.bt.stride, a.blurred, a.cf)
  if (tr) { if (a.forward) PF_LAUNCH_SWEEP_T(true, true); else PF_LAUNCH_SWEEP_T(true, false); }
  else { if (a.forward) PF_LAUNCH_SWEEP_T(false, true); else PF_LAUNCH_SWEEP_T(false, false); }
#undef PF_LAUNCH_SWEEP_T
  return true;
}
