// LambdaRank "group" kernel (round 3): PairwiseLogisticLoss x (N)DCGLambdaWeight with smooth_fraction 0, no topn, a
// built-in monotone gain and no separate mask -- BASELINE config 3, the kernel north_star names.  Included by
// pairwise.hip inside its anonymous namespace (shares PwArgs and the pair-loss helpers).
//
// Reference behaviour restated (losses_impl.py): _compute_ranks :483-500, _pairwise_comparison :503-537,
// DCGLambdaWeight.pair_weights :255-279 / _pair_rank_discount :334-369 (smooth_fraction 0: |D(|r_i-r_j|) - D(|r_i-r_j|+1)|),
// inverse_max_dcg :109-134, _PairwiseLoss._compute_unreduced_loss_impl :871-884, _normalize_weights_impl :917-930,
// PairwiseLogisticLoss._pairwise_loss :936-940; backward = SURVEY.md Appendix B.
//
// What round 2's pairwise_lean_kernel left on the table (profiles/r02_pairwise_phases.txt):
//   (1) the rank-difference discount U[|r_i - r_j|] is a data-dependent LDS gather: 32 lanes with arbitrary ranks hit
//       the 32 banks ~3.3 deep, and together with the 16-byte column records the sweeps asked the CU's one LDS for
//       ~21 cycles per trip and wave against 50-72 cycles of VALU work, four SIMDs sharing it: LDS-bound;
//   (2) one list per wavefront and four such waves per SIMD at B = 4096: the kernel lasted 2.1x the mean wave's
//       lifetime (n^2 varies 4x between lists), i.e. half of the launch was a tail;
//   (3) every pair paid `sub, max, mul` for a weight (G_i - G_j)+ * u whose gain difference is constant over a whole
//       (row, grade segment) rectangle.
// This kernel:
//   * ONE workgroup = W wavefronts = W lists.  Wave w builds list w's LDS image alone (load, compaction, ranks by
//     counting, grade order, ideal DCG, records: no cross-wave traffic), ONE barrier, then the (list, 32-row pass)
//     work items of ALL W lists are dealt round-robin to the W waves: every SIMD of the CU gets an equal share of
//     every list, and the lists of a workgroup are drawn serpentine-wise from the longest-first order (long lists
//     are paired with short ones), so neither SIMDs nor CUs are left with the long lists.
//   * the table U[m] * list_size is replicated once per LDS bank (address = (m * R + lane % R) * 4, R = 32): the
//     gather is conflict-free by construction (2 LDS cycles per wave instruction), `v_sad_u16` on the pre-scaled
//     ranks still forms the address in one instruction.  The table is list independent: one copy per workgroup.
//   * records are split by use: the "hi" sweep reads (B_j, rank_j), the "lo" sweep (A_j, rank_j) -- 8 bytes per
//     column, two columns per ds_read_b128.
//   * grade segments start on 4-column boundaries (padding records with A = B = 0 contribute exactly nothing:
//     w = fma(A_i, 0, 1) = 1 -> log2 = 0, 1 - 1/w = 0), so a row pass sweeps one column SEGMENT at a time and the
//     gain difference is applied once per (row, segment): a hi pair is `sad, fma, rcp, log, fma, sub, fma`
//     (5 plain + 2 transcendental, was 8 + 2), a lo pair `sad, fma, rcp, sub, fma` (4 + 1, was 7 + 1).
//   * more than kMaxRuns distinct label values: the rest forms one unsorted tail segment whose column loops take the
//     gain difference per pair (PG variants).  Lists whose score range exceeds kLeanRange take a per-pair exp body.
#pragma once

constexpr int kGrpMaxSeg = kMaxRuns + 1;        // pure segments + the unsorted tail
constexpr int kGrpSegSlots = 10;
constexpr int kGrpPassRows = 32;                // rows per pass: two lanes per row
constexpr int kGrpMaxPass = 10;
constexpr int kGrpSlack = 16;                   // readable zero records behind every column array (prefetch overrun)

struct GrpHdr {                                 // per list, in LDS (256 bytes)
  int n, np, nseg, flags;                       // valid items; padded grade-order length (multiple of 4); segments; 1 = fast, 2 = tail
  int b, npass;                                 // list index (-1: empty slot), 32-row passes
  float lw;
  int state, next_pass, done_pass;              // 2 = image published; pass tickets handed out / passes completed
  int seg_start[kGrpSegSlots];                  // padded grade positions
  int seg_end[kGrpSegSlots];                    // start + count rounded up to 4
  float seg_gain[kGrpSegSlots];                 // gain * 1 / max DCG of the segment's label value (pure segments)
  float pass_loss[kGrpMaxPass];
  float pass_nnz[kGrpMaxPass];
};
static_assert(sizeof(GrpHdr) <= 256, "GrpHdr");
constexpr int kGrpHdrBytes = 256;

__host__ __device__ inline int grp_lp(int L) { return ((L + 3 * kGrpMaxSeg + 4 + 31) / 32) * 32; }
__host__ __device__ inline size_t grp_list_bytes(int Lp, bool itemw) {
  // recH float2 [Lp + slack], recL float2 [Lp + slack], GS float [Lp + slack], CIS int [Lp], (WS float [Lp + slack])
  return kGrpHdrBytes + (size_t)(Lp + kGrpSlack) * (8 + 8 + 4 + (itemw ? 4 : 0)) + (size_t)Lp * 4;
}
// the replicated rank-difference table [L * R] followed by a plain copy of the discount table D[0 .. L) (ideal DCG)
__host__ __device__ inline size_t grp_table_bytes(int L, int R) { return (((size_t)L * R * 4 + (size_t)L * 4) + 127) & ~(size_t)127; }

typedef const __attribute__((address_space(3))) float grp_lds_cf;

// u = Urep[(|r_i - r_j| * R + lane % R)]: ranks are stored pre-multiplied by 4 R, `ubase` = LDS address of the table
// + 4 (lane % R).  One v_sad_u16 forms the byte address (|a.lo16 - b.lo16| + |a.hi16 - b.hi16| + c).
__device__ __forceinline__ float grp_u(uint32_t ri, float rj_bits, uint32_t ubase) {
  return *(grp_lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(ri, (uint32_t)__float_as_int(rj_bits), ubase);
}

// 2^l - 1 (keras/utils.py:79-92), exact for integer grades like common.h's gain_pow2m1; the general exp2f (a libm
// expansion) is only executed when some lane of the wavefront holds a non-integer or huge label.
__device__ __forceinline__ float grp_gain_pow2m1(float l) {
  const float r = rintf(l);
  const bool exact = (r == l) && fabsf(l) < 120.0f;
  float p = __builtin_amdgcn_ldexpf(1.0f, (int)r);
  if (__ballot(!exact)) p = exact ? p : exp2f(l);
  return p - 1.0f;
}

// wave-wide OR on the DPP network (result wave-uniform, read from lane 63)
__device__ __forceinline__ unsigned grp_wave_or(unsigned v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}

struct GrpAcc { float l, g, w, nz; };

// "hi" sweep of one column segment [s, s + 4 trips): the row item is the preferred one.  Accumulates
// sum u * log2(1 + e^-(x_i - x_j)) and sum u * sigma(-(x_i - x_j)); PG: the gain difference is taken per pair
// (unsorted tail segment) and is part of u.  Lane c of a row takes columns s + 4 t + 2 c, + 1.
template <bool PG, bool AUX>
__device__ __forceinline__ void grp_hi_loop(const float2* recH, const float* GS, int s, int trips, int c, float Ai,
                                            float Gi, uint32_t ri, uint32_t ubase, GrpAcc& acc) {
  float al = 0.f, ag = 0.f, aw = 0.f, anz = 0.f;
  const float4* p = reinterpret_cast<const float4*>(recH + s + 2 * c);      // a trip = 4 records = 2 float4
  const float2* gp = reinterpret_cast<const float2*>(GS + s + 2 * c);
  float4 cr = p[0];
  float u0 = grp_u(ri, cr.y, ubase), u1 = grp_u(ri, cr.w, ubase);
  float4 cn = p[2];
#pragma unroll 2
  for (int t = 0; t < trips; ++t) {                           // uniform trip count: scalar loop control
    const float un0 = grp_u(ri, cn.y, ubase), un1 = grp_u(ri, cn.w, ubase);
    const float4 cnn = p[2 * t + 4];
    if (PG) {
      const float2 gj = gp[2 * t];
      u0 *= fmaxf(Gi - gj.x, 0.0f); u1 *= fmaxf(Gi - gj.y, 0.0f);
    }
    const float w0 = __builtin_fmaf(Ai, cr.x, 1.0f), w1 = __builtin_fmaf(Ai, cr.z, 1.0f);
    const float q0 = __builtin_amdgcn_rcpf(w0), q1 = __builtin_amdgcn_rcpf(w1);
    const float l0 = __builtin_amdgcn_logf(w0), l1 = __builtin_amdgcn_logf(w1);
    al = __builtin_fmaf(u0, l0, al); al = __builtin_fmaf(u1, l1, al);
    ag = __builtin_fmaf(u0, 1.0f - q0, ag); ag = __builtin_fmaf(u1, 1.0f - q1, ag);
    if (AUX) {                                                // padding columns (B = 0) carry a non-zero u: mask them
      const float v0 = (cr.x != 0.0f) ? u0 : 0.0f, v1 = (cr.z != 0.0f) ? u1 : 0.0f;
      aw += v0; aw += v1;
      anz += (v0 != 0.0f) ? 1.0f : 0.0f; anz += (v1 != 0.0f) ? 1.0f : 0.0f;
    }
    cr = cn; cn = cnn; u0 = un0; u1 = un1;
  }
  acc.l = al; acc.g = ag; acc.w = aw; acc.nz = anz;
}

// "lo" sweep of one column segment: the COLUMN item is the preferred one; the row receives sum u * sigma(-(x_j - x_i))
// (times the column's item weight, ITEMW).
template <bool PG, bool ITEMW>
__device__ __forceinline__ float grp_lo_loop(const float2* recL, const float* GS, const float* WS, int s, int trips, int c,
                                             float Bi, float Gi, uint32_t ri, uint32_t ubase) {
  float ag = 0.f;
  const float4* p = reinterpret_cast<const float4*>(recL + s + 2 * c);
  const float2* gp = reinterpret_cast<const float2*>(GS + s + 2 * c);
  const float2* wp = reinterpret_cast<const float2*>(WS + s + 2 * c);
  float4 cr = p[0];
  float u0 = grp_u(ri, cr.y, ubase), u1 = grp_u(ri, cr.w, ubase);
  float4 cn = p[2];
#pragma unroll 2
  for (int t = 0; t < trips; ++t) {
    const float un0 = grp_u(ri, cn.y, ubase), un1 = grp_u(ri, cn.w, ubase);
    const float4 cnn = p[2 * t + 4];
    if (PG) {
      const float2 gj = gp[2 * t];
      u0 *= fmaxf(gj.x - Gi, 0.0f); u1 *= fmaxf(gj.y - Gi, 0.0f);
    }
    if (ITEMW) {
      const float2 wj = wp[2 * t];
      u0 *= wj.x; u1 *= wj.y;                                // the weight of the PREFERRED item (:917-930)
    }
    const float q0 = __builtin_amdgcn_rcpf(__builtin_fmaf(Bi, cr.x, 1.0f));
    const float q1 = __builtin_amdgcn_rcpf(__builtin_fmaf(Bi, cr.z, 1.0f));
    ag = __builtin_fmaf(u0, 1.0f - q0, ag); ag = __builtin_fmaf(u1, 1.0f - q1, ag);
    cr = cn; cn = cnn; u0 = un0; u1 = un1;
  }
  return ag;
}

// The pure segments a row pass needs are CONTIGUOUS columns (a segment ends where the next one starts), so the hi sweep
// is ONE software-pipelined loop over [seg_start[h0], seg_end[h1 - 1]) and the per-segment partial sums are folded into the
// row totals with the segment's gain difference whenever a trip crosses a segment end (a scalar compare per trip):
// no per-rectangle loop start-up (address set-up, two dependent LDS round trips) -- 43 rectangles per list on average.
template <bool AUX>
__device__ __forceinline__ void grp_hi_sweep(const float2* recH, int h0, int h1, int seg_s, int seg_e, float seg_g, int c,
                                             float Ai, float Gi, uint32_t ri, uint32_t ubase, float& accL, float& accG,
                                             float& accW, float& accNZ) {
  if (h0 >= h1) return;
  const int j0 = __builtin_amdgcn_readlane(seg_s, h0), j1 = __builtin_amdgcn_readlane(seg_e, h1 - 1);
  const int trips = (j1 - j0) >> 2;
  int h = h0;
  int tb = (__builtin_amdgcn_readlane(seg_e, h) - j0) >> 2;                   // trip at which segment h ends
  float dG = fmaxf(Gi - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, seg_g), h)), 0.0f);
  float al = 0.f, ag = 0.f, aw = 0.f, anz = 0.f;
  const float4* p = reinterpret_cast<const float4*>(recH + j0 + 2 * c);        // a trip = 4 records = 2 float4
  // pipeline state: CR = records of this trip, U = their rank-difference discounts, CN = records of the next trip.
  // The loop body is written out for two trips with the register roles swapped, so nothing is moved between trips.
#define GRP_HI_TRIP(CR, U0, U1, CN, NU0, NU1, NEWC, T)                                                             \
  do {                                                                                                              \
    if ((T) == tb) {                /* (uniform) the columns of the next segment start here */                      \
      accL = __builtin_fmaf(dG, al, accL); accG = __builtin_fmaf(dG, ag, accG);                                    \
      if (AUX) { accW = __builtin_fmaf(dG, aw, accW); accNZ += (dG != 0.0f) ? anz : 0.0f; aw = 0.f; anz = 0.f; }    \
      al = 0.f; ag = 0.f;                                                                                           \
      ++h;                                                                                                          \
      tb = (__builtin_amdgcn_readlane(seg_e, h) - j0) >> 2;                                                         \
      dG = fmaxf(Gi - __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, seg_g), h)), 0.0f); \
    }                                                                                                               \
    NU0 = grp_u(ri, CN.y, ubase); NU1 = grp_u(ri, CN.w, ubase);                                                     \
    NEWC = p[2 * (T) + 6];                                                                                          \
    const float w0_ = __builtin_fmaf(Ai, CR.x, 1.0f), w1_ = __builtin_fmaf(Ai, CR.z, 1.0f);                         \
    const float q0_ = __builtin_amdgcn_rcpf(w0_), q1_ = __builtin_amdgcn_rcpf(w1_);                                 \
    const float l0_ = __builtin_amdgcn_logf(w0_), l1_ = __builtin_amdgcn_logf(w1_);                                 \
    al = __builtin_fmaf(U0, l0_, al); al = __builtin_fmaf(U1, l1_, al);                                             \
    ag = __builtin_fmaf(U0, 1.0f - q0_, ag); ag = __builtin_fmaf(U1, 1.0f - q1_, ag);                               \
    if (AUX) {                      /* padding columns (B = 0) carry a non-zero u: mask them */                     \
      const float v0_ = (CR.x != 0.0f) ? U0 : 0.0f, v1_ = (CR.z != 0.0f) ? U1 : 0.0f;                               \
      aw += v0_; aw += v1_;                                                                                         \
      anz += (v0_ != 0.0f) ? 1.0f : 0.0f; anz += (v1_ != 0.0f) ? 1.0f : 0.0f;                                       \
    }                                                                                                               \
  } while (0)
  // three trips in flight: records two trips ahead, discounts one trip ahead (an LDS round trip is ~100+ cycles under
  // load, a trip ~50 of VALU work: with the discounts fetched only one trip ahead every trip waited on them)
  float4 ca = p[0], cb = p[2], cc = p[4];
  float ua0 = grp_u(ri, ca.y, ubase), ua1 = grp_u(ri, ca.w, ubase);
  float ub0 = grp_u(ri, cb.y, ubase), ub1 = grp_u(ri, cb.w, ubase);
  int t = 0;
  for (; t + 3 <= trips; t += 3) {                            // uniform trip count: scalar loop control
    float4 cd, ce, cf;
    float uc0, uc1, ud0, ud1, ue0, ue1;
    GRP_HI_TRIP(ca, ua0, ua1, cc, uc0, uc1, cd, t);           // computes trip t, gathers for t + 2, loads records t + 3
    GRP_HI_TRIP(cb, ub0, ub1, cd, ud0, ud1, ce, t + 1);
    GRP_HI_TRIP(cc, uc0, uc1, ce, ue0, ue1, cf, t + 2);
    ca = cd; ua0 = ud0; ua1 = ud1; cb = ce; ub0 = ue0; ub1 = ue1; cc = cf;
  }
  if (t < trips) {
    float4 cd; float uc0, uc1;
    GRP_HI_TRIP(ca, ua0, ua1, cc, uc0, uc1, cd, t);
    if (t + 1 < trips) { float4 ce; float ud0, ud1; GRP_HI_TRIP(cb, ub0, ub1, cd, ud0, ud1, ce, t + 1); (void)ce; (void)ud0; (void)ud1; }
    (void)uc0; (void)uc1;
  }
#undef GRP_HI_TRIP
  accL = __builtin_fmaf(dG, al, accL); accG = __builtin_fmaf(dG, ag, accG);
  if (AUX) { accW = __builtin_fmaf(dG, aw, accW); accNZ += (dG != 0.0f) ? anz : 0.0f; }
}

// The lo sweep over the pure segments [0, h1): one loop over the columns [0, seg_end[h1 - 1]).
template <bool ITEMW>
__device__ __forceinline__ void grp_lo_sweep(const float2* recL, const float* WS, int h1, int seg_e, float seg_g, int c,
                                             float Bi, float Gi, uint32_t ri, uint32_t ubase, float& accG2) {
  if (h1 <= 0) return;
  const int j1 = __builtin_amdgcn_readlane(seg_e, h1 - 1);
  const int trips = j1 >> 2;
  int h = 0;
  int tb = __builtin_amdgcn_readlane(seg_e, 0) >> 2;
  float dG = fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, seg_g), 0)) - Gi, 0.0f);
  float ag = 0.f;
  const float4* p = reinterpret_cast<const float4*>(recL + 2 * c);
  const float2* wp = reinterpret_cast<const float2*>(WS + 2 * c);
#define GRP_LO_TRIP(CR, U0, U1, CN, NU0, NU1, NEWC, T)                                                             \
  do {                                                                                                              \
    if ((T) == tb) {                                                                                                \
      accG2 = __builtin_fmaf(dG, ag, accG2);                                                                        \
      ag = 0.f;                                                                                                     \
      ++h;                                                                                                          \
      tb = __builtin_amdgcn_readlane(seg_e, h) >> 2;                                                                \
      dG = fmaxf(__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, seg_g), h)) - Gi, 0.0f); \
    }                                                                                                               \
    NU0 = grp_u(ri, CN.y, ubase); NU1 = grp_u(ri, CN.w, ubase);                                                     \
    NEWC = p[2 * (T) + 6];                                                                                          \
    float x0_ = U0, x1_ = U1;                                                                                       \
    if (ITEMW) { const float2 wj_ = wp[2 * (T)]; x0_ *= wj_.x; x1_ *= wj_.y; }  /* weight of the PREFERRED item */  \
    const float q0_ = __builtin_amdgcn_rcpf(__builtin_fmaf(Bi, CR.x, 1.0f));                                        \
    const float q1_ = __builtin_amdgcn_rcpf(__builtin_fmaf(Bi, CR.z, 1.0f));                                        \
    ag = __builtin_fmaf(x0_, 1.0f - q0_, ag); ag = __builtin_fmaf(x1_, 1.0f - q1_, ag);                             \
  } while (0)
  float4 ca = p[0], cb = p[2], cc = p[4];                     // three trips in flight (see grp_hi_sweep)
  float ua0 = grp_u(ri, ca.y, ubase), ua1 = grp_u(ri, ca.w, ubase);
  float ub0 = grp_u(ri, cb.y, ubase), ub1 = grp_u(ri, cb.w, ubase);
  int t = 0;
  for (; t + 3 <= trips; t += 3) {
    float4 cd, ce, cf;
    float uc0, uc1, ud0, ud1, ue0, ue1;
    GRP_LO_TRIP(ca, ua0, ua1, cc, uc0, uc1, cd, t);
    GRP_LO_TRIP(cb, ub0, ub1, cd, ud0, ud1, ce, t + 1);
    GRP_LO_TRIP(cc, uc0, uc1, ce, ue0, ue1, cf, t + 2);
    ca = cd; ua0 = ud0; ua1 = ud1; cb = ce; ub0 = ue0; ub1 = ue1; cc = cf;
  }
  if (t < trips) {
    float4 cd; float uc0, uc1;
    GRP_LO_TRIP(ca, ua0, ua1, cc, uc0, uc1, cd, t);
    if (t + 1 < trips) { float4 ce; float ud0, ud1; GRP_LO_TRIP(cb, ub0, ub1, cd, ud0, ud1, ce, t + 1); (void)ce; (void)ud0; (void)ud1; }
    (void)uc0; (void)uc1;
  }
#undef GRP_LO_TRIP
  accG2 = __builtin_fmaf(dG, ag, accG2);
}

// Round 6: the builder for graded relevance (labels = small non-negative integers, at most kMaxRuns distinct values, up
// to 255 items) -- BASELINE config 3.  profiles/r06_lambdarank_counts.txt: the general builder below spends 257 + 434
// vector instructions per list on the compaction and the grade order (a ballot + popcount round per register and per
// grade, every one a VALU -> SALU -> VALU round trip), 15 % of the kernel's vector-pipe time.  Here a lane owns IPL
// CONSECUTIVE items (element order = lane-major order), every item adds a one-hot BYTE for its grade into a 64-bit word
// and ONE inclusive scan of that word over the wavefront (two dwords x six DPP adds) gives, for every grade at once, how
// many items of the grade precede each item: its position inside the grade segment (the byte of its own grade) and its
// compact position among the valid items (the sum of all eight bytes, one v_sad_u8 per dword).  Segment starts come
// from a three-step scan of the per-grade totals on lanes 0 .. 7 and reach the items through one ds_bpermute.
// The image is the one the general builder writes (same segments, same order inside a segment, same padding), and the
// ideal DCG is added up in the general builder's order (the terms go through LDS once), so the outputs are
// bit-identical to it (tests/test_gpu_parity.py compares the two builders).  Returns false (wave-uniform, nothing of the
// image written) for any other label set: the caller then runs the general builder.
template <int IPL>
__device__ __forceinline__ bool grp_build_graded(const PwArgs& a, const int b, const int lane, const int L, const int Lp,
                                                 const int LpS, const int R, const float lw, const float (&lab)[IPL],
                                                 const float (&xin)[IPL], GrpHdr* H, float2* recH, float2* recL,
                                                 float* GS, int* CIS, const float* Dlds) {
  const size_t base = (size_t)b * L;
  const bool unit_t = a.temperature == 1.0f, pow2 = a.gain_kind == TFR_GAIN_POW2M1;
  bool lv[IPL];
  int li[IPL];
  float x[IPL];
  bool bad = false;
  unsigned present = 0u;
  float xmin = INFINITY, xmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = IPL * lane + r;
    lv[r] = e < L && lab[r] >= 0.0f;
    const int q = (int)lab[r];
    const bool ok = (float)q == lab[r] && (unsigned)q < 32u;
    bad = bad || (lv[r] && !ok);
    li[r] = q & 31;
    x[r] = unit_t ? xin[r] : xin[r] / a.temperature;
    if (lv[r]) {
      present |= 1u << li[r];
      xmin = fminf(xmin, x[r]); xmax = fmaxf(xmax, x[r]);
    } else if (e < L) {
      if (a.row_loss) a.row_loss[base + e] = 0.f;
      if (a.dlogits) a.dlogits[base + e] = 0.f;
    }
  }
  if (__ballot(bad)) return false;
  present = grp_wave_or(present);
  const int ng = __popc(present);
  if (ng > kMaxRuns) return false;

  // one-hot grade bytes: grade index k = how many of the present grades lie above the item's
  const unsigned presh = present >> 1;
  unsigned sh[IPL], pl[IPL], ph[IPL];
  unsigned sl = 0u, su = 0u;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    sh[r] = ((unsigned)__popc(presh >> li[r]) & 7u) << 3;      // (k <= 7 for a valid item; the mask only tames invalid ones)
    const unsigned long long o = (unsigned long long)(lv[r] ? 1u : 0u) << sh[r];
    pl[r] = sl; ph[r] = su;
    sl += (unsigned)o; su += (unsigned)(o >> 32);
  }
  const unsigned il = (unsigned)wave_scan_incl_i((int)sl), iu = (unsigned)wave_scan_incl_i((int)su);
  const unsigned Tl = (unsigned)__builtin_amdgcn_readlane((int)il, 63), Tu = (unsigned)__builtin_amdgcn_readlane((int)iu, 63);
  const unsigned el = il - sl, eu = iu - su;
  const int n = (int)((Tl & 0xffu) + ((Tl >> 8) & 0xffu) + ((Tl >> 16) & 0xffu) + (Tl >> 24) +
                      (Tu & 0xffu) + ((Tu >> 8) & 0xffu) + ((Tu >> 16) & 0xffu) + (Tu >> 24));

  // lanes 0 .. 7: the grade's count, label and (sorted position | padded grade position << 16) of its first item
  unsigned long long labs = 0ull;                             // byte g = the g-th largest label present (scalar loop)
  {
    unsigned pm = present;
    for (int g = 0; g < ng; ++g) {
      const int gb = 31 - __builtin_clz(pm);
      pm &= ~(1u << gb);
      labs |= (unsigned long long)gb << (8 * g);
    }
  }
  const unsigned gsh = (unsigned)(lane & 7) << 3;
  const unsigned long long T64 = ((unsigned long long)Tu << 32) | Tl;
  const unsigned cnt_g = lane < ng ? (unsigned)(T64 >> gsh) & 0xffu : 0u;
  const unsigned lab_g = (unsigned)(labs >> gsh) & 0xffu;
  const unsigned pc = cnt_g | (((cnt_g + 3u) & ~3u) << 16);
  unsigned sc = pc;
  sc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sc, 0x111, 0xf, 0xf, true);
  sc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sc, 0x112, 0xf, 0xf, true);
  sc += (unsigned)__builtin_amdgcn_update_dpp(0, (int)sc, 0x114, 0xf, 0xf, true);
  const unsigned startv = sc - pc;
  const int np = ng > 0 ? (int)((unsigned)__builtin_amdgcn_readlane((int)sc, ng - 1) >> 16) : 0;   // padded length (multiple of 4)
  const float gain_g = pow2 ? (__builtin_amdgcn_ldexpf(1.0f, (int)lab_g) - 1.0f) : (float)lab_g;

  // per item: position inside its grade, compact position, sorted / padded grade position
  int posr[IPL], sp[IPL], gp[IPL];
  float g[IPL];
  float* XS = reinterpret_cast<float*>(recH);                 // scratch: compact scores (rank count)
  int* RKS = reinterpret_cast<int*>(recL);                    // scratch: count by compact position
  int* OCC = RKS + Lp;                                        // scratch: how many items share a count
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const unsigned cl = el + pl[r], cu = eu + ph[r];
    const unsigned off = (unsigned)((((unsigned long long)cu << 32) | cl) >> sh[r]) & 0xffu;
    posr[r] = (int)__builtin_amdgcn_sad_u8(cl, 0u, __builtin_amdgcn_sad_u8(cu, 0u, 0u));
    const unsigned st = (unsigned)__builtin_amdgcn_ds_bpermute((int)(sh[r] >> 1), (int)startv);
    sp[r] = (int)((st & 0xffffu) + off);
    gp[r] = (int)((st >> 16) + off);
    g[r] = pow2 ? (__builtin_amdgcn_ldexpf(1.0f, li[r]) - 1.0f) : (float)li[r];
    if (lv[r]) XS[posr[r]] = x[r];
  }
  xmin = wave_min_u(xmin); xmax = wave_max_u(xmax);
  const bool fast = (xmax - xmin) <= kLeanRange;              // wave-uniform
  const float m = 0.5f * (xmax + xmin);
  const int n4 = (n + 3) >> 2;
  for (int p = n + lane; p < n4 * 4 + 4; p += 64) XS[p] = -INFINITY;
  WAVE_LDS_SYNC();

  // ranks by counting (score descending, ties by index) (:483-500): the bucket partition, the sweep as its fallback
  int rk[IPL];
  if (LpS < 192 || !wave_rank_by_bucket<IPL>(XS, n, lane, RKS, OCC, XS + Lp + 8, reinterpret_cast<int*>(GS), true, xmin, xmax))
    wave_rank_by_count(XS, n, lane, RKS, OCC);
#pragma unroll
  for (int r = 0; r < IPL; ++r) rk[r] = lv[r] ? RKS[posr[r]] : 0;
  WAVE_LDS_SYNC();                                            // the scratch is rewritten below

  // ideal DCG (:109-134): the terms g * D[sorted position] through LDS, added up in the general builder's order
  float inv_max_dcg = 1.0f;
  if (a.normalized) {
    float* TS = reinterpret_cast<float*>(recL);               // 64 * IPL floats (2 LpS >= 64 * IPL + 32)
#pragma unroll
    for (int r = 0; r < IPL; ++r) TS[IPL * lane + r] = lv[r] ? g[r] * Dlds[sp[r]] : 0.0f;
    WAVE_LDS_SYNC();
    float idcg = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) idcg += TS[lane + 64 * r];
    idcg = wave_sum_u(idcg);
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
    WAVE_LDS_SYNC();
  }

  // the image: zero records in the (at most three) padding slots behind every segment and in the 32 slots behind the
  // last one (rows of the last pass, prefetch overrun of the column loops); CIS = -1 there
  if (lane < ng) {
    const int e0 = (int)(startv >> 16) + (int)cnt_g, e1 = (int)(startv >> 16) + (int)((cnt_g + 3u) & ~3u);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int p = e0 + j;
      if (p < e1) { recH[p] = make_float2(0.f, 0.f); recL[p] = make_float2(0.f, 0.f); GS[p] = 0.f; CIS[p] = -1; }
    }
    H->seg_start[lane] = (int)(startv >> 16); H->seg_end[lane] = e1; H->seg_gain[lane] = gain_g * inv_max_dcg;
  }
  if (lane < 32) {
    const int p = np + lane;
    if (p < LpS) { recH[p] = make_float2(0.f, 0.f); recL[p] = make_float2(0.f, 0.f); GS[p] = 0.f; }
    if (p < Lp) CIS[p] = -1;
  }
  const int rscale = 4 * R;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    if (!lv[r]) continue;
    const float xv = x[r];
    float Bv, Av;
    if (fast) {
      const float t_hi = xv - m;
      const float bb = t_hi - xv;
      const float t_lo = (xv - (t_hi - bb)) + (-m - bb);
      Bv = exp_df_hw(t_hi, t_lo);                             // e^(x - m), 1 ulp
      Av = __builtin_amdgcn_rcpf(Bv);                        // e^-(x - m)
    } else {
      Bv = xv; Av = 1.0f;                                     // slow path: the score itself / a validity flag
    }
    const float rb = __int_as_float(rk[r] * rscale);
    recH[gp[r]] = make_float2(Bv, rb);
    recL[gp[r]] = make_float2(Av, rb);
    GS[gp[r]] = g[r] * inv_max_dcg;
    CIS[gp[r]] = IPL * lane + r;
  }
  const int npass = (np + kGrpPassRows - 1) / kGrpPassRows;
  if (lane == 0) {
    H->n = n; H->np = np; H->nseg = ng; H->flags = fast ? 1 : 0;
    H->npass = npass; H->lw = lw;
    if (npass == 0) {                                         // no valid item: nothing to sweep, the sums are zero
      if (a.list_loss && !a.sum.out) a.list_loss[b] = 0.f;
    }
  }
  if (npass == 0 && a.sum.out) grid_sum_contribute(a.sum, b, 0.f, lane);
  return true;
}

// BKT (round 4, TFR_LAMBDARANK_BUCKET=0 to switch off): the builder takes its ranks from wave_rank_by_bucket (the NDCG
// metric kernel's rank step since round 4: same integers, ~1/5 of the instructions for 200 items) with the counting
// sweep as the fallback -- outputs bit-identical to the counting builder (tools/lbucket_check.py: the config-3 batch,
// 1024 x 256, tied scores, an outlier, short lists), kernel 47.5 -> 44.5 us at B = 4096, L = 200.
template <int IPL, bool AUX, bool ITEMW, bool BKT = false>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(6))) void lambdarank_group_kernel(const PwArgs a, const int B, const int R, const int Lp,
                                                                const int G, const int kflags) {
  // kflags bit 0: BKT builders take grp_build_graded (lane-major loads) when the labels allow it
  // G lists per workgroup, blockDim.x / 64 >= G wavefronts: waves 0 .. G-1 each BUILD one list's LDS image, then every
  // wave SWEEPS (list, 32-row pass) work items of whichever lists are ready, taken from per-list LDS tickets -- no
  // barrier between the two phases: a wave that has built its (short) list sweeps while a long list is still being
  // ranked, and the latency-bound build of one wave overlaps the VALU-bound sweeps of the others on its SIMD.
  extern __shared__ __attribute__((aligned(128))) unsigned char grp_smem[];
  const int Wt = blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int L = a.L;
  float* Urep = reinterpret_cast<float*>(grp_smem);                         // [L * R]
  float* Dlds = Urep + (size_t)L * R;                                       // [L] the discount table itself (the builders' ideal DCG gathers it)
  const size_t tab_bytes = grp_table_bytes(L, R);
  const size_t per_list = grp_list_bytes(Lp, ITEMW);
  const int LpS = Lp + kGrpSlack;
#ifdef TFR_PROFILE_STAMPS
  unsigned long long grp_t[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int grp_passes = 0, grp_polls = 0;
  const int grp_stop = g_prof_stop_pw;
#define GRP_STAMP(i) do { grp_t[i] = __builtin_amdgcn_s_memtime(); if ((i) >= 1 && (i) <= 5 && grp_stop == (i)) return; } while (0)
#define GRP_SKIP_HI (grp_stop == 7)
#define GRP_SKIP_LO (grp_stop == 6)
#else
#define GRP_STAMP(i) do { } while (0)
#define GRP_SKIP_HI false
#define GRP_SKIP_LO false
#endif
  GRP_STAMP(0);
#define GRP_HDR(l) reinterpret_cast<GrpHdr*>(grp_smem + tab_bytes + (size_t)(l) * per_list)
#define GRP_RECH(l) reinterpret_cast<float2*>(grp_smem + tab_bytes + (size_t)(l) * per_list + kGrpHdrBytes)
  const bool builder = wave < G;

  // ---- 0a. the builder's list (drawn serpentine-wise from the launch order) and its raw loads, issued first.
  int b = -1;
  float lw = 1.0f;
  float lab_raw[IPL], x_raw[IPL], w_raw[IPL];
  if (builder) {
    const int N = gridDim.x;
    const int posn = (wave & 1) ? (wave + 1) * N - 1 - (int)blockIdx.x : wave * N + (int)blockIdx.x;
    b = (posn < B) ? (a.order ? a.order[posn] : posn) : -1;
    if (lane == 0) {                                          // (LDS is not initialised: tickets and state before anyone polls)
      GrpHdr* H = GRP_HDR(wave);
      H->state = 0; H->next_pass = 0; H->done_pass = 0; H->npass = 0; H->b = b; H->n = 0;
    }
  }
  const bool graded = BKT && (kflags & 1) && L <= 255;        // (wave-uniform) lane-major loads for grp_build_graded
#pragma unroll
  for (int r = 0; r < IPL; ++r) { lab_raw[r] = -1.0f; x_raw[r] = 0.0f; w_raw[r] = 1.0f; }
  if (BKT && graded) {
    if (b >= 0) {
      const float* lp = a.labels + (size_t)b * L;
      const float* xp = a.logits + (size_t)b * L;
      if (IPL == 4 && (L & 3) == 0 && (((uintptr_t)a.labels | (uintptr_t)a.logits) & 15) == 0) {
        if (4 * lane < L) {                                   // one 16-byte load per array
          const float4 lv4 = *reinterpret_cast<const float4*>(lp + 4 * lane);
          const float4 xv4 = *reinterpret_cast<const float4*>(xp + 4 * lane);
          lab_raw[0] = lv4.x; lab_raw[1 % IPL] = lv4.y; lab_raw[2 % IPL] = lv4.z; lab_raw[3 % IPL] = lv4.w;
          x_raw[0] = xv4.x; x_raw[1 % IPL] = xv4.y; x_raw[2 % IPL] = xv4.z; x_raw[3 % IPL] = xv4.w;
        }
      } else {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int e = IPL * lane + r;
          if (e < L) { lab_raw[r] = lp[e]; x_raw[r] = xp[e]; }
        }
      }
    }
  } else {
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = lane + 64 * r;
      if (b >= 0 && e < L) {
        lab_raw[r] = a.labels[(size_t)b * L + e];
        x_raw[r] = a.logits[(size_t)b * L + e];
        if (ITEMW) w_raw[r] = a.item_weights[(size_t)b * L + e];
      }
    }
  }
  if (b >= 0 && a.list_weights) lw = a.list_weights[b];

  // ---- 0b. the replicated rank-difference table (list independent).  Thread (part, m) loads D(m), D(m + 1) once and
  // stores its R / P copies of U[m] * list_size, rotated by the lane inside its own slot range so that the lanes of a store
  // spread over the banks.  (P = how many times the L table rows fit into the workgroup, a power of two <= R.)
  {
    int P = 1;
    while (2 * P * L <= (int)blockDim.x && 2 * P <= R) P *= 2;
    const int per = R / P;
    for (int idx = threadIdx.x; idx < L * P; idx += blockDim.x) {            // (one trip unless the workgroup is smaller than L)
      int part = 0, m = idx;
      while (m >= L) { m -= L; ++part; }
      const float dm = a.discount[m];
      const float v = (m >= 1) ? fabsf(a.discount[m - 1] - dm) * (float)L : 0.0f;   // x list_size (:278) folded in
      if (part == 0) Dlds[m] = dm;
      // (the rotation by the lane stays inside the thread's own `per` slots: per = 16 -> 2-way store conflicts, free)
      for (int k = 0; k < per; ++k) Urep[m * R + part * per + ((k + lane) & (per - 1))] = v;
    }
  }
  __syncthreads();                                            // the table and the tickets are in place
#ifdef TFR_PROFILE_STAMPS
  if (grp_stop >= 1 && grp_stop <= 5 && !builder) return;
#endif
  GRP_STAMP(1);

  // ---- 1. wave w < G builds the LDS image of its list.
  if (builder) {
    GrpHdr* H = GRP_HDR(wave);
    float2* recH = GRP_RECH(wave);
    float2* recL = recH + LpS;
    float* GS = reinterpret_cast<float*>(recL + LpS);
    float* WS = GS + LpS;                                                   // ITEMW only
    int* CIS = reinterpret_cast<int*>(GS + LpS + (ITEMW ? LpS : 0));
    bool built = false;
    if (BKT && graded && b >= 0) {
      built = grp_build_graded<IPL>(a, b, lane, L, Lp, LpS, R, lw, lab_raw, x_raw, H, recH, recL, GS, CIS, Dlds);
      if (!built) {                                           // any other label set: the general builder, element e = lane + 64 r
        float* TS = reinterpret_cast<float*>(recH);           // (2 LpS floats >= 128 IPL)
#pragma unroll
        for (int r = 0; r < IPL; ++r) { TS[IPL * lane + r] = lab_raw[r]; TS[64 * IPL + IPL * lane + r] = x_raw[r]; }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int r = 0; r < IPL; ++r) { lab_raw[r] = TS[lane + 64 * r]; x_raw[r] = TS[64 * IPL + lane + 64 * r]; }
        WAVE_LDS_SYNC();
      }
    }
    if (b >= 0 && !built) {
      const size_t base = (size_t)b * L;
      float* XS = reinterpret_cast<float*>(recH);                           // scratch: compact scores (rank count)
      int* RKS = reinterpret_cast<int*>(recL);                              // scratch: count by compact position
      int* OCC = RKS + Lp;                                                  // scratch: how many items share a count
      // 1a. gains, compaction of the valid items (mask == NULL: valid = label >= 0).
      float g[IPL], xr[IPL], labr[IPL], wr[IPL];
      int posr[IPL];
      bool lv[IPL];
      int n = 0;
      float xmin = INFINITY, xmax = -INFINITY;
      const bool unit_t = a.temperature == 1.0f, pow2 = a.gain_kind == TFR_GAIN_POW2M1;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int e = lane + 64 * r;
        g[r] = 0.f; lv[r] = false;
        float x = 0.f, lab = -1.f, w = 0.f;
        if (e < L) {
          lab = lab_raw[r];
          x = unit_t ? x_raw[r] : x_raw[r] / a.temperature;   // (the division is ~10 instructions; T = 1 is the common case)
          lv[r] = lab >= 0.0f;
          if (lv[r]) {
            g[r] = pow2 ? grp_gain_pow2m1(lab) : lab;
            w = ITEMW ? w_raw[r] * lw : lw;
            xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
          } else {
            if (a.row_loss) a.row_loss[base + e] = 0.f;
            if (AUX && a.row_weight) a.row_weight[base + e] = 0.f;
            if (a.dlogits) a.dlogits[base + e] = 0.f;
          }
        }
        const unsigned long long bal = __ballot(lv[r]);
        xr[r] = x; labr[r] = lv[r] ? lab : -1.0f; wr[r] = w;
        posr[r] = n + __popcll(bal & ((1ull << lane) - 1ull));
        if (lv[r]) XS[posr[r]] = x;
        n += __popcll(bal);
      }
      xmin = wave_min_u(xmin); xmax = wave_max_u(xmax);
      const bool fast = (xmax - xmin) <= kLeanRange;                        // wave-uniform
      const float m = 0.5f * (xmax + xmin);
      const int n4 = (n + 3) >> 2;
      for (int p = n + lane; p < n4 * 4 + 4; p += 64) XS[p] = -INFINITY;
      WAVE_LDS_SYNC();
      GRP_STAMP(2);

      // 1b. ranks by counting (score descending, ties by index) (:483-500).
      int rk[IPL];
      // (BKT scratch: bucket-ordered scores behind XS inside recH -- 2 LpS floats, XS takes Lp + 8 of them, n + 32 <=
      //  Lp + 24 -- and the 192 counters in GS, which the records phase writes later; lists with LpS < 192 are short
      //  enough for the sweep)
      if (!BKT || LpS < 192 ||
          !wave_rank_by_bucket<IPL>(XS, n, lane, RKS, OCC, XS + Lp + 8, reinterpret_cast<int*>(GS), true, xmin, xmax))
        wave_rank_by_count(XS, n, lane, RKS, OCC);       // (common.h: all row chunks against each float4 of columns at once)
#pragma unroll
      for (int r = 0; r < IPL; ++r) rk[r] = lv[r] ? RKS[posr[r]] : 0;
      WAVE_LDS_SYNC();                                                       // the scratch is rewritten below
      GRP_STAMP(3);

      // 1c. grade order: repeatedly take the largest remaining label value; its items (in element order) become the
      // next segment, which starts on a 4-column boundary.  After kMaxRuns distinct values the rest forms one
      // unsorted tail segment.  sp = position in the sorted label sequence (ideal DCG), gp = padded grade position.
      int sp[IPL], gp[IPL];
      float rem[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) { rem[r] = labr[r]; sp[r] = 0; gp[r] = 0; }
      int pos = 0, ppos = 0, nseg = 0;
      bool tail = false;
      // graded relevance = small non-negative integers: the set of label values present comes from ONE wave-wide OR of
      // (1 << label) and the rounds below walk its bits from the top -- no wave maximum (a chain of seven dependent
      // DPP steps) per round.  Any other label set keeps the maximum search.
      bool small_int = true;
      unsigned present = 0u;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const bool ok = !lv[r] || (labr[r] < 32.0f && labr[r] == rintf(labr[r]));
        small_int = small_int && ok;
        if (lv[r] && ok) present |= 1u << (int)labr[r];
      }
      small_int = __ballot(!small_int) == 0ull;
      if (small_int) present = grp_wave_or(present);
      for (int it = 0; it <= kMaxRuns; ++it) {
        float v;
        if (small_int && it < kMaxRuns) {
          if (present == 0u) break;
          const int gb = 31 - __builtin_clz(present);
          present &= ~(1u << gb);
          v = (float)gb;
        } else {
          float mx = rem[0];
#pragma unroll
          for (int r = 1; r < IPL; ++r) mx = fmaxf(mx, rem[r]);
          v = wave_max_u(mx);
          if (v < 0.0f) break;
        }
        const bool last = it == kMaxRuns;                                    // everything that is left
        int c = 0;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const bool hit = last ? (rem[r] >= 0.0f) : (rem[r] == v);
          const unsigned long long bal = __ballot(hit);
          if (hit) {
            const int o = c + __popcll(bal & ((1ull << lane) - 1ull));
            sp[r] = pos + o; gp[r] = ppos + o; rem[r] = -2.0f;
          }
          c += __popcll(bal);
        }
        const int pend = (ppos + c + 3) & ~3;
        if (lane == 0) {
          H->seg_start[nseg] = ppos; H->seg_end[nseg] = pend;
          H->seg_gain[nseg] = pow2 ? grp_gain_pow2m1(v) : v;
        }
        pos += c; ppos = pend; ++nseg;
        tail = last;
      }

      // 1d. ideal DCG of the labels (:109-134): the sorted position of a pure-segment item is its grade position;
      // with an unsorted tail the gains are sorted (register bitonic network) instead.
      float inv_max_dcg = 1.0f;
      if (a.normalized) {
        float idcg = 0.f;
        if (!tail) {
#pragma unroll
          for (int r = 0; r < IPL; ++r) if (lv[r]) idcg += g[r] * Dlds[sp[r]];        // (an LDS gather instead of a global one: 8.4 -> 8.0 k cycles for this phase, the grade rounds are what it costs)
        } else {
          uint32_t sk[IPL];
#pragma unroll
          for (int r = 0; r < IPL; ++r) sk[r] = lv[r] ? float_to_ordered(g[r]) : 0u;
          wave_sort_desc_u32<IPL>(sk, lane);
#pragma unroll
          for (int r = 0; r < IPL; ++r) {
            const int e = lane + 64 * r;
            if (e < n) {
              const uint32_t o = sk[r];
              idcg += __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o) * Dlds[e];
            }
          }
        }
        idcg = wave_sum_u(idcg);
        inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
      }
      GRP_STAMP(4);

      // 1e. the image: everything is padding first (A = B = 0, no original index), then the records in grade order.
      for (int p = lane; p < LpS; p += 64) {
        recH[p] = make_float2(0.f, 0.f); recL[p] = make_float2(0.f, 0.f); GS[p] = 0.f;
        if (ITEMW) WS[p] = 0.f;
        if (p < Lp) CIS[p] = -1;
      }
      const int rscale = 4 * R;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        if (!lv[r]) continue;
        const float xv = xr[r];
        float Bv, Av;
        if (fast) {
          const float t_hi = xv - m;
          const float bb = t_hi - xv;
          const float t_lo = (xv - (t_hi - bb)) + (-m - bb);
          Bv = exp_df_hw(t_hi, t_lo);                           // e^(x - m), 1 ulp
          Av = __builtin_amdgcn_rcpf(Bv);                      // e^-(x - m): one more ulp, far inside the 1e-5 budget
        } else {
          Bv = xv; Av = 1.0f;                                                // slow path: the score itself / a validity flag
        }
        const float rb = __int_as_float(rk[r] * rscale);
        recH[gp[r]] = make_float2(Bv, rb);
        recL[gp[r]] = make_float2(Av, rb);
        GS[gp[r]] = g[r] * inv_max_dcg;
        if (ITEMW) WS[gp[r]] = wr[r];
        CIS[gp[r]] = lane + 64 * r;
      }
      if (lane < nseg) H->seg_gain[lane] *= inv_max_dcg;
      const int npass = (ppos + kGrpPassRows - 1) / kGrpPassRows;
      if (lane == 0) {
        H->n = n; H->np = ppos; H->nseg = nseg; H->flags = (fast ? 1 : 0) | (tail ? 2 : 0);
        H->npass = npass; H->lw = lw;
        if (npass == 0) {                                     // no valid item: nothing to sweep, the sums are zero
          if (a.list_loss && !a.sum.out) a.list_loss[b] = 0.f;
          if (AUX && a.nnz) a.nnz[b] = 0.f;
        }
      }
      if (npass == 0 && a.sum.out) grid_sum_contribute(a.sum, b, 0.f, lane);       // (wave-uniform: the list still counts as one entry)
    }
    // publish: release at workgroup scope (every LDS store above is ordered before the flag; the sweepers acquire it)
    WAVE_LDS_SYNC();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    // (the pass count travels in the flag word: state = 2 | npass << 8 -- one LDS read per poll; reading H->npass through
    //  a volatile pointer was a FLAT load with a vmcnt(0) wait on the polling path)
    if (lane == 0) __hip_atomic_store(&H->state, 2 | (b >= 0 ? H->npass << 8 : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  GRP_STAMP(5);

  // ---- 2. sweeps: (list, 32-row pass) work items from the per-list tickets of the lists that are ready.
  const int c = lane & 1;
  const uint32_t ubase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)Urep + 4u * (uint32_t)(lane & (R - 1));
  unsigned pending = (G >= 32) ? 0xffffffffu : ((1u << G) - 1u);            // lists this wave has not seen exhausted yet
  int l = wave < G ? wave : wave % G;                                        // start at the own list / spread the helpers
  int misses = 0;
  long spins = 0;
  while (pending) {
    if (!((pending >> l) & 1u)) { l = (l + 1 == G) ? 0 : l + 1; continue; }
    GrpHdr* H = GRP_HDR(l);
    int q = -1, npass = 0;
    {
      int st = 0;
      if (lane == 0) st = __hip_atomic_load(&H->state, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      st = __builtin_amdgcn_readfirstlane(st);
      if ((st & 0xff) == 2) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");   // the image and the header fields below were written before the flag
        npass = st >> 8;
        int t = 0;
        if (lane == 0) t = (npass > 0) ? atomicAdd(&H->next_pass, 1) : 0;
        q = __builtin_amdgcn_readfirstlane(t);
        if (q >= npass) { pending &= ~(1u << l); l = (l + 1 == G) ? 0 : l + 1; misses = 0; continue; }
      }
    }
    if (q < 0) {                                              // list l is still being built: look at the others first
      l = (l + 1 == G) ? 0 : l + 1;
      if (++misses >= G) {
        misses = 0;
        __builtin_amdgcn_s_sleep(8);
#ifdef TFR_PROFILE_STAMPS
        ++grp_polls;
#endif
        if (++spins > (1l << 22)) {
          // a builder always finishes; if one ever does not, do not spin forever AND do not leave the outputs of the
          // unswept lists uninitialised: poison them (NaN loss / gradient) so that the caller fails loudly
          if (lane == 0) {
            for (int pl = 0; pl < G; ++pl) {
              if (!((pending >> pl) & 1u)) continue;
              const int pb = GRP_HDR(pl)->b;
              if (pb < 0) continue;
              if (a.list_loss) a.list_loss[pb] = __builtin_nanf("");
              if (a.dlogits) a.dlogits[(size_t)pb * L] = __builtin_nanf("");
              if (a.row_loss) a.row_loss[(size_t)pb * L] = __builtin_nanf("");
            }
          }
          break;
        }
      }
      continue;
    }
    misses = 0;
    const float2* recH = GRP_RECH(l);
    const float2* recL = recH + LpS;
    const float* GS = reinterpret_cast<const float*>(recL + LpS);
    const float* WS = GS + LpS;
    const int* CIS = reinterpret_cast<const int*>(GS + LpS + (ITEMW ? LpS : 0));
    const int lb = __builtin_amdgcn_readfirstlane(H->b);
    const int np = __builtin_amdgcn_readfirstlane(H->np);
    const int nseg = __builtin_amdgcn_readfirstlane(H->nseg);
    const int flags = __builtin_amdgcn_readfirstlane(H->flags);
    const float llw = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, H->lw)));
    const bool fast = (flags & 1) != 0, tail = (flags & 2) != 0;
    const size_t base = (size_t)lb * L;
    // the segment table, one segment per lane (read back with v_readlane: no LDS round trip per segment)
    const int sl = lane < kGrpSegSlots ? lane : 0;
    const int seg_s = H->seg_start[sl], seg_e = H->seg_end[sl];
    const float seg_g = H->seg_gain[sl];
    {
#ifdef TFR_PROFILE_STAMPS
      ++grp_passes;
#endif
      const int row0 = q * kGrpPassRows;
      const int row = row0 + (lane >> 1);
      const float2 rh = recH[row], rl = recL[row];
      const float Gi = GS[row];
      const int ci = CIS[row];
      const float wi = ITEMW ? WS[row] : llw;                 // weight of the row item (preferred in the hi sweep)
      const uint32_t ri = (uint32_t)__float_as_int(rh.y);
      const int last_row = (row0 + kGrpPassRows < np ? row0 + kGrpPassRows : np) - 1;
      const int s_first = __popcll(__ballot(lane < nseg && seg_e <= row0));
      const int s_last = __popcll(__ballot(lane < nseg && seg_e <= last_row));
      float accL = 0.f, accG = 0.f, accG2 = 0.f, accW = 0.f, accNZ = 0.f;
      if (fast) {
        const float Ai = rl.x, Bi = rh.x;
        const int npure = tail ? nseg - 1 : nseg;             // pure grade segments; the unsorted tail (if any) is segment nseg - 1
        const bool in_tail = tail && s_last == nseg - 1;      // some row of this pass lies in the tail segment
        // row preferred: the lower grades behind it (pure segments s_first + 1 .. npure - 1 in ONE loop)
        if (!GRP_SKIP_HI)
        grp_hi_sweep<AUX>(recH, s_first + 1, npure, seg_s, seg_e, seg_g, c, Ai, Gi, ri, ubase, accL, accG, accW, accNZ);
        // column preferred: the higher grades in front of it (pure segments 0 .. s_last - 1)
        if (!GRP_SKIP_LO)
        grp_lo_sweep<ITEMW>(recL, WS, s_last < npure ? s_last : npure, seg_e, seg_g, c, Bi, Gi, ri, ubase, accG2);
        if (tail) {                                           // the tail segment: gain difference per pair
          const int s = __builtin_amdgcn_readlane(seg_s, nseg - 1), e = __builtin_amdgcn_readlane(seg_e, nseg - 1);
          GrpAcc r;
          grp_hi_loop<true, AUX>(recH, GS, s, (e - s) >> 2, c, Ai, Gi, ri, ubase, r);
          accL += r.l; accG += r.g;
          if (AUX) { accW += r.w; accNZ += r.nz; }
          if (in_tail) accG2 += grp_lo_loop<true, ITEMW>(recL, GS, WS, s, (e - s) >> 2, c, Bi, Gi, ri, ubase);
        }
      } else {
        // per-pair exponential, numerically safe for any score range (same algebra as pair_loss above): recH.x = x,
        // recL.x = validity flag.  The loss accumulates in log2 units like the fast path.
        const float xi = rh.x;
        for (int j = 2 * c; j < np; j += 4) {
          const float4 ch = *reinterpret_cast<const float4*>(&recH[j]);
          const float4 cl = *reinterpret_cast<const float4*>(&recL[j]);
          const float2 gj = *reinterpret_cast<const float2*>(&GS[j]);
          float2 wj = make_float2(1.f, 1.f);
          if (ITEMW) wj = *reinterpret_cast<const float2*>(&WS[j]);
#pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const float xj = hh ? ch.z : ch.x, rj = hh ? ch.w : ch.y, vj = hh ? cl.z : cl.x;
            const float Gj = hh ? gj.y : gj.x, wjj = hh ? wj.y : wj.x;
            const float u = grp_u(ri, rj, ubase) * vj;
            const float d0 = xi - xj;
            const float ex = __builtin_amdgcn_exp2f(-fabsf(d0) * kLog2e);
            const float w1 = 1.0f + ex;
            const float qq = __builtin_amdgcn_rcpf(w1);
            const float lg = __builtin_amdgcn_logf(w1) + fmaxf(-d0, 0.0f) * kLog2e;
            const float s_neg = (d0 >= 0.0f) ? ex * qq : qq;   // sigma(-d0)
            const float s_pos = (d0 >= 0.0f) ? qq : ex * qq;   // sigma(+d0)
            const float Whi = fmaxf(Gi - Gj, 0.0f) * u;
            float Wlo = fmaxf(Gj - Gi, 0.0f) * u;
            if (ITEMW) Wlo *= wjj;
            accL = __builtin_fmaf(Whi, lg, accL);
            accG = __builtin_fmaf(Whi, s_neg, accG);
            accG2 = __builtin_fmaf(Wlo, s_pos, accG2);
            if (AUX) { accW += Whi; accNZ += (Whi != 0.0f) ? 1.0f : 0.0f; }
          }
        }
      }
      // the two lanes of a row
      accL += TFR_DPP_F(0.f, accL, 0xB1, 0xf, 0xf, false);
      accG += TFR_DPP_F(0.f, accG, 0xB1, 0xf, 0xf, false);
      accG2 += TFR_DPP_F(0.f, accG2, 0xB1, 0xf, 0xf, false);
      if (AUX) { accW += TFR_DPP_F(0.f, accW, 0xB1, 0xf, 0xf, false); accNZ += TFR_DPP_F(0.f, accNZ, 0xB1, 0xf, 0xf, false); }
      float row_l = 0.f, row_nz = 0.f;
      if (c == 0 && ci >= 0) {
        row_l = accL * kLn2 * wi;
        if (a.row_loss) a.row_loss[base + ci] = row_l;
        if (AUX && a.row_weight) a.row_weight[base + ci] = accW * wi;
        const float g2 = ITEMW ? accG2 : accG2 * llw;
        if (a.dlogits) {                                      // (x / 1 = x: the ~13-instruction division only when T != 1, a scalar branch)
          const float dv = g2 - accG * wi;
          if (a.temperature == 1.0f) a.dlogits[base + ci] = dv;
          else a.dlogits[base + ci] = dv / a.temperature;
        }
        if (AUX) row_nz = (wi != 0.0f) ? accNZ : 0.0f;
      }
      const bool want_list = a.list_loss != nullptr, want_nnz = AUX && a.nnz != nullptr;
      if (want_list) { const float sm = wave_sum_u(row_l); if (lane == 0) H->pass_loss[q] = sm; }
      if (want_nnz) { const float sm = wave_sum_u(row_nz); if (lane == 0) H->pass_nnz[q] = sm; }
      if (want_list || want_nnz) {
        // the wave that completes the LAST pass of a list sums the pass partials, in pass order (deterministic):
        // every wave wrote its partial before its own ticket return below, and LDS serves the requests in order
        int done = 0;
        if (lane == 0) done = atomicAdd(&H->done_pass, 1);
        done = __builtin_amdgcn_readfirstlane(done);
        if (done + 1 == npass) {                              // (wave-uniform)
          float tl = 0.f;
          if (want_list) { for (int p2 = 0; p2 < npass; ++p2) tl += H->pass_loss[p2]; }      // every lane: the same sum, in pass order
          if (lane == 0) {
            if (want_list && !a.sum.out) a.list_loss[lb] = tl;
            if (want_nnz) { float t = 0.f; for (int p2 = 0; p2 < npass; ++p2) t += H->pass_nnz[p2]; a.nnz[lb] = t; }
          }
          // the reduced scalar of the launch (round 5): the helper stores list_loss[lb] itself, written through
          if (want_list && a.sum.out) grid_sum_contribute(a.sum, lb, tl, lane);
        }
      }
    }
  }
  GRP_STAMP(7);
#ifdef TFR_PROFILE_STAMPS
  if (lane == 0 && g_prof_buf_pw) {
    grp_t[8] = (unsigned long long)(builder ? GRP_HDR(wave)->n : 0);
    grp_t[9] = (unsigned long long)grp_passes;
    grp_t[10] = (unsigned long long)grp_polls;
    grp_t[11] = (unsigned long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) |          // HW_ID
                ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);     // XCC_ID
    for (int i = 0; i < 12; ++i) g_prof_buf_pw[((size_t)blockIdx.x * Wt + wave) * 12 + i] = grp_t[i];
  }
#endif
#undef GRP_HDR
#undef GRP_RECH
}

// Host side: geometry of the group kernel for a batch.  Returns 0 and fills (W, R, lds) when the kernel applies.
inline bool grp_geometry(int B, int L, bool itemw, int& G, int& Wt, int& R, size_t& lds) {
  const int Lp = grp_lp(L);
  if ((Lp + kGrpPassRows - 1) / kGrpPassRows > kGrpMaxPass) return false;
  static const int env_w = env_int("TFR_LAMBDARANK_WAVES", 0);          // lists (= builder waves) per workgroup
  static const int env_h = env_int("TFR_LAMBDARANK_HELPERS", -1);       // extra sweep-only waves per workgroup
  static const int env_r = env_int("TFR_LAMBDARANK_REP", 0);
  G = env_w > 0 ? env_w : (B >= 2048 ? 8 : (B >= 1024 ? 4 : (B >= 512 ? 2 : 1)));
  if (G > 16) G = 16;                                        // (one workgroup of 16 waves / 16 lists per CU: TFR_LAMBDARANK_WAVES=16)
  Wt = G + (env_h >= 0 ? env_h : G / 2);                      // (measured: 8 + 4 waves beat 8 + 0, 8 + 2 and 8 + 8 at B = 4096 and 16384)
  if (Wt > 16) Wt = 16;
  R = (env_r == 16 || env_r == 32 || env_r == 8) ? env_r : 32;
  // two workgroups per CU (160 KiB of LDS) when the full replication allows it, else the half table
  auto need = [&](int r) { return grp_table_bytes(L, r) + (size_t)G * grp_list_bytes(Lp, itemw); };
  if (env_r == 0 && G == 8 && need(32) > 80 * 1024 && need(16) <= 80 * 1024) R = 16;
  if (L * 4 * R > 65535) R = 16;                              // v_sad_u16 works on 16-bit ranks * 4 R
  if (L * 4 * R > 65535) return false;
  lds = need(R);
  return lds <= 160 * 1024;
}
