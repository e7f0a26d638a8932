// Pairwise logistic loss with optional (N)DCG lambda weights (LambdaRank /
// LambdaLoss), forward + backward fused in one launch, for gfx950.
//
// Reference behaviour restated (losses_impl.py): _compute_ranks :483-500,
// _pairwise_comparison :503-537, AbstractDCGLambdaWeight.pair_weights :255-279,
// DCGLambdaWeight._pair_rank_discount :334-369, inverse_max_dcg :109-134,
// _PairwiseLoss._compute_unreduced_loss_impl :871-884, _normalize_weights_impl
// :917-930, PairwiseLogisticLoss._pairwise_loss :936-940.  Backward: SURVEY.md
// Appendix B (ranks / lambda weights carry no gradient, :882).
//
// Design.  One workgroup per list.  The reference materialises ~35 [B, L, L]
// tensors; here the score-difference / |delta DCG| tile exists only in
// registers.  Per list: one LDS bitonic sort of packed keys for the ranks, one
// for the ideal DCG, the |D(m) - D(m+1)| rank-difference discounts tabulated
// once per list in LDS (L entries instead of L^2 evaluations), then one sweep
// over the pair matrix with each row split over C adjacent lanes.
#include "common.h"
#include "../../include/tfr_hip.h"

#include <stdlib.h>

using namespace tfr;

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

struct PwArgs {
  const float* logits; const float* labels; const uint8_t* mask;
  const float* item_weights; const float* list_weights;
  int lambda_kind; int lambda_sub; int topn; float smooth; int normalized; int gain_kind;
  const float* gains; const float* discount;
  int L; int Lp; int P; float temperature; int C; int kind;
  float* row_loss; float* row_weight; float* nnz; float* dlogits;
  float* list_loss;                      // nullable [B]: sum of the row losses of a list
  const int* order;                      // longest-first launch order (nullable)
  GridSum sum;                           // round 5: sum_b list_loss[b] from the same launch (tfr_pairwise_loss_sum_f32); out == NULL: off
  uint32_t tie_seed;                     // != 0: equal scores ranked in the hashed order of common.h tie_key15 (the workgroup kernel only)
};

__host__ __device__ inline size_t pw_wave_lds(int Lp) { return (size_t)Lp * (16 + 8 + 4 + 4 + 4) + 16; }

__host__ __device__ inline size_t pw_smem_bytes(int Lp, int P) {
  return 256 + (size_t)P * 8 + (size_t)Lp * 4 * 12 + (size_t)Lp * 2 + 32;
}

// Pairwise loss of a preference pair (losses_impl.py:936-958) as a function of the score
// difference d0 = x_row - x_col:  `loss` = loss(t = d0) (only used when the row item is the
// preferred one) and `sel` = -d loss / d t at t = +d0 when the row is preferred (`hi`) and at
// t = -d0 otherwise, so that row gradient += (w_lo - w_hi) * sel.
//   logistic      : loss = relu(-t) + log1p(exp(-|t|)),  -loss' = sigma(-t)
//   hinge         : loss = relu(1 - t),                  -loss' = 1[t < 1]        (relu'(0) = 0)
//   soft zero-one : loss = sigma(-t),                    -loss' = sigma(t) sigma(-t)
__device__ __forceinline__ void pair_loss(const int kind, const float d0, const bool hi, float& loss, float& sel) {
  if ((kind & 0xff) == TFR_PAIR_HINGE) {
    loss = fmaxf(1.0f - d0, 0.0f);
    const float t = hi ? d0 : -d0;
    sel = (t < 1.0f) ? 1.0f : 0.0f;
    return;
  }
  // TFR_PAIR_TIED_ZERO (with TFR_PAIR_LOGISTIC only): the reference's gradient at EXACTLY tied scores.  Its formula
  // relu(-t) + log1p(exp(-|t|)) differentiates to 0 at t = 0 under TF autodiff (relu'(0) = 0, d|t|/dt = sign(0) = 0), not to
  // the analytic -sigma(0) = -1/2: a tied pair contributes its loss log 2 and NO gradient.
  const bool tied_zero = (kind & TFR_PAIR_TIED_ZERO) != 0 && d0 == 0.0f;
  const float e = __builtin_amdgcn_exp2f(-fabsf(d0) * kLog2e);     // exp(-|d|)
  const float w1 = 1.0f + e;
  const float q = __builtin_amdgcn_rcpf(w1);
  const float eq = e * q;
  const bool pos = d0 >= 0.0f;
  if ((kind & 0xff) == TFR_PAIR_SOFT_ZERO_ONE) {
    loss = pos ? eq : q;                                           // sigma(-d0)
    sel = q * eq;
    return;
  }
  // relu(-d) + log1p(exp(-|d|)); fl(1 + e) costs <= 6e-8 absolute per pair, far inside 1e-5
  loss = __builtin_fmaf(__builtin_amdgcn_logf(w1), kLn2, fmaxf(-d0, 0.0f));
  sel = (pos == hi) ? eq : q;                                      // hi: sigma(-d0), lo: sigma(+d0)
  if (tied_zero) sel = 0.0f;
}

// Pair weight of the GENERIC lambda path.  sub 0: DCGLambdaWeight (losses_impl.py:299-369),
// 1: DCGLambdaWeightV2 (:372-394), 2: YetiDCGLambdaWeight (:397-407), 3: PrecisionLambdaWeight
// (:410-454).  ai/aj = 1-based ranks, dpi/dpj = D(rank) (sub 0: zeroed beyond topn), u = |D(m) - D(m+1)|
// at m = |ai - aj|, pg = |gain_i - gain_j| of a label-valid pair (else 0).
#define TFR_SUB_DCG 0
#define TFR_SUB_DCG_V2 1
#define TFR_SUB_YETI 2
#define TFR_SUB_PRECISION 3
__device__ __forceinline__ float lambda_generic_weight(const int sub, const float ai, const float aj, const float dpi,
                                                       const float dpj, const float u, const float pg,
                                                       const float ftopn, const float one_minus_s, const float smooth,
                                                       const float fL) {
  if (sub == TFR_SUB_DCG) {
    const bool in_top = (ai <= ftopn) || (aj <= ftopn);
    const float v = fabsf(dpi - dpj);
    float pd = one_minus_s * u + smooth * v;
    pd = in_top ? pd : 0.0f;
    return (pg * pd) * fL;
  }
  if (sub == TFR_SUB_PRECISION) return ((ai <= ftopn) != (aj <= ftopn)) ? pg : 0.0f;
  const float mx = fmaxf(ai, aj);
  const float dmax = (ai > aj) ? dpi : dpj;
  const float mult = (mx > ftopn) ? (1.0f / (1.0f - dmax)) : 1.0f;
  float pd = (ai != aj) ? u * mult : 0.0f;
  if (sub == TFR_SUB_YETI) pd = (fabsf(ai - aj) == 1.0f) ? pd : 0.0f;
  return (pg * pd) * fL;
}

// One (row i, column j) term.  rec = (x, raw label, gain, item weight); rk = signed
// rank as float (negative: label-invalid item), dp = D'(rank).
template <int LAMBDA, bool GENERIC>
__device__ __forceinline__ void pair_term(const float4 ri, const float rki, const float dpi,
                                          const float4 rj, const float rkj, const float dpj,
                                          const float* __restrict__ U, const float ftopn,
                                          const float one_minus_s, const float smooth, const float fL, const int kind,
                                          const int sub, float& acc_loss, float& acc_w, float& acc_nz, float& acc_g) {
  float wl = 1.0f;
  if (LAMBDA == TFR_LAMBDA_DCG) {
    const float ai = fabsf(rki), aj = fabsf(rkj);
    const int dr = (int)fabsf(ai - aj);
    const float u = U[dr];                                   // U[0] = 0
    if (GENERIC) {
      const float pg = (rki > 0.0f && rkj > 0.0f) ? fabsf(ri.z - rj.z) : 0.0f;
      wl = lambda_generic_weight(sub, ai, aj, dpi, dpj, u, pg, ftopn, one_minus_s, smooth, fL);
    } else {
      wl = (fabsf(ri.z - rj.z) * u) * fL;
    }
  } else if (LAMBDA == TFR_LAMBDA_LABELDIFF) {
    wl = fabsf(ri.y - rj.y);
  }
  const float d0 = ri.x - rj.x;
  if (kind == TFR_PAIR_MSE) {             // :961-998: every ordered pair of two distinct mask-valid items
    const bool pv = (ri.y == ri.y) && (rj.y == rj.y) && (fabsf(rki) != fabsf(rkj));
    const float e = pv ? d0 - (ri.y - rj.y) : 0.0f;      // padding records carry NaN labels
    const float wa = pv ? wl * ri.w : 0.0f, wb = pv ? wl * rj.w : 0.0f;
    acc_loss = __builtin_fmaf(wa, e * e, acc_loss);
    acc_w += wa;
    acc_nz += (wa != 0.0f) ? 1.0f : 0.0f;
    acc_g = __builtin_fmaf(wa + wb, 2.0f * e, acc_g);
    return;
  }
  const bool hi = ri.y > rj.y;            // row item preferred
  const bool lo = rj.y > ri.y;            // column item preferred
  float loss, sel;
  pair_loss(kind, d0, hi, loss, sel);
  const float ww_hi = hi ? wl * ri.w : 0.0f;
  const float ww_lo = lo ? wl * rj.w : 0.0f;
  acc_loss = __builtin_fmaf(ww_hi, loss, acc_loss);
  acc_w += ww_hi;
  acc_nz += (ww_hi != 0.0f) ? 1.0f : 0.0f;
  acc_g = __builtin_fmaf(ww_lo - ww_hi, sel, acc_g);
}

// The same term for the rank-ordered wave kernel: ranks are positions (ai, aj), the
// rank-difference discount `u` was fetched by the caller (already multiplied by list_size when
// !GENERIC), q = (D'(rank), label-valid flag).  AUX: the caller wants sum of weights / pair
// counts; ITEMW: per-item weights differ (otherwise the common list weight is applied once
// per row by the caller).  18 VALU + 3 transcendental instructions per pair in the lean form.
template <int LAMBDA, bool GENERIC, bool AUX, bool ITEMW>
__device__ __forceinline__ void pair_term_ranked(const float4 ri, const float2 qi, const float ai, const float4 rj,
                                                 const float2 qj, const float aj, const float u, const float ftopn,
                                                 const float one_minus_s, const float smooth, const float fL,
                                                 const int kind, const int sub, float& acc_loss, float& acc_w, float& acc_nz,
                                                 float& acc_g) {
  float wl = 1.0f;
  if (LAMBDA == TFR_LAMBDA_DCG) {
    if (GENERIC) {
      const float pg = (qi.y > 0.0f && qj.y > 0.0f) ? fabsf(ri.z - rj.z) : 0.0f;
      wl = lambda_generic_weight(sub, ai, aj, qi.x, qj.x, u, pg, ftopn, one_minus_s, smooth, fL);
    } else {
      wl = fabsf(ri.z - rj.z) * u;
    }
  } else if (LAMBDA == TFR_LAMBDA_LABELDIFF) {
    wl = fabsf(ri.y - rj.y);
  }
  const float d0 = ri.x - rj.x;
  if (kind == TFR_PAIR_MSE) {             // :961-998 (only reached in the AUX && ITEMW variants)
    const bool pv = (ri.y == ri.y) && (rj.y == rj.y) && (ai != aj);
    const float e = pv ? d0 - (ri.y - rj.y) : 0.0f;      // padding records carry NaN labels
    const float wa = pv ? wl * ri.w : 0.0f, wb = pv ? wl * rj.w : 0.0f;
    acc_loss = __builtin_fmaf(wa, e * e, acc_loss);
    acc_w += wa;
    acc_nz += (wa != 0.0f) ? 1.0f : 0.0f;
    acc_g = __builtin_fmaf(wa + wb, 2.0f * e, acc_g);
    return;
  }
  const bool hi = ri.y > rj.y;            // row item preferred
  const bool lo = rj.y > ri.y;            // column item preferred
  float loss, sel;
  pair_loss(kind, d0, hi, loss, sel);
  float ww_hi, ww_lo;
  if (ITEMW) { ww_hi = hi ? wl * ri.w : 0.0f; ww_lo = lo ? wl * rj.w : 0.0f; }
  else       { ww_hi = hi ? wl : 0.0f;        ww_lo = lo ? wl : 0.0f; }
  acc_loss = __builtin_fmaf(ww_hi, loss, acc_loss);
  if (AUX) {
    acc_w += ww_hi;
    acc_nz += (ww_hi != 0.0f) ? 1.0f : 0.0f;
  }
  acc_g = __builtin_fmaf(ww_lo - ww_hi, sel, acc_g);
}

template <int LAMBDA, bool GENERIC>
__global__ void pairwise_logistic_kernel(const PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);            // [32]
  int* wc = reinterpret_cast<int*>(smem_raw + 128);           // [16]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 256);  // [P]
  float* fbase = reinterpret_cast<float*>(keys + a.P);
  const int Lp = a.Lp;
  float4* rec0 = reinterpret_cast<float4*>(fbase);            // [Lp] (x, raw label, gain, item weight)
  float2* rec1 = reinterpret_cast<float2*>(fbase + 4 * Lp);   // [Lp] (D'(rank), rank)
  int* CI = reinterpret_cast<int*>(fbase + 6 * Lp);           // [Lp] compact -> original
  float* U = fbase + 7 * Lp;                                  // [Lp] |D(m) - D(m+1)|, m = index
  float* Xr = fbase + 8 * Lp;                                 // [Lp] original order scratch
  float* Gr = fbase + 9 * Lp;                                 // [Lp] gain, original order
  float* Wr = fbase + 10 * Lp;                                // [Lp] item weight, original order
  int* Rr = reinterpret_cast<int*>(fbase + 11 * Lp);          // [Lp] rank, original order
  uint8_t* MV = reinterpret_cast<uint8_t*>(fbase + 12 * Lp);  // [Lp] mask-valid
  uint8_t* LV = MV + Lp;                                      // [Lp] label-valid

  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = T >> 6;
  const int b = a.order ? a.order[blockIdx.x] : blockIdx.x, L = a.L, P = a.P;
  const size_t base = (size_t)b * L;
  const int topn = (a.topn <= 0 || a.topn > L) ? L : a.topn;
  const float lw = a.list_weights ? a.list_weights[b] : 1.0f;
  const bool dcg_lambda = a.lambda_kind == TFR_LAMBDA_DCG;

  // ---- 1. load; ranks by sorting (valid first, score desc)   (:483-500)
  for (int i = tid; i < P; i += T) {
    uint64_t key = 0;
    if (i < L) {
      const float lab = a.labels[base + i];
      const float x = a.logits[base + i] / a.temperature;
      const bool lv = lab >= 0.0f;
      const bool mv = a.mask ? (a.mask[base + i] != 0) : lv;
      const float labc = lv ? lab : 0.0f;
      float g = 0.f;
      if (dcg_lambda) {
        if (a.gain_kind == TFR_GAIN_CUSTOM) g = a.gains[base + i];
        else if (a.gain_kind == TFR_GAIN_POW2M1) g = gain_pow2m1(labc);
        else g = labc;
      }
      float w = a.item_weights ? a.item_weights[base + i] : 1.0f;
      w = lv ? (w * lw) : 0.0f;                                  // :917-930
      Xr[i] = x; Gr[i] = g; Wr[i] = w; MV[i] = mv; LV[i] = lv;
      key = make_sort_key(mv, x, tie_key15(a.tie_seed, (uint32_t)b, (uint32_t)i), i);
    }
    keys[i] = key;
  }
  block_bitonic_sort_desc(keys, P);
  for (int p = tid; p < L; p += T) Rr[sort_key_index(keys[p])] = p + 1;
  __syncthreads();

  // ---- 2. DCG lambda: optional normalisation by the ideal DCG@topn (:260-266, :109-134)
  float inv_max_dcg = 1.0f;
  if (dcg_lambda) {
    for (int m = tid; m < Lp; m += T)
      U[m] = (m >= 1 && m < L) ? fabsf(a.discount[m - 1] - a.discount[m]) : 0.0f;
    if (a.normalized) {
      for (int i = tid; i < P; i += T) {
        uint64_t key = 0;
        if (i < L) {
          const float lab = a.labels[base + i];
          const float labc = (lab >= 0.0f) ? lab : 0.0f;
          key = ((uint64_t)float_to_ordered(labc) << 32) | (uint64_t)__float_as_uint(Gr[i]);
        }
        keys[i] = key;
      }
      block_bitonic_sort_desc(keys, P);
      float idcg = 0.f;
      for (int p = tid; p < topn; p += T)
        idcg += __uint_as_float((uint32_t)(keys[p] & 0xffffffffull)) * a.discount[p];
      idcg = block_sum(idcg, red);
      inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
    }
  }
  __syncthreads();

  // ---- 3. compaction of the mask-valid items into AoS records.
  int n = 0;
  for (int i0 = 0; i0 < L; i0 += T) {
    const int i = i0 + tid;
    const bool v = (i < L) && (MV[i] != 0);
    const unsigned long long bal = __ballot(v);
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) wc[wid] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const int c = wc[w]; woff += (w < wid) ? c : 0; tot += c; }
    if (v) {
      const int pos = n + woff + lane_prefix;
      const int r = Rr[i];
      // label-invalid but mask-valid items carry gain 0 and a "no lambda pair" flag (NaN-free):
      const float g = LV[i] ? Gr[i] * (a.normalized ? inv_max_dcg : 1.0f) : 0.0f;
      rec0[pos] = make_float4(Xr[i], a.labels[base + i], g, Wr[i]);
      const float dprime = (dcg_lambda && (r <= topn || a.lambda_sub != TFR_SUB_DCG)) ? a.discount[r - 1] : 0.0f;
      rec1[pos] = make_float2(dprime, LV[i] ? (float)r : -(float)r);
      CI[pos] = i;
    }
    n += tot;
  }
  __syncthreads();

  // ---- 4. pair sweep.  Row i (C lanes) against every column j (two columns per trip).
  const int C = a.C;
  const int iters = ((n + C - 1) / C + 1) & ~1;                    // uniform, even trip count
  for (int p = n + tid; p < iters * C; p += T) {                   // neutral padding records
    rec0[p] = make_float4(0.f, NAN, 0.f, 0.f);
    rec1[p] = make_float2(0.f, 1.0f);
  }
  __syncthreads();
  const int rows_per_pass = T / C;
  const int c = tid % C, rsub = tid / C;
  const float fL = (float)L;
  const float one_minus_s = 1.0f - a.smooth;
  float nnz_local = 0.f, list_local = 0.f;
  for (int i = tid; i < L; i += T) {
    if (!MV[i]) {
      if (a.row_loss) a.row_loss[base + i] = 0.f;
      if (a.row_weight) a.row_weight[base + i] = 0.f;
      if (a.dlogits) a.dlogits[base + i] = 0.f;
    }
  }
  const float ftopn = (float)topn;
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    if (!__any(active)) continue;
    float4 ri = rec0[active ? row : 0];
    const float2 qi = rec1[active ? row : 0];
    if (!active) ri.y = NAN;
    float acc_loss = 0.f, acc_w = 0.f, acc_nz = 0.f, acc_g = 0.f;
    for (int it = 0; it < iters; it += 2) {
      const int j0 = c + it * C, j1 = j0 + C;
      const float4 r0 = rec0[j0], r1 = rec0[j1];
      const float2 q0 = rec1[j0], q1 = rec1[j1];
      pair_term<LAMBDA, GENERIC>(ri, qi.y, qi.x, r0, q0.y, q0.x, U, ftopn, one_minus_s, a.smooth, fL, a.kind,
                                 a.lambda_sub, acc_loss, acc_w, acc_nz, acc_g);
      pair_term<LAMBDA, GENERIC>(ri, qi.y, qi.x, r1, q1.y, q1.x, U, ftopn, one_minus_s, a.smooth, fL, a.kind,
                                 a.lambda_sub, acc_loss, acc_w, acc_nz, acc_g);
    }
    for (int o = 1; o < C; o <<= 1) {
      acc_loss += __shfl_xor(acc_loss, o, 64);
      acc_w += __shfl_xor(acc_w, o, 64);
      acc_nz += __shfl_xor(acc_nz, o, 64);
      acc_g += __shfl_xor(acc_g, o, 64);
    }
    if (active && c == 0) {
      const int oi = CI[row];
      if (a.row_loss) a.row_loss[base + oi] = acc_loss;
      if (a.row_weight) a.row_weight[base + oi] = acc_w;
      if (a.dlogits) a.dlogits[base + oi] = acc_g / a.temperature;
      nnz_local += acc_nz;
      list_local += acc_loss;
    }
  }
  nnz_local = block_sum(nnz_local, red);
  if (tid == 0 && a.nnz) a.nnz[b] = nnz_local;
  if (a.list_loss) {
    list_local = block_sum(list_local, red);
    if (tid == 0) a.list_loss[b] = list_local;
    if (a.sum.out && tid < 64) grid_sum_contribute(a.sum, b, list_local, tid);     // wave 0; the value is block-uniform
  }
}


// ===========================================================================
// Wave-per-list variant (L <= 64 * IPL): one wavefront owns a list, no
// workgroup barriers; ranks come from a counting sweep over the packed LDS
// records, the ideal DCG from an in-register bitonic sort, and the pair body is
// branch-free with single-instruction transcendentals (v_exp / v_rcp / v_log;
// the libm log1pf the first version called cost ~120 VALU instructions per pair).
// ===========================================================================

template <int IPL>
__device__ __forceinline__ void wave_sort_desc_u32(uint32_t (&a)[IPL], int lane) {
  constexpr int N = 64 * IPL;
#pragma unroll
  for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int jr = j >> 6;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          if ((r & jr) == 0) {
            const int r2 = r | jr;
            const bool desc = (((r << 6) & kk) == 0);
            const uint32_t hi = a[r] > a[r2] ? a[r] : a[r2];
            const uint32_t lo = a[r] > a[r2] ? a[r2] : a[r];
            a[r] = desc ? hi : lo;
            a[r2] = desc ? lo : hi;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const uint32_t p = (uint32_t)__shfl_xor((int)a[r], j, 64);
          const int e = lane | (r << 6);
          const bool desc = ((e & kk) == 0);
          const bool lower = ((lane & j) == 0);
          const uint32_t mx = a[r] > p ? a[r] : p;
          const uint32_t mn = a[r] > p ? p : a[r];
          a[r] = (lower == desc) ? mx : mn;
        }
      }
    }
  }
}

// Developer aid (never in the product build): see approx_ndcg.hip / tools/phase_profile.py.
#ifdef TFR_PROFILE_STAMPS
__device__ unsigned long long* g_prof_buf_pw = nullptr;
// group kernel, counter runs (tools/phase_profile.py group_counts): 1 .. 5 = every builder returns after that build phase
// (nothing is published, nobody sweeps), 6 = hi sweeps only, 7 = lo sweeps only, 0 = the whole kernel
__device__ int g_prof_stop_pw = 0;
#define PW_STAMP(i) do { if (lane == 0 && wave == 0) prof_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define PW_STAMP(i) do { } while (0)
#endif

// End of a wave kernel: per-list pair count (AUX) and loss sum over the S waves that shared the list.  The only
// workgroup-level exchange: 2 * S floats through LDS, summed in wave order (deterministic).
template <bool AUX>
__device__ __forceinline__ void pw_finish(const PwArgs& a, int b, int lane, int wave, int S, float nnz_local,
                                          float list_local, float* slots) {
  const bool want_list = a.list_loss != nullptr;              // kernel argument: wave-uniform
  if (!AUX && !want_list) return;
  const float nz = AUX ? wave_sum_u(nnz_local) : 0.f;
  const float ls = want_list ? wave_sum_u(list_local) : 0.f;
  if (S == 1) {
    if (lane == 0) {
      if (AUX && a.nnz) a.nnz[b] = nz;
      if (want_list) a.list_loss[b] = ls;
    }
    if (want_list && a.sum.out) grid_sum_contribute(a.sum, b, ls, lane);
    return;
  }
  if (lane == 0) { slots[wave] = nz; slots[S + wave] = ls; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.f, u = 0.f;
    for (int w2 = 0; w2 < S; ++w2) { t += slots[w2]; u += slots[S + w2]; }
    if (AUX && a.nnz) a.nnz[b] = t;
    if (want_list) a.list_loss[b] = u;
  }
  if (want_list && a.sum.out && wave == 0) {                  // the same sum, in the same order, in every lane of wave 0
    float u = 0.f;
    for (int w2 = 0; w2 < S; ++w2) u += slots[S + w2];
    grid_sum_contribute(a.sum, b, u, lane);
  }
}

// KIND: TFR_PAIR_LOGISTIC at compile time (the hot configuration), or -1 = a.kind at run time.
template <int IPL, int LAMBDA, bool GENERIC, bool AUX, bool ITEMW, int KIND>
__global__ __launch_bounds__(256) void pairwise_wave_kernel(const PwArgs a) {
  // blockDim.x = 64 * S: S wavefronts share one list when the batch alone cannot fill the
  // chip.  Every wave repeats the (cheap) per-list set-up in its OWN LDS slice -- no workgroup
  // barrier anywhere -- and sweeps rows wave, wave + S, ... of the pair matrix.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  const int wave = threadIdx.x >> 6, S = blockDim.x >> 6;
  unsigned char* smem_raw = smem_all + (size_t)wave * pw_wave_lds(a.Lp);
  float* nz_slot = reinterpret_cast<float*>(smem_all + (size_t)S * pw_wave_lds(a.Lp));   // [S]
  const int Lp = a.Lp;
  // Per-item data stay in REGISTERS (item e = lane + 64 r) until their rank is known and are then
  // written once, in rank order: 36 bytes of LDS per item -> 7+ list-waves per SIMD.
  float4* recS = reinterpret_cast<float4*>(smem_raw);                  // [Lp] (x, raw label, gain, weight) in RANK order
  float2* auxS = reinterpret_cast<float2*>(recS + Lp);                 // [Lp] (D'(rank), label-valid flag), rank order
  float* XS = reinterpret_cast<float*>(auxS + Lp);                     // [Lp] compact x (pad -inf) for the rank count
  float* U = XS + Lp;                                                  // [Lp] |D(m) - D(m+1)|
  int* CIS = reinterpret_cast<int*>(U + Lp);                           // [Lp] rank position -> original index
  float2* rec1 = auxS;                                                 // scratch alias for the custom-gain ideal DCG
  const int lane = threadIdx.x & 63, b = a.order ? a.order[blockIdx.x] : blockIdx.x, L = a.L;
  const size_t base = (size_t)b * L;
  const int topn = (a.topn <= 0 || a.topn > L) ? L : a.topn;
  const float lw = a.list_weights ? a.list_weights[b] : 1.0f;
#ifdef TFR_PROFILE_STAMPS
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PW_STAMP(0);

  // ---- 1. load; gains; compaction of the mask-valid items.
  float g[IPL], xr[IPL], labr[IPL], wr[IPL];
  int posr[IPL];
  bool lv[IPL], mv[IPL];
  int n = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = lane + 64 * r;
    g[r] = 0.f; lv[r] = false; mv[r] = false;
    float x = 0.f, lab = 0.f, w = 0.f;
    if (e < L) {
      lab = a.labels[base + e];
      x = a.logits[base + e] / a.temperature;
      lv[r] = lab >= 0.0f;
      mv[r] = a.mask ? (a.mask[base + e] != 0) : lv[r];
      const float labc = lv[r] ? lab : 0.0f;
      if (LAMBDA == TFR_LAMBDA_DCG) {
        if (a.gain_kind == TFR_GAIN_CUSTOM) g[r] = a.gains[base + e];
        else if (a.gain_kind == TFR_GAIN_POW2M1) g[r] = gain_pow2m1(labc);
        else g[r] = labc;
      }
      w = a.item_weights ? a.item_weights[base + e] : 1.0f;
      w = lv[r] ? (w * lw) : 0.0f;
      if (!mv[r]) {
        if (a.row_loss) a.row_loss[base + e] = 0.f;
        if (a.row_weight) a.row_weight[base + e] = 0.f;
        if (a.dlogits) a.dlogits[base + e] = 0.f;
      }
    }
    const unsigned long long bal = __ballot(mv[r]);
    xr[r] = x; labr[r] = lab; wr[r] = w;
    posr[r] = n + __popcll(bal & ((1ull << lane) - 1ull));     // compaction position (ties: lower first)
    if (mv[r]) XS[posr[r]] = x;
    n += __popcll(bal);
  }

  PW_STAMP(1);
  // ---- 2. ideal DCG@topn of the cleaned labels (:109-134), in registers.
  float inv_max_dcg = 1.0f;
  if (LAMBDA == TFR_LAMBDA_DCG) {
    for (int m = lane; m < Lp; m += 64) {
      const float um = (m >= 1 && m < L) ? fabsf(a.discount[m - 1] - a.discount[m]) : 0.0f;
      U[m] = GENERIC ? um : um * (float)L;              // the final x list_size (:278) folded in
    }
    if (a.normalized) {
      // gains of monotone gain functions sort like the labels; a custom gain_fn
      // sorts by label and carries the gain (two-key compare through a u32 pair).
      uint32_t sk[IPL];
      float idcg = 0.f;
      if (a.gain_kind != TFR_GAIN_CUSTOM) {
        // built-in gains are non-negative and graded labels have a handful of distinct values:
        // run-length form of sum_p sorted(g)[p] * D(p+1), p < topn; bitonic sort as the fallback.
        float gc[IPL], tbl[IPL];
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int e = lane + 64 * r;
          gc[r] = (e < L && lv[r]) ? g[r] : 0.0f;
          tbl[r] = (e < topn) ? a.discount[e] : 0.0f;
        }
        if (!wave_sorted_dot_runs<IPL>(gc, tbl, lane, L, 8, idcg)) {
#pragma unroll
          for (int r = 0; r < IPL; ++r) sk[r] = (lane + 64 * r < L) ? float_to_ordered(g[r]) : 0u;
          wave_sort_desc_u32<IPL>(sk, lane);
          idcg = 0.f;
#pragma unroll
          for (int r = 0; r < IPL; ++r) {
            const int e = lane + 64 * r;
            if (e < topn) {
              const uint32_t o = sk[r];
              const float gv = __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o);
              idcg += gv * a.discount[e];
            }
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) idcg += __shfl_xor(idcg, o, 64);
        }
        inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
      } else {
        // rank of every label by counting (ties: lower index first), then scatter by rank.
        float* sl = reinterpret_cast<float*>(rec1);          // [2*Lp] scratch (rec1 is filled in step 3)
        WAVE_LDS_SYNC();
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int e = lane + 64 * r;
          if (e < L) { const float lab = a.labels[base + e]; sl[e] = lab >= 0.0f ? lab : 0.0f; }
        }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const int e = lane + 64 * r;
          if (e < L) {
            const float me = sl[e];
            int cnt = 0;
            for (int j = 0; j < L; ++j) {
              const float o = sl[j];
              cnt += (o > me || (o == me && j < e)) ? 1 : 0;
            }
            if (cnt < topn) idcg += g[r] * a.discount[cnt];
          }
        }
        WAVE_LDS_SYNC();
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) idcg += __shfl_xor(idcg, o, 64);
        inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
      }
    }
  }
  WAVE_LDS_SYNC();

  PW_STAMP(2);
  // ---- 3. ranks by counting (valid first, score desc, ties by index) (:483-500); the
  // records are then re-homed in RANK order so that in the sweep the rank of a column is its
  // position and the |rank_i - rank_j| discount is an affine (data independent) LDS address.
  const int n4 = (n + 3) >> 2;
  for (int p = n + lane; p < n4 * 4 + 4 && p < Lp; p += 64) XS[p] = -INFINITY;
  WAVE_LDS_SYNC();
  int cntr[IPL];
  {
    int* RKS = reinterpret_cast<int*>(auxS);                 // scratch (auxS / CIS are filled by the re-homing below)
    int* OCC = CIS;
    wave_rank_by_count(XS, n, lane, RKS, OCC);
#pragma unroll
    for (int r = 0; r < IPL; ++r) cntr[r] = mv[r] ? RKS[posr[r]] : 0;
    WAVE_LDS_SYNC();
  }
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    if (!mv[r]) continue;
    const float xi = xr[r];
    const int cnt = cntr[r];
    float dprime = 0.f;
    float gz = lv[r] ? g[r] : 0.0f;
    if (LAMBDA == TFR_LAMBDA_DCG) {
      dprime = (cnt < topn || a.lambda_sub != TFR_SUB_DCG) ? a.discount[cnt] : 0.0f;
      if (a.normalized) gz *= inv_max_dcg;
    }
    recS[cnt] = make_float4(xi, labr[r], gz, wr[r]);
    auxS[cnt] = make_float2(dprime, lv[r] ? 1.0f : 0.0f);
    CIS[cnt] = lane + 64 * r;
  }
  const int C = a.C;
  const int npad = ((n + 2 * C - 1) / (2 * C)) * (2 * C);            // two columns per trip per lane
  for (int p = n + lane; p < npad && p < Lp; p += 64) {               // neutral padding: label NaN -> no pair
    recS[p] = make_float4(0.f, NAN, 0.f, 0.f);
    auxS[p] = make_float2(0.f, 0.0f);
  }
  WAVE_LDS_SYNC();

  PW_STAMP(3);
  // ---- 4. pair sweep: row = C adjacent lanes, 64/C rows per pass, two columns per trip.
  const int rows_per_pass = 64 / C;
  const int c = lane % C, rsub = lane / C;
  const int trips = npad / (2 * C);
  const float fL = (float)L;
  const float one_minus_s = 1.0f - a.smooth;
  const float ftopn = (float)topn;
  const int kind = (KIND >= 0) ? KIND : a.kind;
  float nnz_local = 0.f, list_local = 0.f;
  for (int row0 = wave * rows_per_pass; row0 < n; row0 += S * rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    float4 ri = recS[active ? row : 0];
    const float2 qi = auxS[active ? row : 0];
    if (!active) ri.y = NAN;                                   // compares false: contributes nothing
    float acc_loss = 0.f, acc_w = 0.f, acc_nz = 0.f, acc_g = 0.f;
    for (int it = 0; it < trips; ++it) {                      // uniform trip count: scalar loop control
      const int j0 = c + it * 2 * C;
      const int j1 = j0 + C;
      const float4 r0 = recS[j0], r1 = recS[j1];
      float u0 = 0.f, u1 = 0.f;
      if (LAMBDA == TFR_LAMBDA_DCG) { const int rc = active ? row : 0; u0 = U[abs(rc - j0)]; u1 = U[abs(rc - j1)]; }
      float2 q0 = make_float2(0.f, 1.f), q1 = q0;
      if (GENERIC) { q0 = auxS[j0]; q1 = auxS[j1]; }
      pair_term_ranked<LAMBDA, GENERIC, AUX, ITEMW>(ri, qi, (float)(row + 1), r0, q0, (float)(j0 + 1), u0, ftopn,
                                                    one_minus_s, a.smooth, fL, kind, a.lambda_sub, acc_loss, acc_w, acc_nz, acc_g);
      pair_term_ranked<LAMBDA, GENERIC, AUX, ITEMW>(ri, qi, (float)(row + 1), r1, q1, (float)(j1 + 1), u1, ftopn,
                                                    one_minus_s, a.smooth, fL, kind, a.lambda_sub, acc_loss, acc_w, acc_nz, acc_g);
    }
    for (int o = 1; o < C; o <<= 1) {
      acc_loss += __shfl_xor(acc_loss, o, 64);
      acc_w += __shfl_xor(acc_w, o, 64);
      acc_nz += __shfl_xor(acc_nz, o, 64);
      acc_g += __shfl_xor(acc_g, o, 64);
    }
    if (active && c == 0) {
      const int oi = CIS[row];
      if (!ITEMW) { acc_loss *= lw; acc_w *= lw; acc_g *= lw; if (lw == 0.0f) acc_nz = 0.f; }
      if (a.row_loss) a.row_loss[base + oi] = acc_loss;
      if (AUX && a.row_weight) a.row_weight[base + oi] = acc_w;
      if (a.dlogits) a.dlogits[base + oi] = acc_g / a.temperature;
      nnz_local += acc_nz;
      list_local += acc_loss;
    }
  }
  PW_STAMP(4);
#ifdef TFR_PROFILE_STAMPS
  if (lane == 0 && wave == 0 && g_prof_buf_pw) {
    prof_t[7] = (unsigned long long)n;
    for (int i = 0; i < 8; ++i) g_prof_buf_pw[(size_t)b * 8 + i] = prof_t[i];
  }
#endif
  pw_finish<AUX>(a, b, lane, wave, S, nnz_local, list_local, nz_slot);
}

// ===========================================================================
// LambdaRank fast path: PairwiseLogisticLoss x DCGLambdaWeight (smooth_fraction 0, no topn, built-in
// monotone gain, no separate mask) -- the configuration of BASELINE config 3 / keras NDCGLambdaWeight().
//
// Only ordered pairs with l_i > l_j carry a weight (losses_impl.py:503-537): ~40 % of the n^2 pairs of a
// list with 5 uniform grades.  The general kernel above sweeps all n^2 and pays exp + log + rcp on each;
// this one
//   * re-homes the items in GRADE order (label descending; a handful of distinct values -> a few ballot /
//     popcount rounds, no sort), so that the pairs a row needs are two contiguous column ranges:
//     lower grades behind its own segment ("hi" sweep: the row is the preferred item -> loss and its own
//     gradient share) and higher grades in front of it ("lo" sweep: the gradient share it receives as
//     the non-preferred item; no log needed);
//   * factorises the exponential: with A = e^{-(x - m)}, B = e^{x - m} per ITEM (double-float argument, 1 ulp),
//     1 + e^{-(x_i - x_j)} = fma(A_i, B_j, 1): a pair costs fma, rcp, log (hi) or fma, rcp (lo) instead of
//     sub, mul, exp, add, rcp, log, max, fma, mul, cmp, select;
//   * needs no per-pair predicate: for a monotone gain max(G_i - G_j, 0) is already 0 on equal or inverted
//     grades and on the padding records (gain +1e30 / -1e30), so ranges may be rounded outwards freely.
// The rank-difference discount |D(m) - D(m+1)| is an LDS gather on |rank_i - rank_j| (v_sad + ds_read_b32).
// Lists whose score range exceeds kLeanRange (e^{range} would leave fp32) take the per-pair exp form.
// ===========================================================================
constexpr float kLeanRange = 80.0f;
constexpr float kBigGain = 1e30f;
constexpr int kMaxRuns = 8;

__host__ __device__ inline size_t pw_lean_wave_lds(int Lp, bool itemw) {
  return (size_t)Lp * (16 + 4 + 4 + 4 + 4 + (itemw ? 4 : 0)) + 16;
}

template <int IPL, bool AUX, bool ITEMW>
__global__ __launch_bounds__(256) void pairwise_lean_kernel(const PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_all[];
  // blockDim.x = 64 * S: S wavefronts COOPERATE on one list through ONE shared LDS image when the batch alone cannot
  // fill the chip (B < 8192): the loads, the grade order and the ideal DCG are cheap and repeated per wave in
  // registers; the rank count and the two pair sweeps -- 85 % of the work -- are split.  S == 1: no workgroup barrier.
  const int wave = threadIdx.x >> 6, S = blockDim.x >> 6;
  const int Lp = a.Lp;
  unsigned char* smem_raw = smem_all;
  float* nz_slot = reinterpret_cast<float*>(smem_all + pw_lean_wave_lds(Lp, ITEMW));   // [2 S]
#define LIST_SYNC() do { if (S > 1) __syncthreads(); else WAVE_LDS_SYNC(); } while (0)
  float4* rec = reinterpret_cast<float4*>(smem_raw);          // [Lp] grade order: (B, A, gain, 4 * rank as bits)
  float* U = reinterpret_cast<float*>(rec + Lp);              // [Lp] |D(m) - D(m+1)| * list_size
  float* XS = U + Lp;                                         // [Lp] compact x (rank count) / x in grade order (slow path)
  int* CIS = reinterpret_cast<int*>(XS + Lp);                 // [Lp] grade position -> original index
  int* SEG = CIS + Lp;                                        // [Lp] (lo-sweep end | hi-sweep start << 16)
  float* WS = reinterpret_cast<float*>(SEG + Lp);             // [Lp] item weight, grade order (ITEMW only)
  const int lane = threadIdx.x & 63, b = a.order ? a.order[blockIdx.x] : blockIdx.x, L = a.L;
  const size_t base = (size_t)b * L;
  const float lw = a.list_weights ? a.list_weights[b] : 1.0f;
#ifdef TFR_PROFILE_STAMPS
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  PW_STAMP(0);

  // ---- 1. load, gains, compaction of the valid items (mask == NULL: valid = label >= 0).
  float g[IPL], xr[IPL], labr[IPL], wr[IPL];
  int posr[IPL];
  bool lv[IPL];
  int n = 0;
  float xmin = INFINITY, xmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = lane + 64 * r;
    g[r] = 0.f; lv[r] = false;
    float x = 0.f, lab = -1.f, w = 0.f;
    if (e < L) {
      lab = a.labels[base + e];
      x = a.logits[base + e] / a.temperature;
      lv[r] = lab >= 0.0f;
      if (lv[r]) {
        g[r] = (a.gain_kind == TFR_GAIN_POW2M1) ? gain_pow2m1(lab) : lab;
        w = ITEMW ? a.item_weights[base + e] * lw : lw;
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x);
      } else if (wave == 0) {
        if (a.row_loss) a.row_loss[base + e] = 0.f;
        if (AUX && a.row_weight) a.row_weight[base + e] = 0.f;
        if (a.dlogits) a.dlogits[base + e] = 0.f;
      }
    }
    const unsigned long long bal = __ballot(lv[r]);
    xr[r] = x; labr[r] = lv[r] ? lab : -1.0f; wr[r] = w;
    posr[r] = n + __popcll(bal & ((1ull << lane) - 1ull));
    if (lv[r] && wave == 0) XS[posr[r]] = x;
    n += __popcll(bal);
  }
  xmin = wave_min_u(xmin); xmax = wave_max_u(xmax);
  const bool fast = (xmax - xmin) <= kLeanRange;             // wave-uniform
  const float m = 0.5f * (xmax + xmin);
  for (int q = threadIdx.x; q < Lp; q += 64 * S) {
    const float um = (q >= 1 && q < L) ? fabsf(a.discount[q - 1] - a.discount[q]) : 0.0f;
    U[q] = um * (float)L;                                    // the final x list_size (:278) folded in
  }
  const int n4 = (n + 3) >> 2;
  if (wave == 0) for (int p = n + lane; p < n4 * 4 + 4 && p < Lp; p += 64) XS[p] = -INFINITY;
  int* const RKS = CIS;                                      // scratch: count by compact position (CIS is filled in step 5)
  int* const OCC = SEG;                                      // scratch: how many items share a count (SEG: step 5)
  for (int p = threadIdx.x; p < n; p += 64 * S) OCC[p] = 0;
  LIST_SYNC();

  PW_STAMP(1);
  // ---- 2. ranks by counting (score descending, ties by index) (:483-500).  Compact item p = lane + 64 q
  // counts the scores above its own: one v_cmp + half a carry-add per compare, columns as float4 LDS broadcasts.
  // Items with tied scores end up with the SAME count: an occupancy table over the counts finds them (rare), and
  // only those add their earlier equals.
  int rk[IPL];
  {
    // (wave_rank_by_count of common.h with its passes dealt to the S waves)
    const float4* X4 = reinterpret_cast<const float4*>(XS);
    for (int q0 = 64 * wave; q0 < n; q0 += 64 * S) {
      const int p = q0 + lane;
      const bool on = p < n;
      const float xi = on ? XS[p] : INFINITY;
      int cnt = 0;
      for (int gq = 0; gq < n4; ++gq) {
        const float4 xx = X4[gq];
        cnt += (xx.x > xi) ? 1 : 0; cnt += (xx.y > xi) ? 1 : 0;
        cnt += (xx.z > xi) ? 1 : 0; cnt += (xx.w > xi) ? 1 : 0;
      }
      if (on) { RKS[p] = cnt; atomicAdd(&OCC[cnt], 1); }
    }
    LIST_SYNC();
    for (int q0 = 64 * wave; q0 < n; q0 += 64 * S) {
      const int p = q0 + lane;
      const bool tie = p < n && OCC[RKS[p]] > 1;
      if (__ballot(tie)) {                                   // wave-uniform: some item of this pass shares its score
        if (tie) {
          const float xi = XS[p];
          int cnt = RKS[p];
          for (int j = 0; j < p; ++j) cnt += (XS[j] == xi) ? 1 : 0;
          RKS[p] = cnt;                                      // (others read OCC at their OWN first count only)
        }
      }
    }
    LIST_SYNC();
#pragma unroll
    for (int r = 0; r < IPL; ++r) rk[r] = lv[r] ? RKS[posr[r]] : 0;
    LIST_SYNC();                                             // RKS (= CIS) and OCC (= SEG) are rewritten below
  }

  PW_STAMP(2);
  // ---- 3. grade order: repeatedly take the largest remaining label value; its items (in element order)
  // become the next segment.  After kMaxRuns distinct values the rest forms one unsorted tail segment whose
  // rows sweep conservatively (every column is a candidate; the gain difference decides).
  int sp[IPL], seg[IPL];
  float rem[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) { rem[r] = labr[r]; sp[r] = 0; seg[r] = 0; }
  int pos = 0;
  bool tail = false;
  for (int it = 0; it <= kMaxRuns; ++it) {
    float mx = rem[0];
#pragma unroll
    for (int r = 1; r < IPL; ++r) mx = fmaxf(mx, rem[r]);
    const float v = wave_max_u(mx);
    if (v < 0.0f) break;
    const bool last = it == kMaxRuns;                        // everything that is left
    int c = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const bool hit = last ? (rem[r] >= 0.0f) : (rem[r] == v);
      const unsigned long long bal = __ballot(hit);
      if (hit) { sp[r] = pos + c + __popcll(bal & ((1ull << lane) - 1ull)); rem[r] = -2.0f; }
      c += __popcll(bal);
      seg[r] = hit ? -1 : seg[r];                            // marks "assigned in this round"
    }
#pragma unroll
    for (int r = 0; r < IPL; ++r)
      if (seg[r] == -1) seg[r] = last ? (n | (pos << 16)) : (pos | ((pos + c) << 16));
    pos += c;
    tail = last;
  }

  // ---- 4. ideal DCG of the labels (:109-134): the grade position IS the sorted position (equal labels have
  // equal gains); with an unsorted tail segment the gains are sorted (register bitonic network) instead.
  float inv_max_dcg = 1.0f;
  if (a.normalized) {
    float idcg = 0.f;
    if (!tail) {
#pragma unroll
      for (int r = 0; r < IPL; ++r) if (lv[r]) idcg += g[r] * a.discount[sp[r]];
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
          idcg += __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o) * a.discount[e];
        }
      }
    }
    idcg = wave_sum_u(idcg);
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
  }

  // ---- 5. records in grade order.
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    if (!lv[r] || (r % S) != wave) continue;                  // register r's records are written by wave r mod S
    const float xv = xr[r];
    float Bv, Av;
    if (fast) {
      const float t_hi = xv - m;
      const float bb = t_hi - xv;
      const float t_lo = (xv - (t_hi - bb)) + (-m - bb);
      Bv = exp_df_hw(t_hi, t_lo);
      Av = exp_df_hw(-t_hi, -t_lo);
    } else {
      Bv = xv; Av = xv;                                      // the slow path works on the scores themselves
    }
    rec[sp[r]] = make_float4(Bv, Av, g[r] * inv_max_dcg, __int_as_float(rk[r] * 4));
    CIS[sp[r]] = lane + 64 * r;
    SEG[sp[r]] = seg[r];
    if (ITEMW) WS[sp[r]] = wr[r];
  }
  const int C = a.C;
  const int npad = ((n + 2 * C - 1) / (2 * C)) * (2 * C);    // two columns per trip per lane
  for (int p = n + threadIdx.x; p < npad && p < Lp; p += 64 * S) {      // neutral padding: B = A = 0 -> w1 = 1; gain decides
    rec[p] = make_float4(0.f, 0.f, kBigGain, __int_as_float(0));
    if (ITEMW) WS[p] = 0.f;
  }
  LIST_SYNC();

  PW_STAMP(3);
  // ---- 6. pair sweeps: row = C adjacent lanes, 64 / C rows per pass, two columns per trip.
  const int rows_per_pass = 64 / C;
  const int c = lane % C, rsub = lane / C;
  const int trips = npad / (2 * C);
  typedef const __attribute__((address_space(3))) float lds_cf;
  const uint32_t ubase = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)U;
  float nnz_local = 0.f, list_local = 0.f;
  for (int row0 = wave * rows_per_pass; row0 < n; row0 += S * rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    const float4 ri = rec[active ? row : 0];
    const float Ghi = active ? ri.z : -kBigGain;             // inactive rows: no positive gain difference either way
    const float Glo = active ? ri.z : kBigGain;
    const uint32_t r4i = (uint32_t)__float_as_int(ri.w);
    const int last_row = (row0 + rows_per_pass < n ? row0 + rows_per_pass : n) - 1;
    const int it_h0 = (SEG[row0] >> 16) / (2 * C);           // hi sweep: columns behind the first row's segment
    const int it_l1 = ((SEG[last_row] & 0xffff) + 2 * C - 1) / (2 * C);   // lo sweep: columns before the last row's
    float acc_l = 0.f, acc_g = 0.f, acc_w = 0.f, acc_nz = 0.f, acc_g2 = 0.f;
    if (fast) {
      for (int it = it_h0; it < trips; ++it) {               // row preferred: 1 + e^{-(x_i - x_j)} = fma(A_i, B_j, 1)
        const int j0 = c + it * 2 * C, j1 = j0 + C;
        const float4 r0 = rec[j0], r1 = rec[j1];
        const float u0 = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(r0.w), ubase);
        const float u1 = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(r1.w), ubase);
        const float w10 = __builtin_fmaf(ri.y, r0.x, 1.0f), w11 = __builtin_fmaf(ri.y, r1.x, 1.0f);
        const float q0 = __builtin_amdgcn_rcpf(w10), q1 = __builtin_amdgcn_rcpf(w11);
        const float lg0 = __builtin_amdgcn_logf(w10), lg1 = __builtin_amdgcn_logf(w11);
        const float W0 = fmaxf(Ghi - r0.z, 0.0f) * u0, W1 = fmaxf(Ghi - r1.z, 0.0f) * u1;
        acc_l = __builtin_fmaf(W0, lg0, acc_l); acc_l = __builtin_fmaf(W1, lg1, acc_l);
        acc_g = __builtin_fmaf(W0, 1.0f - q0, acc_g); acc_g = __builtin_fmaf(W1, 1.0f - q1, acc_g);
        if (AUX) {
          acc_w += W0; acc_w += W1;
          acc_nz += (W0 != 0.0f) ? 1.0f : 0.0f; acc_nz += (W1 != 0.0f) ? 1.0f : 0.0f;
        }
      }
      for (int it = 0; it < it_l1; ++it) {                   // column preferred: 1 + e^{-(x_j - x_i)} = fma(B_i, A_j, 1)
        const int j0 = c + it * 2 * C, j1 = j0 + C;
        const float4 r0 = rec[j0], r1 = rec[j1];
        const float u0 = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(r0.w), ubase);
        const float u1 = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(r1.w), ubase);
        const float q0 = __builtin_amdgcn_rcpf(__builtin_fmaf(ri.x, r0.y, 1.0f));
        const float q1 = __builtin_amdgcn_rcpf(__builtin_fmaf(ri.x, r1.y, 1.0f));
        float W0 = fmaxf(r0.z - Glo, 0.0f) * u0, W1 = fmaxf(r1.z - Glo, 0.0f) * u1;
        if (ITEMW) { W0 *= WS[j0]; W1 *= WS[j1]; }          // the weight of the PREFERRED item (:917-930)
        acc_g2 = __builtin_fmaf(W0, 1.0f - q0, acc_g2); acc_g2 = __builtin_fmaf(W1, 1.0f - q1, acc_g2);
      }
    } else {
      // per-pair exponential, numerically safe for any score range (same algebra as pair_loss above);
      // rec.x = rec.y = x.  The loss accumulates in log2 units like the fast path.
      for (int it = it_h0; it < trips; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = c + it * 2 * C + h * C;
          const float4 rj = rec[j];
          const float u = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(rj.w), ubase);
          const bool padded = j >= n;
          const float d0 = padded ? 0.0f : ri.x - rj.x;
          const float e = __builtin_amdgcn_exp2f(-fabsf(d0) * kLog2e);
          const float w1 = 1.0f + e;
          const float q = __builtin_amdgcn_rcpf(w1);
          const float lg = __builtin_amdgcn_logf(w1) + fmaxf(-d0, 0.0f) * kLog2e;
          const float sel = (d0 >= 0.0f) ? e * q : q;         // sigma(-d0)
          const float W = fmaxf(Ghi - rj.z, 0.0f) * u;
          acc_l = __builtin_fmaf(W, lg, acc_l);
          acc_g = __builtin_fmaf(W, sel, acc_g);
          if (AUX) { acc_w += W; acc_nz += (W != 0.0f) ? 1.0f : 0.0f; }
        }
      }
      for (int it = 0; it < it_l1; ++it) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int j = c + it * 2 * C + h * C;
          const float4 rj = rec[j];
          const float u = *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(r4i, (uint32_t)__float_as_int(rj.w), ubase);
          const bool padded = j >= n;
          const float d0 = padded ? 0.0f : rj.x - ri.x;        // preferred (column) minus row
          const float e = __builtin_amdgcn_exp2f(-fabsf(d0) * kLog2e);
          const float q = __builtin_amdgcn_rcpf(1.0f + e);
          const float sel = (d0 >= 0.0f) ? e * q : q;
          float W = fmaxf(rj.z - Glo, 0.0f) * u;
          if (padded) W = 0.0f;
          if (ITEMW) W *= WS[j];
          acc_g2 = __builtin_fmaf(W, sel, acc_g2);
        }
      }
    }
    for (int o = 1; o < C; o <<= 1) {
      acc_l += __shfl_xor(acc_l, o, 64);
      acc_g += __shfl_xor(acc_g, o, 64);
      acc_g2 += __shfl_xor(acc_g2, o, 64);
      if (AUX) { acc_w += __shfl_xor(acc_w, o, 64); acc_nz += __shfl_xor(acc_nz, o, 64); }
    }
    if (active && c == 0) {
      const int oi = CIS[row];
      const float wi = ITEMW ? WS[row] : lw;                 // weight of the row item (preferred in the hi sweep)
      const float row_l = acc_l * kLn2 * wi;
      list_local += row_l;
      if (a.row_loss) a.row_loss[base + oi] = row_l;
      if (AUX && a.row_weight) a.row_weight[base + oi] = acc_w * wi;
      const float g2 = ITEMW ? acc_g2 : acc_g2 * lw;
      if (a.dlogits) a.dlogits[base + oi] = (g2 - acc_g * wi) / a.temperature;
      if (AUX) nnz_local += (wi != 0.0f) ? acc_nz : 0.0f;
    }
  }
  PW_STAMP(4);
#ifdef TFR_PROFILE_STAMPS
  if (lane == 0 && wave == 0 && g_prof_buf_pw) {
    prof_t[7] = (unsigned long long)n;
    for (int i = 0; i < 8; ++i) g_prof_buf_pw[(size_t)b * 8 + i] = prof_t[i];
  }
#endif
  pw_finish<AUX>(a, b, lane, wave, S, nnz_local, list_local, nz_slot);
#undef LIST_SYNC
}

int env_int(const char* name, int dflt);

#include "lambdarank_group.h"

// One launch of the group kernel (dynamic LDS beyond 64 KiB needs the attribute once per instantiation).
template <int IPL, bool AUX, bool IW>
int launch_grp(const PwArgs& a, int B, int G, int Wt, int R, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024) {                                     // per device and cheap: set whenever it is needed (thread-safe, any GPU)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lambdarank_group_kernel<IPL, AUX, IW>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const int N = (B + G - 1) / G;
  hipLaunchKernelGGL((lambdarank_group_kernel<IPL, AUX, IW>), dim3(N), dim3(64 * Wt), lds, stream, a, B, R, grp_lp(a.L), G, 0);
  return (int)hipGetLastError();
}

// The same launch with the builder's ranks from wave_rank_by_bucket (plain case only: no aux outputs, no item weights).
template <int IPL>
int launch_grp_bucket(const PwArgs& a, int B, int G, int Wt, int R, size_t lds, hipStream_t stream) {
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&lambdarank_group_kernel<IPL, false, false, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const int N = (B + G - 1) / G;
  static const int env_graded = env_int("TFR_LAMBDARANK_GRADED", 1);     // 0: every list through the general builder
  hipLaunchKernelGGL((lambdarank_group_kernel<IPL, false, false, true>), dim3(N), dim3(64 * Wt), lds, stream, a, B, R,
                     grp_lp(a.L), G, env_graded ? 1 : 0);
  return (int)hipGetLastError();
}

template <int IPL>
int launch_pw_wave(const PwArgs& a, int B, hipStream_t stream) {
  static const int env_s = env_int("TFR_PAIRWISE_WAVES_PER_LIST", 0);
  int S = env_s > 0 ? env_s : 1;     // (S > 1 measured slower at every batch size tried)
  if (S > 4) S = 4;
  if (a.L <= 64) S = 1;
  while (S > 1 && (size_t)S * pw_wave_lds(a.Lp) + 32 > 60 * 1024) S >>= 1;
  const size_t lds = (size_t)S * pw_wave_lds(a.Lp) + 32;
  if (lds > 64 * 1024) return -3;          // (caller falls back to the workgroup kernel)
  const bool generic = (a.lambda_kind == TFR_LAMBDA_DCG) &&
                       (a.lambda_sub != TFR_SUB_DCG || a.smooth != 0.0f || (a.topn > 0 && a.topn < a.L) ||
                        a.mask != nullptr);
  const bool aux = a.row_weight != nullptr || a.nnz != nullptr;
  const bool itemw = generic || a.item_weights != nullptr || a.mask != nullptr;
  static const int env_lean = env_int("TFR_PAIRWISE_LEAN", 1);
  if (env_lean && a.kind == TFR_PAIR_LOGISTIC && a.lambda_kind == TFR_LAMBDA_DCG && !generic &&
      a.gain_kind != TFR_GAIN_CUSTOM) {
    // LambdaRank fast path (grade-segmented, factorised exponential)
    const bool iw = a.item_weights != nullptr;
    static const int env_grp = env_int("TFR_LAMBDARANK_GROUP", 1);
    static const int env_grp_min = env_int("TFR_LAMBDARANK_GROUP_MIN_B", 512);
    int gG = 0, gWt = 0, gR = 0;
    size_t glds = 0;
    if (env_grp && B >= env_grp_min && grp_geometry(B, a.L, iw, gG, gWt, gR, glds)) {
      // G lists per workgroup, barrier-free build / sweep, conflict-free rank-difference gather (lambdarank_group.h)
      static const int env_bkt = env_int("TFR_LAMBDARANK_BUCKET", 1);
      if (env_bkt && !aux && !iw) return launch_grp_bucket<IPL>(a, B, gG, gWt, gR, glds, stream);
      if (aux) return iw ? launch_grp<IPL, true, true>(a, B, gG, gWt, gR, glds, stream)
                         : launch_grp<IPL, true, false>(a, B, gG, gWt, gR, glds, stream);
      return iw ? launch_grp<IPL, false, true>(a, B, gG, gWt, gR, glds, stream)
                : launch_grp<IPL, false, false>(a, B, gG, gWt, gR, glds, stream);
    }
    // smaller batches: S waves cooperate on one list (round-2 kernel)
    static const int env_ls = env_int("TFR_PAIRWISE_LEAN_WAVES", 0);
    // waves per list: 1 when the batch alone gives every SIMD its 8 list-waves; else 2 / 4 cooperate (shared LDS image)
    int Sl = env_ls > 0 ? env_ls : (B >= 8192 ? 1 : (B >= 2048 ? 2 : 4));
    if (a.L <= 64) Sl = 1;
    if (Sl > 4) Sl = 4;
    if (Sl == 3) Sl = 2;
    const size_t ll = pw_lean_wave_lds(a.Lp, iw) + 32;
    if (ll <= 64 * 1024) {
#define PW_LEAN(AUX, IW) hipLaunchKernelGGL((pairwise_lean_kernel<IPL, AUX, IW>), dim3(B), dim3(64 * Sl), ll, stream, a)
      if (aux) { if (iw) PW_LEAN(true, true); else PW_LEAN(true, false); }
      else { if (iw) PW_LEAN(false, true); else PW_LEAN(false, false); }
#undef PW_LEAN
      return (int)hipGetLastError();
    }
  }
#define PW_L2(LAM, GEN, AUX, IW) hipLaunchKernelGGL((pairwise_wave_kernel<IPL, LAM, GEN, AUX, IW, TFR_PAIR_LOGISTIC>), dim3(B), dim3(64 * S), lds, stream, a)
#define PW_RT(LAM, GEN) hipLaunchKernelGGL((pairwise_wave_kernel<IPL, LAM, GEN, true, true, -1>), dim3(B), dim3(64 * S), lds, stream, a)
#define PW_LAUNCH(LAM, GEN) do { if (a.kind != TFR_PAIR_LOGISTIC) PW_RT(LAM, GEN);                              \
                                 else if (aux) { if (itemw) PW_L2(LAM, GEN, true, true); else PW_L2(LAM, GEN, true, false); } \
                                 else { if (itemw) PW_L2(LAM, GEN, false, true); else PW_L2(LAM, GEN, false, false); } } while (0)
  if (a.lambda_kind == TFR_LAMBDA_DCG) { if (generic) PW_LAUNCH(TFR_LAMBDA_DCG, true); else PW_LAUNCH(TFR_LAMBDA_DCG, false); }
  else if (a.lambda_kind == TFR_LAMBDA_LABELDIFF) PW_LAUNCH(TFR_LAMBDA_LABELDIFF, false);
  else PW_LAUNCH(TFR_LAMBDA_NONE, false);
#undef PW_RT
#undef PW_L2
#undef PW_LAUNCH
  return (int)hipGetLastError();
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

static int pairwise_dispatch(int kind, const float* logits, const float* labels, const uint8_t* mask,
                                         const float* item_weights, const float* list_weights,
                                         int lambda_kind, int topn, float smooth_fraction,
                                         int normalized, int gain_kind, const float* gains,
                                         const float* discount, int B, int L, float temperature,
                                         float* row_loss_out, float* row_weight_out, float* nnz_out,
                                         float* dlogits_out, const int* order, float* list_loss_out, void* stream,
                                         float* loss_sum_out = nullptr, uint32_t* ticket = nullptr, uint32_t tie_seed = 0u) {
  if (!logits || !labels || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (loss_sum_out && (!list_loss_out || !ticket)) return TFR_EINVAL;       // the per-list sums are the entries that are added up
  if ((kind & ~TFR_PAIR_TIED_ZERO) < TFR_PAIR_LOGISTIC || (kind & ~TFR_PAIR_TIED_ZERO) > TFR_PAIR_MSE) return TFR_EINVAL;
  if ((kind & TFR_PAIR_TIED_ZERO) && (kind & 0xff) != TFR_PAIR_LOGISTIC) return TFR_EINVAL;     // the flag is the logistic loss's
  if (lambda_kind < TFR_LAMBDA_NONE || lambda_kind > TFR_LAMBDA_PRECISION) return TFR_EINVAL;
  // DCGLambdaWeightV2 / YetiDCGLambdaWeight / PrecisionLambdaWeight run as sub-kinds of the generic DCG path
  int lambda_sub = TFR_SUB_DCG;
  if (lambda_kind == TFR_LAMBDA_DCG_V2) { lambda_sub = TFR_SUB_DCG_V2; lambda_kind = TFR_LAMBDA_DCG; smooth_fraction = 0.0f; }
  else if (lambda_kind == TFR_LAMBDA_YETI_DCG) { lambda_sub = TFR_SUB_YETI; lambda_kind = TFR_LAMBDA_DCG; smooth_fraction = 0.0f; }
  else if (lambda_kind == TFR_LAMBDA_PRECISION) {
    if (topn <= 0) return TFR_EINVAL;                   // PrecisionLambdaWeight(topn) is mandatory (:413)
    lambda_sub = TFR_SUB_PRECISION; lambda_kind = TFR_LAMBDA_DCG; smooth_fraction = 0.0f; normalized = 0;
  }
  if (lambda_kind == TFR_LAMBDA_DCG) {
    if (!discount) return TFR_EINVAL;
    if (gain_kind == TFR_GAIN_CUSTOM && !gains) return TFR_EINVAL;
    if (!(smooth_fraction >= 0.0f && smooth_fraction <= 1.0f)) return TFR_EINVAL;   // :329-331
  }
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) {
    if (loss_sum_out) return (int)hipMemsetAsync(loss_sum_out, 0, sizeof(float), (hipStream_t)stream);
    return TFR_OK;
  }
  const GridSum gsum = {loss_sum_out, list_loss_out, ticket, B};
  static const int env_threads = env_int("TFR_PAIRWISE_THREADS", 0);
  static const int env_lanes = env_int("TFR_PAIRWISE_LANES", 0);
  static const int env_wave = env_int("TFR_PAIRWISE_WAVE", 1);
  const int C = env_lanes > 0 ? env_lanes : 2;
  if (C > 64 || (C & (C - 1))) return TFR_EINVAL;
  // (a tie seed -- _compute_ranks(shuffle_ties=True), :483-500: equal scores ranked in a random order -- goes to the
  // workgroup kernel, whose ranks come from a key sort with a tie field; the wave kernels rank by counting, index order)
  if (env_wave && env_threads == 0 && L <= 256 && tie_seed == 0u) {      // longer lists: workgroup-per-list kernel below
    PwArgs w;
    w.tie_seed = 0u;
    w.logits = logits; w.labels = labels; w.mask = mask; w.item_weights = item_weights;
    w.list_weights = list_weights; w.lambda_kind = lambda_kind; w.lambda_sub = lambda_sub; w.topn = topn;
    w.smooth = smooth_fraction; w.normalized = normalized; w.gain_kind = gain_kind; w.gains = gains;
    const int c2 = (2 * C > 4) ? 2 * C : 4;
    w.discount = discount; w.L = L; w.Lp = ((L + c2 - 1) / c2) * c2 + 4; w.P = 0;
    w.temperature = temperature; w.C = C; w.kind = kind; w.row_loss = row_loss_out; w.row_weight = row_weight_out;
    w.nnz = nnz_out; w.dlogits = dlogits_out; w.order = order; w.list_loss = list_loss_out; w.sum = gsum;
    hipStream_t st = (hipStream_t)stream;
    if (L <= 64) return launch_pw_wave<1>(w, B, st);
    if (L <= 128) return launch_pw_wave<2>(w, B, st);
    return launch_pw_wave<4>(w, B, st);
  }
  const int T = env_threads > 0 ? env_threads : (L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 512 ? 256 : 512)));
  if (T % 64 || T > 1024) return TFR_EINVAL;
  PwArgs a;
  a.logits = logits; a.labels = labels; a.mask = mask; a.item_weights = item_weights;
  a.list_weights = list_weights; a.lambda_kind = lambda_kind; a.lambda_sub = lambda_sub; a.topn = topn;
  a.smooth = smooth_fraction; a.normalized = normalized; a.gain_kind = gain_kind; a.gains = gains;
  a.discount = discount; a.L = L; a.Lp = ((L + 3) / 4) * 4 + 4; a.P = pow2_ceil(L < 2 ? 2 : L);
  a.temperature = temperature; a.C = C; a.kind = kind; a.row_loss = row_loss_out; a.row_weight = row_weight_out;
  a.nnz = nnz_out; a.dlogits = dlogits_out; a.order = order; a.list_loss = list_loss_out; a.sum = gsum;
  a.tie_seed = tie_seed;
  const size_t lds = pw_smem_bytes(a.Lp, a.P);
  if (lds > 160 * 1024) return TFR_ETOOLARGE;
  const bool generic = (lambda_kind == TFR_LAMBDA_DCG) &&
                       (lambda_sub != TFR_SUB_DCG || smooth_fraction != 0.0f || (topn > 0 && topn < L) ||
                        mask != nullptr);
#define PW_BLOCK(LAM, GEN)                                                                        \
  do {                                                                                            \
    if (lds > 64 * 1024) {                                                                        \
      hipError_t e = hipFuncSetAttribute(                                                         \
          reinterpret_cast<const void*>(&pairwise_logistic_kernel<LAM, GEN>),                     \
          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                                  \
      if (e != hipSuccess) return (int)e;                                                         \
    }                                                                                             \
    hipLaunchKernelGGL((pairwise_logistic_kernel<LAM, GEN>), dim3(B), dim3(T), lds,               \
                       (hipStream_t)stream, a);                                                   \
  } while (0)
  if (lambda_kind == TFR_LAMBDA_DCG) { if (generic) PW_BLOCK(TFR_LAMBDA_DCG, true); else PW_BLOCK(TFR_LAMBDA_DCG, false); }
  else if (lambda_kind == TFR_LAMBDA_LABELDIFF) PW_BLOCK(TFR_LAMBDA_LABELDIFF, false);
  else PW_BLOCK(TFR_LAMBDA_NONE, false);
#undef PW_BLOCK
  return (int)hipGetLastError();
}

extern "C" int tfr_pairwise_logistic_f32(const float* logits, const float* labels, const uint8_t* mask,
                                         const float* item_weights, const float* list_weights,
                                         int lambda_kind, int topn, float smooth_fraction,
                                         int normalized, int gain_kind, const float* gains,
                                         const float* discount, int B, int L, float temperature,
                                         float* row_loss_out, float* row_weight_out, float* nnz_out,
                                         float* dlogits_out, void* stream) {
  return pairwise_dispatch(TFR_PAIR_LOGISTIC, logits, labels, mask, item_weights, list_weights, lambda_kind, topn,
                           smooth_fraction, normalized, gain_kind, gains, discount, B, L, temperature,
                           row_loss_out, row_weight_out, nnz_out, dlogits_out, nullptr, nullptr, stream);
}

extern "C" int tfr_pairwise_loss_f32(int loss_kind, const float* logits, const float* labels, const uint8_t* mask,
                                     const float* item_weights, const float* list_weights,
                                     int lambda_kind, int topn, float smooth_fraction,
                                     int normalized, int gain_kind, const float* gains,
                                     const float* discount, int B, int L, float temperature,
                                     float* row_loss_out, float* row_weight_out, float* nnz_out,
                                     float* dlogits_out, const int32_t* list_order, float* list_loss_out,
                                     uint32_t tie_seed, void* stream) {
  return pairwise_dispatch(loss_kind, logits, labels, mask, item_weights, list_weights, lambda_kind, topn,
                           smooth_fraction, normalized, gain_kind, gains, discount, B, L, temperature,
                           row_loss_out, row_weight_out, nnz_out, dlogits_out, list_order, list_loss_out, stream,
                           nullptr, nullptr, tie_seed);
}

extern "C" int tfr_pairwise_loss_sum_f32(int loss_kind, const float* logits, const float* labels, const uint8_t* mask,
                                         const float* item_weights, const float* list_weights,
                                         int lambda_kind, int topn, float smooth_fraction,
                                         int normalized, int gain_kind, const float* gains,
                                         const float* discount, int B, int L, float temperature,
                                         float* row_loss_out, float* row_weight_out, float* nnz_out,
                                         float* dlogits_out, const int32_t* list_order, float* list_loss_out,
                                         float* loss_sum_out, uint32_t* ticket, uint32_t tie_seed, void* stream) {
  if (!loss_sum_out || !ticket || !list_loss_out) return TFR_EINVAL;
  return pairwise_dispatch(loss_kind, logits, labels, mask, item_weights, list_weights, lambda_kind, topn,
                           smooth_fraction, normalized, gain_kind, gains, discount, B, L, temperature,
                           row_loss_out, row_weight_out, nnz_out, dlogits_out, list_order, list_loss_out, stream,
                           loss_sum_out, ticket, tie_seed);
}

#ifdef TFR_PROFILE_STAMPS
extern "C" int tfr_prof_set_buffer_pw(void* device_u64_buffer) {
  unsigned long long* p = (unsigned long long*)device_u64_buffer;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_buf_pw), &p, sizeof(p));
}
extern "C" int tfr_prof_set_stop_pw(int phase) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_stop_pw), &phase, sizeof(phase));
}
#endif
