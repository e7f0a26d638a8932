// ApproxNDCG loss, forward + backward fused in one launch, for gfx950.
//
// Reference behaviour restated (losses_impl.py): approx_ranks :77-106,
// inverse_max_dcg :109-134, ndcg :137-167, _safe_default_gain_fn :33-49,
// ApproxNDCGLoss._compute_unreduced_loss_impl :1587-1603, get_logits :773-785.
// Backward formulas: SURVEY.md Appendix B (the reference relies on TF autodiff).
//
// Design.  One workgroup per list.  The reference materialises four
// [B, L, L] tensors forward (tile, tile, sub, sigmoid) and as many again
// backward; here a list costs 12*L + 12 bytes of HBM traffic and the L^2 pair
// work never leaves registers:
//   * valid items are compacted to the front of LDS arrays (ragged lists cost
//     n_valid^2, not L^2);
//   * the pair sigmoid is factorised, sigma(x_j - x_i) = 1/(1 + E_i*F_j) with
//     E_i = exp(x_i - m), F_j = exp(m - x_j) computed ONCE per item (to ~1ulp
//     through a double-float argument), so a pair costs one v_fma + one v_rcp
//     + one v_add instead of sub/exp/add/rcp/add -- the transcendental count
//     per pair drops from 2 to 1.  Lists whose logit range exceeds 160 (where
//     E/F would leave the fp32 normal range) take a per-pair exp path;
//   * a row of the pair matrix is split over `C` adjacent lanes (columns in
//     float4 groups read from LDS as ds_read_b128) so that ragged row counts
//     still fill 64-wide waves; the C partial sums are combined with
//     wave shuffles.
#include "common.h"
#include "../../include/tfr_hip.h"

#include <stdlib.h>

using namespace tfr;

#define TFR_APPROX_NDCG 0
#define TFR_APPROX_MRR 1

namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kFastRange = 160.0f;
// Two sigmoids from ONE v_rcp_f32 (round 4): 1/a = b * rcp(a b), 1/b = a * rcp(a b).  The reciprocal is a
// quarter-rate instruction (8.5 issue cycles against 2.4 for an fma, tools/ubench.hip), so a pair of columns costs
// one rcp + one multiply + two fmas into the accumulators instead of two rcps + two adds.  a, b = 1 + E_i F_j must
// stay FINITE (inf * rcp(inf) = NaN): the logit range of the list is bounded so that E_i F_j <= e^87 < FLT_MAX; a
// product a b that overflows has both sigmoids below 2^-64 and yields 0 for both.  Lists beyond that range keep one
// rcp per pair (kFastRange), beyond that the per-pair exponential.
constexpr float kPairRange = 87.0f;
// ... and r = rcp(a b) must stay NORMAL (v_rcp_f32 flushes a denormal result to 0, and 0 times a huge partner is 0
// where the true sigmoid may be 1/3): a and b are carried scaled by c = 2^-40 (a' = c + (c E_i) F_j, the same fma),
// so a' b' lies in [2^-80, 2^171): a product beyond FLT_MAX (or one whose reciprocal flushes) has BOTH sigmoids below
// 2^-80.  The row sums accumulate sigma / c (the fma into the accumulator is unchanged) and are scaled back once per row.
constexpr float kPairScale = 0x1p-40f;

struct Smem {
  float* red;      // [32]
  int* wc;         // [16]
  float* Xr;       // [Lp] x = logit / T, original order
  float* Lb;       // [Lp] cleaned labels, original order
  uint8_t* V;      // [Lp] validity, original order
  uint32_t* sortbuf;  // [P]
  float* CX;       // [Lp] compact x       (pad -inf)
  int* CI;         // [Lp] compact -> original index
  float* CG;       // [Lp] compact gain
  float* E;        // [Lp] exp(x - m)
  float* F;        // [Lp] exp(m - x)      (pad +inf)
  float* A;        // [Lp] d loss / d rank (pad 0)
  float* Rk;       // [Lp] approx ranks
};

__host__ __device__ inline size_t smem_bytes(int Lp, int P) {
  return 128 + 64 + (size_t)Lp * 4 * 9 + (size_t)Lp + 16 + (size_t)P * 4;
}

__device__ __forceinline__ Smem carve(unsigned char* raw, int Lp, int P) {
  Smem s;
  s.red = reinterpret_cast<float*>(raw);
  s.wc = reinterpret_cast<int*>(raw + 128);
  float* f = reinterpret_cast<float*>(raw + 192);
  s.Xr = f; f += Lp;
  s.Lb = f; f += Lp;
  s.CX = f; f += Lp;
  s.CI = reinterpret_cast<int*>(f); f += Lp;
  s.CG = f; f += Lp;
  s.E = f; f += Lp;
  s.F = f; f += Lp;
  s.A = f; f += Lp;
  s.Rk = f; f += Lp;
  s.sortbuf = reinterpret_cast<uint32_t*>(f); f += P;
  s.V = reinterpret_cast<uint8_t*>(f);
  return s;
}

__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// sum_j sigma(x_j - x_i) over a float4 column group, factorised form.
__device__ __forceinline__ void pair_fwd_fast(float Ei, const float4 f, float& a0, float& a1,
                                              float& a2, float& a3) {
  a0 += fast_rcp(__builtin_fmaf(Ei, f.x, 1.0f));
  a1 += fast_rcp(__builtin_fmaf(Ei, f.y, 1.0f));
  a2 += fast_rcp(__builtin_fmaf(Ei, f.z, 1.0f));
  a3 += fast_rcp(__builtin_fmaf(Ei, f.w, 1.0f));
}
__device__ __forceinline__ float sig_slow(float xi, float xj) {
  // sigma(xj - xi) = 1 / (1 + exp(xi - xj))
  return fast_rcp(1.0f + __builtin_amdgcn_exp2f((xi - xj) * kLog2e));
}

__global__ void approx_ndcg_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                   const uint8_t* __restrict__ mask, const float* __restrict__ inv_log1p,
                                   const float* __restrict__ list_scale, int L, int Lp, int P,
                                   float temperature, int C, float* __restrict__ loss_out,
                                   float* __restrict__ weight_out, float* __restrict__ dlogits_out, int metric,
                                   const int* __restrict__ order, float* __restrict__ loss_sum,
                                   unsigned int* __restrict__ ticket, int B) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const Smem s = carve(smem_raw, Lp, P);
  const int tid = threadIdx.x, T = blockDim.x, lane = tid & 63, wid = tid >> 6, nw = T >> 6;
  const int b = order ? order[blockIdx.x] : blockIdx.x;
  const size_t base = (size_t)b * L;

  // ---- 1. load, clean (losses_impl.py:1589-1594), per-list label statistics.
  float lmax = -INFINITY, lsum = 0.f, xmin = INFINITY, xmax = -INFINITY;
  for (int i = tid; i < L; i += T) {
    const float lab = labels[base + i];
    const float x = logits[base + i] / temperature;
    const bool v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
    const float labc = v ? lab : 0.0f;
    s.Xr[i] = x; s.Lb[i] = labc; s.V[i] = v ? 1 : 0;
    lmax = fmaxf(lmax, labc); lsum += labc;
    if (v) { xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); }
  }
  lmax = block_max(lmax, s.red);
  lsum = block_sum(lsum, s.red);
  xmin = block_min(xmin, s.red);
  xmax = block_max(xmax, s.red);
  const bool nonzero = lsum > 0.0f;
  if (!nonzero) lmax = 1e-10f;                       // labels := 1e-10 everywhere (:1598-1599)

  // ---- 2. gains (safe gain :33-49) and inverse max DCG (:109-134): sort the gains.
  // (ApproxMRR, :1606-1632: the cleaned labels and 1 / their sum instead.)
  float inv_max_dcg;
  if (metric == TFR_APPROX_MRR) {
    for (int i = tid; i < L; i += T) s.Lb[i] = nonzero ? s.Lb[i] : 1e-10f;
    __syncthreads();
    inv_max_dcg = 1.0f / (nonzero ? lsum : (float)L * 1e-10f);
  } else {
    const float g0 = exp2f(-lmax);
    for (int i = tid; i < P; i += T) {
      float g = 0.f;
      if (i < L) {
        const float labc = nonzero ? s.Lb[i] : 1e-10f;
        g = exp2f(labc - lmax) - g0;
        s.Lb[i] = g;                                    // Lb now holds the gain
      }
      s.sortbuf[i] = __float_as_uint(g);                // g >= 0: bit order == value order
    }
    block_bitonic_sort_desc(s.sortbuf, P);
    float idcg = 0.f;
    for (int p = tid; p < L; p += T) idcg += __uint_as_float(s.sortbuf[p]) * inv_log1p[p];
    idcg = block_sum(idcg, s.red);
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
  }

  // ---- 3. stable compaction of the valid items.
  int n = 0;
  for (int i0 = 0; i0 < L; i0 += T) {
    const int i = i0 + tid;
    const bool v = (i < L) && (s.V[i] != 0);
    const unsigned long long bal = __ballot(v);
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s.wc[wid] = __popcll(bal);
    __syncthreads();
    int woff = 0, tot = 0;
    for (int w = 0; w < nw; ++w) { const int c = s.wc[w]; woff += (w < wid) ? c : 0; tot += c; }
    if (v) {
      const int pos = n + woff + lane_prefix;
      s.CX[pos] = s.Xr[i]; s.CI[pos] = i; s.CG[pos] = s.Lb[i];
    }
    n += tot;
  }
  __syncthreads();
  const int n4 = (n + 3) >> 2;                        // float4 column groups

  // ---- 4. per-item exponentials of the factorised sigmoid.
  const float m = 0.5f * (xmax + xmin);
  const bool fast = (xmax - xmin) <= kFastRange;
  for (int i = tid; i < n4 * 4; i += T) {
    float e = 0.f, f = INFINITY, x = -INFINITY;
    if (i < n) {
      x = s.CX[i];
      const float t_hi = x - m;
      const float bb = t_hi - x;
      const float t_lo = (x - (t_hi - bb)) + (-m - bb);
      e = exp_df(t_hi, t_lo);
      f = exp_df(-t_hi, -t_lo);
    }
    s.E[i] = e; s.F[i] = f; s.A[i] = 0.f;
    if (i >= n) s.CX[i] = x;
  }
  __syncthreads();

  // ---- 5. approximate ranks r_i = 0.5 + sum_j sigma(x_j - x_i)   (:77-106)
  const int rows_per_pass = T / C;
  const int c = tid % C, rsub = tid / C;
  const float4* F4 = reinterpret_cast<const float4*>(s.F);
  const float4* X4 = reinterpret_cast<const float4*>(s.CX);
  const float4* A4 = reinterpret_cast<const float4*>(s.A);
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    if (!__any(active)) continue;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (fast) {
      const float Ei = active ? s.E[row] : 0.f;
      for (int g = c; g < n4; g += C) pair_fwd_fast(Ei, F4[g], a0, a1, a2, a3);
    } else {
      const float xi = active ? s.CX[row] : 0.f;
      for (int g = c; g < n4; g += C) {
        const float4 xx = X4[g];
        a0 += sig_slow(xi, xx.x); a1 += sig_slow(xi, xx.y);
        a2 += sig_slow(xi, xx.z); a3 += sig_slow(xi, xx.w);
      }
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc = lanes_sum_c(acc, C);
    if (active && c == 0) s.Rk[row] = acc + 0.5f;
  }
  __syncthreads();

  // ---- 6. loss = -(sum_i G_i / log1p(r_i)) * invMaxDCG  (:137-167); a_i = dloss/dr_i.
  float dcg = 0.f;
  for (int i = tid; i < n; i += T) {
    const float r = s.Rk[i];
    const float g = s.CG[i];
    if (metric == TFR_APPROX_MRR) {
      dcg += g * (1.0f / r);
      s.A[i] = (g * inv_max_dcg) / (r * r);
    } else {
      const float lr = log1pf(r);
      dcg += g * (1.0f / lr);
      s.A[i] = (g * inv_max_dcg) / (lr * lr * (1.0f + r));
    }
  }
  dcg = block_sum(dcg, s.red);          // (ends with the barrier that publishes A)
  if (tid == 0) {
    if (!loss_sum) loss_out[b] = -(dcg * inv_max_dcg);
    weight_out[b] = nonzero ? 1.0f : 0.0f;
  }
  if (loss_sum && wid == 0) grid_weighted_sum_last(loss_out, b, -(dcg * inv_max_dcg), list_scale, B, loss_sum, ticket, lane);
  if (!dlogits_out) return;
  __syncthreads();

  // ---- 7. backward: dloss/ds_k = (1/T) * sum_i (a_i - a_k) * sigma'(x_i - x_k).
  const float scale = list_scale ? list_scale[b] : 1.0f;
  for (int i = tid; i < L; i += T)
    if (!s.V[i]) dlogits_out[base + i] = 0.0f;
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    if (!__any(active)) continue;
    const float ak = active ? s.A[row] : 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (fast) {
      const float Ek = active ? s.E[row] : 0.f;
      for (int g = c; g < n4; g += C) {
        const float4 f = F4[g];
        const float4 aj = A4[g];
        const float s0 = fast_rcp(__builtin_fmaf(Ek, f.x, 1.0f));
        const float s1 = fast_rcp(__builtin_fmaf(Ek, f.y, 1.0f));
        const float s2 = fast_rcp(__builtin_fmaf(Ek, f.z, 1.0f));
        const float s3 = fast_rcp(__builtin_fmaf(Ek, f.w, 1.0f));
        a0 = __builtin_fmaf(aj.x - ak, __builtin_fmaf(-s0, s0, s0), a0);
        a1 = __builtin_fmaf(aj.y - ak, __builtin_fmaf(-s1, s1, s1), a1);
        a2 = __builtin_fmaf(aj.z - ak, __builtin_fmaf(-s2, s2, s2), a2);
        a3 = __builtin_fmaf(aj.w - ak, __builtin_fmaf(-s3, s3, s3), a3);
      }
    } else {
      const float xk = active ? s.CX[row] : 0.f;
      for (int g = c; g < n4; g += C) {
        const float4 xx = X4[g];
        const float4 aj = A4[g];
        const float s0 = sig_slow(xk, xx.x), s1 = sig_slow(xk, xx.y);
        const float s2 = sig_slow(xk, xx.z), s3 = sig_slow(xk, xx.w);
        a0 = __builtin_fmaf(aj.x - ak, __builtin_fmaf(-s0, s0, s0), a0);
        a1 = __builtin_fmaf(aj.y - ak, __builtin_fmaf(-s1, s1, s1), a1);
        a2 = __builtin_fmaf(aj.z - ak, __builtin_fmaf(-s2, s2, s2), a2);
        a3 = __builtin_fmaf(aj.w - ak, __builtin_fmaf(-s3, s3, s3), a3);
      }
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc = lanes_sum_c(acc, C);
    if (active && c == 0) dlogits_out[base + s.CI[row]] = scale * (acc / temperature);
  }
}


// ===========================================================================
// Wave-per-list variant (L <= 64 * IPL).  One 64-lane wavefront owns a list:
// no workgroup barriers, reductions are wave shuffles, the ideal-DCG sort is a
// bitonic network held entirely in registers (IPL keys per lane), and the
// per-item scalar work uses single-instruction transcendentals.  Profiling the
// block kernel showed ~50 % of its VALU time in the per-list set-up (LDS sort,
// block reductions, libm calls) rather than in the pair sweep; this variant
// removes most of it and is used whenever there are enough lists to fill the
// chip with single waves.
// ===========================================================================
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wmin(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Descending bitonic sort of 64*IPL non-negative floats (as uint32 bit patterns),
// element e = lane + 64*r lives in a[r] of `lane`.
template <int IPL>
__device__ __forceinline__ void wave_sort_desc(uint32_t (&a)[IPL], int lane) {
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
          const uint32_t p = wave_xor_exchange(a[r], j);
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

// exp(t_hi + t_lo) with hardware exp2 / ldexp (1 ulp), see exp_df in common.h.
__device__ __forceinline__ float exp_df_fast(float t_hi, float t_lo) {
  const float LOG2E_HI = 1.44269502162933349609375f;
  const float LOG2E_LO = 1.92596299112661746e-08f;
  const float y_hi = t_hi * LOG2E_HI;
  float y_lo = __builtin_fmaf(t_hi, LOG2E_HI, -y_hi);
  y_lo = __builtin_fmaf(t_hi, LOG2E_LO, y_lo);
  y_lo = __builtin_fmaf(t_lo, LOG2E_HI, y_lo);
  const float n = rintf(y_hi);
  const float f = (y_hi - n) + y_lo;
  return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// Developer aid (never in the product build): -DTFR_PROFILE_STAMPS makes lane 0 record
// s_memtime at the phase boundaries into a buffer set with tfr_prof_set_buffer().
#ifdef TFR_PROFILE_STAMPS
__device__ unsigned long long* g_prof_buf = nullptr;
#define TFR_STAMP(i) do { if (lane == 0) prof_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TFR_STAMP(i) do { } while (0)
#endif

// CT: the lanes per row as a compile-time constant (2, the default geometry; 0 = the run-time value).  Round 5: with a
// run-time C the column index `c + it * C` of every LDS read of the sweeps was re-formed with VALU adds -- one v_add_u32 per
// trip of the forward sweep and FOUR per trip of the backward sweep (two reads, two address forms), 5 of the 24 vector
// instructions of a trip pair, in a kernel whose VALU pipe is busy 102 % of the time (profiles/r05_pmc.txt); with a constant
// stride the unrolled reads take immediate offsets from one base.
template <int IPL, int CT>
__global__ __launch_bounds__(64) void approx_ndcg_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ inv_log1p, const float* __restrict__ list_scale, int L, int Lp,
    float temperature, int Crt, float* __restrict__ loss_out, float* __restrict__ weight_out,
    float* __restrict__ dlogits_out, int max_runs, int metric, const int* __restrict__ order, int pair_rcp,
    float* __restrict__ loss_sum, unsigned int* __restrict__ ticket, int B, int fast_labels) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* X = reinterpret_cast<float*>(smem_raw);   // [Lp] compact x (pad -inf)
  float* E = X + Lp;                               // [Lp] exp(x - m)
  float* F = E + Lp;                               // [Lp] exp(m - x)   (pad +inf)
  float* A = F + Lp;                               // [Lp] dloss/drank  (pad 0)
  float* G = A + Lp;                               // [Lp] compact gain
  int* CI = reinterpret_cast<int*>(G + Lp);        // [Lp] compact -> original index
  const int lane = threadIdx.x;
  const int C = CT ? CT : Crt;
  const int b = order ? order[blockIdx.x] : blockIdx.x;      // longest-first launch order (tfr_list_order_i32)
  const size_t base = (size_t)b * L;
  constexpr float kLn2 = 0.69314718055994530942f;
#ifdef TFR_PROFILE_STAMPS
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  TFR_STAMP(0);

  // ---- 1. load + clean; label / logit statistics (wave reductions).
  float x[IPL], g[IPL];
  bool v[IPL];
  float lmax = -INFINITY, lsum = 0.f, xmin = INFINITY, xmax = -INFINITY;
  unsigned gbits = 0u;                               // grades present (bit l), when every cleaned label is an integer 0 .. 31
  bool small_int = true;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = lane + 64 * r;
    x[r] = 0.f; g[r] = 0.f; v[r] = false;
    if (e < L) {
      const float lab = labels[base + e];
      x[r] = logits[base + e] / temperature;
      v[r] = mask ? (mask[base + e] != 0) : (lab >= 0.0f);
      g[r] = v[r] ? lab : 0.0f;                      // cleaned label for now
      lmax = fmaxf(lmax, g[r]); lsum += g[r];
      if (v[r]) { xmin = fminf(xmin, x[r]); xmax = fmaxf(xmax, x[r]); }
      const int li = (int)g[r];
      const bool ok = g[r] >= 0.0f && g[r] < 32.0f && (float)li == g[r];
      small_int = small_int && ok;
      gbits |= ok ? (1u << li) : 0u;
    }
  }
  xmin = wave_min_u(xmin); xmax = wave_max_u(xmax);
  // Graded relevance labels -- small non-negative integers -- are the common case (round 4): the label maximum and the
  // sign of the label sum come out of ONE wave-wide OR of the grade bits, and the ideal DCG below needs no reduction per
  // run.  (In-kernel stamps: the statistics and the run-length ideal DCG were 17 % + 25 % of a wave's lifetime, ten dependent
  // DPP reduction chains for what is five grade counts.)  Anything else -- fractional, negative or huge labels, ApproxMRR
  // (which needs the label sum itself) -- takes the general reductions.
  const bool int_path = fast_labels && metric == TFR_APPROX_NDCG && !__ballot(!small_int);
  unsigned present = 0u;
  bool nonzero;
  if (int_path) {
    present = wave_or_u(gbits);
    nonzero = (present >> 1) != 0u;
    lmax = nonzero ? (float)(31 - __builtin_clz(present)) : 1e-10f;
  } else {
    lmax = wave_max_u(lmax); lsum = wave_sum_u(lsum);
    nonzero = lsum > 0.0f;
    if (!nonzero) lmax = 1e-10f;
  }

  TFR_STAMP(1);
  // ---- 2. gains and normaliser.  NDCG: safe gains + inverse ideal DCG (:33-49, :109-134);
  // MRR (ApproxMRRLoss, :1606-1632): the cleaned labels themselves, normalised by their sum.
  float inv_max_dcg;
  if (int_path) {
    // gain of grade l = 2^(l - lmax) - 2^-lmax, exact (ldexp), 0 for grade 0 and for every item of a list without a
    // relevant one (the reference's 1e-10 labels give 2^0 - 2^-1e-10 = 0 in fp32).  Ideal DCG: the sorted gains are runs
    // of equal grades, highest first; position e of run [pos, pos + c) carries that grade's gain -- every lane looks its
    // positions up against the (wave-uniform) run boundaries and ONE reduction adds gain(e) * discount(e).
    const int lmi = nonzero ? 31 - __builtin_clz(present) : 0;
    const float g0 = __builtin_amdgcn_ldexpf(1.0f, -lmi);
    float tbl[IPL], gpos[IPL];
    int li_[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = lane + 64 * r;
      tbl[r] = (e < L) ? inv_log1p[e] : 0.0f;
      li_[r] = (e < L) ? (int)g[r] : -1;
      g[r] = (e < L && nonzero) ? __builtin_amdgcn_ldexpf(1.0f, li_[r] - lmi) - g0 : 0.0f;
      gpos[r] = 0.0f;
    }
    int pos = 0;
    for (unsigned rest = nonzero ? (present & ~1u) : 0u; rest != 0u;) {     // grades >= 1, highest first (scalar loop)
      const int gq = 31 - __builtin_clz(rest);
      rest &= ~(1u << gq);
      const float val = __builtin_amdgcn_ldexpf(1.0f, gq - lmi) - g0;
      int c = 0;
#pragma unroll
      for (int r = 0; r < IPL; ++r) c += __popcll(__ballot(li_[r] == gq));
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int e = lane + 64 * r;
        gpos[r] = (e >= pos && e < pos + c) ? val : gpos[r];
      }
      pos += c;
    }
    float t = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) t = __builtin_fmaf(gpos[r], tbl[r], t);
    const float idcg = wave_sum_u(t);
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
  } else if (metric == TFR_APPROX_MRR) {
#pragma unroll
    for (int r = 0; r < IPL; ++r) g[r] = (lane + 64 * r < L) ? (nonzero ? g[r] : 1e-10f) : 0.f;
    const float denom = nonzero ? lsum : (float)L * 1e-10f;
    inv_max_dcg = 1.0f / denom;
  } else {
    const float g0 = exp2f(-lmax);
    uint32_t sk[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = lane + 64 * r;
      float gg = 0.f;
      if (e < L) gg = exp2f((nonzero ? g[r] : 1e-10f) - lmax) - g0;
      g[r] = gg;
      sk[r] = __float_as_uint(gg);
    }
    float idcg = 0.f;
    float tbl[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) tbl[r] = (lane + 64 * r < L) ? inv_log1p[lane + 64 * r] : 0.0f;
    // graded labels: a handful of distinct gains -> run-length form, no sort
    if (!wave_sorted_dot_runs<IPL>(g, tbl, lane, L, max_runs, idcg)) {
      wave_sort_desc<IPL>(sk, lane);
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < IPL; ++r) t += __uint_as_float(sk[r]) * tbl[r];
      idcg = wsum(t);
    }
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
  }

  TFR_STAMP(2);
  // ---- 3. stable compaction of valid items into LDS (order = original index).
  int n = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const unsigned long long bal = __ballot(v[r]);
    if (v[r]) {
      const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
      X[pos] = x[r]; G[pos] = g[r]; CI[pos] = lane + 64 * r;
    }
    n += __popcll(bal);
  }
  __syncthreads();
  const int n4 = (n + 3) >> 2;
  const int iters = (n4 + C - 1) / C;                 // uniform trip count of the pair sweeps: every lane of a row
  const int n4p = iters * C;                          // group walks `iters` float4 column groups; the padding
  const float m = 0.5f * (xmax + xmin);               // groups hold F = +inf (sigma = 0) and A = 0
  const bool fast = (xmax - xmin) <= kFastRange;
  const bool pair2 = pair_rcp && (xmax - xmin) <= kPairRange;      // two sigmoids per reciprocal in the forward sweep
  const bool pair2b = pair2 && pair_rcp >= 2;                       // ... and in the backward sweep (round 5)
  // padding columns: F = +inf (sigma = 0); on the pair path F = 0 (a = 1, sigma = 1 EXACTLY when both columns of a
  // pair are padding, within an ulp next to a real column) and the padding count is taken off the rank afterwards
  const float f_pad = pair2 ? 0.0f : INFINITY;
  for (int i = lane; i < n4p * 4; i += 64) {
    float e = 0.f, f = f_pad;
    if (i < n) {
      const float xv = X[i];
      const float t_hi = xv - m;
      const float bb = t_hi - xv;
      const float t_lo = (xv - (t_hi - bb)) + (-m - bb);
      e = exp_df_fast(t_hi, t_lo);
      f = exp_df_fast(-t_hi, -t_lo);
    } else {
      X[i] = -INFINITY;
    }
    E[i] = e; F[i] = f; A[i] = 0.f;
  }
  __syncthreads();

  TFR_STAMP(3);
  // ---- 4. ranks + loss terms.  Row = C adjacent lanes; 64/C rows per pass.
  const int rows_per_pass = 64 / C;
  const int c = lane % C, rsub = lane / C;
  const float4* F4 = reinterpret_cast<const float4*>(F);
  const float4* X4 = reinterpret_cast<const float4*>(X);
  const float4* A4 = reinterpret_cast<const float4*>(A);
  const float pad_cols = pair2 ? (float)(n4p * 4 - n) : 0.0f;       // every padding column added sigma = 1 to a row's sum
  float dcg = 0.f;
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (pair2) {
      // columns (0, 2) and (1, 3) of a float4 share a reciprocal: d01 * d23 = {a0 a2, a1 a3}, r = 1 / that,
      // {s0, s1} = d23 * r, {s2, s3} = d01 * r -- 2 v_pk_fma + 1 v_pk_mul + 2 v_rcp + 2 v_pk_fma per 4 pairs
      const float Ei = (active ? E[row] : 0.f) * kPairScale;
      const f32x2 Ei2 = {Ei, Ei}, one2 = {kPairScale, kPairScale};
      f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
        const float4 f = F4[c + it * C];
        const f32x2 d01 = __builtin_elementwise_fma(Ei2, f32x2{f.x, f.y}, one2);
        const f32x2 d23 = __builtin_elementwise_fma(Ei2, f32x2{f.z, f.w}, one2);
        const f32x2 pq = d01 * d23;
        const f32x2 r = {fast_rcp(pq.x), fast_rcp(pq.y)};
        acc01 = __builtin_elementwise_fma(d23, r, acc01);
        acc23 = __builtin_elementwise_fma(d01, r, acc23);
      }
      a0 = ((acc01.x + acc01.y) + (acc23.x + acc23.y)) * kPairScale;    // (one scaling per row; a1 .. a3 stay 0)
    } else if (fast) {
      // packed fp32 (v_pk_fma_f32 / v_pk_add_f32): 2 + 2 full-rate instructions and 4 v_rcp_f32 per 4 pairs
      const float Ei = active ? E[row] : 0.f;
      const f32x2 Ei2 = {Ei, Ei}, one2 = {1.0f, 1.0f};
      f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
        const float4 f = F4[c + it * C];
        const f32x2 d01 = __builtin_elementwise_fma(Ei2, f32x2{f.x, f.y}, one2);
        const f32x2 d23 = __builtin_elementwise_fma(Ei2, f32x2{f.z, f.w}, one2);
        acc01 += f32x2{fast_rcp(d01.x), fast_rcp(d01.y)};
        acc23 += f32x2{fast_rcp(d23.x), fast_rcp(d23.y)};
      }
      a0 = acc01.x; a1 = acc01.y; a2 = acc23.x; a3 = acc23.y;
    } else {
      const float xi = active ? X[row] : 0.f;
      for (int gI = c; gI < n4; gI += C) {
        const float4 xx = X4[gI];
        a0 += sig_slow(xi, xx.x); a1 += sig_slow(xi, xx.y);
        a2 += sig_slow(xi, xx.z); a3 += sig_slow(xi, xx.w);
      }
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc = lanes_sum_c(acc, C);
    if (active && c == 0) {
      const float r = (acc - pad_cols) + 0.5f;                           // (pad_cols = 0 off the pair path)
      const float gg = G[row];
      if (metric == TFR_APPROX_MRR) {                                 // term = l / r, d term / d r = -l / r^2
        const float ir = 1.0f / r;                                    // (per row, not per pair: exact division)
        dcg = __builtin_fmaf(gg, ir, dcg);
        A[row] = (gg * inv_max_dcg) / (r * r);
      } else {
        const float lr = __builtin_amdgcn_logf(1.0f + r) * kLn2;      // log1p(r), r >= 1
        const float ilr = fast_rcp(lr);
        dcg = __builtin_fmaf(gg, ilr, dcg);
        A[row] = (gg * inv_max_dcg) * ilr * ilr * fast_rcp(1.0f + r);   // not read until step 5
      }
    }
  }
  dcg = wave_sum_u(dcg);
  const float list_loss = -(dcg * inv_max_dcg);
  if (lane == 0) {
    if (!loss_sum) loss_out[b] = list_loss;
    weight_out[b] = nonzero ? 1.0f : 0.0f;
  }
  // sum_b loss_b * list_scale_b (the scalar the reduced loss returns): by the last wave to get here, while the others
  // are in their backward sweeps (the helper stores loss_out[b] itself, written through)
  if (loss_sum) grid_weighted_sum_last(loss_out, b, list_loss, list_scale, B, loss_sum, ticket, lane);
  TFR_STAMP(4);
  if (!dlogits_out) return;
  __syncthreads();

  // ---- 5. backward.
  const float gscale = (list_scale ? list_scale[b] : 1.0f) * (1.0f / temperature);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = lane + 64 * r;
    if (e < L && !v[r]) dlogits_out[base + e] = 0.0f;
  }
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    const float ak = active ? A[row] : 0.f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (pair2b) {
      // two sigmoids per reciprocal here too (round 5; the forward sweep's trick, same scaling and range condition):
      // t = d_partner' * rcp(d0' d2') = sigma / c, sigma = c t, sigma' = sigma - sigma^2 -- per four pairs 13 packed
      // instructions + 2 v_rcp_f32 instead of 8 + 4 (a reciprocal costs 3.5 plain instructions).  Padding columns hold
      // F = 0 on this path: sigma = 1 exactly, sigma' = 0; a product d0' d2' beyond FLT_MAX has both sigmas below 2^-80.
      const float Ek = (active ? E[row] : 0.f) * kPairScale;
      const f32x2 Ek2 = {Ek, Ek}, one2 = {kPairScale, kPairScale}, ak2 = {ak, ak};
      f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
        const float4 f = F4[c + it * C];
        const float4 aj = A4[c + it * C];
        const f32x2 d01 = __builtin_elementwise_fma(Ek2, f32x2{f.x, f.y}, one2);
        const f32x2 d23 = __builtin_elementwise_fma(Ek2, f32x2{f.z, f.w}, one2);
        const f32x2 pq = d01 * d23;
        const f32x2 r = {fast_rcp(pq.x), fast_rcp(pq.y)};
        const f32x2 s01 = (d23 * r) * one2;
        const f32x2 s23 = (d01 * r) * one2;
        const f32x2 w01 = __builtin_elementwise_fma(-s01, s01, s01);          // sigma' = s - s^2
        const f32x2 w23 = __builtin_elementwise_fma(-s23, s23, s23);
        acc01 = __builtin_elementwise_fma(f32x2{aj.x, aj.y} - ak2, w01, acc01);
        acc23 = __builtin_elementwise_fma(f32x2{aj.z, aj.w} - ak2, w23, acc23);
      }
      a0 = acc01.x; a1 = acc01.y; a2 = acc23.x; a3 = acc23.y;
    } else if (fast) {
      const float Ek = active ? E[row] : 0.f;
      const f32x2 Ek2 = {Ek, Ek}, one2 = {1.0f, 1.0f}, ak2 = {ak, ak};
      f32x2 acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
      for (int it = 0; it < iters; ++it) {
        const float4 f = F4[c + it * C];
        const float4 aj = A4[c + it * C];
        const f32x2 d01 = __builtin_elementwise_fma(Ek2, f32x2{f.x, f.y}, one2);
        const f32x2 d23 = __builtin_elementwise_fma(Ek2, f32x2{f.z, f.w}, one2);
        const f32x2 s01 = {fast_rcp(d01.x), fast_rcp(d01.y)};
        const f32x2 s23 = {fast_rcp(d23.x), fast_rcp(d23.y)};
        const f32x2 w01 = __builtin_elementwise_fma(-s01, s01, s01);          // sigma' = s - s^2
        const f32x2 w23 = __builtin_elementwise_fma(-s23, s23, s23);
        acc01 = __builtin_elementwise_fma(f32x2{aj.x, aj.y} - ak2, w01, acc01);
        acc23 = __builtin_elementwise_fma(f32x2{aj.z, aj.w} - ak2, w23, acc23);
      }
      a0 = acc01.x; a1 = acc01.y; a2 = acc23.x; a3 = acc23.y;
    } else {
      const float xk = active ? X[row] : 0.f;
      for (int gI = c; gI < n4; gI += C) {
        const float4 xx = X4[gI];
        const float4 aj = A4[gI];
        const float s0 = sig_slow(xk, xx.x), s1 = sig_slow(xk, xx.y);
        const float s2 = sig_slow(xk, xx.z), s3 = sig_slow(xk, xx.w);
        a0 = __builtin_fmaf(aj.x - ak, __builtin_fmaf(-s0, s0, s0), a0);
        a1 = __builtin_fmaf(aj.y - ak, __builtin_fmaf(-s1, s1, s1), a1);
        a2 = __builtin_fmaf(aj.z - ak, __builtin_fmaf(-s2, s2, s2), a2);
        a3 = __builtin_fmaf(aj.w - ak, __builtin_fmaf(-s3, s3, s3), a3);
      }
    }
    float acc = (a0 + a1) + (a2 + a3);
    acc = lanes_sum_c(acc, C);
    if (active && c == 0) dlogits_out[base + CI[row]] = acc * gscale;
  }
  TFR_STAMP(5);
#ifdef TFR_PROFILE_STAMPS
  if (lane == 0 && g_prof_buf) {
    prof_t[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    prof_t[7] = (unsigned long long)n;
    for (int i = 0; i < 8; ++i) g_prof_buf[(size_t)b * 8 + i] = prof_t[i];
  }
#endif
}

int env_int(const char* name, int dflt);

template <int IPL>
int launch_wave(const float* logits, const float* labels, const uint8_t* mask, const float* inv_log1p,
                const float* list_scale, int B, int L, float temperature, int C, float* loss_out,
                float* weight_out, float* dlogits_out, hipStream_t stream, int metric, const int* order,
                float* loss_sum, unsigned int* ticket) {
  const int Lp = ((L + 3) / 4 + C) * 4;           // room for the padding column groups of the uniform sweeps
  const size_t lds = (size_t)Lp * 4 * 6;
  static const int max_runs = env_int("TFR_APPROX_MAX_RUNS", 8);   // 0 forces the sort (A/B measurements)
  // 0: one reciprocal per pair everywhere (round 3); 1 (default): two sigmoids per reciprocal in the forward sweep (round 4);
  // 2: in the backward sweep too -- built and measured in round 5: SLOWER (kernel 0.1057 -> 0.1103 ms, twice each on one box:
  // the 5 extra packed multiplies of a trip cost more than the 2 reciprocals they replace), kept behind the switch
  static const int pair_rcp = env_int("TFR_APPROX_PAIR_RCP", 1);
  static const int int_labels = env_int("TFR_APPROX_INT_LABELS", 1);   // 0: label statistics / ideal DCG by the general reductions
  if (C == 2)
    hipLaunchKernelGGL((approx_ndcg_wave_kernel<IPL, 2>), dim3(B), dim3(64), lds, stream, logits, labels, mask,
                       inv_log1p, list_scale, L, Lp, temperature, C, loss_out, weight_out, dlogits_out, max_runs, metric, order,
                       pair_rcp, loss_sum, ticket, B, int_labels);
  else
    hipLaunchKernelGGL((approx_ndcg_wave_kernel<IPL, 0>), dim3(B), dim3(64), lds, stream, logits, labels, mask,
                       inv_log1p, list_scale, L, Lp, temperature, C, loss_out, weight_out, dlogits_out, max_runs, metric, order,
                       pair_rcp, loss_sum, ticket, B, int_labels);
  return (int)hipGetLastError();
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

static int approx_dispatch(int metric, const float* logits, const float* labels, const uint8_t* mask,
                           const float* inv_log1p, const float* list_scale, int B, int L,
                           float temperature, int lanes_per_row, float* loss_out,
                           float* weight_out, float* dlogits_out, const int* order, void* stream,
                           float* loss_sum = nullptr, unsigned int* ticket = nullptr) {
  if (!logits || !labels || (!inv_log1p && metric == TFR_APPROX_NDCG) || !loss_out || !weight_out || B < 0 || L <= 0)
    return TFR_EINVAL;
  if (!(temperature > 0.0f)) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if ((loss_sum != nullptr) != (ticket != nullptr)) return TFR_EINVAL;
  if (B == 0) {
    if (loss_sum) return (int)hipMemsetAsync(loss_sum, 0, sizeof(float), (hipStream_t)stream);
    return TFR_OK;
  }
  static const int env_threads = env_int("TFR_APPROX_THREADS", 0);
  static const int env_lanes = env_int("TFR_APPROX_LANES", 0);
  static const int env_wave = env_int("TFR_APPROX_WAVE", 1);      // 0 forces the block kernel
  static const int env_wave_min_b = env_int("TFR_APPROX_WAVE_MIN_B", 2048);
  int C = lanes_per_row > 0 ? lanes_per_row : (env_lanes > 0 ? env_lanes : 2);
  if (C > 64 || (C & (C - 1))) return TFR_EINVAL;
  // Wave-per-list kernel: always for L <= 256; for L <= 1024 only when the batch
  // alone fills the chip with single waves (otherwise several waves share a list).
  if (env_wave && env_threads == 0 && (L <= 256 || (L <= 1024 && B >= env_wave_min_b))) {
    hipStream_t st = (hipStream_t)stream;
    if (L <= 64) return launch_wave<1>(logits, labels, mask, inv_log1p, list_scale, B, L, temperature, C, loss_out, weight_out, dlogits_out, st, metric, order, loss_sum, ticket);
    if (L <= 128) return launch_wave<2>(logits, labels, mask, inv_log1p, list_scale, B, L, temperature, C, loss_out, weight_out, dlogits_out, st, metric, order, loss_sum, ticket);
    if (L <= 256) return launch_wave<4>(logits, labels, mask, inv_log1p, list_scale, B, L, temperature, C, loss_out, weight_out, dlogits_out, st, metric, order, loss_sum, ticket);
    if (L <= 512) return launch_wave<8>(logits, labels, mask, inv_log1p, list_scale, B, L, temperature, C, loss_out, weight_out, dlogits_out, st, metric, order, loss_sum, ticket);
    return launch_wave<16>(logits, labels, mask, inv_log1p, list_scale, B, L, temperature, C, loss_out, weight_out, dlogits_out, st, metric, order, loss_sum, ticket);
  }
  int T = env_threads > 0 ? env_threads : (L <= 128 ? 64 : (L <= 512 ? 128 : 512));
  if (T % 64 || T > 1024) return TFR_EINVAL;
  const int Lp = ((L + 3) / 4) * 4 + 4;
  const int P = pow2_ceil(L < 2 ? 2 : L);
  const size_t lds = smem_bytes(Lp, P);
  if (lds > 160 * 1024) return TFR_ETOOLARGE;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(approx_ndcg_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(approx_ndcg_kernel, dim3(B), dim3(T), lds, (hipStream_t)stream, logits, labels,
                     mask, inv_log1p, list_scale, L, Lp, P, temperature, C, loss_out, weight_out,
                     dlogits_out, metric, order, loss_sum, ticket, B);
  return (int)hipGetLastError();
}


extern "C" int tfr_approx_ndcg_f32(const float* logits, const float* labels, const uint8_t* mask,
                                   const float* inv_log1p, const float* list_scale, int B, int L,
                                   float temperature, int lanes_per_row, float* loss_out,
                                   float* weight_out, float* dlogits_out, const int32_t* list_order,
                                   void* stream) {
  return approx_dispatch(TFR_APPROX_NDCG, logits, labels, mask, inv_log1p, list_scale, B, L, temperature,
                         lanes_per_row, loss_out, weight_out, dlogits_out, list_order, stream);
}

extern "C" int tfr_grid_sum_state_ints(void) { return kGridSumStateInts; }

extern "C" int tfr_approx_ndcg_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                                       const float* inv_log1p, const float* list_scale, int B, int L,
                                       float temperature, int lanes_per_row, float* loss_out,
                                       float* weight_out, float* dlogits_out, const int32_t* list_order,
                                       float* loss_sum_out, uint32_t* ticket, void* stream) {
  if (!loss_sum_out || !ticket) return TFR_EINVAL;
  return approx_dispatch(TFR_APPROX_NDCG, logits, labels, mask, inv_log1p, list_scale, B, L, temperature,
                         lanes_per_row, loss_out, weight_out, dlogits_out, list_order, stream, loss_sum_out, ticket);
}

extern "C" int tfr_approx_mrr_f32(const float* logits, const float* labels, const uint8_t* mask,
                                  const float* list_scale, int B, int L, float temperature,
                                  float* loss_out, float* weight_out, float* dlogits_out,
                                  const int32_t* list_order, void* stream) {
  return approx_dispatch(TFR_APPROX_MRR, logits, labels, mask, nullptr, list_scale, B, L, temperature, 0,
                         loss_out, weight_out, dlogits_out, list_order, stream);
}

#ifdef TFR_PROFILE_STAMPS
extern "C" int tfr_prof_set_buffer(void* device_u64_buffer) {
  unsigned long long* p = (unsigned long long*)device_u64_buffer;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_buf), &p, sizeof(p));
}
#endif
