// Softmax (ListNet) loss fwd+bwd and the Gumbel-softmax sampler fwd / bwd for
// gfx950.  These are the O(L) members of the hot path: one workgroup per list,
// a handful of 64-wide wave reductions, every HBM byte touched exactly once.
//
// Reference behaviour restated (losses_impl.py): SoftmaxLoss.precompute
// :1122-1137, _compute_unreduced_loss_impl :1139-1158, AbstractDCGLambdaWeight
// .individual_weights :281-296, _compute_ranks :483-500, inverse_max_dcg
// :109-134, GumbelSampler.sample :556-644, _sample_gumbel :647-649.
#include "common.h"
#include <stdlib.h>
#include "../../include/tfr_hip.h"

using namespace tfr;

namespace {

constexpr float kLogEps10 = -23.02585092994045684f;   // ln(1e-10)  (losses_impl.py:28,1130)
constexpr float kLogEps20 = -46.0517018598809137f;    // ln(1e-20)  (losses_impl.py:603)

struct SmArgs {
  const float* logits; const float* labels; const uint8_t* mask; const float* item_weights;
  int weights_per_list; int lambda_kind; int topn; int normalized; int gain_kind;
  const float* gains; const float* discount; int L; int Lp; int P; float temperature;
  float* loss; float* weight; float* dlogits;
  float poly_eps;                 // PolyOneSoftmaxLoss (losses_impl.py:1200-1247): loss += eps * (1 - sum_i p_i softmax_i)
  GridSum sum;                    // round 5: sum_b loss_b * weight_b from the same launch (tfr_softmax_loss_sum_f32); out == NULL: off
};

__global__ void softmax_loss_kernel(const SmArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P] (lambda only)
  const bool dcg_lambda = a.lambda_kind == TFR_LAMBDA_DCG;
  float* fbase = reinterpret_cast<float*>(keys + (dcg_lambda ? a.P : 0));
  float* Z = fbase;                 // [Lp] logits for softmax
  float* Y = fbase + a.Lp;          // [Lp] labels for softmax (un-normalised)
  float* Gr = fbase + 2 * a.Lp;     // [Lp] gain (lambda only)
  int* Rk = reinterpret_cast<int*>(fbase + 3 * a.Lp);   // [Lp] ranks (lambda only)
  uint8_t* MV = reinterpret_cast<uint8_t*>(fbase + 4 * a.Lp);

  const int tid = threadIdx.x, T = blockDim.x;
  const int b = blockIdx.x, L = a.L, P = a.P;
  const size_t base = (size_t)b * L;
  const int topn = (a.topn <= 0 || a.topn > L) ? L : a.topn;

  // ---- precompute (:1122-1137)
  for (int i = tid; i < L; i += T) {
    const float lab = a.labels[base + i];
    const bool mv = a.mask ? (a.mask[base + i] != 0) : (lab >= 0.0f);
    const float x = a.logits[base + i] / a.temperature;
    MV[i] = mv;
    Z[i] = mv ? x : kLogEps10;
    Y[i] = mv ? lab : 0.0f;
    if (dcg_lambda) keys[i] = make_sort_key(mv, x, 0, i);
  }
  if (dcg_lambda) {
    for (int i = L + tid; i < P; i += T) keys[i] = 0;
    block_bitonic_sort_desc(keys, P);                 // ranks by logits (:483-500)
    for (int p = tid; p < L; p += T) Rk[sort_key_index(keys[p])] = p + 1;
    __syncthreads();
    // individual_weights (:281-296): (gain(clean label) [* invMaxDCG]) * discount(rank)
    for (int i = tid; i < P; i += T) {
      uint64_t key = 0;
      if (i < L) {
        const float yl = Y[i];
        const float labc = (yl >= 0.0f) ? yl : 0.0f;
        float g;
        if (a.gain_kind == TFR_GAIN_CUSTOM) g = a.gains[base + i];
        else if (a.gain_kind == TFR_GAIN_POW2M1) g = gain_pow2m1(labc);
        else g = labc;
        Gr[i] = g;
        key = ((uint64_t)float_to_ordered(labc) << 32) | (uint64_t)__float_as_uint(g);
      }
      keys[i] = key;
    }
    float inv = 1.0f;
    if (a.normalized) {
      block_bitonic_sort_desc(keys, P);               // ideal order (:109-134)
      float idcg = 0.f;
      for (int p = tid; p < topn; p += T)
        idcg += __uint_as_float((uint32_t)(keys[p] & 0xffffffffull)) * a.discount[p];
      idcg = block_sum(idcg, red);
      inv = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
    }
    __syncthreads();
    for (int i = tid; i < L; i += T) {
      float g = Gr[i];
      if (a.normalized) g = g * inv;
      Y[i] = g * a.discount[Rk[i] - 1];
    }
  }
  __syncthreads();
  // item / list weights (:1135-1136)
  if (a.item_weights) {
    const float wl = a.weights_per_list ? a.item_weights[b] : 1.0f;
    for (int i = tid; i < L; i += T)
      Y[i] = Y[i] * (a.weights_per_list ? wl : a.item_weights[base + i]);
  }
  __syncthreads();

  // ---- _compute_unreduced_loss_impl (:1139-1158)
  float lsum = 0.f, zmax = -INFINITY;
  for (int i = tid; i < L; i += T) { lsum += Y[i]; zmax = fmaxf(zmax, Z[i]); }
  lsum = block_sum(lsum, red);
  zmax = block_max(zmax, red);
  const bool nonzero = lsum > 0.0f;
  float psum = 0.f, esum = 0.f;
  for (int i = tid; i < L; i += T) {
    float y = nonzero ? Y[i] : 1e-10f;
    y = MV[i] ? y : 0.0f;
    Y[i] = y;
    psum += y;
    esum += expf(Z[i] - zmax);
  }
  psum = block_sum(psum, red);
  esum = block_sum(esum, red);
  const float lse = logf(esum);
  float loss = 0.f, ptot = 0.f, pt = 0.f;
  for (int i = tid; i < L; i += T) {
    const float p = (psum != 0.0f) ? (Y[i] / psum) : 0.0f;   // divide_no_nan
    loss += p * (lse - (Z[i] - zmax));
    ptot += p;
    if (a.poly_eps != 0.0f) pt += p * (expf(Z[i] - zmax) / esum);
    Y[i] = p;
  }
  loss = block_sum(loss, red);
  ptot = block_sum(ptot, red);
  if (a.poly_eps != 0.0f) {                                    // wave-uniform
    pt = block_sum(pt, red);
    loss += a.poly_eps * (1.0f - pt);
  }
  if (tid == 0) { a.loss[b] = loss; a.weight[b] = lsum; }
  if (a.sum.out) { if (tid < 64) grid_sum_contribute(a.sum, b, loss * lsum, tid); }    // (wave 0, converged; values are block-uniform)
  else if (a.sum.vec && tid == 0) a.sum.vec[b] = loss * lsum;                          // partials mode: the product, for a short tfr_list_dot_f32
  if (!a.dlogits) return;
  // ---- backward: d(weight * loss)/d logits_k = (w/T) (sum_p * softmax_k - p_k), valid k;
  // poly-1 adds  -eps * softmax_k * (p_k - pt).
  for (int i = tid; i < L; i += T) {
    float g = 0.f;
    if (MV[i]) {
      const float sm = expf(Z[i] - zmax) / esum;
      float d = ptot * sm - Y[i];
      if (a.poly_eps != 0.0f) d -= a.poly_eps * sm * (Y[i] - pt);
      g = lsum * (d / a.temperature);
    }
    a.dlogits[base + i] = g;
  }
}

// ------------------------------------------------------------------ Gumbel
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return __umulhi(a, b); }

// Philox4x32-10 (Salmon et al. 2011); returns the first output word.
__device__ __forceinline__ uint32_t philox_first(uint64_t ctr_lo, uint64_t ctr_hi, uint64_t key) {
  uint32_t c0 = (uint32_t)ctr_lo, c1 = (uint32_t)(ctr_lo >> 32);
  uint32_t c2 = (uint32_t)ctr_hi, c3 = (uint32_t)(ctr_hi >> 32);
  uint32_t k0 = (uint32_t)key, k1 = (uint32_t)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return c0;
}

__global__ void gumbel_sample_kernel(const float* __restrict__ logits, const float* __restrict__ labels,
                                     const uint8_t* __restrict__ mask, const float* __restrict__ uniform,
                                     uint64_t seed, uint64_t offset, int S, int L, int Lp,
                                     float gumbel_temperature, float* __restrict__ sampled_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);
  float* Z = reinterpret_cast<float*>(smem_raw + 128);
  const int tid = threadIdx.x, T = blockDim.x;
  const int bs = blockIdx.x;               // b * S + s
  const int b = bs / S;
  const size_t ibase = (size_t)b * L, obase = (size_t)bs * L;
  float zmax = -INFINITY;
  for (int i = tid; i < L; i += T) {
    const float lab = labels[ibase + i];
    const bool v = mask ? (mask[ibase + i] != 0) : (lab >= 0.0f);
    float u;
    if (uniform) u = uniform[obase + i];
    else u = (float)(philox_first(obase + i, offset, seed) >> 8) * (1.0f / 16777216.0f);
    const float g = -logf(-logf(u + 1e-20f) + 1e-20f);            // :647-649
    const float z = v ? ((logits[ibase + i] + g) / gumbel_temperature) : kLogEps20;
    Z[i] = z;
    zmax = fmaxf(zmax, z);
  }
  zmax = block_max(zmax, red);
  float esum = 0.f;
  for (int i = tid; i < L; i += T) esum += expf(Z[i] - zmax);
  esum = block_sum(esum, red);
  for (int i = tid; i < L; i += T)
    sampled_out[obase + i] = logf(expf(Z[i] - zmax) / esum + 1e-20f);   // :605
}

__global__ void gumbel_sample_bwd_kernel(const float* __restrict__ sampled, const float* __restrict__ labels,
                                         const uint8_t* __restrict__ mask, const float* __restrict__ upstream,
                                         int S, int L, float gumbel_temperature,
                                         float* __restrict__ dlogits_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);
  float* ACC = reinterpret_cast<float*>(smem_raw + 128);
  const int tid = threadIdx.x, T = blockDim.x;
  const int b = blockIdx.x;
  const size_t ibase = (size_t)b * L;
  for (int i = tid; i < L; i += T) ACC[i] = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t obase = ((size_t)b * S + s) * L;
    // out_k = log(p_k + eps): d out_k / d z_j = c_k (delta_kj - p_j), c_k = p_k / (p_k + eps)
    float dot = 0.f;
    for (int i = tid; i < L; i += T) {
      const float e = expf(sampled[obase + i]);           // p + eps
      const float p = fmaxf(e - 1e-20f, 0.0f);
      dot += upstream[obase + i] * (p / e);
    }
    dot = block_sum(dot, red);
    for (int i = tid; i < L; i += T) {
      const float e = expf(sampled[obase + i]);
      const float p = fmaxf(e - 1e-20f, 0.0f);
      ACC[i] += upstream[obase + i] * (p / e) - p * dot;
    }
  }
  for (int i = tid; i < L; i += T) {
    const float lab = labels[ibase + i];
    const bool v = mask ? (mask[ibase + i] != 0) : (lab >= 0.0f);
    dlogits_out[ibase + i] = v ? (ACC[i] / gumbel_temperature) : 0.0f;
  }
}

// Wave-per-row forms of the Gumbel sampler (list_size <= 64 * IPL): one wavefront per sampled row (b, s) forward,
// per list b backward; values in registers, wave reductions, four rows per workgroup.  Same arithmetic and the same
// Philox counters (one per output element) as the workgroup kernels above, which stay for longer lists.
template <int IPL>
__global__ __launch_bounds__(256) void gumbel_sample_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ uniform, uint64_t seed, uint64_t offset, int BS, int S, int L,
    float gumbel_temperature, float* __restrict__ sampled_out, const unsigned long long* __restrict__ step,
    float* __restrict__ labels_out) {
  const int lane = threadIdx.x & 63;
  const int bs = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (bs >= BS) return;
  if (step) offset += *step;                                 // round 6: the Philox offset of a replayed step lives on the device
  const int b = bs / S;
  const size_t ibase = (size_t)b * L, obase = (size_t)bs * L;
  float z[IPL];
  float zmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    z[r] = -INFINITY;
    if (i < L) {
      const float lab = labels[ibase + i];
      if (labels_out) labels_out[obase + i] = lab;           // the labels of the S copies of a list (was an expand + copy launch)
      const bool v = mask ? (mask[ibase + i] != 0) : (lab >= 0.0f);
      float u;
      if (uniform) u = uniform[obase + i];
      else u = (float)(philox_first(obase + i, offset, seed) >> 8) * (1.0f / 16777216.0f);
      const float g = -logf(-logf(u + 1e-20f) + 1e-20f);            // :647-649
      z[r] = v ? ((logits[ibase + i] + g) / gumbel_temperature) : kLogEps20;
      zmax = fmaxf(zmax, z[r]);
    }
  }
  zmax = wave_max_u(zmax);
  float esum = 0.f, e[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) { e[r] = (lane + 64 * r < L) ? expf(z[r] - zmax) : 0.0f; esum += e[r]; }
  esum = wave_sum_u(esum);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    if (i < L) sampled_out[obase + i] = logf(e[r] / esum + 1e-20f);   // :605
  }
}

template <int IPL>
__global__ __launch_bounds__(256) void gumbel_sample_bwd_wave_kernel(
    const float* __restrict__ sampled, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ upstream, int B, int S, int L, float gumbel_temperature,
    float* __restrict__ dlogits_out, unsigned long long* __restrict__ step_inc) {
  // the backward of a step runs after its forward and before the next one: it is where the step's Philox offset advances
  if (step_inc && blockIdx.x == 0 && threadIdx.x == 0) *step_inc += 1ull;
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const size_t ibase = (size_t)b * L;
  float acc[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) acc[r] = 0.f;
  for (int s = 0; s < S; ++s) {
    const size_t obase = ((size_t)b * S + s) * L;
    // out_k = log(p_k + eps): d out_k / d z_j = c_k (delta_kj - p_j), c_k = p_k / (p_k + eps)
    float p[IPL], t[IPL], dot = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int i = lane + 64 * r;
      p[r] = 0.f; t[r] = 0.f;
      if (i < L) {
        const float e = expf(sampled[obase + i]);           // p + eps
        p[r] = fmaxf(e - 1e-20f, 0.0f);
        t[r] = upstream[obase + i] * (p[r] / e);
        dot += t[r];
      }
    }
    dot = wave_sum_u(dot);
#pragma unroll
    for (int r = 0; r < IPL; ++r) acc[r] += t[r] - p[r] * dot;
  }
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    if (i < L) {
      const float lab = labels[ibase + i];
      const bool v = mask ? (mask[ibase + i] != 0) : (lab >= 0.0f);
      dlogits_out[ibase + i] = v ? (acc[r] / gumbel_temperature) : 0.0f;
    }
  }
}

inline int threads_for(int L) {
  int t = ((L + 63) / 64) * 64;
  if (t > 256) t = 256;
  return t;
}

// Wave-per-list softmax loss for the plain case (no lambda weight, list_size <= 64 * IPL): everything of a list
// lives in the registers of one wavefront -- no LDS, no workgroup barrier, four lists per 256-thread workgroup.
// Same arithmetic as softmax_loss_kernel (which stays for DCGLambdaWeight.individual_weights and long lists):
// masked logits = ln(1e-10), labels (x weights), all-zero lists -> 1e-10 on the valid entries, log-sum-exp,
// gradient w (ptot * softmax - p) / T;  + poly-1 term.  12 B read + 4 B written per item: HBM-bound.
// (A 16-byte-access form -- 4 consecutive items per lane, one dwordx4 load per array for a 200-item list -- measured
// SLOWER on MI355X: 25.1 vs 20.1 us per step at B = 16384, L = 200; only 50 of 64 lanes carry data and the four
// items of a lane serialise the exp / divide chain.  Dropped.  So was a form with 2 / 4 lists per wavefront and all
// loads issued up front: 12.5 / 13.7 us against 11.9 us per launch at B = 16384, L = 100 -- the kernel is not short of
// bytes in flight; ~4 us of every launch are fixed cost and the rest streams at ~2.5 TB/s out of the Infinity Cache.)
// NT (round 4): the launch streams more than the Infinity Cache holds (B * L * 12 B > 128 MB) -- loads and the gradient
// store carry the non-temporal hint, so that lines which will not be touched again do not evict each other.
template <int IPL, bool NT>
__global__ __launch_bounds__(256) void softmax_wave_kernel(const SmArgs a, int B) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const int L = a.L;
  const size_t base = (size_t)b * L;
  const float wl = (a.item_weights && a.weights_per_list) ? a.item_weights[b] : 1.0f;
  // one reciprocal per list instead of a division per item (x / T, y / sum y, e / sum e, d / T: four IEEE divisions of
  // ~10 instructions each on every item) and the single-instruction exponential: the kernel was issue-bound on them
  const float inv_t = 1.0f / a.temperature;
  float z[IPL], y[IPL];
  bool mv[IPL];
  float lsum = 0.f, zmax = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {                        // all loads of the list are issued before the first use
    const int i = lane + 64 * r;
    z[r] = -INFINITY; y[r] = 0.f; mv[r] = false;
    if (i < L) {
      const float lab = NT ? __builtin_nontemporal_load(a.labels + base + i) : a.labels[base + i];
      const float x = NT ? __builtin_nontemporal_load(a.logits + base + i) : a.logits[base + i];
      mv[r] = a.mask ? (a.mask[base + i] != 0) : (lab >= 0.0f);
      float w = 1.0f;
      if (a.item_weights) w = a.weights_per_list ? wl : a.item_weights[base + i];
      z[r] = mv[r] ? x * inv_t : kLogEps10;
      y[r] = (mv[r] ? lab : 0.0f) * (a.item_weights ? w : 1.0f);
      lsum += y[r];
      zmax = fmaxf(zmax, z[r]);
    }
  }
  lsum = wave_sum_u(lsum);
  zmax = wave_max_u(zmax);
  const bool nonzero = lsum > 0.0f;
  float psum = 0.f, esum = 0.f, e[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const bool in = lane + 64 * r < L;
    float yy = nonzero ? y[r] : 1e-10f;
    yy = (in && mv[r]) ? yy : 0.0f;
    y[r] = yy;
    psum += yy;
    e[r] = in ? __builtin_amdgcn_exp2f((z[r] - zmax) * 1.44269504088896340736f) : 0.0f;
    esum += e[r];
  }
  psum = wave_sum_u(psum);
  esum = wave_sum_u(esum);
  const float lse = logf(esum);
  const float inv_p = (psum != 0.0f) ? 1.0f / psum : 0.0f;            // divide_no_nan
  const float inv_e = 1.0f / esum;
  float loss = 0.f, ptot = 0.f, pt = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const bool in = lane + 64 * r < L;
    const float p = y[r] * inv_p;
    if (in) {
      loss += p * (lse - (z[r] - zmax));
      ptot += p;
      pt += p * (e[r] * inv_e);
    }
    y[r] = p;
  }
  loss = wave_sum_u(loss);
  ptot = wave_sum_u(ptot);
  if (a.poly_eps != 0.0f) {
    pt = wave_sum_u(pt);
    loss += a.poly_eps * (1.0f - pt);
  }
  if (lane == 0) { a.loss[b] = loss; a.weight[b] = lsum; }
  if (a.sum.out) grid_sum_contribute(a.sum, b, loss * lsum, lane);     // one contributor per list
  else if (a.sum.vec && lane == 0) a.sum.vec[b] = loss * lsum;
  if (!a.dlogits) return;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    if (i < L) {
      float g = 0.f;
      if (mv[r]) {
        const float sm = e[r] * inv_e;
        float d = ptot * sm - y[r];
        if (a.poly_eps != 0.0f) d -= a.poly_eps * sm * (y[r] - pt);
        g = (lsum * inv_t) * d;
      }
      if (NT) __builtin_nontemporal_store(g, a.dlogits + base + i); else a.dlogits[base + i] = g;
    }
  }
}

// Streaming form of the same kernel for LARGE batches of the plain case (round 4, TFR_SOFTMAX_STREAM: no mask array,
// no per-item weights (per-list ones: LW), gradient requested, more than 4 * kSmStreamGroups lists): at most kSmStreamGroups workgroups; a
// wavefront walks the lists w, w + W, w + 2 W, ... with the label / logit loads of its next D lists in flight (a slot
// is re-loaded as soon as its values are consumed), so that every resident wave has loads in flight while it reduces
// and a launch of 65 536 lists dispatches 2 048 workgroups instead of 16 384.  gfx950 counts loads AND stores on one in-order counter
// (vmcnt), so the overlap only exists if the compiler can count the memory operations between a load and its first
// use: every load and store below is unconditional (out-of-range slots re-read / re-write item 0 of the list with
// item 0's value; a wave past the end re-reads its last list) -- no branch around a memory instruction.  Same
// arithmetic, in the same order, as softmax_wave_kernel: bit-identical results (tests/test_gpu_parity.py).
constexpr int kSmStreamGroups = 2048;               // 8 workgroups of 4 waves per CU

template <int IPL, bool NT, int D, bool LW>
__global__ __launch_bounds__(256) void softmax_stream_kernel(const SmArgs a, int B) {
  const int lane = threadIdx.x & 63;
  const int W = gridDim.x * 4;
  int b = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));     // (the grid never exceeds the lists: b < B)
  const int wave_id = b;
  float wsum = 0.f;                                      // sum of loss * weight over this wave's lists, in the order it walks them
  const int L = a.L;
  const float inv_t = 1.0f / a.temperature;
  int off[IPL];
  bool in[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) { const int i = lane + 64 * r; in[r] = i < L; off[r] = in[r] ? i : 0; }
  // D lists in flight per wave: slot d holds list b + d W (a list past the end re-reads the wave's first one)
  float lab[D][IPL], x[D][IPL], wl[D];                   // (LW: one weight per list, item_weights[b])
#pragma unroll
  for (int d = 0; d < D; ++d) {
    const int bl = (b + d * W < B) ? b + d * W : b;
    wl[d] = LW ? a.item_weights[bl] : 1.0f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const float* pl = a.labels + (size_t)bl * L + off[r];
      const float* px = a.logits + (size_t)bl * L + off[r];
      lab[d][r] = NT ? __builtin_nontemporal_load(pl) : *pl;
      x[d][r] = NT ? __builtin_nontemporal_load(px) : *px;
    }
  }
  auto one_list = [&](const int d, const int cur, const bool reload) {
      float z[IPL], y[IPL];
      bool mv[IPL];
      float lsum = 0.f, zmax = -INFINITY;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {                      // selects, no branches (see above); x + 0 and max(x, -inf) are x
        mv[r] = in[r] && lab[d][r] >= 0.0f;
        z[r] = mv[r] ? x[d][r] * inv_t : (in[r] ? kLogEps10 : -INFINITY);
        y[r] = mv[r] ? lab[d][r] : 0.0f;
        if (LW) y[r] *= wl[d];
        lsum += y[r];
        zmax = fmaxf(zmax, z[r]);
      }
      if (reload) {                                        // the slot is free: the list D W further on
        const int bl = (cur + D * W < B) ? cur + D * W : cur;
        if (LW) wl[d] = a.item_weights[bl];
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const float* pl = a.labels + (size_t)bl * L + off[r];
          const float* px = a.logits + (size_t)bl * L + off[r];
          lab[d][r] = NT ? __builtin_nontemporal_load(pl) : *pl;
          x[d][r] = NT ? __builtin_nontemporal_load(px) : *px;
        }
      }
      lsum = wave_sum_u(lsum);
      zmax = wave_max_u(zmax);
      const bool nonzero = lsum > 0.0f;
      float psum = 0.f, esum = 0.f, e[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        float yy = nonzero ? y[r] : 1e-10f;
        yy = (in[r] && mv[r]) ? yy : 0.0f;
        y[r] = yy;
        psum += yy;
        e[r] = in[r] ? __builtin_amdgcn_exp2f((z[r] - zmax) * 1.44269504088896340736f) : 0.0f;
        esum += e[r];
      }
      psum = wave_sum_u(psum);
      esum = wave_sum_u(esum);
      const float lse = logf(esum);
      const float inv_p = (psum != 0.0f) ? 1.0f / psum : 0.0f;            // divide_no_nan
      const float inv_e = 1.0f / esum;
      float loss = 0.f, ptot = 0.f, pt = 0.f;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float p = y[r] * inv_p;                      // (0 in an out-of-range slot; e is 0 there too)
        loss += in[r] ? p * (lse - (z[r] - zmax)) : 0.0f;
        ptot += p;
        pt += p * (e[r] * inv_e);
        y[r] = p;
      }
      loss = wave_sum_u(loss);
      ptot = wave_sum_u(ptot);
      if (a.poly_eps != 0.0f) {
        pt = wave_sum_u(pt);
        loss += a.poly_eps * (1.0f - pt);
      }
      a.loss[cur] = loss; a.weight[cur] = lsum;            // (every lane, one address, one value)
      wsum = __builtin_fmaf(loss, lsum, wsum);
      float g[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float sm = e[r] * inv_e;
        float dd = ptot * sm - y[r];
        if (a.poly_eps != 0.0f) dd -= a.poly_eps * sm * (y[r] - pt);
        g[r] = mv[r] ? (lsum * inv_t) * dd : 0.0f;
      }
      const float g_first = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, g[0])));
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float v = in[r] ? g[r] : g_first;            // an out-of-range slot re-writes item 0 with item 0's value
        float* pd = a.dlogits + (size_t)cur * L + off[r];
        if (NT) __builtin_nontemporal_store(v, pd); else *pd = v;
      }
  };
  // the main loop has no exit inside its body (all D lists exist): straight-line code, exact vmcnt counts, nothing drains
  for (; b + (D - 1) * W < B; b += D * W) {
#pragma unroll
    for (int d = 0; d < D; ++d) one_list(d, b + d * W, true);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {                          // the last fewer-than-D lists of this wave
    if (b + d * W < B) one_list(d, b + d * W, false);
  }
  // one contributor per WAVE (W of them; the walk order of a wave is fixed by the grid, so the sum is reproducible for a
  // given launch geometry): a ticket per list would put two dependent memory round trips into every trip of the stream
  if (a.sum.out) grid_sum_contribute(a.sum, wave_id, wsum, lane);
  else if (a.sum.vec && lane == 0) a.sum.vec[wave_id] = wsum;     // partials mode (the default): W values for tfr_list_dot_f32 instead of 2 B
}

// Packed form (round 5) of the plain case WITHOUT weights: 64 / LG lists side by side in one wavefront (LG = 32 or 16 lanes per
// list, IPL items per lane).  Why: the streaming kernel above moves 79 MB in 24.8 us (3.2 TB/s, 40 % of 8 TB/s) and is flat in
// lists in flight and in workgroups -- it is bound by instruction issue, not by bytes in flight: a 100-item list occupies a
// whole wavefront (100 of 128 item slots) through six wave-wide reductions of ~13 DPP / readlane instructions each plus ~60
// element-wise instructions, ~200 instructions per list = 12 800 per SIMD for 65 536 lists.  Two lists per wavefront halve both
// parts (every vector instruction serves two lists; a 32-lane reduction is the same DPP steps with the last cross-row step
// dropped), four lists per wavefront for list_size <= 64 quarter them.  A wavefront reads / writes the rows of its lists as ONE
// contiguous run of memory (lists g S .. g S + S - 1 are adjacent rows).
// The reductions of a list run over its own LG lanes only, in an order that does not depend on the lane group or on the batch:
// a row's results are the same bits whatever surrounds it.  (They are NOT the bits of softmax_wave_kernel -- another
// summation order, both within a few ulp of the fp64 arbiter; every plain unweighted batch takes this kernel, so there is no
// mixing.)  Persistent like the streaming form: a wavefront walks groups w, w + W, ... with the next group's loads in flight.
template <int LG> struct SegOps;
template <> struct SegOps<32> {      // totals of lanes [0, 32) and [32, 64), returned to every lane of the segment
  static __device__ __forceinline__ float sum(float v, bool hi) {
    v += TFR_DPP_F(0.f, v, 0x111, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x112, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x114, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x118, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x142, 0xa, 0xf, false);                   // row_bcast:15 into rows 1 and 3
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return hi ? b : a;
  }
  static __device__ __forceinline__ float max(float v, bool hi) {
    const float ninf = -INFINITY;
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x111, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x112, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x114, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x118, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x142, 0xa, 0xf, false));
    const float a = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float b = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    return hi ? b : a;
  }
};
template <> struct SegOps<16> {      // totals of the four 16-lane rows
  static __device__ __forceinline__ float pick(float v, int seg) {
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 15));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 31));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 47));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
    const float lo = (seg & 1) ? r1 : r0, hi = (seg & 1) ? r3 : r2;
    return (seg & 2) ? hi : lo;
  }
  static __device__ __forceinline__ float sum(float v, int seg) {
    v += TFR_DPP_F(0.f, v, 0x111, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x112, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x114, 0xf, 0xf, true);
    v += TFR_DPP_F(0.f, v, 0x118, 0xf, 0xf, true);
    return pick(v, seg);
  }
  static __device__ __forceinline__ float max(float v, int seg) {
    const float ninf = -INFINITY;
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x111, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x112, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x114, 0xf, 0xf, false));
    v = fmaxf(v, TFR_DPP_F(ninf, v, 0x118, 0xf, 0xf, false));
    return pick(v, seg);
  }
};

// V4 (round 6; list_size % 4 == 0, 16-byte aligned rows, IPL == 4): lane sl owns the four ADJACENT items
// 4 sl .. 4 sl + 3 and moves them as ONE 16-byte access per array -- 25 lanes x 16 B for list_size 100 -- instead of four
// dword accesses (items sl, sl + LG, ...).  The texture addresser takes a wave's dword access in the same 16 cycles as a
// dwordx4 one (four lanes per cycle, tools/fill_bench.hip): per list 12 wave-instructions of address processing become 3.
// The sums of a list run over another item order than without V4 (same few-ulp distance to the fp64 arbiter).
typedef float sm_f4 __attribute__((ext_vector_type(4)));

template <int LG, int IPL, bool NT, bool LW, int D, bool V4 = false>      // D groups of lists in flight per wavefront (TFR_SOFTMAX_PACK_DEPTH)
__global__ __launch_bounds__(256) void softmax_pack_kernel(const SmArgs a, int B) {
  static_assert(!V4 || IPL == 4, "V4: four adjacent items per lane");
  constexpr int S = 64 / LG;                              // lists per wavefront
  const int lane = threadIdx.x & 63;
  const int seg = lane / LG, sl = lane % LG;
  const int W = gridDim.x * 4;
  const int G = (B + S - 1) / S;                          // groups of S adjacent lists
  int g = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));    // (the grid never exceeds the groups)
  const int wave_id = g;
  const int L = a.L;
  const float inv_t = 1.0f / a.temperature;
  int off[IPL];
  bool in[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) { const int i = V4 ? 4 * sl + r : sl + LG * r; in[r] = i < L; off[r] = in[r] ? i : 0; }
  float lab_n[D][IPL], x_n[D][IPL], wl_n[D];              // (LW: one weight per list, item_weights[b])
  auto fetch = [&](const int d, int gg) {
    gg = gg < G ? gg : (wave_id < G ? wave_id : 0);        // a group past the end re-reads the wavefront's first one
    int bl = gg * S + seg;
    bl = bl < B ? bl : B - 1;                              // an absent list of the last group re-reads the last list
    wl_n[d] = LW ? a.item_weights[(uint32_t)bl] : 1.0f;
    // (B L < 2^30, checked by the launcher: the BYTE offset of an item fits 32 bits, so an access is the uniform base
    // pointer + a 32-bit lane offset -- `global_load_dword v, v_off, s[base:base+1]` -- instead of a 64-bit address per
    // lane and item: 16 v_lshl_add_u64 + their moves per group of lists were address arithmetic)
    const uint32_t row = (uint32_t)bl * (uint32_t)L;
    if constexpr (V4) {                                    // (list_size % 4 == 0: a lane's four items are all inside or all outside)
      const uint32_t boff = (row + (uint32_t)off[0]) * 4u;
      const sm_f4* pl = reinterpret_cast<const sm_f4*>(reinterpret_cast<const char*>(a.labels) + boff);
      const sm_f4* px = reinterpret_cast<const sm_f4*>(reinterpret_cast<const char*>(a.logits) + boff);
      const sm_f4 l4 = NT ? __builtin_nontemporal_load(pl) : *pl;
      const sm_f4 x4 = NT ? __builtin_nontemporal_load(px) : *px;
#pragma unroll
      for (int r = 0; r < 4; ++r) { lab_n[d][r] = l4[r]; x_n[d][r] = x4[r]; }
      return;
    }
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const uint32_t boff = (row + (uint32_t)off[r]) * 4u;
      const float* pl = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.labels) + boff);
      const float* px = reinterpret_cast<const float*>(reinterpret_cast<const char*>(a.logits) + boff);
      lab_n[d][r] = NT ? __builtin_nontemporal_load(pl) : *pl;
      x_n[d][r] = NT ? __builtin_nontemporal_load(px) : *px;
    }
  };
  float wacc = 0.f;                                       // lanes with sl == 0: sum of loss * weight over the lists of that lane group
  // (up to three wavefronts of the last workgroup have no group: they fall through both loops and contribute a zero sum)
#pragma unroll
  for (int d = 0; d < D; ++d) fetch(d, g + d * W);
  auto one_group = [&](const int d, const int g, const bool reload) {
    const int b = g * S + seg;
    const bool have = b < B;
    float z[IPL], y[IPL], e[IPL];
    bool mv[IPL];
    float lsum = 0.f, zmax = -INFINITY;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {                       // the slot's values are consumed HERE (no copy of them is kept) ...
      mv[r] = in[r] && lab_n[d][r] >= 0.0f;
      z[r] = mv[r] ? x_n[d][r] * inv_t : (in[r] ? kLogEps10 : -INFINITY);
      y[r] = mv[r] ? lab_n[d][r] : 0.0f;
      if (LW) y[r] *= wl_n[d];
      lsum += y[r];
      zmax = fmaxf(zmax, z[r]);
    }
    if (reload) fetch(d, g + D * W);                      // ... so the slot is free: the group D W further on
    const auto sg = [&](float v) { if constexpr (LG == 32) return SegOps<32>::sum(v, seg != 0); else return SegOps<16>::sum(v, seg); };
    const auto mg = [&](float v) { if constexpr (LG == 32) return SegOps<32>::max(v, seg != 0); else return SegOps<16>::max(v, seg); };
    lsum = sg(lsum);
    zmax = mg(zmax);
    const bool nonzero = lsum > 0.0f;
    float psum = 0.f, esum = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      float yy = nonzero ? y[r] : 1e-10f;
      yy = mv[r] ? yy : 0.0f;
      y[r] = yy;
      psum += yy;
      e[r] = in[r] ? __builtin_amdgcn_exp2f((z[r] - zmax) * 1.44269504088896340736f) : 0.0f;
      esum += e[r];
    }
    // (every list of the wavefront has a positive label sum -- the common case: the normalised labels ARE the labels, their
    // sum is lsum, bit for bit: one segmented reduction less)
    psum = __ballot(!nonzero) ? sg(psum) : lsum;
    esum = sg(esum);
    // single-instruction log / reciprocals (1 ulp each: the loss and the gradient stay within a few 1e-7 of the fp64
    // arbiter; the libm logf and two IEEE divisions were ~35 of the ~300 vector instructions of a group)
    const float lse = __builtin_amdgcn_logf(esum) * 0.69314718055994530942f;
    const float inv_p = (psum != 0.0f) ? __builtin_amdgcn_rcpf(psum) : 0.0f;      // divide_no_nan
    const float inv_e = __builtin_amdgcn_rcpf(esum);
    float loss = 0.f, pt = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const float p = y[r] * inv_p;
      loss += in[r] ? p * (lse - (z[r] - zmax)) : 0.0f;
      y[r] = p;
    }
    loss = sg(loss);
    // sum_i p_i: 1 for a list with a valid item (its labels were normalised by their sum, or replaced by a uniform
    // 1e-10), 0 for a list without one -- in exact arithmetic; the per-list kernels add the p_i up (1 +- 1e-7).  The packed
    // form takes the exact value: one accumulation per item and one segmented reduction less.
    const float ptot = (psum != 0.0f) ? 1.0f : 0.0f;
    if (a.poly_eps != 0.0f) {                                // (uniform; PolyOneSoftmax only)
#pragma unroll
      for (int r = 0; r < IPL; ++r) pt += y[r] * (e[r] * inv_e);
      pt = sg(pt);
      loss += a.poly_eps * (1.0f - pt);
    }
    if (have && sl == 0) { a.loss[b] = loss; a.weight[b] = lsum; wacc = __builtin_fmaf(loss, lsum, wacc); }
    sm_f4 g4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const float sm = e[r] * inv_e;
      float dd = ptot * sm - y[r];
      if (a.poly_eps != 0.0f) dd -= a.poly_eps * sm * (y[r] - pt);
      const float gv = mv[r] ? (lsum * inv_t) * dd : 0.0f;
      if constexpr (V4) { g4[r & 3] = gv; continue; }
      if (have && in[r]) {
        float* pd = reinterpret_cast<float*>(reinterpret_cast<char*>(a.dlogits) + ((uint32_t)b * (uint32_t)L + (uint32_t)off[r]) * 4u);
        if (NT) __builtin_nontemporal_store(gv, pd); else *pd = gv;
      }
    }
    if constexpr (V4) {
      if (have && in[0]) {
        sm_f4* pd = reinterpret_cast<sm_f4*>(reinterpret_cast<char*>(a.dlogits) + ((uint32_t)b * (uint32_t)L + (uint32_t)off[0]) * 4u);
        if (NT) __builtin_nontemporal_store(g4, pd); else *pd = g4;
      }
    }
  };
  for (; g + (D - 1) * W < G; g += D * W) {
#pragma unroll
    for (int d = 0; d < D; ++d) one_group(d, g + d * W, true);
  }
#pragma unroll
  for (int d = 0; d < D; ++d) {                           // the last fewer-than-D groups of this wavefront
    if (g + d * W < G) one_group(d, g + d * W, false);
  }
  // one contributor per wavefront (see softmax_stream_kernel): its lane groups' sums added in lane order
  if (a.sum.out || a.sum.vec) {
    const float wsum = wave_sum_u(wacc);
    if (a.sum.out) grid_sum_contribute(a.sum, wave_id, wsum, lane);
    else if (lane == 0) a.sum.vec[wave_id] = wsum;
  }
}

}  // namespace

extern "C" int tfr_poly1_softmax_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                                          const float* item_weights, int weights_per_list, int lambda_kind,
                                          int topn, int normalized, int gain_kind, const float* gains,
                                          const float* discount, int B, int L, float temperature, float epsilon,
                                          float* loss_out, float* weight_out, float* dlogits_out, void* stream);
static int softmax_dispatch(const float* logits, const float* labels, const uint8_t* mask,
                            const float* item_weights, int weights_per_list, int lambda_kind,
                            int topn, int normalized, int gain_kind, const float* gains,
                            const float* discount, int B, int L, float temperature, float epsilon,
                            float* loss_out, float* weight_out, float* dlogits_out, float* loss_sum_out,
                            float* sum_scratch, uint32_t* ticket, void* stream);

extern "C" int tfr_softmax_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                                    const float* item_weights, int weights_per_list, int lambda_kind,
                                    int topn, int normalized, int gain_kind, const float* gains,
                                    const float* discount, int B, int L, float temperature,
                                    float* loss_out, float* weight_out, float* dlogits_out,
                                    void* stream) {
  return tfr_poly1_softmax_loss_f32(logits, labels, mask, item_weights, weights_per_list, lambda_kind, topn,
                                    normalized, gain_kind, gains, discount, B, L, temperature, 0.0f, loss_out,
                                    weight_out, dlogits_out, stream);
}

extern "C" int tfr_poly1_softmax_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                                          const float* item_weights, int weights_per_list, int lambda_kind,
                                          int topn, int normalized, int gain_kind, const float* gains,
                                          const float* discount, int B, int L, float temperature, float epsilon,
                                          float* loss_out, float* weight_out, float* dlogits_out, void* stream) {
  return softmax_dispatch(logits, labels, mask, item_weights, weights_per_list, lambda_kind, topn, normalized, gain_kind,
                          gains, discount, B, L, temperature, epsilon, loss_out, weight_out, dlogits_out, nullptr, nullptr,
                          nullptr, stream);
}

extern "C" int tfr_softmax_loss_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                                        const float* item_weights, int weights_per_list, int lambda_kind,
                                        int topn, int normalized, int gain_kind, const float* gains,
                                        const float* discount, int B, int L, float temperature, float epsilon,
                                        float* loss_out, float* weight_out, float* dlogits_out, float* loss_sum_out,
                                        float* sum_scratch, uint32_t* ticket, void* stream) {
  if (!sum_scratch || (loss_sum_out != nullptr) != (ticket != nullptr)) return TFR_EINVAL;
  return softmax_dispatch(logits, labels, mask, item_weights, weights_per_list, lambda_kind, topn, normalized, gain_kind,
                          gains, discount, B, L, temperature, epsilon, loss_out, weight_out, dlogits_out, loss_sum_out,
                          sum_scratch, ticket, stream);
}

// streaming form: plain case, gradient requested, more lists than four per workgroup of the persistent grid
static int sm_stream_groups() {
  static const int env_groups = [] { const char* e = getenv("TFR_SOFTMAX_STREAM_GROUPS"); return (e && *e) ? atoi(e) : kSmStreamGroups; }();
  return env_groups;
}
static bool sm_streams(int B, int L, bool has_mask, bool per_item_weights, int lambda_kind, bool want_grad) {
  static const int env_wave = [] { const char* e = getenv("TFR_SOFTMAX_WAVE"); return (e && *e) ? atoi(e) : 1; }();
  static const int env_stream = [] { const char* e = getenv("TFR_SOFTMAX_STREAM"); return (e && *e) ? atoi(e) : 1; }();
  const int groups = sm_stream_groups();
  return env_wave && lambda_kind == TFR_LAMBDA_NONE && L <= 256 && env_stream && !has_mask && !per_item_weights && want_grad &&
         (B + 3) / 4 > groups && groups >= 1;
}

// packed form: plain case (no per-item weights), list_size <= 256; lanes per list / items per lane by list size
// lanes per list of the packed form: 16 (four lists per wavefront) up to 64 items, beyond that TFR_SOFTMAX_PACK_LG (32 = two
// lists per wavefront, 8 items per lane at most; 16 = four lists, up to 16 items per lane)
static int sm_pack_lg() {
  static const int env_lg = [] { const char* e = getenv("TFR_SOFTMAX_PACK_LG"); return (e && *e) ? atoi(e) : 32; }();
  return env_lg == 16 ? 16 : 32;
}
static int sm_pack_lists_per_wave(int L) { return (L <= 64 || sm_pack_lg() == 16) ? 4 : 2; }
// (from 4 096 wavefronts' worth of lists on: below that the one-list-per-wavefront kernel has more wavefronts to hide latency
// behind -- B = 4 096, L = 100: 12.4 us per step against 13.3 packed; B = 65 536: 24.9 -> 22 us per launch, profiles/r05_softmax_ab.txt)
static bool sm_packs(int B, int L, bool has_mask, bool has_weights, int lambda_kind, bool want_grad) {
  static const int env_wave = [] { const char* e = getenv("TFR_SOFTMAX_WAVE"); return (e && *e) ? atoi(e) : 1; }();
  static const int env_pack = [] { const char* e = getenv("TFR_SOFTMAX_PACK"); return (e && *e) ? atoi(e) : 1; }();
  static const int env_min = [] { const char* e = getenv("TFR_SOFTMAX_PACK_MIN_WAVES"); return (e && *e) ? atoi(e) : 4096; }();
  return env_wave && env_pack && lambda_kind == TFR_LAMBDA_NONE && L <= 256 && !has_mask && !has_weights && want_grad &&
         (long)B * L < (1L << 30) && (B + sm_pack_lists_per_wave(L) - 1) / sm_pack_lists_per_wave(L) >= env_min;
}
static int sm_pack_grid(int B, int L) {
  const int S = sm_pack_lists_per_wave(L);
  const int groups = (B + S - 1) / S;
  const int wg = (groups + 3) / 4;
  // (round 5: 65 536 x 100 from HBM 20.7 us with 2 048 workgroups, 20.0 with 1 024 or 1 536 -- profiles/r05_softmax_ab.txt;
  //  round 6, with the 16-byte accesses and non-temporal accesses: 17.1 / 16.8 / 16.8 / 17.1 us with 768 / 1 024 / 1 280 / 1 536)
  static const int env_pg = [] { const char* e = getenv("TFR_SOFTMAX_PACK_GROUPS"); return (e && *e) ? atoi(e) : 1024; }();
  const int cap = env_pg >= 1 ? env_pg : 1024;
  return wg < cap ? wg : cap;
}

// has_weights: 0 none, 1 per list, 2 per item
extern "C" int tfr_softmax_sum_contributors(int B, int L, int has_mask, int has_weights, int lambda_kind, int want_grad) {
  if (B < 0 || L <= 0) return TFR_EINVAL;
  if (B == 0) return 0;
  if (sm_packs(B, L, has_mask != 0, has_weights == 2, lambda_kind, want_grad != 0)) return 4 * sm_pack_grid(B, L);
  return sm_streams(B, L, has_mask != 0, has_weights == 2, lambda_kind, want_grad != 0) ? 4 * sm_stream_groups() : B;
}

static int softmax_dispatch(const float* logits, const float* labels, const uint8_t* mask,
                            const float* item_weights, int weights_per_list, int lambda_kind,
                            int topn, int normalized, int gain_kind, const float* gains,
                            const float* discount, int B, int L, float temperature, float epsilon,
                            float* loss_out, float* weight_out, float* dlogits_out, float* loss_sum_out,
                            float* sum_scratch, uint32_t* ticket, void* stream) {
  if (!logits || !labels || !loss_out || !weight_out || B < 0 || L <= 0 || !(temperature > 0.0f))
    return TFR_EINVAL;
  if (lambda_kind != TFR_LAMBDA_NONE && lambda_kind != TFR_LAMBDA_DCG) return TFR_EINVAL;
  if (lambda_kind == TFR_LAMBDA_DCG && (!discount || (gain_kind == TFR_GAIN_CUSTOM && !gains)))
    return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) {
    if (loss_sum_out) return (int)hipMemsetAsync(loss_sum_out, 0, sizeof(float), (hipStream_t)stream);
    return TFR_OK;
  }
  SmArgs a;
  a.sum.out = loss_sum_out; a.sum.vec = sum_scratch; a.sum.st = ticket; a.sum.n = B;      // one contributor per list ...
  a.logits = logits; a.labels = labels; a.mask = mask; a.item_weights = item_weights;
  a.weights_per_list = weights_per_list; a.lambda_kind = lambda_kind; a.topn = topn;
  a.normalized = normalized; a.gain_kind = gain_kind; a.gains = gains; a.discount = discount;
  a.L = L; a.Lp = ((L + 3) / 4) * 4; a.P = pow2_ceil(L < 2 ? 2 : L); a.temperature = temperature;
  a.loss = loss_out; a.weight = weight_out; a.dlogits = dlogits_out; a.poly_eps = epsilon;
  static const int env_wave = [] { const char* e = getenv("TFR_SOFTMAX_WAVE"); return (e && *e) ? atoi(e) : 1; }();
  if (env_wave && lambda_kind == TFR_LAMBDA_NONE && L <= 1024) {
    hipStream_t st = (hipStream_t)stream;
    static const int env_nt = [] { const char* e = getenv("TFR_SOFTMAX_NT"); return (e && *e) ? atoi(e) : -1; }();
    // non-temporal from 64 MB per launch on (round 6: 65 536 x 100 = 79 MB per launch, 18.0 -> 17.2 us; it was 128 MB)
    const bool nt = env_nt >= 0 ? env_nt != 0 : ((long)B * L * 12 > (64L << 20));
    const int env_groups = sm_stream_groups();
    if (sm_packs(B, L, mask != nullptr, item_weights && !weights_per_list, lambda_kind, dlogits_out != nullptr)) {
      const int grid = sm_pack_grid(B, L);
      a.sum.n = grid * 4;                                           // one contributor per wavefront
      static const int env_pdepth = [] { const char* e = getenv("TFR_SOFTMAX_PACK_DEPTH"); return (e && *e) ? atoi(e) : 1; }();
#define SPK3(G_, I_, N_, W_) do { if (env_pdepth >= 2) hipLaunchKernelGGL((softmax_pack_kernel<G_, I_, N_, W_, 2>), dim3(grid), dim3(256), 0, st, a, B); \
                                 else hipLaunchKernelGGL((softmax_pack_kernel<G_, I_, N_, W_, 1>), dim3(grid), dim3(256), 0, st, a, B); } while (0)
#define SPK2(G_, I_, N_) do { if (item_weights) SPK3(G_, I_, N_, true); else SPK3(G_, I_, N_, false); } while (0)
#define SPK(G_, I_) do { if (nt) SPK2(G_, I_, true); else SPK2(G_, I_, false); } while (0)
      static const int env_v4 = [] { const char* e = getenv("TFR_SOFTMAX_V4"); return (e && *e) ? atoi(e) : 1; }();
      const bool v4 = env_v4 && (L % 4) == 0 && ((((uintptr_t)logits | (uintptr_t)labels | (uintptr_t)dlogits_out) & 15u) == 0);
#define SPV3(G_, N_, W_) do { if (env_pdepth >= 2) hipLaunchKernelGGL((softmax_pack_kernel<G_, 4, N_, W_, 2, true>), dim3(grid), dim3(256), 0, st, a, B); \
                              else hipLaunchKernelGGL((softmax_pack_kernel<G_, 4, N_, W_, 1, true>), dim3(grid), dim3(256), 0, st, a, B); } while (0)
#define SPV(G_) do { if (nt) { if (item_weights) SPV3(G_, true, true); else SPV3(G_, true, false); } \
                     else { if (item_weights) SPV3(G_, false, true); else SPV3(G_, false, false); } } while (0)
      if (v4 && L <= 64) { SPV(16); return (int)hipGetLastError(); }                       // 16 lanes x 4 adjacent items
      if (v4 && L <= 128 && sm_pack_lg() != 16) { SPV(32); return (int)hipGetLastError(); } // 32 lanes x 4 adjacent items
#undef SPV
#undef SPV3
      if (L <= 16) SPK(16, 1); else if (L <= 32) SPK(16, 2); else if (L <= 64) SPK(16, 4);
      else if (sm_pack_lg() == 16) { if (L <= 112) SPK(16, 7); else if (L <= 128) SPK(16, 8); else if (L <= 208) SPK(16, 13); else SPK(16, 16); }
      else if (L <= 128) SPK(32, 4); else SPK(32, 8);
#undef SPK
#undef SPK2
#undef SPK3
      return (int)hipGetLastError();
    }
    if (sm_streams(B, L, mask != nullptr, item_weights && !weights_per_list, lambda_kind, dlogits_out != nullptr)) {
      static const int env_depth = [] { const char* e = getenv("TFR_SOFTMAX_STREAM_DEPTH"); return (e && *e) ? atoi(e) : 2; }();
      a.sum.n = env_groups * 4;                                     // ... per wave in the streaming form
#define SMK(I, N, D, W) hipLaunchKernelGGL((softmax_stream_kernel<I, N, D, W>), dim3(env_groups), dim3(256), 0, st, a, B)
#define SMS(I, D) do { if (item_weights) { if (nt) SMK(I, true, D, true); else SMK(I, false, D, true); } \
                       else { if (nt) SMK(I, true, D, false); else SMK(I, false, D, false); } } while (0)
#define SMD(I) do { if (env_depth >= 4) SMS(I, 4); else if (env_depth >= 2) SMS(I, 2); else SMS(I, 1); } while (0)
      if (L <= 64) SMD(1); else if (L <= 128) SMD(2); else SMD(4);
#undef SMD
#undef SMS
#undef SMK
      return (int)hipGetLastError();
    }
#define SMW(I) do { if (nt) hipLaunchKernelGGL((softmax_wave_kernel<I, true>), dim3((B + 3) / 4), dim3(256), 0, st, a, B); \
                    else hipLaunchKernelGGL((softmax_wave_kernel<I, false>), dim3((B + 3) / 4), dim3(256), 0, st, a, B); } while (0)
    if (L <= 64) SMW(1); else if (L <= 128) SMW(2); else if (L <= 256) SMW(4); else if (L <= 512) SMW(8); else SMW(16);
#undef SMW
    return (int)hipGetLastError();
  }
  const size_t lds = 128 + (lambda_kind == TFR_LAMBDA_DCG ? (size_t)a.P * 8 : 0) +
                     (size_t)a.Lp * 17 + 16;
  if (lds > 160 * 1024) return TFR_ETOOLARGE;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(softmax_loss_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  int T = threads_for(L);
  if (lambda_kind == TFR_LAMBDA_DCG && a.P / 2 > T) T = a.P / 2 > 1024 ? 1024 : a.P / 2;
  hipLaunchKernelGGL(softmax_loss_kernel, dim3(B), dim3(T), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

extern "C" int tfr_gumbel_sample_step_f32(const float* logits, const float* labels, const uint8_t* mask,
                                          const float* uniform, uint64_t seed, uint64_t offset, const uint64_t* step,
                                          int B, int S, int L, float gumbel_temperature, float* sampled_out,
                                          float* labels_out, void* stream);
extern "C" int tfr_gumbel_sample_f32(const float* logits, const float* labels, const uint8_t* mask,
                                     const float* uniform, uint64_t seed, uint64_t offset, int B,
                                     int S, int L, float gumbel_temperature, float* sampled_out,
                                     void* stream) {
  return tfr_gumbel_sample_step_f32(logits, labels, mask, uniform, seed, offset, nullptr, B, S, L, gumbel_temperature,
                                    sampled_out, nullptr, stream);
}

extern "C" int tfr_gumbel_sample_step_f32(const float* logits, const float* labels, const uint8_t* mask,
                                          const float* uniform, uint64_t seed, uint64_t offset, const uint64_t* step,
                                          int B, int S, int L, float gumbel_temperature, float* sampled_out,
                                          float* labels_out, void* stream) {
  if (!logits || !labels || !sampled_out || B < 0 || S <= 0 || L <= 0 || !(gumbel_temperature > 0.0f))
    return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if ((step || labels_out) && L > 1024) return TFR_EINVAL;    // the wavefront kernels only (list_size <= 1024)
  if (B == 0) return TFR_OK;
  if (L <= 1024) {
    hipStream_t st = (hipStream_t)stream;
    const int BS = B * S;
#define GW(I) hipLaunchKernelGGL(gumbel_sample_wave_kernel<I>, dim3((BS + 3) / 4), dim3(256), 0, st, logits, labels, mask, uniform, seed, offset, BS, S, L, gumbel_temperature, sampled_out, (const unsigned long long*)step, labels_out)
    if (L <= 64) GW(1); else if (L <= 128) GW(2); else if (L <= 256) GW(4); else if (L <= 512) GW(8); else GW(16);
#undef GW
    return (int)hipGetLastError();
  }
  const int Lp = ((L + 3) / 4) * 4;
  hipLaunchKernelGGL(gumbel_sample_kernel, dim3(B * S), dim3(threads_for(L)), 128 + (size_t)Lp * 4,
                     (hipStream_t)stream, logits, labels, mask, uniform, seed, offset, S, L, Lp,
                     gumbel_temperature, sampled_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_gumbel_sample_bwd_step_f32(const float* sampled, const float* labels, const uint8_t* mask,
                                              const float* upstream, int B, int S, int L, float gumbel_temperature,
                                              float* dlogits_out, uint64_t* step_inc, void* stream);
extern "C" int tfr_gumbel_sample_bwd_f32(const float* sampled, const float* labels, const uint8_t* mask,
                                         const float* upstream, int B, int S, int L,
                                         float gumbel_temperature, float* dlogits_out, void* stream) {
  return tfr_gumbel_sample_bwd_step_f32(sampled, labels, mask, upstream, B, S, L, gumbel_temperature, dlogits_out, nullptr,
                                        stream);
}

extern "C" int tfr_gumbel_sample_bwd_step_f32(const float* sampled, const float* labels, const uint8_t* mask,
                                              const float* upstream, int B, int S, int L, float gumbel_temperature,
                                              float* dlogits_out, uint64_t* step_inc, void* stream) {
  if (!sampled || !labels || !upstream || !dlogits_out || B < 0 || S <= 0 || L <= 0 ||
      !(gumbel_temperature > 0.0f))
    return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (step_inc && (L > 1024 || B == 0)) return TFR_EINVAL;
  if (B == 0) return TFR_OK;
  if (L <= 1024) {
    hipStream_t st = (hipStream_t)stream;
#define GB(I) hipLaunchKernelGGL(gumbel_sample_bwd_wave_kernel<I>, dim3((B + 3) / 4), dim3(256), 0, st, sampled, labels, mask, upstream, B, S, L, gumbel_temperature, dlogits_out, (unsigned long long*)step_inc)
    if (L <= 64) GB(1); else if (L <= 128) GB(2); else if (L <= 256) GB(4); else if (L <= 512) GB(8); else GB(16);
#undef GB
    return (int)hipGetLastError();
  }
  const int Lp = ((L + 3) / 4) * 4;
  hipLaunchKernelGGL(gumbel_sample_bwd_kernel, dim3(B), dim3(threads_for(L)), 128 + (size_t)Lp * 4,
                     (hipStream_t)stream, sampled, labels, mask, upstream, S, L, gumbel_temperature,
                     dlogits_out);
  return (int)hipGetLastError();
}
