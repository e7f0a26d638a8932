// NeuralSort losses, forward + backward fused, wave-per-list (gfx950); a workgroup per list beyond 2048 items.
//
// Reference behaviour restated (losses_impl.py:1716-1801 neural_sort, :1635-1673
// NeuralSortCrossEntropyLoss, :1676-1713 NeuralSortNDCGLoss, :137-167 ndcg(perm_mat=...),
// :33-49 _safe_default_gain_fn, :109-134 inverse_max_dcg).  With n valid items of a list,
// scores s (= logits / temperature), A_k = sum_j |s_k - s_j| and c_t = n + 1 - 2 t (t = 1..n):
//     P[t, k] = softmax_k( c_t * s_k - A_k )                      (row t = soft "item at rank t")
// rows of invalid positions are pushed behind the n valid ones and only spread over invalid
// columns, whose gains / true-permutation entries are zero: they never reach a loss value.
//   NeuralSortNDCG :  loss = - ( sum_t D_t * sum_k P[t,k] g_k ) * inverse_max_dcg,  D_t = 1/log1p(t)
//   NeuralSortCE   :  loss = (1/n) sum_{t<=n} - sum_k T[t,k] * log(1e-20 + P[t,k]),
//                     T = the same construction on the (cleaned) labels.
//     (the log-sum-exp of softmax_cross_entropy_with_logits over log(1e-20 + P) is
//      log(1 + L * 1e-20) = 0 in fp32 and its gradient cancels in the softmax Jacobian.)
//
// The reference materialises five [B, L, L] tensors forward and as many backward.  Here one
// wavefront owns a list and nothing of size L^2 exists anywhere:
//   A. lanes = items:   A_k (fp64 accumulation: it is the exponent of everything below) and the
//                       descending rank of s_k; the row maximum is known in closed form -- row t
//                       peaks at the t-th largest score (the NeuralSort theorem) -- so no
//                       max-reduction is ever run:  m_t = c_t s_(t) - A_(t).
//   B. lanes = rows t:  every lane sweeps the columns (LDS broadcast reads) and keeps Z_t and the
//                       row statistics in registers -- no cross-lane reduction in the sweep.
//   C. lanes = columns: sweeps the rows and accumulates Q_k = sum_t dL/dz[t,k], R_k = sum_t c_t dL/dz[t,k].
//   D. lanes = items:   dL/ds_k = R_k - sum_j sign(s_k - s_j) (Q_k + Q_j)      (d|x|/dx = sign, 0 at 0).
// Per list: ~5 n^2 / 64 exponentials (NDCG: 2), no workgroup barrier, 44-60 B of LDS per item.
#include "common.h"
#include "../../include/tfr_hip.h"

using namespace tfr;

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr float kTiny = 1e-20f;

#define NS_LDS_SYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

__host__ __device__ inline size_t ns_lds_bytes(int Lp, int kind) {
  return (size_t)Lp * (kind == TFR_NEURAL_SORT_NDCG ? (16 + 16 + 4 + 4) : (16 + 16 + 16 + 4 + 4 + 4)) + 16;
}

__device__ __forceinline__ float ns_exp(float d) { return __builtin_amdgcn_exp2f(d * kLog2e); }

template <int IPL, int KIND>
__global__ __launch_bounds__(64) void neural_sort_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ inv_log1p, const float* __restrict__ list_scale, int L, int Lp, float temperature,
    float* __restrict__ loss_out, float* __restrict__ dlogits_out, int max_runs) {
  constexpr bool CE = (KIND == TFR_NEURAL_SORT_CE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float4* COL = reinterpret_cast<float4*>(smem_raw);      // [Lp] (s, A, g | y, A^y), compact order
  float4* ROW = COL + Lp;                                  // [Lp] per-row statistics; later (s, Q) pairs
  float4* ROW2 = ROW + Lp;                                 // [Lp] CE only
  float* MS = reinterpret_cast<float*>(CE ? (ROW2 + Lp) : ROW2);   // [Lp] row maximum (scores)
  int* CI = reinterpret_cast<int*>(MS + Lp);               // [Lp] compact -> original index
  float* MY = reinterpret_cast<float*>(CI + Lp);           // [Lp] CE only: row maximum (labels)
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;

  // ---- 1. load + clean (:1642-1643, :1703-1704); label statistics.
  float x[IPL], y[IPL];
  bool v[IPL];
  float lmax = -INFINITY, lsum = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int e = lane + 64 * r;
    x[r] = 0.f; y[r] = 0.f; v[r] = false;
    if (e < L) {
      const float lab = labels[base + e];
      v[r] = mask ? (mask[base + e] != 0) : (lab >= 0.0f);
      x[r] = v[r] ? logits[base + e] / temperature : 0.0f;
      y[r] = v[r] ? lab : 0.0f;
      lmax = fmaxf(lmax, y[r]); lsum += y[r];
    }
  }
  lmax = wave_max_u(lmax); lsum = wave_sum_u(lsum);
  const bool nonzero = lsum > 0.0f;

  // ---- 2. NDCG: safe gains (:33-49) and the inverse ideal DCG (:109-134).
  float inv_max_dcg = 0.f;
  if (!CE) {
    if (!nonzero) lmax = 1e-10f;                             // :1708-1709 labels := 1e-10
    const float g0 = exp2f(-lmax);
    uint32_t sk[IPL];
    float tbl[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = lane + 64 * r;
      float gg = 0.f;
      if (e < L) gg = exp2f((nonzero ? y[r] : 1e-10f) - lmax) - g0;
      y[r] = gg;                                             // y now holds the gain
      sk[r] = __float_as_uint(gg);
      tbl[r] = (e < L) ? inv_log1p[e] : 0.0f;
    }
    float idcg = 0.f;
    if (!wave_sorted_dot_runs<IPL>(y, tbl, lane, L, max_runs, idcg)) {
      wave_bitonic_sort_desc<uint32_t, IPL>(sk, lane);
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < IPL; ++r) t += __uint_as_float(sk[r]) * tbl[r];
      idcg = wave_sum_u(t);
    }
    inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
  }

  // ---- 3. stable compaction of the valid items.
  int n = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const unsigned long long bal = __ballot(v[r]);
    if (v[r]) {
      const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
      COL[pos] = make_float4(x[r], 0.f, y[r], 0.f);
      CI[pos] = lane + 64 * r;
    }
    n += __popcll(bal);
  }
  const float scale = list_scale ? list_scale[b] : 1.0f;
  if (dlogits_out) {
#pragma unroll
    for (int r = 0; r < IPL; ++r)
      if (lane + 64 * r < L && !v[r]) dlogits_out[base + lane + 64 * r] = 0.0f;
  }
  if (n == 0) {                                              // divide_no_nan / zero gains
    if (lane == 0) loss_out[b] = 0.0f;
    return;
  }
  NS_LDS_SYNC();

  // ---- A. per item: A_k = sum_j |s_k - s_j| (and A^y_k), descending rank, closed-form row maxima.
  {
    float sk_[IPL], ak[IPL], gk[IPL], ayk[IPL];
    double accA[IPL], accY[IPL];
    int cnt[IPL], cnty[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int k = lane + 64 * r;
      const float4 c = COL[k < n ? k : 0];
      sk_[r] = c.x; gk[r] = c.z;
      accA[r] = 0.0; accY[r] = 0.0; cnt[r] = 0; cnty[r] = 0;
    }
    for (int j = 0; j < n; ++j) {
      const float4 cj = COL[j];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int k = lane + 64 * r;
        accA[r] += (double)fabsf(sk_[r] - cj.x);
        cnt[r] += (cj.x > sk_[r] || (cj.x == sk_[r] && j < k)) ? 1 : 0;
        if (CE) {
          accY[r] += (double)fabsf(gk[r] - cj.z);
          cnty[r] += (cj.z > gk[r] || (cj.z == gk[r] && j < k)) ? 1 : 0;
        }
      }
    }
    NS_LDS_SYNC();
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int k = lane + 64 * r;
      ak[r] = (float)accA[r]; ayk[r] = CE ? (float)accY[r] : 0.f;
      if (k < n) {
        COL[k] = make_float4(sk_[r], ak[r], gk[r], ayk[r]);
        MS[cnt[r]] = __builtin_fmaf((float)(n - 1 - 2 * cnt[r]), sk_[r], -ak[r]);
        if (CE) MY[cnty[r]] = __builtin_fmaf((float)(n - 1 - 2 * cnty[r]), gk[r], -ayk[r]);
      }
    }
    NS_LDS_SYNC();
  }

  // ---- B. per row t (lanes = rows): Z_t and the row statistics.
  float ct[IPL], ms[IPL], my[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int t = lane + 64 * r;
    ct[r] = (float)(n - 1 - 2 * t);
    ms[r] = (t < n) ? MS[t] : 0.f;
    my[r] = (CE && t < n) ? MY[t] : 0.f;
    if (t >= n) ct[r] = 0.f;
  }
  float loss;
  if (!CE) {
    float Z[IPL], N[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) { Z[r] = 0.f; N[r] = 0.f; }
    for (int k = 0; k < n; ++k) {
      const float4 c = COL[k];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float e = ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]);
        Z[r] += e;
        N[r] = __builtin_fmaf(e, c.z, N[r]);
      }
    }
    float dcg = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int t = lane + 64 * r;
      if (t < n) {
        const float G = N[r] / Z[r];
        const float D = inv_log1p[t];
        dcg = __builtin_fmaf(G, D, dcg);
        ROW[t] = make_float4(ms[r], -(inv_max_dcg * D) / Z[r], G, 0.f);
      }
    }
    dcg = wave_sum_u(dcg);
    loss = -(dcg * inv_max_dcg);
  } else {
    float Z[IPL], ZY[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) { Z[r] = 0.f; ZY[r] = 0.f; }
    for (int k = 0; k < n; ++k) {
      const float4 c = COL[k];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        Z[r] += ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]);
        ZY[r] += ns_exp(__builtin_fmaf(ct[r], c.z, -c.w) - my[r]);
      }
    }
    float rz[IPL], rzy[IPL], H[IPL], ll[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) { rz[r] = 1.0f / Z[r]; rzy[r] = 1.0f / ZY[r]; H[r] = 0.f; ll[r] = 0.f; }
    for (int k = 0; k < n; ++k) {
      const float4 c = COL[k];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float p = ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]) * rz[r];
        const float tt = ns_exp(__builtin_fmaf(ct[r], c.z, -c.w) - my[r]) * rzy[r];
        const float den = kTiny + p;
        ll[r] = __builtin_fmaf(-tt * kLn2, __builtin_amdgcn_logf(den), ll[r]);      // -T log(1e-20 + P)
        H[r] = __builtin_fmaf(tt * p, __builtin_amdgcn_rcpf(den), H[r]);
      }
    }
    float tot = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int t = lane + 64 * r;
      if (t < n) {
        tot += ll[r];
        ROW[t] = make_float4(ms[r], rz[r], H[r], 0.f);
        ROW2[t] = make_float4(my[r], rzy[r], 0.f, 0.f);
      }
    }
    tot = wave_sum_u(tot);
    loss = tot / (float)n;
  }
  if (lane == 0) loss_out[b] = loss;
  if (!dlogits_out) return;
  NS_LDS_SYNC();

  // ---- C. per column k (lanes = columns): Q_k = sum_t dL/dz[t,k],  R_k = sum_t c_t dL/dz[t,k].
  float Q[IPL], R[IPL], sk_[IPL], ak[IPL], gk[IPL], ayk[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int k = lane + 64 * r;
    const float4 c = COL[k < n ? k : 0];
    sk_[r] = c.x; ak[r] = c.y; gk[r] = c.z; ayk[r] = c.w;
    Q[r] = 0.f; R[r] = 0.f;
  }
  const float inv_n = 1.0f / (float)n;
  for (int t = 0; t < n; ++t) {
    const float c = (float)(n - 1 - 2 * t);
    const float4 row = ROW[t];
    if (!CE) {
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float e = ns_exp(__builtin_fmaf(c, sk_[r], -ak[r]) - row.x);
        const float u = (row.y * e) * (gk[r] - row.z);        // -inv D_t P[t,k] (g_k - G_t)
        Q[r] += u;
        R[r] = __builtin_fmaf(c, u, R[r]);
      }
    } else {
      const float4 row2 = ROW2[t];
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const float p = ns_exp(__builtin_fmaf(c, sk_[r], -ak[r]) - row.x) * row.y;
        const float tt = ns_exp(__builtin_fmaf(c, gk[r], -ayk[r]) - row2.x) * row2.y;
        const float rho = p * __builtin_amdgcn_rcpf(kTiny + p);
        const float u = inv_n * __builtin_fmaf(p, row.z, -tt * rho);   // (P H_t - T rho) / n
        Q[r] += u;
        R[r] = __builtin_fmaf(c, u, R[r]);
      }
    }
  }
  NS_LDS_SYNC();
  float2* SQ = reinterpret_cast<float2*>(ROW);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int k = lane + 64 * r;
    if (k < n) SQ[k] = make_float2(sk_[r], Q[r]);
  }
  NS_LDS_SYNC();

  // ---- D. dL/ds_k = R_k - sum_j sign(s_k - s_j) (Q_k + Q_j).
  float acc[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) acc[r] = 0.f;
  for (int j = 0; j < n; ++j) {
    const float2 sq = SQ[j];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const float sg = (sk_[r] > sq.x) ? 1.0f : ((sk_[r] < sq.x) ? -1.0f : 0.0f);
      acc[r] = __builtin_fmaf(sg, Q[r] + sq.y, acc[r]);
    }
  }
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int k = lane + 64 * r;
    if (k < n) dlogits_out[base + CI[k]] = scale * ((R[r] - acc[r]) / temperature);
  }
}

// ------------------------------------------------------------------------------------------
// Workgroup form for TFR_LDS_LIST_SIZE_NEURAL_SORT < list_size <= 8192 (round 5): the register arrays of the wave kernel
// stop at 32 items per lane.  1024 threads own a list, 8 items / rows / columns per thread (e = tid + 1024 r); the
// column image COL (16 B per item: the operand of every sweep) stays in LDS and is read by broadcast as in the wave
// kernel; everything read once per sweep STEP of another phase -- the row statistics, the closed-form row maxima, the
// (s, Q) pairs of phase D -- lives in a slot of the caller's workspace (32 / 52 B per item) and reaches the sweeps
// through 512-row LDS tiles.  The two sorts (ideal DCG of the gains; the stable compaction of the valid items) run on the
// COL region before it is filled.  Same arithmetic as the wave kernel, phase by phase; block reductions associate
// differently, so the two agree to rounding.  A launch has one workgroup per workspace slot.
constexpr int NSB_T = 1024, NSB_IPL = 8, NSB_TILE = 512, HP = 4;

// LDSWS: list sizes up to 2048 (P <= 2048) keep that slot in LDS behind the tiles (32 / 52 B per item: 152 KB in all for the
// CE kind) -- no workspace, and the form the launcher takes from 1025 items on: the one-wavefront kernel needs 40 / 60 B of LDS
// per item, i.e. ONE wavefront per CU beyond 1024 items, where this form runs sixteen on the same list (measured at 16 lists:
// 1.99 ms here at 2049 items against 3.82 ms there at 2048, profiles/r05_long_lists.txt).
template <int KIND, int ITEMS>      // ITEMS: 0 = up to 8 items per thread, their side arrays in the workspace; 1 / 2 = P = 1024 / 2048, all in LDS
__global__ __launch_bounds__(NSB_T) void neural_sort_block_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ inv_log1p, const float* __restrict__ list_scale, int B, int L, int P, float temperature,
    float* __restrict__ loss_out, float* __restrict__ dlogits_out, unsigned char* __restrict__ ws, long slot_bytes) {
  constexpr bool CE = (KIND == TFR_NEURAL_SORT_CE);
  constexpr int IPL = NSB_IPL;
  constexpr bool LDSWS = ITEMS != 0;
  constexpr int RM = LDSWS ? ITEMS : NSB_IPL;           // items a thread can own (P / NSB_T)
  constexpr int HPc = LDSWS ? ITEMS : HP;                   // items of a thread per pass of phases A - C (P <= 2048: two per thread)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);                       // [32]
  float4* COL = reinterpret_cast<float4*>(smem_raw + 128);                // [P] (s, A, g | y, A^y), compact order
  uint64_t* keys = reinterpret_cast<uint64_t*>(COL);                      // [P] before COL is filled: compaction keys
  uint32_t* G32 = reinterpret_cast<uint32_t*>(COL);                       // [P] before that: the gains' bits (ideal DCG)
  float4* TILE = COL + P;                                                 // [NSB_TILE] rows of ROW / (s, Q) pairs
  float4* TILE2 = TILE + NSB_TILE;                                        // [NSB_TILE] rows of ROW2 (CE)
  unsigned char* slot = LDSWS ? reinterpret_cast<unsigned char*>(TILE2 + NSB_TILE)
                              : ws + (size_t)blockIdx.x * (size_t)slot_bytes;
  float* XS = reinterpret_cast<float*>(slot);                             // [P] score by original index
  float* YS = XS + P;                                                     // [P] label / gain by original index
  float* MS = YS + P;                                                     // [P] row maximum (scores)
  int* CI = reinterpret_cast<int*>(MS + P);                               // [P] compact -> original index
  float* MY = reinterpret_cast<float*>(CI + P);                           // [P] CE only: row maximum (labels)
  float4* ROW = reinterpret_cast<float4*>(CE ? (MY + P) : MY);            // [P] per-row statistics; later (s, Q) pairs
  float4* ROW2 = ROW + P;                                                 // [P] CE only
  const int tid = threadIdx.x;

  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    __syncthreads();
    const size_t base = (size_t)b * L;
    // ---- 1. load + clean; label statistics; compaction keys are written after the gains' sort.
    float lmax = -INFINITY, lsum = 0.f, nvf = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = tid + NSB_T * r;
      if (e < P) {
        float x = 0.f, y = 0.f;
        if (e < L) {
          const float lab = labels[base + e];
          const bool v = mask ? (mask[base + e] != 0) : (lab >= 0.0f);
          x = v ? logits[base + e] / temperature : 0.0f;
          y = v ? lab : 0.0f;
          nvf += v ? 1.0f : 0.0f;
          lmax = fmaxf(lmax, y); lsum += y;
          if (dlogits_out && !v) dlogits_out[base + e] = 0.0f;
        }
        XS[e] = x; YS[e] = y;
      }
    }
    lmax = block_max(lmax, red); lsum = block_sum(lsum, red);
    const int n = (int)(block_sum(nvf, red) + 0.5f);                     // < 2^24: exact
    const bool nonzero = lsum > 0.0f;

    // ---- 2. NDCG: safe gains and the inverse ideal DCG (the gains sorted descending, as bits: they are >= 0).
    float inv_max_dcg = 0.f;
    if (!CE) {
      if (!nonzero) lmax = 1e-10f;
      const float g0 = exp2f(-lmax);
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int e = tid + NSB_T * r;
        if (e < P) {
          float gg = 0.f;
          if (e < L) gg = exp2f((nonzero ? YS[e] : 1e-10f) - lmax) - g0;
          YS[e] = gg;                                                     // YS now holds the gain
          G32[e] = __float_as_uint(gg);
        }
      }
      block_bitonic_sort_desc(G32, P);
      float t = 0.f;
      for (int p = tid; p < L; p += NSB_T) t = __builtin_fmaf(__uint_as_float(G32[p]), inv_log1p[p], t);
      const float idcg = block_sum(t, red);
      inv_max_dcg = (idcg > 0.0f) ? (1.0f / idcg) : 0.0f;
      __syncthreads();
    }
    if (n == 0) {                                                        // divide_no_nan / zero gains
      if (tid == 0) loss_out[b] = 0.0f;
      continue;
    }

    // ---- 3. stable compaction of the valid items: sort (valid, index); then COL over the key region.
    for (int e = tid; e < P; e += NSB_T) {
      bool v = false;
      if (e < L) { const float lab = labels[base + e]; v = mask ? (mask[base + e] != 0) : (lab >= 0.0f); }
      keys[e] = make_sort_key(v, 0.0f, 0, e);
    }
    block_bitonic_sort_desc(keys, P);                                    // valid first, ascending index
    {
      int idx[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) { const int p = tid + NSB_T * r; idx[r] = (p < n) ? sort_key_index(keys[p]) : 0; }
      __syncthreads();
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int p = tid + NSB_T * r;
        if (p < n) { COL[p] = make_float4(XS[idx[r]], 0.f, YS[idx[r]], 0.f); CI[p] = idx[r]; }
      }
    }
    const float scale = list_scale ? list_scale[b] : 1.0f;
    __syncthreads();

    // ---- A. per item: A_k = sum_j |s_k - s_j| (and A^y_k), descending rank, closed-form row maxima.  (Phases A - C
    // take a thread's 8 items in two halves of HPc = 4: at 128 registers per thread the full set spilled 100-150 words.)
    for (int h = 0; h < IPL / HPc; ++h) {
      if (NSB_T * HPc * h >= n) break;                                    // (uniform)
      float sk_[HPc], gk[HPc];
      double accA[HPc], accY[HPc];
      int cnt[HPc], cnty[HPc];
#pragma unroll
      for (int r = 0; r < HPc; ++r) {
        const int k = tid + NSB_T * (HPc * h + r);
        const float4 c = COL[k < n ? k : 0];
        sk_[r] = c.x; gk[r] = c.z;
        accA[r] = 0.0; accY[r] = 0.0; cnt[r] = 0; cnty[r] = 0;
      }
      for (int j = 0; j < n; ++j) {
        const float4 cj = COL[j];
#pragma unroll
        for (int r = 0; r < HPc; ++r) {
          const int k = tid + NSB_T * (HPc * h + r);
          accA[r] += (double)fabsf(sk_[r] - cj.x);
          cnt[r] += (cj.x > sk_[r] || (cj.x == sk_[r] && j < k)) ? 1 : 0;
          if (CE) {
            accY[r] += (double)fabsf(gk[r] - cj.z);
            cnty[r] += (cj.z > gk[r] || (cj.z == gk[r] && j < k)) ? 1 : 0;
          }
        }
      }
      __syncthreads();                                                   // (the sweeps read .x / .z only; .y / .w change below)
#pragma unroll
      for (int r = 0; r < HPc; ++r) {
        const int k = tid + NSB_T * (HPc * h + r);
        if (k < n) {
          const float ak = (float)accA[r], ayk = CE ? (float)accY[r] : 0.f;
          COL[k] = make_float4(sk_[r], ak, gk[r], ayk);
          MS[cnt[r]] = __builtin_fmaf((float)(n - 1 - 2 * cnt[r]), sk_[r], -ak);
          if (CE) MY[cnty[r]] = __builtin_fmaf((float)(n - 1 - 2 * cnty[r]), gk[r], -ayk);
        }
      }
    }
    __syncthreads();

    // ---- B. per row t (threads = rows): Z_t and the row statistics.
    float part = 0.f;
    for (int h = 0; h < IPL / HPc; ++h) {
      if (NSB_T * HPc * h >= n) break;
      float ct[HPc], ms[HPc], my[HPc];
#pragma unroll
      for (int r = 0; r < HPc; ++r) {
        const int t = tid + NSB_T * (HPc * h + r);
        ct[r] = (t < n) ? (float)(n - 1 - 2 * t) : 0.f;
        ms[r] = (t < n) ? MS[t] : 0.f;
        my[r] = (CE && t < n) ? MY[t] : 0.f;
      }
      if (!CE) {
        float Z[HPc], N[HPc];
#pragma unroll
        for (int r = 0; r < HPc; ++r) { Z[r] = 0.f; N[r] = 0.f; }
        for (int k = 0; k < n; ++k) {
          const float4 c = COL[k];
#pragma unroll
          for (int r = 0; r < HPc; ++r) {
            const float e = ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]);
            Z[r] += e;
            N[r] = __builtin_fmaf(e, c.z, N[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < HPc; ++r) {
          const int t = tid + NSB_T * (HPc * h + r);
          if (t < n) {
            const float G = N[r] / Z[r];
            const float D = inv_log1p[t];
            part = __builtin_fmaf(G, D, part);
            ROW[t] = make_float4(ms[r], -(inv_max_dcg * D) / Z[r], G, 0.f);
          }
        }
      } else {
        float Z[HPc], ZY[HPc];
#pragma unroll
        for (int r = 0; r < HPc; ++r) { Z[r] = 0.f; ZY[r] = 0.f; }
        for (int k = 0; k < n; ++k) {
          const float4 c = COL[k];
#pragma unroll
          for (int r = 0; r < HPc; ++r) {
            Z[r] += ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]);
            ZY[r] += ns_exp(__builtin_fmaf(ct[r], c.z, -c.w) - my[r]);
          }
        }
        float rz[HPc], rzy[HPc], H[HPc], ll[HPc];
#pragma unroll
        for (int r = 0; r < HPc; ++r) { rz[r] = 1.0f / Z[r]; rzy[r] = 1.0f / ZY[r]; H[r] = 0.f; ll[r] = 0.f; }
        for (int k = 0; k < n; ++k) {
          const float4 c = COL[k];
#pragma unroll
          for (int r = 0; r < HPc; ++r) {
            const float p = ns_exp(__builtin_fmaf(ct[r], c.x, -c.y) - ms[r]) * rz[r];
            const float tt = ns_exp(__builtin_fmaf(ct[r], c.z, -c.w) - my[r]) * rzy[r];
            const float den = kTiny + p;
            ll[r] = __builtin_fmaf(-tt * kLn2, __builtin_amdgcn_logf(den), ll[r]);      // -T log(1e-20 + P)
            H[r] = __builtin_fmaf(tt * p, __builtin_amdgcn_rcpf(den), H[r]);
          }
        }
#pragma unroll
        for (int r = 0; r < HPc; ++r) {
          const int t = tid + NSB_T * (HPc * h + r);
          if (t < n) {
            part += ll[r];
            ROW[t] = make_float4(ms[r], rz[r], H[r], 0.f);
            ROW2[t] = make_float4(my[r], rzy[r], 0.f, 0.f);
          }
        }
      }
    }
    part = block_sum(part, red);
    const float loss = CE ? part / (float)n : -(part * inv_max_dcg);
    if (tid == 0) loss_out[b] = loss;
    if (!dlogits_out) continue;

    // ---- C. per column k (threads = columns): Q_k = sum_t dL/dz[t,k],  R_k = sum_t c_t dL/dz[t,k].
    float Q[IPL], R[IPL], sk_[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int k = tid + NSB_T * r;
      sk_[r] = COL[k < n ? k : 0].x;
      Q[r] = 0.f; R[r] = 0.f;
    }
    {
      const float inv_n = 1.0f / (float)n;
      for (int t0 = 0; t0 < n; t0 += NSB_TILE) {
        __syncthreads();                                               // the previous tile is consumed; ROW is written
        if (tid < NSB_TILE && t0 + tid < n) {
          TILE[tid] = ROW[t0 + tid];
          if (CE) TILE2[tid] = ROW2[t0 + tid];
        }
        __syncthreads();
        const int tn = (n - t0 < NSB_TILE) ? n - t0 : NSB_TILE;
#pragma unroll
        for (int h = 0; h < IPL / HPc; ++h) {
          if (NSB_T * HPc * h < n) {
            float ak[HPc], gk[HPc], ayk[HPc];
#pragma unroll
            for (int r = 0; r < HPc; ++r) {
              const int k = tid + NSB_T * (HPc * h + r);
              const float4 c = COL[k < n ? k : 0];
              ak[r] = c.y; gk[r] = c.z; ayk[r] = c.w;
            }
            for (int tt0 = 0; tt0 < tn; ++tt0) {
              const float c = (float)(n - 1 - 2 * (t0 + tt0));
              const float4 row = TILE[tt0];
              if (!CE) {
#pragma unroll
                for (int r = 0; r < HPc; ++r) {
                  const float e = ns_exp(__builtin_fmaf(c, sk_[HPc * h + r], -ak[r]) - row.x);
                  const float u = (row.y * e) * (gk[r] - row.z);        // -inv D_t P[t,k] (g_k - G_t)
                  Q[HPc * h + r] += u;
                  R[HPc * h + r] = __builtin_fmaf(c, u, R[HPc * h + r]);
                }
              } else {
                const float4 row2 = TILE2[tt0];
#pragma unroll
                for (int r = 0; r < HPc; ++r) {
                  const float p = ns_exp(__builtin_fmaf(c, sk_[HPc * h + r], -ak[r]) - row.x) * row.y;
                  const float tt = ns_exp(__builtin_fmaf(c, gk[r], -ayk[r]) - row2.x) * row2.y;
                  const float rho = p * __builtin_amdgcn_rcpf(kTiny + p);
                  const float u = inv_n * __builtin_fmaf(p, row.z, -tt * rho);   // (P H_t - T rho) / n
                  Q[HPc * h + r] += u;
                  R[HPc * h + r] = __builtin_fmaf(c, u, R[HPc * h + r]);
                }
              }
            }
          }
        }
      }
    }
    __syncthreads();                                                   // every thread is done with ROW
    float2* SQ = reinterpret_cast<float2*>(ROW);
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int k = tid + NSB_T * r;
      if (k < n) SQ[k] = make_float2(sk_[r], Q[r]);
    }

    // ---- D. dL/ds_k = R_k - sum_j sign(s_k - s_j) (Q_k + Q_j).
    float acc[IPL];
#pragma unroll
    for (int r = 0; r < RM; ++r) acc[r] = 0.f;
    float2* T2 = reinterpret_cast<float2*>(TILE);                       // [2 * NSB_TILE] pairs
    for (int j0 = 0; j0 < n; j0 += 2 * NSB_TILE) {
      __syncthreads();                                                 // SQ is written / the previous tile is consumed
      if (j0 + tid < n) T2[tid] = SQ[j0 + tid];
      __syncthreads();
      const int jn = (n - j0 < 2 * NSB_TILE) ? n - j0 : 2 * NSB_TILE;
      for (int jj = 0; jj < jn; ++jj) {
        const float2 sq = T2[jj];
#pragma unroll
        for (int r = 0; r < RM; ++r) {
          const float sg = (sk_[r] > sq.x) ? 1.0f : ((sk_[r] < sq.x) ? -1.0f : 0.0f);
          acc[r] = __builtin_fmaf(sg, Q[r] + sq.y, acc[r]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < RM; ++r) {
      const int k = tid + NSB_T * r;
      if (k < n) dlogits_out[base + CI[k]] = scale * ((R[r] - acc[r]) / temperature);
    }
  }
}

int env_int_ns(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

extern "C" int tfr_neural_sort_loss_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                                        const float* inv_log1p, const float* list_scale, int B, int L,
                                        float temperature, float* loss_out, float* dlogits_out, void* workspace,
                                        long workspace_bytes, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (kind != TFR_NEURAL_SORT_NDCG && kind != TFR_NEURAL_SORT_CE) return TFR_EINVAL;
  if (kind == TFR_NEURAL_SORT_NDCG && !inv_log1p) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const long slot = tfr_list_workspace_bytes(kind == TFR_NEURAL_SORT_NDCG ? TFR_WS_NEURAL_SORT_NDCG : TFR_WS_NEURAL_SORT_CE, L);
  if (slot && (!workspace || workspace_bytes < slot)) return TFR_ETOOLARGE;   // beyond one wavefront's 32 items per lane
  if (B == 0) return TFR_OK;
  static const int max_runs = env_int_ns("TFR_APPROX_MAX_RUNS", 8);
  hipStream_t st = (hipStream_t)stream;
  if (slot) {                                       // one workgroup per list, row statistics in the workspace
    const int P = pow2_ceil(L);
    const size_t lds = 128 + (size_t)P * 16 + 2 * (size_t)NSB_TILE * 16;
    auto fn = kind == TFR_NEURAL_SORT_NDCG ? neural_sort_block_kernel<TFR_NEURAL_SORT_NDCG, 0>
                                           : neural_sort_block_kernel<TFR_NEURAL_SORT_CE, 0>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(big_slots(B, (size_t)workspace_bytes, (size_t)slot)), dim3(NSB_T), lds, st, logits, labels, mask,
                       inv_log1p, list_scale, B, L, P, temperature, loss_out, dlogits_out, (unsigned char*)workspace, slot);
    return (int)hipGetLastError();
  }
  static const int env_block = env_int_ns("TFR_NEURAL_SORT_BLOCK", 1);
  // 512 < L <= 2048: the workgroup form with everything in LDS -- sixteen wavefronts on a list instead of one.  Always from 1025
  // items on (the wave kernel's 40 / 60 B of LDS per item leave ONE wavefront per CU there: 3.8-4.4x measured); between 513
  // and 1024 items a list runs 2.8-3.1x faster on the workgroup form, but only 256 lists are resident against two to six
  // wavefronts per CU of the wave kernel: the workgroup form is taken when the batch fits two rounds of it or the wave
  // kernel would not get three wavefronts per CU (measured at 600 / 1024 items, 16 - 1024 lists: profiles/r05_long_lists.txt)
  static const int env_block_min = env_int_ns("TFR_NEURAL_SORT_BLOCK_MIN", 513);
  const bool few_waves = 3 * ns_lds_bytes(((L + 3) / 4) * 4 + 4, kind) > (size_t)160 * 1024;
  if (env_block && L >= env_block_min && L > 512 && (L > 1024 || B <= 512 || few_waves)) {
    const int P = L > 1024 ? 2048 : 1024;
    const size_t per_item = (kind == TFR_NEURAL_SORT_NDCG) ? 32 : 52;
    const size_t lds = 128 + (size_t)P * 16 + 2 * (size_t)NSB_TILE * 16 + (size_t)P * per_item;
    auto fn = kind == TFR_NEURAL_SORT_NDCG
                  ? (P == 2048 ? neural_sort_block_kernel<TFR_NEURAL_SORT_NDCG, 2> : neural_sort_block_kernel<TFR_NEURAL_SORT_NDCG, 1>)
                  : (P == 2048 ? neural_sort_block_kernel<TFR_NEURAL_SORT_CE, 2> : neural_sort_block_kernel<TFR_NEURAL_SORT_CE, 1>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(B < 65535 * 16 ? B : 65535 * 16), dim3(NSB_T), lds, st, logits, labels, mask, inv_log1p, list_scale, B,
                       L, P, temperature, loss_out, dlogits_out, (unsigned char*)nullptr, 0L);
    return (int)hipGetLastError();
  }
  const int Lp = ((L + 3) / 4) * 4 + 4;
  const size_t lds = ns_lds_bytes(Lp, kind);
#define NS(I, K) if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&neural_sort_wave_kernel<I, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return (int)e; } hipLaunchKernelGGL((neural_sort_wave_kernel<I, K>), dim3(B), dim3(64), lds, st, logits, labels, mask, inv_log1p, list_scale, L, Lp, temperature, loss_out, dlogits_out, max_runs)
#define NS_K(K) do { if (L <= 64) { NS(1, K); } else if (L <= 128) { NS(2, K); } else if (L <= 256) { NS(4, K); } else if (L <= 512) { NS(8, K); } else if (L <= 1024) { NS(16, K); } else { NS(32, K); } } while (0)
  if (kind == TFR_NEURAL_SORT_NDCG) NS_K(TFR_NEURAL_SORT_NDCG); else NS_K(TFR_NEURAL_SORT_CE);
#undef NS_K
#undef NS
  return (int)hipGetLastError();
}
