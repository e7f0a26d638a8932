// NeuralSort losses, forward + backward fused, wave-per-list (gfx950).
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

int env_int_ns(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

extern "C" int tfr_neural_sort_loss_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                                        const float* inv_log1p, const float* list_scale, int B, int L,
                                        float temperature, float* loss_out, float* dlogits_out, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (kind != TFR_NEURAL_SORT_NDCG && kind != TFR_NEURAL_SORT_CE) return TFR_EINVAL;
  if (kind == TFR_NEURAL_SORT_NDCG && !inv_log1p) return TFR_EINVAL;
  if (L > TFR_MAX_LIST_SIZE_NEURAL_SORT) return TFR_ETOOLARGE;            // 40 / 60 B of LDS per item, one wavefront per list (32 items per lane)
  if (B == 0) return TFR_OK;
  static const int max_runs = env_int_ns("TFR_APPROX_MAX_RUNS", 8);
  hipStream_t st = (hipStream_t)stream;
  const int Lp = ((L + 3) / 4) * 4 + 4;
  const size_t lds = ns_lds_bytes(Lp, kind);
#define NS(I, K) if (lds > 64 * 1024) { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&neural_sort_wave_kernel<I, K>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return (int)e; } hipLaunchKernelGGL((neural_sort_wave_kernel<I, K>), dim3(B), dim3(64), lds, st, logits, labels, mask, inv_log1p, list_scale, L, Lp, temperature, loss_out, dlogits_out, max_runs)
#define NS_K(K) do { if (L <= 64) { NS(1, K); } else if (L <= 128) { NS(2, K); } else if (L <= 256) { NS(4, K); } else if (L <= 512) { NS(8, K); } else if (L <= 1024) { NS(16, K); } else { NS(32, K); } } while (0)
  if (kind == TFR_NEURAL_SORT_NDCG) NS_K(TFR_NEURAL_SORT_NDCG); else NS_K(TFR_NEURAL_SORT_CE);
#undef NS_K
#undef NS
  return (int)hipGetLastError();
}
