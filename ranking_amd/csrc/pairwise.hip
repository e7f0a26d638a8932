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

struct PwArgs {
  const float* logits; const float* labels; const uint8_t* mask;
  const float* item_weights; const float* list_weights;
  int lambda_kind; int topn; float smooth; int normalized; int gain_kind;
  const float* gains; const float* discount;
  int L; int Lp; int P; float temperature; int C;
  float* row_loss; float* row_weight; float* nnz; float* dlogits;
};

__host__ __device__ inline size_t pw_smem_bytes(int Lp, int P) {
  return 256 + (size_t)P * 8 + (size_t)Lp * 4 * 12 + (size_t)Lp * 2 + 32;
}

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
  const int b = blockIdx.x, L = a.L, P = a.P;
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
      key = make_sort_key(mv, x, 0, i);
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
      const float dprime = (dcg_lambda && r <= topn) ? a.discount[r - 1] : 0.0f;
      rec1[pos] = make_float2(dprime, __int_as_float(LV[i] ? r : -r));
      CI[pos] = i;
    }
    n += tot;
  }
  __syncthreads();

  // ---- 4. pair sweep.  Row i (C lanes) against every column j.
  const int C = a.C;
  const int rows_per_pass = T / C;
  const int c = tid % C, rsub = tid / C;
  const float fL = (float)L;
  const float one_minus_s = 1.0f - a.smooth;
  float nnz_local = 0.f;
  for (int i = tid; i < L; i += T) {
    if (!MV[i]) {
      if (a.row_loss) a.row_loss[base + i] = 0.f;
      if (a.row_weight) a.row_weight[base + i] = 0.f;
      if (a.dlogits) a.dlogits[base + i] = 0.f;
    }
  }
  for (int row0 = 0; row0 < n; row0 += rows_per_pass) {
    const int row = row0 + rsub;
    const bool active = row < n;
    if (!__any(active)) continue;
    const float4 ri = rec0[active ? row : 0];
    const float2 qi = rec1[active ? row : 0];
    const int rsi = __float_as_int(qi.y);
    const int ranki = rsi < 0 ? -rsi : rsi;
    const bool lvi = rsi > 0;
    float acc_loss = 0.f, acc_w = 0.f, acc_g = 0.f;
    for (int j = c; j < n; j += C) {
      const float4 rj = rec0[j];
      const bool hi = ri.y > rj.y;            // row item is the preferred one
      const bool lo = rj.y > ri.y;            // column item is the preferred one
      float wl = 1.0f;
      if (a.lambda_kind == TFR_LAMBDA_DCG) {
        const float2 qj = rec1[j];
        const int rsj = __float_as_int(qj.y);
        const int rankj = rsj < 0 ? -rsj : rsj;
        const bool lvj = rsj > 0;
        const int dr = ranki > rankj ? ranki - rankj : rankj - ranki;
        const bool in_top = (ranki <= topn) || (rankj <= topn);
        const float u = in_top ? U[dr] : 0.0f;               // U[0] = 0
        const float v = fabsf(qi.x - qj.x);
        float pd = one_minus_s * u + a.smooth * v;
        pd = in_top ? pd : 0.0f;
        const float pg = (lvi && lvj) ? fabsf(ri.z - rj.z) : 0.0f;
        wl = (pg * pd) * fL;
      } else if (a.lambda_kind == TFR_LAMBDA_LABELDIFF) {
        wl = fabsf(ri.y - rj.y);
      }
      const float d = hi ? (ri.x - rj.x) : (rj.x - ri.x);    // winner - loser
      const float ww = wl * (hi ? ri.w : rj.w);               // winner's item weight
      const float u = __builtin_amdgcn_exp2f(-fabsf(d) * kLog2e);
      const float q = __builtin_amdgcn_rcpf(1.0f + u);
      const float loss = fmaxf(-d, 0.0f) + log1pf(u);         // :936-940
      const float sig_neg = (d >= 0.0f) ? u * q : q;          // sigma(-d) = -loss'(d)
      const float gterm = ww * sig_neg;
      if (hi) {
        acc_loss = __builtin_fmaf(ww, loss, acc_loss);
        acc_w += ww;
        nnz_local += (active && ww != 0.0f) ? 1.0f : 0.0f;
        acc_g -= gterm;
      } else if (lo) {
        acc_g += gterm;
      }
    }
    for (int o = 1; o < C; o <<= 1) {
      acc_loss += __shfl_xor(acc_loss, o, 64);
      acc_w += __shfl_xor(acc_w, o, 64);
      acc_g += __shfl_xor(acc_g, o, 64);
    }
    if (active && c == 0) {
      const int oi = CI[row];
      if (a.row_loss) a.row_loss[base + oi] = acc_loss;
      if (a.row_weight) a.row_weight[base + oi] = acc_w;
      if (a.dlogits) a.dlogits[base + oi] = acc_g / a.temperature;
    }
  }
  nnz_local = block_sum(nnz_local, red);
  if (tid == 0 && a.nnz) a.nnz[b] = nnz_local;
}

int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && *v) ? atoi(v) : dflt;
}

}  // namespace

extern "C" int tfr_pairwise_logistic_f32(const float* logits, const float* labels, const uint8_t* mask,
                                         const float* item_weights, const float* list_weights,
                                         int lambda_kind, int topn, float smooth_fraction,
                                         int normalized, int gain_kind, const float* gains,
                                         const float* discount, int B, int L, float temperature,
                                         float* row_loss_out, float* row_weight_out, float* nnz_out,
                                         float* dlogits_out, void* stream) {
  if (!logits || !labels || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (lambda_kind != TFR_LAMBDA_NONE && lambda_kind != TFR_LAMBDA_DCG &&
      lambda_kind != TFR_LAMBDA_LABELDIFF) return TFR_EINVAL;
  if (lambda_kind == TFR_LAMBDA_DCG) {
    if (!discount) return TFR_EINVAL;
    if (gain_kind == TFR_GAIN_CUSTOM && !gains) return TFR_EINVAL;
    if (!(smooth_fraction >= 0.0f && smooth_fraction <= 1.0f)) return TFR_EINVAL;   // :329-331
  }
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  static const int env_threads = env_int("TFR_PAIRWISE_THREADS", 0);
  static const int env_lanes = env_int("TFR_PAIRWISE_LANES", 0);
  const int C = env_lanes > 0 ? env_lanes : 4;
  if (C > 64 || (C & (C - 1))) return TFR_EINVAL;
  const int T = env_threads > 0 ? env_threads : (L <= 64 ? 64 : (L <= 128 ? 128 : (L <= 512 ? 256 : 512)));
  if (T % 64 || T > 1024) return TFR_EINVAL;
  PwArgs a;
  a.logits = logits; a.labels = labels; a.mask = mask; a.item_weights = item_weights;
  a.list_weights = list_weights; a.lambda_kind = lambda_kind; a.topn = topn;
  a.smooth = smooth_fraction; a.normalized = normalized; a.gain_kind = gain_kind; a.gains = gains;
  a.discount = discount; a.L = L; a.Lp = ((L + 3) / 4) * 4 + 4; a.P = pow2_ceil(L < 2 ? 2 : L);
  a.temperature = temperature; a.C = C; a.row_loss = row_loss_out; a.row_weight = row_weight_out;
  a.nnz = nnz_out; a.dlogits = dlogits_out;
  const size_t lds = pw_smem_bytes(a.Lp, a.P);
  if (lds > 160 * 1024) return TFR_ETOOLARGE;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise_logistic_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(pairwise_logistic_kernel, dim3(B), dim3(T), lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
