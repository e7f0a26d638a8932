// Masked descending sort / ranks and the sort-based ranking metrics (NDCG@k,
// MRR@k) for gfx950.  One workgroup per list; the packed 64-bit sort keys live
// in LDS and are ordered by an in-LDS bitonic network.
//
// Reference behaviour restated: utils.py:84-195 (sort_by_scores, sorted_ranks),
// metrics_impl.py:228-266 (_prepare_and_validate_params), :122-151 (DCG),
// :429-459 (MRR), :631-670 (NDCG).
#include "common.h"
#include "../../include/tfr_hip.h"

#include <stdlib.h>

using namespace tfr;

namespace {

struct TopN { int k[TFR_MAX_TOPN]; int n; uint32_t tie_seed; };      // tie_seed: 0 = equal predictions in index order, else hashed (common.h tie_key15)

__device__ __forceinline__ bool item_valid(const float* labels, const uint8_t* mask, size_t off) {
  if (mask) return mask[off] != 0;
  if (labels) return labels[off] >= 0.0f;
  return true;
}

__global__ void sort_ranks_kernel(const float* __restrict__ scores, const float* __restrict__ labels,
                                  const uint8_t* __restrict__ mask, const int32_t* __restrict__ tiebreak,
                                  int L, int P, int32_t* __restrict__ ranks_out,
                                  int32_t* __restrict__ order_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw);
  const size_t base = (size_t)blockIdx.x * L;
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    uint64_t key = 0;
    if (i < L) {
      const bool v = item_valid(labels, mask, base + i);
      key = make_sort_key(v, scores[base + i], tiebreak ? tiebreak[base + i] : 0, i);
    }
    keys[i] = key;
  }
  block_bitonic_sort_desc(keys, P);
  for (int p = threadIdx.x; p < L; p += blockDim.x) {
    const int idx = sort_key_index(keys[p]);
    if (order_out) order_out[base + p] = idx;
    if (ranks_out) ranks_out[base + idx] = p + 1;
  }
}

// kind 0 = NDCG, 1 = MRR.  LDS per item: the 64-bit sort key, one tree-sum scratch float and the product
// w * gain -- 16 bytes (round 2: 25, with separate weight / gain / mask / un-cut-term arrays), so that the workgroup
// form reaches the 8192 items of the loss kernels (P = 8192: 128 KiB).  Weights, gains and the mask are recomputed
// from the (cache resident) global rows where they are needed; the sums run in the same order as before.
template <int KIND>
__global__ void rank_metric_kernel(const float* __restrict__ labels, const float* __restrict__ predictions,
                                   const float* __restrict__ weights, int weights_per_list,
                                   const uint8_t* __restrict__ mask, const float* __restrict__ gains,
                                   const float* __restrict__ discount, TopN topn, int B, int L, int P,
                                   float* __restrict__ metric_out, float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32] reduction scratch + dcg[8]
  float* dcg = red + 20;
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* term = reinterpret_cast<float*>(keys + P);               // [P] tree-sum scratch
  float* WG = term + P;                                           // [P] example weight x gain (relevance for MRR)

  const int b = blockIdx.x;
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;

  // ---- _prepare_and_validate_params (metrics_impl.py:228-266) of item i
  auto item = [&](int i, float& w, float& g, bool& m) {
    w = 0.f; g = 0.f; m = false;
    if (i < L) {
      const float lab = labels[base + i];
      w = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      const bool v0 = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      m = v0 && (w > 0.0f);
      const float labc = m ? lab : 0.0f;
      if (KIND == 0) g = gains ? gains[base + i] : gain_pow2m1(labc);
      else g = (labc >= 1.0f) ? 1.0f : 0.0f;
    }
  };

  // ---- per-list weight statistics, tree_sum order over the original index.
  float s_w, s_g, s_wg;
  for (int i = threadIdx.x; i < P; i += blockDim.x) { float w, g; bool m; item(i, w, g, m); term[i] = w; }
  block_tree_sum(term, P); s_w = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += blockDim.x) { float w, g; bool m; item(i, w, g, m); term[i] = g; }
  block_tree_sum(term, P); s_g = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += blockDim.x) { float w, g; bool m; item(i, w, g, m); WG[i] = w * g; term[i] = w * g; }
  block_tree_sum(term, P); s_wg = term[0]; __syncthreads();
  if (threadIdx.x == 0) {
    stats_out[(size_t)b * 3 + 0] = s_w;
    stats_out[(size_t)b * 3 + 1] = s_g;
    stats_out[(size_t)b * 3 + 2] = s_wg;
  }

  // ---- sort by prediction (masked entries last): utils.py:115-164.
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    float w, g; bool m; item(i, w, g, m);
    keys[i] = (i < L) ? make_sort_key(m, predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i) : 0ull;
    if (KIND == 1) WG[i] = g;                                  // MRR asks for the relevance of the sorted items
  }
  block_bitonic_sort_desc(keys, P);

  if (KIND == 1) {
    // MRR: first sorted position holding a relevant item (metrics_impl.py:443-459).
    float pmin = INFINITY;
    for (int p = threadIdx.x; p < L; p += blockDim.x)
      if (WG[sort_key_index(keys[p])] > 0.0f) pmin = fminf(pmin, (float)p);
    pmin = block_min(pmin, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < topn.n; ++q) {
        const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
        metric_out[(size_t)q * B + b] = (pmin < (float)k) ? (1.0f / (pmin + 1.0f)) : 0.0f;
      }
    }
    return;
  }

  // ---- DCG terms in sorted order: (w * gain) * discount(rank)  (:122-151), cut at every topn
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    for (int p = threadIdx.x; p < P; p += blockDim.x)
      term[p] = (p < k) ? WG[sort_key_index(keys[p])] * discount[p] : 0.0f;
    block_tree_sum(term, P);
    if (threadIdx.x == 0) dcg[q] = term[0];
    __syncthreads();
  }

  // ---- ideal ordering: sort by weighted gain (metrics_impl.py:660-666)
  for (int i = threadIdx.x; i < P; i += blockDim.x) {
    float w, g; bool m; item(i, w, g, m);
    keys[i] = (i < L) ? make_sort_key(m, WG[i], 0, i) : 0ull;
  }
  block_bitonic_sort_desc(keys, P);
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    for (int p = threadIdx.x; p < P; p += blockDim.x)
      term[p] = (p < k) ? WG[sort_key_index(keys[p])] * discount[p] : 0.0f;
    block_tree_sum(term, P);
    const float idcg = term[0];
    if (threadIdx.x == 0)
      metric_out[(size_t)q * B + b] = (idcg != 0.0f) ? (dcg[q] / idcg) : 0.0f;   // divide_no_nan
    __syncthreads();
  }
}

// ===========================================================================
// Wave-per-list variants (L <= 64 * IPL <= 1024): one wavefront owns a list, the
// packed 64-bit keys live in registers and are ordered by an in-register bitonic
// network (wave shuffles), the fixed-order tree sums run on registers + shuffles:
// no workgroup barriers at all.  Bit-identical to the workgroup kernels above.
// ===========================================================================
template <int IPL>
__global__ __launch_bounds__(64) void sort_ranks_wave_kernel(const float* __restrict__ scores,
                                                             const float* __restrict__ labels,
                                                             const uint8_t* __restrict__ mask,
                                                             const int32_t* __restrict__ tiebreak, int L,
                                                             int32_t* __restrict__ ranks_out,
                                                             int32_t* __restrict__ order_out) {
  const int lane = threadIdx.x;
  const size_t base = (size_t)blockIdx.x * L;
  uint64_t key[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    key[r] = 0;
    if (i < L) {
      const bool v = item_valid(labels, mask, base + i);
      key[r] = make_sort_key(v, scores[base + i], tiebreak ? tiebreak[base + i] : 0, i);
    }
  }
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    if (p < L) {
      const int idx = sort_key_index(key[r]);
      if (order_out) order_out[base + p] = idx;
      if (ranks_out) ranks_out[base + idx] = p + 1;
    }
  }
}

template <int KIND, int IPL>
__global__ __launch_bounds__(64) void rank_metric_wave_kernel(
    const float* __restrict__ labels, const float* __restrict__ predictions, const float* __restrict__ weights,
    int weights_per_list, const uint8_t* __restrict__ mask, const float* __restrict__ gains,
    const float* __restrict__ discount, TopN topn, int B, int L, int P, float* __restrict__ metric_out,
    float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* WG = reinterpret_cast<float*>(smem_raw);          // [64 * IPL] w * gain by original index
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;

  // ---- _prepare_and_validate_params (metrics_impl.py:228-266)
  float w[IPL], g[IPL], wg[IPL];
  bool m[IPL];
  uint64_t key[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    w[r] = 0.f; g[r] = 0.f; m[r] = false; key[r] = 0;
    if (i < L) {
      const float lab = labels[base + i];
      w[r] = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      const bool v0 = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      m[r] = v0 && (w[r] > 0.0f);
      const float labc = m[r] ? lab : 0.0f;
      if (KIND == 0) g[r] = gains ? gains[base + i] : gain_pow2m1(labc);
      else g[r] = (labc >= 1.0f) ? 1.0f : 0.0f;
      key[r] = make_sort_key(m[r], predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i);
    }
    wg[r] = w[r] * g[r];
    WG[i] = (KIND == 0) ? wg[r] : g[r];
  }
  // ---- per-list weight statistics, tree_sum order over the original index.
  {
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = w[r];
    const float s_w = wave_tree_sum<IPL>(t, P);
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = g[r];
    const float s_g = wave_tree_sum<IPL>(t, P);
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = wg[r];
    const float s_wg = wave_tree_sum<IPL>(t, P);
    if (lane == 0) {
      stats_out[(size_t)b * 3 + 0] = s_w;
      stats_out[(size_t)b * 3 + 1] = s_g;
      stats_out[(size_t)b * 3 + 2] = s_wg;
    }
  }
  __syncthreads();                                      // WG visible (single wave: a waitcnt)

  if (KIND == 1) {
    // MRR (metrics_impl.py:443-459): position of the first relevant item in the order
    // "prediction descending, masked last" = number of keys above the best relevant key
    // (keys are unique: they end in the item index) -- no sort needed.
    uint64_t best = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r)
      if (g[r] > 0.0f && key[r] > best) best = key[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    int above = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r) above += __popcll(__ballot(key[r] > best));
    const float pmin = (best != 0) ? (float)above : INFINITY;
    if (lane == 0) {
      for (int q = 0; q < topn.n; ++q) {
        const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
        metric_out[(size_t)q * B + b] = (pmin < (float)k) ? (1.0f / (pmin + 1.0f)) : 0.0f;
      }
    }
    return;
  }

  // ---- sort by prediction (masked entries last): utils.py:115-164.
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);

  // ---- DCG terms in sorted order: (w * gain) * discount(rank)  (:122-151)
  float term[IPL], dcg[TFR_MAX_TOPN];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    term[r] = (p < L) ? WG[sort_key_index(key[r])] * discount[p] : 0.0f;
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? term[r] : 0.0f;
    dcg[q] = wave_tree_sum<IPL>(t, P);
  }
  // ---- ideal ordering: sort by weighted gain (metrics_impl.py:660-666)
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    key[r] = (i < L) ? make_sort_key(m[r], wg[r], 0, i) : 0ull;
  }
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    term[r] = (p < L) ? WG[sort_key_index(key[r])] * discount[p] : 0.0f;
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? term[r] : 0.0f;
    const float idcg = wave_tree_sum<IPL>(t, P);
    if (lane == 0) metric_out[(size_t)q * B + b] = (idcg != 0.0f) ? (dcg[q] / idcg) : 0.0f;   // divide_no_nan
  }
}

// NDCG without the two 64-bit register sorts (the default for list_size <= 512): the DCG needs the RANK of every
// item, not the sorted sequence -- a counting sweep over the compact predictions (wave_rank_by_count: one compare + half
// a carry-add per pair, ties by index exactly like the sort keys) -- and the ideal DCG needs the weighted gains in
// descending order, which for graded labels are a handful of runs of equal values (largest remaining value + its
// multiplicity, <= kNdcgRuns rounds; more distinct values, e.g. per-item weights: ONE register sort for that half).
// The terms are scattered to their sorted position in LDS and summed with the same tree_sum pairing as before, so the
// result is bit-identical to rank_metric_wave_kernel<0> (and to the oracle).
constexpr int kNdcgRuns = 8;

// BUCKET (round 4, TFR_NDCG_BUCKET): the ranks from wave_rank_by_bucket (64-bucket partition of the score range, same
// integers) with the counting sweep as the fallback for lists it declines.
template <int IPL, bool BUCKET>
__global__ __launch_bounds__(64) void ndcg_count_wave_kernel(
    const float* __restrict__ labels, const float* __restrict__ predictions, const float* __restrict__ weights,
    int weights_per_list, const uint8_t* __restrict__ mask, const float* __restrict__ gains,
    const float* __restrict__ discount, TopN topn, int B, int L, int P, float* __restrict__ metric_out,
    float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int N = 64 * IPL;
  float* TERM = reinterpret_cast<float*>(smem_raw);        // [N] term by sorted position
  float* XS = TERM + N;                                    // [N + 8] compact predictions (pad -inf)
  int* RKS = reinterpret_cast<int*>(XS + N + 8);           // [N]
  int* OCC = RKS + N;                                      // [N]
  float* WG = reinterpret_cast<float*>(OCC + N);           // [N] w * gain by original index (sort fallback)
  float* DISC = WG + N;                                    // [N] the discount table: the gathers by rank / by sorted position
  float* BX = DISC + N;                                    // BUCKET: [N + kRankBucketMax] scores in bucket order, then
  int* HB = reinterpret_cast<int*>(BX + N + kRankBucketMax);   //     [192] histogram / fill counters / bucket starts
  const int lane = threadIdx.x, b = blockIdx.x;            //     below are LDS reads, not a dependent global round trip each
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) { const int i = lane + 64 * r; DISC[i] = (i < L) ? discount[i] : 0.0f; }

  float w[IPL], g[IPL], wg[IPL], pr[IPL];
  bool m[IPL];
  int posr[IPL];
  int n = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    w[r] = 0.f; g[r] = 0.f; m[r] = false; pr[r] = 0.f;
    if (i < L) {
      const float lab = labels[base + i];
      w[r] = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      const bool v0 = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      m[r] = v0 && (w[r] > 0.0f);
      const float labc = m[r] ? lab : 0.0f;
      g[r] = gain_pow2m1(labc);                              // (custom gains: rank_metric_wave_kernel<0>, a masked
      pr[r] = predictions[base + i];                         //  item may then carry a non-zero gain_fn(0))
    }
    wg[r] = w[r] * g[r];
    WG[i] = wg[r];
    const unsigned long long bal = __ballot(m[r]);
    posr[r] = n + __popcll(bal & ((1ull << lane) - 1ull));
    if (m[r]) XS[posr[r]] = pr[r] + 0.0f;                  // -0 -> +0 like float_to_ordered: tied zeros compare equal
    n += __popcll(bal);
  }
  {
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = w[r];
    const float s_w = wave_tree_sum<IPL>(t, P);
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = g[r];
    const float s_g = wave_tree_sum<IPL>(t, P);
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = wg[r];
    const float s_wg = wave_tree_sum<IPL>(t, P);
    if (lane == 0) {
      stats_out[(size_t)b * 3 + 0] = s_w;
      stats_out[(size_t)b * 3 + 1] = s_g;
      stats_out[(size_t)b * 3 + 2] = s_wg;
    }
  }
  const int n4 = (n + 3) >> 2;
  for (int p = n + lane; p < n4 * 4 + 4 && p < N + 8; p += 64) XS[p] = -INFINITY;
  for (int p = lane; p < N; p += 64) TERM[p] = 0.0f;
  WAVE_LDS_SYNC();

  // ---- DCG: rank of every metric-valid item among them (prediction descending, ties by index); the masked items
  // follow in the sorted order and carry w * gain = 0: their terms are the zeros TERM starts from.
  if (!BUCKET || !wave_rank_by_bucket<IPL>(XS, n, lane, RKS, OCC, BX, HB)) wave_rank_by_count(XS, n, lane, RKS, OCC);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    if (m[r]) { const int rk = RKS[posr[r]]; TERM[rk] = wg[r] * DISC[rk]; }
  }
  WAVE_LDS_SYNC();
  float term[IPL], dcg[TFR_MAX_TOPN];
#pragma unroll
  for (int r = 0; r < IPL; ++r) term[r] = TERM[lane + 64 * r];
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? term[r] : 0.0f;
    dcg[q] = wave_tree_sum<IPL>(t, P);
  }

  // ---- ideal DCG: the weighted gains of the metric-valid items in descending order = runs of equal values
  float rv[kNdcgRuns];
  int rs[kNdcgRuns];
  float rem[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) rem[r] = m[r] ? wg[r] : -INFINITY;
  int nruns = 0, pos = 0;
  bool complete = false;
#pragma unroll
  for (int it = 0; it < kNdcgRuns; ++it) {
    if (complete) break;                                     // (wave-uniform; rv / rs beyond nruns are never read)
    float mx = rem[0];
#pragma unroll
    for (int r = 1; r < IPL; ++r) mx = fmaxf(mx, rem[r]);
    const float v = wave_max_u(mx);
    rv[it] = v; rs[it] = pos;
    if (!complete) {
      if (v == -INFINITY) { complete = true; }
      else {
        int c = 0;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const bool hit = rem[r] == v;
          c += __popcll(__ballot(hit));
          rem[r] = hit ? -INFINITY : rem[r];
        }
        pos += c; nruns = it + 1;
      }
    }
  }
  if (!complete) {                                           // all kNdcgRuns rounds found a value: anything left?
    float mx = rem[0];
#pragma unroll
    for (int r = 1; r < IPL; ++r) mx = fmaxf(mx, rem[r]);
    complete = wave_max_u(mx) == -INFINITY;
  }
  if (complete) {
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      float val = 0.0f;
#pragma unroll
      for (int it = 0; it < kNdcgRuns; ++it) val = (it < nruns && p >= rs[it]) ? rv[it] : val;   // rs ascends
      term[r] = (p < pos) ? val * DISC[p] : 0.0f;        // beyond the metric-valid items: w * gain(0) = 0
    }
  } else {
    uint64_t key[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int i = lane + 64 * r;
      key[r] = (i < L) ? make_sort_key(m[r], wg[r], 0, i) : 0ull;
    }
    wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      term[r] = (p < L) ? WG[sort_key_index(key[r])] * DISC[p] : 0.0f;
    }
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    float t[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? term[r] : 0.0f;
    const float idcg = wave_tree_sum<IPL>(t, P);
    if (lane == 0) metric_out[(size_t)q * B + b] = (idcg != 0.0f) ? (dcg[q] / idcg) : 0.0f;   // divide_no_nan
  }
}

// ------------------------------------------------------------------------------------------------------------------
// NDCG@{k...} "lean" form (round 5) for the common case: built-in gain, validity from the labels, no per-item weights,
// list_size <= 256.  Same results, bit for bit, as ndcg_count_wave_kernel / rank_metric_wave_kernel<0> / the oracle (the
// terms are the same products, summed with the same tree_sum pairing); what changed is the instruction count -- the
// counters of round 4 (profiles/r04_pmc.txt) put the kernel at 2 100 VALU + 700 SALU instructions per 200-item list, of
// which ~1 000 were the THIRTEEN tree sums of a list (five cut-offs x (DCG + ideal DCG) + three statistics, ~80
// instructions each with their cut-off select chain, kernarg loads and run-time P tests), ~200 the ideal-DCG run rounds
// (a seven-step DPP maximum per distinct label value) and the load phase waited for every global load on its own:
//  * cut-offs k <= 16 (NDCG@1/3/5/10) need only positions 0..15: tree_sum over P >= 16 values that are zero beyond k
//    IS the 16-wide tree over the first 16 (x + 0.0f is exact), and four of those run at once, one per 16-lane DPP row,
//    in FOUR v_add_f32_dpp (row_shl 8, 4, 2, 1 = the pairs (i, i + 8), (i, i + 4), ... of block_tree_sum);
//  * without weights the statistics are one tree sum (sum w = list_size exactly, sum w g = sum g);
//  * graded relevance = small integers: the distinct label values come from ONE wave-wide OR of (1 << label), the run
//    lengths from ballots, the value of a sorted position from a compare / select per (run, register);
//  * P = 64 * IPL at compile time for lists beyond 64 items: no run-time level tests in the tree sums;
//  * four lists (wavefronts) per workgroup share the discount table in LDS, and a wavefront walks lists b, b + W, ...
//    with the NEXT list's labels / predictions requested before the current one is processed (unconditional loads: an
//    out-of-range lane re-reads item 0), so no load is waited for where it is issued.
// Lists whose labels are not small non-negative integers take the generic gain and a run-by-run scatter of the ideal
// terms (any number of distinct values).  Per-item weights, a mask array, custom gains, list_size > 256: the kernels above.
struct NdcgCut {
  unsigned small_k, small_q;      // up to four cut-offs <= 16, one per byte (0 = slot unused) / their rows in metric_out
  int n_large, large_k[TFR_MAX_TOPN], large_q[TFR_MAX_TOPN];     // the others (k = min(topn, list_size))
};

__device__ __forceinline__ float row16_tree(float v) {        // lane 16 q: tree_sum of the 16 values of DPP row q
#define TFR_ROW_SHL(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(x)), 0x100 + (n), 0xf, 0xf, true))
  v = v + TFR_ROW_SHL(v, 8);
  v = v + TFR_ROW_SHL(v, 4);
  v = v + TFR_ROW_SHL(v, 2);
  v = v + TFR_ROW_SHL(v, 1);
#undef TFR_ROW_SHL
  return v;
}

template <int IPL>
__global__ __launch_bounds__(256) void ndcg_lean_kernel(
    const float* __restrict__ labels, const float* __restrict__ predictions, const float* __restrict__ list_weights,
    const float* __restrict__ discount, const NdcgCut cut, const int B, const int L, const int Prt,
    float* __restrict__ metric_out, float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  constexpr int N = 64 * IPL;
  constexpr int kPerWave = 5 * N + 8 + kRankBucketMax + 192;      // 32-bit words of LDS per wavefront
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  float* DISC = reinterpret_cast<float*>(smem_raw);                 // [N] shared by the four waves
  float* TERM = DISC + N + wave * kPerWave;                         // [N] w * gain * discount by sorted position
  float* XS = TERM + N;                                             // [N + 8] compact predictions (pad -inf) / generic ideal terms
  int* RKS = reinterpret_cast<int*>(XS + N + 8);                    // [N]
  int* OCC = RKS + N;                                               // [N]
  float* BX = reinterpret_cast<float*>(OCC + N);                    // [N + kRankBucketMax]
  int* HB = reinterpret_cast<int*>(BX + N + kRankBucketMax);        // [192]
  for (int i = threadIdx.x; i < N; i += 256) DISC[i] = (i < L) ? discount[i] : 0.0f;
  __syncthreads();
  const int P = (IPL >= 2) ? N : Prt;                               // tree_sum width: pow2_ceil(L) (= N beyond 64 items)
  float disc[IPL];
  int off[IPL];
  bool in[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) { const int i = lane + 64 * r; in[r] = i < L; off[r] = in[r] ? i : 0; disc[r] = DISC[i]; }
  const int W = gridDim.x * 4;
  int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const int j16 = lane & 15, q16 = lane >> 4;
  const int kq = (int)((cut.small_k >> (8 * q16)) & 0xffu), oq = (int)((cut.small_q >> (8 * q16)) & 0xffu);

  float lab_n[IPL], pr_n[IPL], wl_n = 1.0f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) { lab_n[r] = labels[(size_t)b * L + off[r]]; pr_n[r] = predictions[(size_t)b * L + off[r]]; }
  if (list_weights) wl_n = list_weights[b];

  for (; b < B; b += W) {
    float lab[IPL], pr[IPL];
    const float wl = wl_n;
#pragma unroll
    for (int r = 0; r < IPL; ++r) { lab[r] = lab_n[r]; pr[r] = pr_n[r]; }
    {                                                               // the next list of this wave (the last one re-reads itself)
      const int bn = (b + W < B) ? b + W : b;
#pragma unroll
      for (int r = 0; r < IPL; ++r) { lab_n[r] = labels[(size_t)bn * L + off[r]]; pr_n[r] = predictions[(size_t)bn * L + off[r]]; }
      if (list_weights) wl_n = list_weights[bn];
    }
    // ---- _prepare_and_validate_params (metrics_impl.py:228-266): valid = label >= 0 and weight > 0
    const bool wpos = wl > 0.0f;
    bool m[IPL];
    float labm[IPL], g[IPL], wg[IPL];
    bool okint = true;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      m[r] = in[r] && (lab[r] >= 0.0f) && wpos;
      labm[r] = m[r] ? lab[r] : -1.0f;
      okint = okint && (!m[r] || (lab[r] < 31.0f && lab[r] == rintf(lab[r])));
    }
    const bool small_int = __ballot(!okint) == 0ull;                // wave-uniform
    unsigned present = 0u;
    if (small_int) {
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int e = m[r] ? (int)lab[r] : 0;
        g[r] = __builtin_amdgcn_ldexpf(1.0f, e) - 1.0f;              // 2^l - 1, exact (gain_pow2m1's integer branch)
        present |= m[r] ? (1u << e) : 0u;
      }
      present = wave_or_u(present);
    } else {
#pragma unroll
      for (int r = 0; r < IPL; ++r) g[r] = gain_pow2m1(m[r] ? lab[r] : 0.0f);
    }
    int posr[IPL];
    int n = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      wg[r] = in[r] ? wl * g[r] : 0.0f;                             // (1.0f * g is g: the unweighted products are the gains)
      const unsigned long long bal = __ballot(m[r]);
      posr[r] = n + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
      if (m[r]) XS[posr[r]] = pr[r] + 0.0f;                          // -0 -> +0 like float_to_ordered
      n += __popcll(bal);
    }
    // ---- per-list weight statistics (tree_sum over the original index)
    {
      float t[IPL];
      float s_w, s_g, s_wg;
#pragma unroll
      for (int r = 0; r < IPL; ++r) t[r] = in[r] ? g[r] : 0.0f;
      s_g = wave_tree_sum<IPL>(t, P);
      if (list_weights) {
#pragma unroll
        for (int r = 0; r < IPL; ++r) t[r] = in[r] ? wl : 0.0f;
        s_w = wave_tree_sum<IPL>(t, P);
#pragma unroll
        for (int r = 0; r < IPL; ++r) t[r] = wg[r];
        s_wg = wave_tree_sum<IPL>(t, P);
      } else {
        s_w = (float)L;                                             // L ones: exact in any order
        s_wg = s_g;                                                 // w = 1
      }
      if (lane == 0) {
        stats_out[(size_t)b * 3 + 0] = s_w;
        stats_out[(size_t)b * 3 + 1] = s_g;
        stats_out[(size_t)b * 3 + 2] = s_wg;
      }
    }
    {
      const int n4 = (n + 3) >> 2;
      for (int p = n + lane; p < n4 * 4 + 4 && p < N + 8; p += 64) XS[p] = -INFINITY;
    }
    WAVE_LDS_SYNC();
    // ---- DCG: rank of every valid item (prediction descending, ties by index); its term goes to its sorted position.
    // The ranks 0 .. n - 1 are a permutation: every position below n is written, the ones beyond are never read.
    if (!wave_rank_by_bucket<IPL>(XS, n, lane, RKS, OCC, BX, HB)) wave_rank_by_count(XS, n, lane, RKS, OCC);
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      if (m[r]) { const int rk = RKS[posr[r]]; TERM[rk] = wg[r] * DISC[rk]; }
    }
    WAVE_LDS_SYNC();
    float term[IPL], termi[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) term[r] = (lane + 64 * r < n) ? TERM[lane + 64 * r] : 0.0f;
    const float dsm = (j16 < kq && j16 < n) ? TERM[j16] : 0.0f;     // the cut-offs <= 16: row q16 holds positions 0 .. 15

    // ---- ideal DCG terms: the weighted gains of the valid items in descending order = runs of equal values
    if (small_int) {
      float val[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) val[r] = 0.0f;
      int pos = 0;
      unsigned bits = present & ~1u;                                // grade 0 has gain 0: its run contributes zeros
      while (bits) {                                                // (wave-uniform; one trip per distinct grade >= 1)
        const int gb = 31 - __builtin_clz(bits);
        bits &= ~(1u << gb);
        const float gv = (float)gb;
        const float v = wl * (__builtin_amdgcn_ldexpf(1.0f, gb) - 1.0f);      // the same product as the items' wg
        int c = 0;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          c += __popcll(__ballot(labm[r] == gv));
          val[r] = (lane + 64 * r >= pos) ? v : val[r];
        }
        pos += c;
      }
#pragma unroll
      for (int r = 0; r < IPL; ++r) termi[r] = (lane + 64 * r >= pos) ? 0.0f : val[r] * disc[r];
    } else {
      // any label values: repeatedly take the largest remaining w * gain; its items go to the next sorted positions
      // (equal values: any order among them gives the same terms).  XS is free again: the ranks are done.
      float rem[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) rem[r] = m[r] ? wg[r] : -INFINITY;
      int pos = 0;
      for (int it = 0; it <= N; ++it) {                             // (every trip retires >= 1 item: the bound is never reached)
        float mx = rem[0];
#pragma unroll
        for (int r = 1; r < IPL; ++r) mx = fmaxf(mx, rem[r]);
        const float v = wave_max_u(mx);
        if (!(v > -INFINITY)) break;
        int c = 0;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const bool hit = rem[r] == v;
          const unsigned long long bal = __ballot(hit);
          if (hit) {
            const int sp = pos + c + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(bal >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)bal, 0u));
            XS[sp] = v * DISC[sp];
            rem[r] = -INFINITY;
          }
          c += __popcll(bal);
        }
        pos += c;
      }
      WAVE_LDS_SYNC();
#pragma unroll
      for (int r = 0; r < IPL; ++r) termi[r] = (lane + 64 * r < pos) ? XS[lane + 64 * r] : 0.0f;
    }
    // ---- the cut-offs <= 16, four at a time (one per DPP row)
    if (cut.small_k) {
      const float ti = __shfl(termi[0], j16, 64);
      const float ism = (j16 < kq) ? ti : 0.0f;
      const float dsum = row16_tree(dsm), isum = row16_tree(ism);
      if (j16 == 0 && kq > 0) metric_out[(size_t)oq * B + b] = (isum != 0.0f) ? (dsum / isum) : 0.0f;   // divide_no_nan
    }
    // ---- the others (k = list_size: NDCG over the whole list)
    for (int ql = 0; ql < cut.n_large; ++ql) {
      const int k = cut.large_k[ql];
      float t[IPL];
#pragma unroll
      for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? term[r] : 0.0f;
      const float dcg = wave_tree_sum<IPL>(t, P);
#pragma unroll
      for (int r = 0; r < IPL; ++r) t[r] = (lane + 64 * r < k) ? termi[r] : 0.0f;
      const float idcg = wave_tree_sum<IPL>(t, P);
      if (lane == 0) metric_out[(size_t)cut.large_q[ql] * B + b] = (idcg != 0.0f) ? (dcg / idcg) : 0.0f;
    }
    WAVE_LDS_SYNC();                                                // TERM / XS are rewritten by the next list
  }
}

// The other sort-based metrics of metrics_impl.py on the same wave-per-list machinery
// (kind is wave-uniform at run time):
//   TFR_METRIC_DCG       :673-705  sum_{p<k} w gain disc(p)   (divided by the list weight by the caller)
//   TFR_METRIC_HITS      :462-506  1{some relevant item in the top k}                 (no sort needed)
//   TFR_METRIC_RECALL    :154-177, 539-561   #relevant in top k / #relevant
//   TFR_METRIC_PRECISION :180-207, 564-586   #relevant in top k / min(k, #valid)
//   TFR_METRIC_MAP       :589-628  sum_{p<k} prec@p w rel / sum w rel
//   TFR_METRIC_ARP       :509-536  sum_p p (w l) / sum_p (w l), sums in SORTED order; stats[2] = that denominator
// "relevant" = label >= 1.  Sums of floats use the shared tree_sum order (bit-reproducible).
template <int IPL>
__global__ __launch_bounds__(64) void rank_metric2_wave_kernel(
    int kind, const float* __restrict__ labels, const float* __restrict__ predictions,
    const float* __restrict__ weights, int weights_per_list, const uint8_t* __restrict__ mask,
    const float* __restrict__ gains, const float* __restrict__ discount, TopN topn, int B, int L, int P,
    float* __restrict__ metric_out, float* __restrict__ stats_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* WG = reinterpret_cast<float*>(smem_raw);          // [64 * IPL] w * rel by original index
  float* G = WG + 64 * IPL;                                // [64 * IPL] rel by original index
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;

  float w[IPL], g[IPL], wg[IPL];
  bool m[IPL];
  uint64_t key[IPL];
  int nmask = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    w[r] = 0.f; g[r] = 0.f; m[r] = false; key[r] = 0;
    if (i < L) {
      const float lab = labels[base + i];
      w[r] = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      const bool v0 = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      m[r] = v0 && (w[r] > 0.0f);
      const float labc = m[r] ? lab : 0.0f;
      if (kind == TFR_METRIC_DCG) g[r] = gains ? gains[base + i] : gain_pow2m1(labc);
      else if (kind == TFR_METRIC_ARP || kind == TFR_METRIC_PWA) g[r] = labc;
      else g[r] = (labc >= 1.0f) ? 1.0f : 0.0f;
      key[r] = make_sort_key(m[r], predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i);
    }
    wg[r] = w[r] * g[r];
    WG[i] = wg[r];
    G[i] = g[r];
    nmask += __popcll(__ballot(m[r]));
  }
  float t[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = w[r];
  const float s_w = wave_tree_sum<IPL>(t, P);
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = g[r];
  const float s_g = wave_tree_sum<IPL>(t, P);
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = wg[r];
  const float s_wg = wave_tree_sum<IPL>(t, P);
  if (lane == 0) {
    stats_out[(size_t)b * 3 + 0] = s_w;
    stats_out[(size_t)b * 3 + 1] = s_g;
    if (kind != TFR_METRIC_ARP) stats_out[(size_t)b * 3 + 2] = s_wg;
  }
  __syncthreads();

  if (kind == TFR_METRIC_OPA) {
    // OPAMetric (:708-743): sum over ordered pairs (i, j) of valid items with y_i > y_j of w_i, and of
    // w_i [s_i > s_j].  Lanes own the items i and sweep j through LDS (label = +inf marks "not a valid j").
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int i = lane + 64 * r;
      if (i < L) { G[i] = m[r] ? labels[base + i] : INFINITY; WG[i] = predictions[base + i]; }
    }
    __syncthreads();
    float pw[IPL], cw[IPL];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int i = lane + 64 * r;
      int np = 0, nc = 0;
      if (m[r]) {
        const float li = G[i], si = WG[i];
        for (int j = 0; j < L; ++j) {
          const bool gt = li > G[j];
          np += gt ? 1 : 0;
          nc += (gt && si > WG[j]) ? 1 : 0;
        }
      }
      pw[r] = w[r] * (float)np;
      cw[r] = w[r] * (float)nc;
    }
    const float tw = wave_tree_sum<IPL>(pw, P);
    const float tc = wave_tree_sum<IPL>(cw, P);
    if (lane == 0) {
      stats_out[(size_t)b * 3 + 2] = tw;
      metric_out[b] = (tw != 0.0f) ? tc / tw : 0.0f;
    }
    return;
  }

  if (kind == TFR_METRIC_HITS) {
    uint64_t best = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r)
      if (g[r] > 0.0f && key[r] > best) best = key[r];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const uint64_t other = __shfl_xor(best, o, 64);
      best = other > best ? other : best;
    }
    int above = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r) above += __popcll(__ballot(key[r] > best));
    if (lane == 0) {
      for (int q = 0; q < topn.n; ++q) {
        const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
        metric_out[(size_t)q * B + b] = (best != 0 && above < k) ? 1.0f : 0.0f;
      }
    }
    return;
  }

  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);
  float rel[IPL], wr[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    const int idx = sort_key_index(key[r]);
    rel[r] = (p < L) ? G[idx] : 0.0f;
    wr[r] = (p < L) ? WG[idx] : 0.0f;
  }
  const bool bpref = (kind == TFR_METRIC_BPREF || kind == TFR_METRIC_BPREF_NONTREC);
  float cum[IPL];                             // inclusive prefix count of relevant (MAP) / irrelevant (BPref) items
  if (kind == TFR_METRIC_MAP || bpref) {
    float carry = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      float v = bpref ? ((lane + 64 * r < nmask) ? 1.0f - rel[r] : 0.0f) : rel[r];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const float u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
      }
      cum[r] = v + carry;
      carry += __shfl(v, 63, 64);
    }
  }
  float arp_den = 0.f;
  if (kind == TFR_METRIC_ARP) {
#pragma unroll
    for (int r = 0; r < IPL; ++r) t[r] = wr[r];
    arp_den = wave_tree_sum<IPL>(t, P);
    if (lane == 0) stats_out[(size_t)b * 3 + 2] = arp_den;
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      float v = 0.f;
      if (p < k) {
        if (kind == TFR_METRIC_DCG) v = wr[r] * discount[p];
        else if (kind == TFR_METRIC_MAP) v = (cum[r] / (float)(p + 1)) * wr[r];
        else if (kind == TFR_METRIC_ARP) v = (float)(p + 1) * wr[r];
        else if (bpref) {                                    // (1 - min(#irrelevant above, R) / den) * rel   (:879-892)
          const float num = fminf(cum[r], s_g);
          const float den = (kind == TFR_METRIC_BPREF) ? fminf((float)nmask - s_g, s_g) : s_g;
          v = (1.0f - ((den != 0.0f) ? num / den : 0.0f)) * rel[r];
        } else if (kind == TFR_METRIC_PWA) v = (p < nmask) ? rel[r] * (1.0f / (float)(p + 1)) : 0.0f;   // (:946-961)
        else v = rel[r];                                     // recall / precision: count
      }
      t[r] = v;
    }
    const float total = wave_tree_sum<IPL>(t, P);
    float out = total;
    if (bpref) out = (s_g != 0.0f) ? total / s_g : 0.0f;
    if (kind == TFR_METRIC_PWA) {
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        const int p = lane + 64 * r;
        t[r] = (p < k && p < nmask) ? 1.0f / (float)(p + 1) : 0.0f;
      }
      const float den = wave_tree_sum<IPL>(t, P);
      out = (den != 0.0f) ? total / den : 0.0f;
    }
    if (kind == TFR_METRIC_RECALL) out = (s_g != 0.0f) ? total / s_g : 0.0f;
    else if (kind == TFR_METRIC_PRECISION) { const int d = k < nmask ? k : nmask; out = d > 0 ? total / (float)d : 0.0f; }
    else if (kind == TFR_METRIC_MAP) out = (s_wg != 0.0f) ? total / s_wg : 0.0f;
    else if (kind == TFR_METRIC_ARP) out = (arp_den != 0.0f) ? total / arp_den : 0.0f;
    if (lane == 0) metric_out[(size_t)q * B + b] = out;
  }
}

// Diversity metrics on [B, L, S] subtopic labels (metrics_impl.py:313-426 _DivRankingMetric, :36-59
// _alpha_dcg_gain_fn, :785-822 AlphaDCGMetric, :746-782 PrecisionIAMetric).  One wavefront per list:
// sort by prediction (valid first), then one exclusive scan per subtopic along the ranking.
//   TFR_DIV_ALPHA_DCG   : metric_out = sum_{p<k} w_p * (sum_s y_ps (1 - alpha)^{#covered_s before p}) * discount[p]
//                         (the caller divides by the per-list weight, like TFR_METRIC_DCG)
//   TFR_DIV_PRECISION_IA: metric_out = sum_{p<k} sum_s [y_ps >= 1] / (min(k, #valid) * #subtopics with a relevant item)
// stats_out[b] = (sum w, sum rel, sum w*rel) with rel_i = any_s [y_is >= 1]: the per-list weight statistics.
template <int IPL>
__global__ __launch_bounds__(64) void div_metric_wave_kernel(
    int kind, const float* __restrict__ labels, const float* __restrict__ predictions,
    const float* __restrict__ weights, int weights_per_list, const uint8_t* __restrict__ mask,
    const float* __restrict__ discount, float alpha, TopN topn, int B, int L, int S, int P,
    float* __restrict__ metric_out, float* __restrict__ stats_out) {
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;

  float w[IPL], g[IPL];
  bool m[IPL];
  uint64_t key[IPL];
  int nmask = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    w[r] = 0.f; g[r] = 0.f; m[r] = false; key[r] = 0;
    if (i < L) {
      const float* lab = labels + (base + i) * (size_t)S;
      w[r] = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      bool any_valid = false, any_rel = false;
      for (int t = 0; t < S; ++t) { const float y = lab[t]; any_valid |= (y >= 0.0f); any_rel |= (y >= 1.0f); }
      m[r] = mask ? (mask[base + i] != 0) : any_valid;        // :354-358 (a rank-3 mask is reduced by the caller)
      g[r] = (m[r] && any_rel) ? 1.0f : 0.0f;
      key[r] = make_sort_key(m[r], predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i);
    }
    nmask += __popcll(__ballot(m[r]));
  }
  float t[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = w[r];
  const float s_w = wave_tree_sum<IPL>(t, P);
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = g[r];
  const float s_g = wave_tree_sum<IPL>(t, P);
#pragma unroll
  for (int r = 0; r < IPL; ++r) t[r] = w[r] * g[r];
  const float s_wg = wave_tree_sum<IPL>(t, P);
  if (lane == 0) {
    stats_out[(size_t)b * 3 + 0] = s_w;
    stats_out[(size_t)b * 3 + 1] = s_g;
    stats_out[(size_t)b * 3 + 2] = s_wg;
  }

  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);
  int idx[IPL];
  float ws[IPL], val[IPL];
  bool ms[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    idx[r] = sort_key_index(key[r]);
    ms[r] = p < nmask;                                        // valid-first order
    ws[r] = (p < L) ? (weights ? (weights_per_list ? wl : weights[base + idx[r]]) : 1.0f) : 0.0f;
    val[r] = 0.f;
  }
  float n_sub = 0.f;
  const float one_m_alpha = 1.0f - alpha;
  for (int st = 0; st < S; ++st) {
    float y[IPL];
    bool any = false;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      y[r] = (p < L && ms[r]) ? labels[(base + idx[r]) * (size_t)S + st] : 0.0f;
      any |= (y[r] >= 1.0f);
    }
    if (kind == TFR_DIV_PRECISION_IA) {
      n_sub += (__ballot(any) != 0ull) ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < IPL; ++r) val[r] += (y[r] >= 1.0f) ? 1.0f : 0.0f;
    } else {
      float carry = 0.f;
#pragma unroll
      for (int r = 0; r < IPL; ++r) {
        float v = y[r];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const float u = __shfl_up(v, o, 64);
          if (lane >= o) v += u;
        }
        float ex = __shfl_up(v, 1, 64);
        if (lane == 0) ex = 0.0f;
        const float cum = ex + carry;                          // tf.cumsum(exclusive=True) along the ranking
        carry += __shfl(v, 63, 64);
        val[r] += y[r] * powf(one_m_alpha, cum);
      }
    }
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      float v = 0.f;
      if (p < k) v = (kind == TFR_DIV_PRECISION_IA) ? val[r] : (ws[r] * val[r]) * discount[p];
      t[r] = v;
    }
    const float total = wave_tree_sum<IPL>(t, P);
    float out = total;
    if (kind == TFR_DIV_PRECISION_IA) {
      const float den = (float)(k < nmask ? k : nmask) * n_sub;
      out = (den != 0.0f) ? total / den : 0.0f;
    }
    if (lane == 0) metric_out[(size_t)q * B + b] = out;
  }
}

// ===========================================================================
// Workgroup-per-list forms of the two kernels above for LONG lists (list_size > 512): the register bitonic
// network of the wave kernels does not fit the register file beyond 8 keys per lane (the 16-key forms spilled
// 0.6 - 1.7 KB per lane).  Keys, the per-item values and the sorted-order values live in LDS (32 B per item:
// list_size <= 4096), sorts are block_bitonic_sort_desc, sums block_tree_sum over the same zero-padded power of
// two -- the same pairing as wave_tree_sum, so results are bit-identical to the wave kernels.
// ===========================================================================
// inclusive prefix sum of a[0 .. P) (Hillis-Steele ping-pong between `a` and `scratch`, P a power of two; the result
// ends in `a`).  Values are item counts / small label sums: the wave kernels scan in a different association, equal
// for integer-valued data.
__device__ __forceinline__ void block_inclusive_scan(float* a, float* scratch, int P) {
  float* src = a;
  float* dst = scratch;
  for (int o = 1; o < P; o <<= 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) dst[i] = src[i] + ((i >= o) ? src[i - o] : 0.0f);
    float* t = src; src = dst; dst = t;
  }
  __syncthreads();
  if (src != a) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) a[i] = src[i];
    __syncthreads();
  }
}

// BIG (round 5, list_size > 4096): the six per-position arrays live in a slot of the caller's workspace (global memory)
// instead of LDS -- see csrc/listwise.hip; one workgroup per slot walks the lists with a grid stride.
template <bool BIG>
__global__ void rank_metric2_block_kernel(
    int kind, const float* __restrict__ labels, const float* __restrict__ predictions,
    const float* __restrict__ weights, int weights_per_list, const uint8_t* __restrict__ mask,
    const float* __restrict__ gains, const float* __restrict__ discount, TopN topn, int B, int L, int P,
    float* __restrict__ metric_out, float* __restrict__ stats_out, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* W = BIG ? ws + (size_t)blockIdx.x * 6 * P : reinterpret_cast<float*>(keys + P);   // [P] weight by original index
  float* G = W + P;                                               // [P] relevance / gain by original index
  float* REL = G + P;                                             // [P] relevance in sorted order
  float* WR = REL + P;                                            // [P] w * rel in sorted order
  float* CUM = WR + P;                                            // [P] prefix counts
  float* term = CUM + P;                                          // [P] tree-sum scratch
  const int T = blockDim.x;
 for (int b = blockIdx.x; b < B; b += gridDim.x) {                // (one list per workgroup unless BIG)
  if (BIG) __syncthreads();
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;

  float nm = 0.f;
  for (int i = threadIdx.x; i < P; i += T) {
    float w = 0.f, g = 0.f;
    bool m = false;
    uint64_t key = 0;
    if (i < L) {
      const float lab = labels[base + i];
      w = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      const bool v0 = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      m = v0 && (w > 0.0f);
      const float labc = m ? lab : 0.0f;
      if (kind == TFR_METRIC_DCG) g = gains ? gains[base + i] : gain_pow2m1(labc);
      else if (kind == TFR_METRIC_ARP || kind == TFR_METRIC_PWA) g = labc;
      else g = (labc >= 1.0f) ? 1.0f : 0.0f;
      key = make_sort_key(m, predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i);
    }
    W[i] = w; G[i] = g; keys[i] = key;
    nm += m ? 1.0f : 0.0f;
  }
  const int nmask = (int)block_sum(nm, red);
  __syncthreads();
  float s_w, s_g, s_wg;
  for (int i = threadIdx.x; i < P; i += T) term[i] = W[i];
  block_tree_sum(term, P); s_w = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += T) term[i] = G[i];
  block_tree_sum(term, P); s_g = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += T) term[i] = W[i] * G[i];
  block_tree_sum(term, P); s_wg = term[0]; __syncthreads();
  if (threadIdx.x == 0) {
    stats_out[(size_t)b * 3 + 0] = s_w;
    stats_out[(size_t)b * 3 + 1] = s_g;
    if (kind != TFR_METRIC_ARP) stats_out[(size_t)b * 3 + 2] = s_wg;
  }

  if (kind == TFR_METRIC_OPA) {
    // (:708-743) REL = label (+inf: not a valid j), WR = prediction, by original index; CUM / term = the two sums
    for (int i = threadIdx.x; i < P; i += T) {
      const bool m = i < L && (keys[i] >> 63) != 0;
      REL[i] = m ? labels[base + i] : INFINITY;
      WR[i] = i < L ? predictions[base + i] : 0.0f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += T) {
      int np = 0, nc = 0;
      if (i < L && (keys[i] >> 63) != 0) {
        const float li = REL[i], si = WR[i];
        for (int j = 0; j < L; ++j) {
          const bool gt = li > REL[j];
          np += gt ? 1 : 0;
          nc += (gt && si > WR[j]) ? 1 : 0;
        }
      }
      CUM[i] = W[i] * (float)np;
      term[i] = W[i] * (float)nc;
    }
    block_tree_sum(term, P);
    const float tc = term[0];
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += T) term[i] = CUM[i];
    block_tree_sum(term, P);
    const float tw = term[0];
    if (threadIdx.x == 0) {
      stats_out[(size_t)b * 3 + 2] = tw;
      metric_out[b] = (tw != 0.0f) ? tc / tw : 0.0f;
    }
    continue;
  }

  block_bitonic_sort_desc(keys, P);                              // (Hits too: the first relevant sorted position)
  for (int p = threadIdx.x; p < P; p += T) {
    float r = 0.f, wr = 0.f;
    if (p < L) { const int idx = sort_key_index(keys[p]); r = G[idx]; wr = W[idx] * G[idx]; }
    REL[p] = r; WR[p] = wr;
  }
  __syncthreads();
  if (kind == TFR_METRIC_HITS) {
    float pmin = INFINITY;
    for (int p = threadIdx.x; p < L; p += T) if (REL[p] > 0.0f && (keys[p] >> 63) != 0) pmin = fminf(pmin, (float)p);
    pmin = block_min(pmin, red);
    if (threadIdx.x == 0) {
      for (int q = 0; q < topn.n; ++q) {
        const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
        metric_out[(size_t)q * B + b] = (pmin < (float)k) ? 1.0f : 0.0f;
      }
    }
    continue;
  }
  const bool bpref = (kind == TFR_METRIC_BPREF || kind == TFR_METRIC_BPREF_NONTREC);
  if (kind == TFR_METRIC_MAP || bpref) {
    for (int p = threadIdx.x; p < P; p += T) CUM[p] = bpref ? ((p < nmask) ? 1.0f - REL[p] : 0.0f) : REL[p];
    block_inclusive_scan(CUM, term, P);
  }
  float arp_den = 0.f;
  if (kind == TFR_METRIC_ARP) {
    for (int p = threadIdx.x; p < P; p += T) term[p] = WR[p];
    block_tree_sum(term, P);
    arp_den = term[0];
    __syncthreads();
    if (threadIdx.x == 0) stats_out[(size_t)b * 3 + 2] = arp_den;
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    for (int p = threadIdx.x; p < P; p += T) {
      float v = 0.f;
      if (p < k) {
        if (kind == TFR_METRIC_DCG) v = WR[p] * discount[p];
        else if (kind == TFR_METRIC_MAP) v = (CUM[p] / (float)(p + 1)) * WR[p];
        else if (kind == TFR_METRIC_ARP) v = (float)(p + 1) * WR[p];
        else if (bpref) {
          const float num = fminf(CUM[p], s_g);
          const float den = (kind == TFR_METRIC_BPREF) ? fminf((float)nmask - s_g, s_g) : s_g;
          v = (1.0f - ((den != 0.0f) ? num / den : 0.0f)) * REL[p];
        } else if (kind == TFR_METRIC_PWA) v = (p < nmask) ? REL[p] * (1.0f / (float)(p + 1)) : 0.0f;
        else v = REL[p];
      }
      term[p] = v;
    }
    block_tree_sum(term, P);
    const float total = term[0];
    __syncthreads();
    float out = total;
    if (bpref) out = (s_g != 0.0f) ? total / s_g : 0.0f;
    if (kind == TFR_METRIC_PWA) {
      for (int p = threadIdx.x; p < P; p += T) term[p] = (p < k && p < nmask) ? 1.0f / (float)(p + 1) : 0.0f;
      block_tree_sum(term, P);
      const float den = term[0];
      __syncthreads();
      out = (den != 0.0f) ? total / den : 0.0f;
    }
    if (kind == TFR_METRIC_RECALL) out = (s_g != 0.0f) ? total / s_g : 0.0f;
    else if (kind == TFR_METRIC_PRECISION) { const int d = k < nmask ? k : nmask; out = d > 0 ? total / (float)d : 0.0f; }
    else if (kind == TFR_METRIC_MAP) out = (s_wg != 0.0f) ? total / s_wg : 0.0f;
    else if (kind == TFR_METRIC_ARP) out = (arp_den != 0.0f) ? total / arp_den : 0.0f;
    if (threadIdx.x == 0) metric_out[(size_t)q * B + b] = out;
  }
 }
}

template <bool BIG>
__global__ void div_metric_block_kernel(
    int kind, const float* __restrict__ labels, const float* __restrict__ predictions,
    const float* __restrict__ weights, int weights_per_list, const uint8_t* __restrict__ mask,
    const float* __restrict__ discount, float alpha, TopN topn, int B, int L, int S, int P,
    float* __restrict__ metric_out, float* __restrict__ stats_out, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* W = BIG ? ws + (size_t)blockIdx.x * 5 * P : reinterpret_cast<float*>(keys + P);   // [P] weight by original index, then in sorted order
  float* G = W + P;                                               // [P] rel by original index
  float* VAL = G + P;                                             // [P] per-position value, sorted order
  float* Y = VAL + P;                                             // [P] one subtopic's labels / their prefix
  float* term = Y + P;                                            // [P]
  const int T = blockDim.x;
 for (int b = blockIdx.x; b < B; b += gridDim.x) {                // (one list per workgroup unless BIG)
  if (BIG) __syncthreads();
  const size_t base = (size_t)b * L;
  const float wl = (weights && weights_per_list) ? weights[b] : 1.0f;
  float nm = 0.f;
  for (int i = threadIdx.x; i < P; i += T) {
    float w = 0.f, g = 0.f;
    uint64_t key = 0;
    if (i < L) {
      const float* lab = labels + (base + i) * (size_t)S;
      w = weights ? (weights_per_list ? wl : weights[base + i]) : 1.0f;
      bool any_valid = false, any_rel = false;
      for (int t = 0; t < S; ++t) { const float y = lab[t]; any_valid |= (y >= 0.0f); any_rel |= (y >= 1.0f); }
      const bool m = mask ? (mask[base + i] != 0) : any_valid;
      g = (m && any_rel) ? 1.0f : 0.0f;
      key = make_sort_key(m, predictions[base + i], tie_key15(topn.tie_seed, (uint32_t)b, (uint32_t)i), i);
      nm += m ? 1.0f : 0.0f;
    }
    W[i] = w; G[i] = g; keys[i] = key;
  }
  const int nmask = (int)block_sum(nm, red);
  __syncthreads();
  float s_w, s_g, s_wg;
  for (int i = threadIdx.x; i < P; i += T) term[i] = W[i];
  block_tree_sum(term, P); s_w = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += T) term[i] = G[i];
  block_tree_sum(term, P); s_g = term[0]; __syncthreads();
  for (int i = threadIdx.x; i < P; i += T) term[i] = W[i] * G[i];
  block_tree_sum(term, P); s_wg = term[0]; __syncthreads();
  if (threadIdx.x == 0) {
    stats_out[(size_t)b * 3 + 0] = s_w;
    stats_out[(size_t)b * 3 + 1] = s_g;
    stats_out[(size_t)b * 3 + 2] = s_wg;
  }
  block_bitonic_sort_desc(keys, P);
  for (int p = threadIdx.x; p < P; p += T) {
    float ws = 0.f;
    if (p < L) { const int idx = sort_key_index(keys[p]); ws = weights ? (weights_per_list ? wl : weights[base + idx]) : 1.0f; }
    G[p] = ws;                                                   // G is free now: weight in sorted order
    VAL[p] = 0.f;
  }
  __syncthreads();
  float n_sub = 0.f;
  const float one_m_alpha = 1.0f - alpha;
  for (int st = 0; st < S; ++st) {
    float anyf = 0.f;
    for (int p = threadIdx.x; p < P; p += T) {
      float y = 0.f;
      if (p < L && p < nmask) y = labels[(base + sort_key_index(keys[p])) * (size_t)S + st];
      Y[p] = y;
      term[p] = y;
      anyf = fmaxf(anyf, (y >= 1.0f) ? 1.0f : 0.0f);
    }
    if (kind == TFR_DIV_PRECISION_IA) {
      n_sub += block_max(anyf, red);
      __syncthreads();
      for (int p = threadIdx.x; p < P; p += T) VAL[p] += (Y[p] >= 1.0f) ? 1.0f : 0.0f;
      __syncthreads();
    } else {
      block_inclusive_scan(term, W, P);                          // inclusive (W is free: scratch); exclusive below
      for (int p = threadIdx.x; p < P; p += T) {
        const float cum = (p > 0) ? term[p - 1] : 0.0f;          // tf.cumsum(exclusive=True) along the ranking
        VAL[p] += Y[p] * powf(one_m_alpha, cum);
      }
      __syncthreads();
    }
  }
  for (int q = 0; q < topn.n; ++q) {
    const int k = (topn.k[q] <= 0 || topn.k[q] > L) ? L : topn.k[q];
    for (int p = threadIdx.x; p < P; p += T) {
      float v = 0.f;
      if (p < k) v = (kind == TFR_DIV_PRECISION_IA) ? VAL[p] : (G[p] * VAL[p]) * discount[p];
      term[p] = v;
    }
    block_tree_sum(term, P);
    const float total = term[0];
    __syncthreads();
    float out = total;
    if (kind == TFR_DIV_PRECISION_IA) {
      const float den = (float)(k < nmask ? k : nmask) * n_sub;
      out = (den != 0.0f) ? total / den : 0.0f;
    }
    if (threadIdx.x == 0) metric_out[(size_t)q * B + b] = out;
  }
 }
}

// Per-list metric weights from the per-list statistics (metrics_impl.py:63-119
// _per_example_weights_to_per_list_weights): w_b = sum(w rel) / sum(rel) for a list with relevance, the batch mean
// of those for a list without, 0 for a list whose weights are all 0.  Two passes over [B, 3]; the batch mean (pass 1)
// is computed by EVERY workgroup in the same fixed order (identical in all of them), pass 2 is split over the
// workgroups: the one-workgroup form was a 12.6 us serial tail of the NDCG step at B = 16384.
__global__ __launch_bounds__(1024) void metric_list_weights_kernel(const float* __restrict__ stats, int B,
                                                                    float* __restrict__ out) {
  __shared__ float red[2][16];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  float cnt = 0.f, sum = 0.f;
  int b = tid;
  // Round 6: SIXTEEN lists in flight per thread (B = 16 384: the whole of pass 1 behind ONE memory round trip instead of
  // four dependent ones -- the kernel is 6.6 us of latency, 20 % of the NDCG metric step), still added in index order
  for (; b + 15 * 1024 < B; b += 16 * 1024) {
    float sw[16], sr[16], swr[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t o = (size_t)(b + u * 1024) * 3;
      sw[u] = stats[o]; sr[u] = stats[o + 1]; swr[u] = stats[o + 2];
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      cnt += (sw[u] > 0.0f && sr[u] > 0.0f) ? 1.0f : 0.0f;
      sum += (sr[u] != 0.0f) ? swr[u] / sr[u] : 0.0f;
    }
  }
  for (; b + 3 * 1024 < B; b += 4 * 1024) {                 // four lists in flight per thread, added in index order
    float sw[4], sr[4], swr[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const size_t o = (size_t)(b + u * 1024) * 3;
      sw[u] = stats[o]; sr[u] = stats[o + 1]; swr[u] = stats[o + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      cnt += (sw[u] > 0.0f && sr[u] > 0.0f) ? 1.0f : 0.0f;
      sum += (sr[u] != 0.0f) ? swr[u] / sr[u] : 0.0f;
    }
  }
  for (; b < B; b += 1024) {
    const float sw = stats[(size_t)b * 3], sr = stats[(size_t)b * 3 + 1], swr = stats[(size_t)b * 3 + 2];
    cnt += (sw > 0.0f && sr > 0.0f) ? 1.0f : 0.0f;
    sum += (sr != 0.0f) ? swr / sr : 0.0f;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sum += __shfl_xor(sum, o, 64); }
  if (lane == 0) { red[0][wid] = cnt; red[1][wid] = sum; }
  __syncthreads();
  float tc = 0.f, ts = 0.f;
  for (int w = 0; w < 16; ++w) { tc += red[0][w]; ts += red[1][w]; }
  const float avg = (tc > 0.0f) ? ts / tc : 1.0f;
  for (int b2 = blockIdx.x * 1024 + tid; b2 < B; b2 += gridDim.x * 1024) {      // (L1 / L2 hits: pass 1 read every line)
    const float sw = stats[(size_t)b2 * 3], sr = stats[(size_t)b2 * 3 + 1], swr = stats[(size_t)b2 * 3 + 2];
    out[b2] = (sw > 0.0f) ? ((sr > 0.0f) ? swr / sr : avg) : 0.0f;
  }
}

template <int KIND, int IPL>
void launch_metric_wave(const float* labels, const float* predictions, const float* weights, int weights_per_list,
                        const uint8_t* mask, const float* gains, const float* discount, const TopN& tn, int B,
                        int L, int P, float* metric_out, float* stats_out, hipStream_t st) {
  static const int env_count = [] { const char* e = getenv("TFR_NDCG_COUNT"); return (e && *e) ? atoi(e) : 1; }();
  static const int env_lean = [] { const char* e = getenv("TFR_NDCG_LEAN"); return (e && *e) ? atoi(e) : 1; }();
  // (a tie seed -- equal predictions in a hashed order -- goes through the sort kernel below: its key has the tie field)
  if (KIND == 0 && IPL <= 4 && env_lean && env_count && !gains && !mask && (!weights || weights_per_list) && !tn.tie_seed) {
    // the lean form (round 5): cut-offs <= 16 packed four to a launch pass, the rest through full tree sums
    NdcgCut cut;
    cut.small_k = 0u; cut.small_q = 0u; cut.n_large = 0;
    int ns = 0;
    for (int q = 0; q < tn.n; ++q) {
      const int k = (tn.k[q] <= 0 || tn.k[q] > L) ? L : tn.k[q];
      if (k <= 16 && ns < 4) { cut.small_k |= (unsigned)k << (8 * ns); cut.small_q |= (unsigned)q << (8 * ns); ++ns; }
      else { cut.large_k[cut.n_large] = k; cut.large_q[cut.n_large] = q; ++cut.n_large; }
    }
    for (int q = cut.n_large; q < TFR_MAX_TOPN; ++q) { cut.large_k[q] = 0; cut.large_q[q] = 0; }
    constexpr int N = 64 * IPL;
    constexpr size_t lds = (size_t)(N + 4 * (5 * N + 8 + kRankBucketMax + 192)) * sizeof(float);
    // a wavefront walks `per` lists.  Measured (B = 16 384, L = 200, 1.3 GB working set, profiles/r05_ndcg_ab.txt): 2 048
    // wavefronts 42.5 us, 4 096 28.9, 6 144 (one resident round) 31.9, 8 192 27.4, 16 384 (one list per wavefront) 26.6 --
    // the dispatcher balances short and long lists better than a static walk, so a wavefront only walks several lists in
    // batches beyond 65 536 lists
    static const int env_waves = [] { const char* e = getenv("TFR_NDCG_LEAN_WAVES"); return (e && *e) ? atoi(e) : 65536; }();
    const int per = (B + env_waves - 1) / env_waves;
    const int waves = (B + per - 1) / per;
    hipLaunchKernelGGL((ndcg_lean_kernel<IPL>), dim3((waves + 3) / 4), dim3(256), lds, st, labels, predictions, weights,
                       discount, cut, B, L, P, metric_out, stats_out);
    return;
  }
  if (KIND == 0 && env_count && !gains && !tn.tie_seed) {          // NDCG with the built-in gain: ranks by counting, no register sort
    constexpr size_t lds = (size_t)64 * IPL * 6 * sizeof(float) + 8 * sizeof(float);
    static const int env_bucket = [] { const char* e = getenv("TFR_NDCG_BUCKET"); return (e && *e) ? atoi(e) : 1; }();
    if (env_bucket) {
      constexpr size_t lds_b = lds + (size_t)(64 * IPL + kRankBucketMax + 192) * sizeof(float);
      hipLaunchKernelGGL((ndcg_count_wave_kernel<IPL, true>), dim3(B), dim3(64), lds_b, st, labels, predictions, weights,
                         weights_per_list, mask, gains, discount, tn, B, L, P, metric_out, stats_out);
      return;
    }
    hipLaunchKernelGGL((ndcg_count_wave_kernel<IPL, false>), dim3(B), dim3(64), lds, st, labels, predictions, weights,
                       weights_per_list, mask, gains, discount, tn, B, L, P, metric_out, stats_out);
    return;
  }
  hipLaunchKernelGGL((rank_metric_wave_kernel<KIND, IPL>), dim3(B), dim3(64), (size_t)64 * IPL * sizeof(float), st,
                     labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P,
                     metric_out, stats_out);
}

template <int KIND>
void dispatch_metric_wave(int L, const float* labels, const float* predictions, const float* weights,
                          int weights_per_list, const uint8_t* mask, const float* gains, const float* discount,
                          const TopN& tn, int B, int P, float* metric_out, float* stats_out, hipStream_t st) {
#define MW(I) launch_metric_wave<KIND, I>(labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P, metric_out, stats_out, st)
  if (L <= 64) MW(1); else if (L <= 128) MW(2); else if (L <= 256) MW(4);
  else if (KIND == 0) MW(8);                      // 256 < L <= 512 (NDCG only; longer lists: the workgroup kernel)
#undef MW
}

// Longest-first launch order for the O(n^2) loss kernels.  Their work per list goes with the
// square of its valid length, so with lists dispatched in index order the last wavefronts to
// start can be the longest ones and the kernel ends on a long low-occupancy tail (measured:
// -10 % ApproxNDCG kernel time).  Two small launches:
// (1) list_count_kernel: one wave per list counts its valid items (fully parallel);
// (2) list_scatter_kernel: ONE workgroup buckets the lists into kOrderClasses (64) length classes
//     (longest first) with LDS atomics: class histogram, prefix, then cursor fetch-adds.  Within a
//     class the order is whatever the LDS atomics gave -- the loss kernels write each list to its
//     own rows, so their results do not depend on it.  (A histogram by GLOBAL atomics cost 50 us
//     here: ~100 hot addresses; ballot counting on one CU 31 us -- measured twice, the second time
//     atomic-free with 16 loads in flight; issuing the 16 loads of a thread before its LDS atomics: 16.5 us
//     against 12 us for this plain loop.)
constexpr int kOrderClasses = 64;   // one per lane of the scanning wave; valid lengths U{L/2..L} touch half of them

__global__ __launch_bounds__(256) void list_count_kernel(const float* __restrict__ labels,
                                                         const uint8_t* __restrict__ mask, int B, int L,
                                                         int* __restrict__ nvalid) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * (blockDim.x >> 6) + wave;
  if (b >= B) return;
  const size_t base = (size_t)b * L;
  int n = 0;
  for (int c = 0; c < L; c += 64) {                         // wave-uniform trip count
    const int i = c + lane;
    const bool v = (i < L) && (mask ? (mask[base + i] != 0) : (labels[base + i] >= 0.0f));
    n += __popcll(__ballot(v));
  }
  if (lane == 0) nvalid[b] = n;
}

__global__ __launch_bounds__(1024) void list_scatter_kernel(int B, int L, const int* __restrict__ nvalid,
                                                            int* __restrict__ order_out) {
  __shared__ int s_hist[kOrderClasses];                     // class counts, then running cursors
  auto cls_of = [&](int n) { return kOrderClasses - 1 - (n * kOrderClasses) / (L + 1); };   // 0 = longest
  if (threadIdx.x < kOrderClasses) s_hist[threadIdx.x] = 0;
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += 1024) atomicAdd(&s_hist[cls_of(nvalid[i])], 1);
  __syncthreads();
  if (threadIdx.x < 64) {                                   // exclusive prefix over the classes: one wave scan
    const int v = s_hist[threadIdx.x];
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(inc, d); if ((int)threadIdx.x >= d) inc += t; }
    s_hist[threadIdx.x] = inc - v;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < B; i += 1024) order_out[atomicAdd(&s_hist[cls_of(nvalid[i])], 1)] = i;
}

// The same order from two FULLY PARALLEL launches (B >= kOrderParallelMin): the one-workgroup scatter above spends
// 12 us in two dependent sweeps of conflicting LDS atomics over all B lists.  Here every workgroup owns 256 lists:
// (1) list_class_kernel: four threads per list count its valid items, the class of every list goes to
//     `cls` (one byte per list) and the workgroup's class histogram (LDS atomics on 256 values only) to
//     `partial[blk][64]`;
// (2) list_place_kernel: a workgroup derives its base offset per class from the partial histograms (class prefix
//     over the column sums + the column sums of the workgroups before it: lane = class, coalesced 256-byte rows) and
//     places its 256 lists with LDS cursors.  Workspace: cls (B bytes) + partial (64 ints per workgroup) <= B ints.
constexpr int kOrderParallelMin = 512;
constexpr int kOrderLists = 256;    // lists per workgroup (LPB below; 128 from 8 192 lists on: twice the workgroups, round 6)

template <bool DEEP, int LPB>
__global__ __launch_bounds__(4 * LPB) void list_class_kernel(const float* __restrict__ labels,
                                                          const uint8_t* __restrict__ mask, int B, int L,
                                                          uint8_t* __restrict__ cls, int* __restrict__ partial) {
  __shared__ int s_hist[kOrderClasses];
  if (threadIdx.x < kOrderClasses) s_hist[threadIdx.x] = 0;
  __syncthreads();
  // FOUR THREADS per list (a DPP quad), interleaved over the row: independent loads (many in
  // flight per thread), no wave-wide ballot to wait for.  (A wave per list, 64 lists one after the other, measured
  // 90 us here -- every ballot waits for its own load; one thread per list 11 us.)
  const int t = threadIdx.x & 3;
  const int b = blockIdx.x * LPB + (threadIdx.x >> 2);
  int n = 0;
  if (b < B) {
    const size_t base = (size_t)b * L;
    if (!mask && (L & 3) == 0 && ((reinterpret_cast<uintptr_t>(labels) & 15) == 0)) {
      const float4* p = reinterpret_cast<const float4*>(labels + base);     // the quad reads 64 contiguous bytes a step
      // round 6, DEEP (fewer than 32 workgroups, i.e. most of the chip idle and the launch latency-bound): 16 loads of a
      // thread issued before the first add -- a list of up to 256 items = ONE memory round trip instead of four
      // (6.7 -> 4.9 us at B = 4096).  With 64 workgroups the deep form measured SLOWER (6.3 -> 7.2 us at B = 16 384: the
      // 64 CUs that run it are bandwidth-bound), so larger batches keep the four-deep loop.
      const int q = L / 4;
      int i = t;
      if (!DEEP) {
#pragma unroll 4
        for (; i < q; i += 4) {
          const float4 v = p[i];
          n += (v.x >= 0.0f) + (v.y >= 0.0f) + (v.z >= 0.0f) + (v.w >= 0.0f);
        }
      }
      for (; DEEP && i + 60 < q; i += 64) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[i + 4 * k];
#pragma unroll
        for (int k = 0; k < 16; ++k) n += (v[k].x >= 0.0f) + (v[k].y >= 0.0f) + (v[k].z >= 0.0f) + (v[k].w >= 0.0f);
      }
      if (DEEP) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (i + 4 * k < q) ? p[i + 4 * k] : make_float4(-1.f, -1.f, -1.f, -1.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) n += (v[k].x >= 0.0f) + (v[k].y >= 0.0f) + (v[k].z >= 0.0f) + (v[k].w >= 0.0f);
      }
    } else {
      const int per = (L + 3) / 4, lo = t * per, hi = (lo + per < L) ? lo + per : L;
      if (mask) { for (int i = lo; i < hi; ++i) n += mask[base + i] != 0; }
      else { for (int i = lo; i < hi; ++i) n += labels[base + i] >= 0.0f; }
    }
  }
  n += __shfl_xor(n, 1, 64);
  n += __shfl_xor(n, 2, 64);
  if (b < B && t == 0) {
    const int c = kOrderClasses - 1 - (n * kOrderClasses) / (L + 1);                   // 0 = longest
    cls[b] = (uint8_t)c;
    atomicAdd(&s_hist[c], 1);
  }
  __syncthreads();
  if (threadIdx.x < kOrderClasses) partial[blockIdx.x * kOrderClasses + threadIdx.x] = s_hist[threadIdx.x];
}

template <int LPB>
__global__ __launch_bounds__(256) void list_place_kernel(int B, int nblk, const uint8_t* __restrict__ cls,
                                                         const int* __restrict__ partial, int* __restrict__ order_out) {
  __shared__ int s_cur[kOrderClasses];
  __shared__ int s_part[4][2][kOrderClasses];               // per wave: (total, before-me) partial column sums
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int tot = 0, before = 0;
  for (int r0 = wave; r0 < nblk; r0 += 32) {                // lane = class: coalesced rows of the histogram matrix,
    int v[8];                                                // eight rows of a wave in flight (round 6)
#pragma unroll
    for (int k = 0; k < 8; ++k) { const int r = r0 + 4 * k; v[k] = r < nblk ? partial[r * kOrderClasses + lane] : 0; }
#pragma unroll
    for (int k = 0; k < 8; ++k) { tot += v[k]; before += (r0 + 4 * k < (int)blockIdx.x) ? v[k] : 0; }
  }
  s_part[wave][0][lane] = tot; s_part[wave][1][lane] = before;
  __syncthreads();
  if (wave == 0) {
    const int t = s_part[0][0][lane] + s_part[1][0][lane] + s_part[2][0][lane] + s_part[3][0][lane];
    const int bf = s_part[0][1][lane] + s_part[1][1][lane] + s_part[2][1][lane] + s_part[3][1][lane];
    int inc = t;                                             // exclusive prefix over the classes: one wave scan
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const int u = __shfl_up(inc, d); if (lane >= d) inc += u; }
    s_cur[lane] = inc - t + bf;
  }
  __syncthreads();
  const int b = blockIdx.x * LPB + threadIdx.x;
  if ((int)threadIdx.x < LPB && b < B) order_out[atomicAdd(&s_cur[cls[b]], 1)] = b;
}

// Round 6: the launch order from ONE launch with no global step (tfr_list_order_interleaved_i32).  Every workgroup sorts
// ITS lists by length class (LDS counting sort: histogram, class prefix, cursors) and writes them interleaved with the other
// workgroups' lists: the r-th longest list of segment s goes to position r * nseg + s -- the first nseg positions hold the
// longest list of every segment, the next nseg the second longest, ...  The lengths of different segments are draws from the
// same batch, so the interleaved order is longest-first up to the spread between segments' r-th order statistics, which is
// all the load balance of the O(n^2) kernels needs (their results do not depend on the order), and the second launch + the
// partial-histogram round trip of the exact order go away (headline step 0.119 -> see DESIGN 4.2).  The last segment may
// be short (m lists): ranks below m interleave over all nseg segments, the rest over the nseg - 1 full ones.
template <bool DEEP, int LPB>
__global__ __launch_bounds__(4 * LPB) void list_order_local_kernel(const float* __restrict__ labels,
                                                                   const uint8_t* __restrict__ mask, int B, int L,
                                                                   int* __restrict__ order_out, int C) {
  __shared__ int s_hist[kOrderClasses];
  __shared__ int s_cur[kOrderClasses];
  if (threadIdx.x < kOrderClasses) s_hist[threadIdx.x] = 0;
  __syncthreads();
  const int t = threadIdx.x & 3;
  const int b = blockIdx.x * LPB + (threadIdx.x >> 2);
  int n = 0;
  if (b < B) {
    const size_t base = (size_t)b * L;
    if (!mask && (L & 3) == 0 && ((reinterpret_cast<uintptr_t>(labels) & 15) == 0)) {
      const float4* p = reinterpret_cast<const float4*>(labels + base);
      const int q = L / 4;
      int i = t;
      if (!DEEP) {
#pragma unroll 4
        for (; i < q; i += 4) {
          const float4 v = p[i];
          n += (v.x >= 0.0f) + (v.y >= 0.0f) + (v.z >= 0.0f) + (v.w >= 0.0f);
        }
      }
      for (; DEEP && i + 60 < q; i += 64) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = p[i + 4 * k];
#pragma unroll
        for (int k = 0; k < 16; ++k) n += (v[k].x >= 0.0f) + (v[k].y >= 0.0f) + (v[k].z >= 0.0f) + (v[k].w >= 0.0f);
      }
      if (DEEP) {
        float4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) v[k] = (i + 4 * k < q) ? p[i + 4 * k] : make_float4(-1.f, -1.f, -1.f, -1.f);
#pragma unroll
        for (int k = 0; k < 16; ++k) n += (v[k].x >= 0.0f) + (v[k].y >= 0.0f) + (v[k].z >= 0.0f) + (v[k].w >= 0.0f);
      }
    } else {
      const int per = (L + 3) / 4, lo = t * per, hi = (lo + per < L) ? lo + per : L;
      if (mask) { for (int i = lo; i < hi; ++i) n += mask[base + i] != 0; }
      else { for (int i = lo; i < hi; ++i) n += labels[base + i] >= 0.0f; }
    }
  }
  n += __shfl_xor(n, 1, 64);
  n += __shfl_xor(n, 2, 64);
  const bool own = b < B && t == 0;
  int c = 0;
  if (own) {
    c = kOrderClasses - 1 - (n * kOrderClasses) / (L + 1);                             // 0 = longest
    atomicAdd(&s_hist[c], 1);
  }
  __syncthreads();
  if (threadIdx.x < 64) {                                     // class starts inside the segment: one wave scan
    const int v = s_hist[threadIdx.x];
    s_cur[threadIdx.x] = wave_scan_incl_i(v) - v;
  }
  __syncthreads();
  if (own) {
    const int r = atomicAdd(&s_cur[c], 1);                    // rank of the list inside its segment (arbitrary inside a class)
    const int nseg = gridDim.x, seg = blockIdx.x;
    const int m = B - (nseg - 1) * LPB;                       // lists of the last segment (1 .. LPB)
    // interleaved in chunks of C consecutive ranks (C divides LPB; default 1, TFR_ORDER_CHUNK): position = how many lists
    // come before (row major, then segment, then rank inside the chunk); only the last segment can be short.
    const int base = (r / C) * C;
    const int pos = (nseg - 1) * base + (m < base ? m : base) + (seg < nseg - 1 ? seg : nseg - 1) * C + (r - base);
    order_out[pos] = b;
  }
}

inline int block_threads_for(int P) {
  int t = P / 2;
  if (t < 64) t = 64;
  if (t > 1024) t = 1024;
  return t;
}

}  // namespace

extern "C" int tfr_hip_abi_version(void) { return TFR_HIP_ABI_VERSION; }

extern "C" int tfr_sort_ranks_f32(const float* scores, const float* labels, const uint8_t* mask,
                                  const int32_t* tiebreak, int B, int L, int32_t* ranks_out,
                                  int32_t* order_out, void* stream) {
  if (!scores || B < 0 || L <= 0 || (!ranks_out && !order_out)) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  const int P = pow2_ceil(L < 2 ? 2 : L);
  static const int env_wave = [] { const char* e = getenv("TFR_SORT_WAVE"); return (e && *e) ? atoi(e) : 1; }();
  if (env_wave && L <= 1024 && (L <= 256 || B >= 1024)) {
    hipStream_t st = (hipStream_t)stream;
#define SW(I) hipLaunchKernelGGL(sort_ranks_wave_kernel<I>, dim3(B), dim3(64), 0, st, scores, labels, mask, tiebreak, L, ranks_out, order_out)
    if (L <= 64) SW(1); else if (L <= 128) SW(2); else if (L <= 256) SW(4); else if (L <= 512) SW(8); else SW(16);
#undef SW
    return (int)hipGetLastError();
  }
  const int T = block_threads_for(P);
  hipLaunchKernelGGL(sort_ranks_kernel, dim3(B), dim3(T), (size_t)P * sizeof(uint64_t),
                     (hipStream_t)stream, scores, labels, mask, tiebreak, L, P, ranks_out, order_out);
  return (int)hipGetLastError();
}

static int launch_metric(int kind, const float* labels, const float* predictions, const float* weights,
                         int weights_per_list, const uint8_t* mask, const float* gains,
                         const float* discount, const int32_t* topn_host, int K, int B, int L,
                         float* metric_out, float* stats_out, uint32_t tie_seed, void* stream) {
  if (!labels || !predictions || !metric_out || !stats_out || B < 0 || L <= 0) return TFR_EINVAL;
  if (K < 1 || K > TFR_MAX_TOPN || !topn_host) return TFR_EINVAL;
  if (kind == 0 && !discount) return TFR_EINVAL;
  if (L > TFR_MAX_LIST_SIZE) return TFR_ETOOLARGE;            // NDCG / MRR: 16 B of LDS per item in the workgroup kernel
  if (B == 0) return TFR_OK;
  TopN tn; tn.n = K; tn.tie_seed = tie_seed;
  for (int q = 0; q < TFR_MAX_TOPN; ++q) tn.k[q] = (q < K) ? topn_host[q] : 0;
  const int P = pow2_ceil(L < 2 ? 2 : L);
  static const int env_wave = [] { const char* e = getenv("TFR_SORT_WAVE"); return (e && *e) ? atoi(e) : 1; }();
  // (the MRR wave kernel is only instantiated usefully up to IPL = 4: hipcc spills the IPL >= 8 forms)
  if (env_wave && L <= (kind == 0 ? 512 : 256) && (L <= 256 || B >= 1024)) {
    if (kind == 0) dispatch_metric_wave<0>(L, labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, P, metric_out, stats_out, (hipStream_t)stream);
    else dispatch_metric_wave<1>(L, labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, P, metric_out, stats_out, (hipStream_t)stream);
    return (int)hipGetLastError();
  }
  const int T = block_threads_for(P);
  const size_t lds = 128 + (size_t)P * (sizeof(uint64_t) + 2 * sizeof(float));
  if (lds > 160 * 1024) return TFR_ETOOLARGE;
  if (lds > 64 * 1024) {
    hipError_t e = (kind == 0)
        ? hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_metric_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)
        : hipFuncSetAttribute(reinterpret_cast<const void*>(&rank_metric_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  if (kind == 0)
    hipLaunchKernelGGL(rank_metric_kernel<0>, dim3(B), dim3(T), lds, (hipStream_t)stream, labels,
                       predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P,
                       metric_out, stats_out);
  else
    hipLaunchKernelGGL(rank_metric_kernel<1>, dim3(B), dim3(T), lds, (hipStream_t)stream, labels,
                       predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P,
                       metric_out, stats_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_ndcg_metric_f32(const float* labels, const float* predictions, const float* weights,
                                   int weights_per_list, const uint8_t* mask, const float* gains,
                                   const float* discount, const int32_t* topn_host, int K, int B,
                                   int L, float* ndcg_out, float* stats_out, uint32_t tie_seed, void* stream) {
  return launch_metric(0, labels, predictions, weights, weights_per_list, mask, gains, discount,
                       topn_host, K, B, L, ndcg_out, stats_out, tie_seed, stream);
}

extern "C" int tfr_mrr_metric_f32(const float* labels, const float* predictions, const float* weights,
                                  int weights_per_list, const uint8_t* mask, const int32_t* topn_host,
                                  int K, int B, int L, float* mrr_out, float* stats_out, uint32_t tie_seed, void* stream) {
  return launch_metric(1, labels, predictions, weights, weights_per_list, mask, nullptr, nullptr,
                       topn_host, K, B, L, mrr_out, stats_out, tie_seed, stream);
}

extern "C" int tfr_rank_metric_f32(int kind, const float* labels, const float* predictions, const float* weights,
                                   int weights_per_list, const uint8_t* mask, const float* gains,
                                   const float* discount, const int32_t* topn_host, int K, int B, int L,
                                   float* metric_out, float* stats_out, uint32_t tie_seed, void* workspace,
                                   long workspace_bytes, void* stream) {
  if (kind == TFR_METRIC_NDCG || kind == TFR_METRIC_MRR)
    return launch_metric(kind, labels, predictions, weights, weights_per_list, mask, gains, discount, topn_host, K,
                         B, L, metric_out, stats_out, tie_seed, stream);
  if (kind < TFR_METRIC_DCG || kind > TFR_METRIC_OPA) return TFR_EINVAL;
  if (kind == TFR_METRIC_OPA && K != 1) return TFR_EINVAL;
  if (!labels || !predictions || !metric_out || !stats_out || B < 0 || L <= 0) return TFR_EINVAL;
  if (K < 1 || K > TFR_MAX_TOPN || !topn_host) return TFR_EINVAL;
  if (kind == TFR_METRIC_DCG && !discount) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const size_t slot = (size_t)tfr_list_workspace_bytes(TFR_WS_RANK_METRIC, L);   // > 0: 32 B per item outgrow LDS
  if (slot && (!workspace || workspace_bytes < (long)slot)) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  TopN tn; tn.n = K; tn.tie_seed = tie_seed;
  for (int q = 0; q < TFR_MAX_TOPN; ++q) tn.k[q] = (q < K) ? topn_host[q] : 0;
  const int P = pow2_ceil(L < 2 ? 2 : L);
  hipStream_t st = (hipStream_t)stream;
  if (L > 512) {                                  // long lists: one workgroup per list, everything in LDS (or the workspace)
    const size_t lds = 128 + (size_t)P * (slot ? 8 : 32);
    auto fn = slot ? rank_metric2_block_kernel<true> : rank_metric2_block_kernel<false>;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3(slot ? big_slots(B, (size_t)workspace_bytes, slot) : B), dim3(block_threads_for(P)), lds, st, kind,
                       labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P, metric_out, stats_out,
                       (float*)workspace);
    return (int)hipGetLastError();
  }
#define M2(I) hipLaunchKernelGGL(rank_metric2_wave_kernel<I>, dim3(B), dim3(64), (size_t)2 * 64 * I * sizeof(float), st, kind, labels, predictions, weights, weights_per_list, mask, gains, discount, tn, B, L, P, metric_out, stats_out)
  if (L <= 64) M2(1); else if (L <= 128) M2(2); else if (L <= 256) M2(4); else M2(8);
#undef M2
  return (int)hipGetLastError();
}

extern "C" int tfr_div_metric_f32(int kind, const float* labels, const float* predictions, const float* weights,
                                  int weights_per_list, const uint8_t* mask, const float* discount, float alpha,
                                  const int32_t* topn_host, int K, int B, int L, int S, float* metric_out,
                                  float* stats_out, uint32_t tie_seed, void* workspace, long workspace_bytes, void* stream) {
  if (kind != TFR_DIV_ALPHA_DCG && kind != TFR_DIV_PRECISION_IA) return TFR_EINVAL;
  if (!labels || !predictions || !metric_out || !stats_out || B < 0 || L <= 0 || S <= 0) return TFR_EINVAL;
  if (K < 1 || K > TFR_MAX_TOPN || !topn_host) return TFR_EINVAL;
  if (kind == TFR_DIV_ALPHA_DCG && !discount) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const size_t slot = (size_t)tfr_list_workspace_bytes(TFR_WS_DIV_METRIC, L);    // > 0: 28 B per item outgrow LDS
  if (slot && (!workspace || workspace_bytes < (long)slot)) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  TopN tn; tn.n = K; tn.tie_seed = tie_seed;
  for (int q = 0; q < TFR_MAX_TOPN; ++q) tn.k[q] = (q < K) ? topn_host[q] : 0;
  const int P = pow2_ceil(L < 2 ? 2 : L);
  hipStream_t st = (hipStream_t)stream;
  if (L > 512) {                                  // long lists: one workgroup per list
    const size_t lds = 128 + (size_t)P * (slot ? 8 : 28);
    auto fn = slot ? div_metric_block_kernel<true> : div_metric_block_kernel<false>;
    if (lds > 64 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(fn, dim3(slot ? big_slots(B, (size_t)workspace_bytes, slot) : B), dim3(block_threads_for(P)), lds, st, kind,
                       labels, predictions, weights, weights_per_list, mask, discount, alpha, tn, B, L, S, P, metric_out, stats_out,
                       (float*)workspace);
    return (int)hipGetLastError();
  }
#define DM(I) hipLaunchKernelGGL(div_metric_wave_kernel<I>, dim3(B), dim3(64), 0, st, kind, labels, predictions, weights, weights_per_list, mask, discount, alpha, tn, B, L, S, P, metric_out, stats_out)
  if (L <= 64) DM(1); else if (L <= 128) DM(2); else if (L <= 256) DM(4); else DM(8);
#undef DM
  return (int)hipGetLastError();
}

extern "C" int tfr_metric_list_weights_f32(const float* stats, int B, float* weights_out, void* stream) {
  if (!stats || !weights_out || B < 0) return TFR_EINVAL;
  if (B == 0) return TFR_OK;
  const int nwg = (B + 1023) / 1024 < 16 ? (B + 1023) / 1024 : 16;
  hipLaunchKernelGGL(metric_list_weights_kernel, dim3(nwg), dim3(1024), 0, (hipStream_t)stream, stats, B, weights_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_list_order_i32(const float* labels, const uint8_t* mask, int B, int L, int32_t* order_out,
                                  int32_t* workspace, void* stream) {
  if ((!labels && !mask) || !order_out || !workspace || B < 0 || L <= 0) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  hipStream_t st = (hipStream_t)stream;
  static const int env_par = [] { const char* e = getenv("TFR_ORDER_PARALLEL"); return (e && *e) ? atoi(e) : 1; }();
  if (env_par && B >= kOrderParallelMin) {
    // 128 lists per workgroup from 8 192 lists on: with 256 a batch of 16 384 lists ran on 64 of the 256 CUs
    static const int env_lpb = [] { const char* e = getenv("TFR_ORDER_LPB"); return (e && *e) ? atoi(e) : 0; }();
    const bool half = env_lpb ? env_lpb == 128 : B >= 8192;
    const int lpb = half ? 128 : kOrderLists;
    const int nblk = (B + lpb - 1) / lpb;
    uint8_t* cls = reinterpret_cast<uint8_t*>(workspace);                    // B bytes
    int* partial = reinterpret_cast<int*>(workspace) + (B + 3) / 4;         // nblk * 64 ints (<= B / 2 + 64 <= B - B / 4)
    static const int env_deep = [] { const char* e = getenv("TFR_ORDER_DEEP"); return (e && *e) ? atoi(e) : -1; }();
    if (half) {
      if (env_deep == 1) hipLaunchKernelGGL((list_class_kernel<true, 128>), dim3(nblk), dim3(512), 0, st, labels, mask, B, L, cls, partial);
      else hipLaunchKernelGGL((list_class_kernel<false, 128>), dim3(nblk), dim3(512), 0, st, labels, mask, B, L, cls, partial);
      hipLaunchKernelGGL(list_place_kernel<128>, dim3(nblk), dim3(256), 0, st, B, nblk, (const uint8_t*)cls,
                         (const int*)partial, (int*)order_out);
    } else {
      if (nblk < 32) hipLaunchKernelGGL((list_class_kernel<true, 256>), dim3(nblk), dim3(1024), 0, st, labels, mask, B, L, cls, partial);
      else hipLaunchKernelGGL((list_class_kernel<false, 256>), dim3(nblk), dim3(1024), 0, st, labels, mask, B, L, cls, partial);
      hipLaunchKernelGGL(list_place_kernel<256>, dim3(nblk), dim3(256), 0, st, B, nblk, (const uint8_t*)cls,
                         (const int*)partial, (int*)order_out);
    }
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL(list_count_kernel, dim3((B + 3) / 4), dim3(256), 0, st, labels, mask, B, L, (int*)workspace);
  hipLaunchKernelGGL(list_scatter_kernel, dim3(1), dim3(1024), 0, st, B, L, (const int*)workspace, (int*)order_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_list_order_interleaved_i32(const float* labels, const uint8_t* mask, int B, int L, int32_t* order_out,
                                              void* stream) {
  if ((!labels && !mask) || !order_out || B < 0 || L <= 0) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  hipStream_t st = (hipStream_t)stream;
  static const int env_c = [] { const char* e = getenv("TFR_ORDER_CHUNK"); return (e && *e) ? atoi(e) : 1; }();
  // (measured, headline step: 0.1159 / 0.1165 / 0.1175 / 0.1198 ms with chunks of 1 / 4 / 8 / 16 ranks; exact order 0.1193)
  const int C = (env_c == 1 || env_c == 2 || env_c == 4 || env_c == 8 || env_c == 16 || env_c == 32) ? env_c : 1;
  static const int env_il_lpb = [] { const char* e = getenv("TFR_ORDER_IL_LPB"); return (e && *e) ? atoi(e) : 0; }();
  if (env_il_lpb ? env_il_lpb == 128 : B >= 8192) {
    hipLaunchKernelGGL((list_order_local_kernel<false, 128>), dim3((B + 127) / 128), dim3(512), 0, st, labels, mask, B, L, (int*)order_out, C);
  } else if ((B + 255) / 256 < 32) {
    hipLaunchKernelGGL((list_order_local_kernel<true, 256>), dim3((B + 255) / 256), dim3(1024), 0, st, labels, mask, B, L, (int*)order_out, C);
  } else {
    hipLaunchKernelGGL((list_order_local_kernel<false, 256>), dim3((B + 255) / 256), dim3(1024), 0, st, labels, mask, B, L, (int*)order_out, C);
  }
  return (int)hipGetLastError();
}
