// ListMLE loss, forward + backward fused, wave-per-list (gfx950).
//
// Reference behaviour restated (losses_impl.py:1541-1576 ListMLELoss, :457-480
// ListMLELambdaWeight): masked labels := 0, masked logits := log(1e-10); items sorted by label
// (descending; the reference shuffles ties at random, op seed 37 under the graph seed -- `tie_seed` != 0 orders equal
// labels by a counter-based hash of (tie_seed, list, item), 0 keeps index order; the TF random stream itself is not
// reproducible: "parity unpinned" like every tie rule); with s the sorted logits,
//     loss = sum_p w_p * ( log sum_{q >= p} exp(s_q) - s_p ),   w_p = rank_discount(p + 1) or 1.
// Backward (autodiff in the reference):  d loss / d s_p = exp(s_p) * sum_{q <= p} w_q / S_q - w_p.
//
// One wavefront per list: in-register bitonic sort of packed keys, a reverse scan for S, a
// forward scan for the gradient; every item of the list takes part (masked ones as constants).
#include "common.h"
#include "../../include/tfr_hip.h"

using namespace tfr;

namespace {

constexpr float kLogEps = -23.025850929940457f;        // log(1e-10)

template <int IPL>
__global__ __launch_bounds__(64) void list_mle_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ pos_weight, const float* __restrict__ list_scale, int L, float temperature,
    float* __restrict__ loss_out, float* __restrict__ dlogits_out, const GridSum sum, const uint32_t tie_seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* XS = reinterpret_cast<float*>(smem_raw);          // [64 * IPL] logits by original index
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;

  uint64_t key[IPL];
  bool valid[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    key[r] = 0; valid[r] = false;
    if (i < L) {
      const float lab = labels[base + i];
      const bool v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      valid[r] = v;
      XS[i] = v ? logits[base + i] / temperature : kLogEps;
      key[r] = make_sort_key(v, v ? lab : 0.0f, tie_key15(tie_seed, (uint32_t)b, (uint32_t)i), i);   // valid first, label desc, then the tie key, then index
    }
  }
  __syncthreads();
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);

  // sorted logits, max, exp
  float xs[IPL], e[IPL], w[IPL];
  int idx[IPL];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    idx[r] = sort_key_index(key[r]);
    xs[r] = (p < L) ? XS[idx[r]] : -INFINITY;
    w[r] = (p < L) ? (pos_weight ? pos_weight[p] : 1.0f) : 0.0f;
    mx = fmaxf(mx, xs[r]);
  }
  mx = wave_max_u(mx);
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    xs[r] -= mx;
    e[r] = (lane + 64 * r < L) ? expf(xs[r]) : 0.0f;
  }
  // S_p = sum_{q >= p} e_q : reverse inclusive scan in position order (p = lane + 64 r)
  float S[IPL];
  float carry = 0.f;
#pragma unroll
  for (int r = IPL - 1; r >= 0; --r) {
    float v = e[r];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_down(v, o, 64);
      if (lane + o < 64) v += u;
    }
    S[r] = v + carry;
    carry += __shfl(v, 0, 64);
  }
  float term = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r)
    if (lane + 64 * r < L) term += w[r] * (logf(S[r]) - xs[r]);
  const float loss = wave_sum_u(term);
  if (lane == 0) loss_out[b] = loss;
  if (sum.out) grid_weighted_sum_last(loss_out, b, loss, list_scale, sum.n, sum.out, sum.st, lane);   // sum_b loss_b * list_scale_b
  if (!dlogits_out) return;

  // C_p = sum_{q <= p} w_q / S_q : forward inclusive scan;  grad_p = e_p * C_p - w_p
  const float gscale = (list_scale ? list_scale[b] : 1.0f) / temperature;
  carry = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    float v = (p < L) ? w[r] / S[r] : 0.0f;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    const float C = v + carry;
    carry += __shfl(v, 63, 64);
    if (p < L) {
      const bool vld = (key[r] >> 63) != 0;
      dlogits_out[base + idx[r]] = vld ? (e[r] * C - w[r]) * gscale : 0.0f;
    }
  }
}


// ------------------------------------------------------------------------------------------
// UniqueSoftmaxLoss (losses_impl.py:1250-1281): every document i competes, in its own softmax,
// only against the documents with a strictly LOWER label:
//     loss = sum_i (2^{l_i} - 1) * ( log( e^{s_i} + sum_{j: l_j < l_i} e^{s_j} ) - s_i ),   valid i, j.
// The reference builds the [B, L, L+1] denominator tensor; sorted by label (descending) the
// inner sum is a SUFFIX sum starting at the end of i's tie group, and the backward
//     d loss / d s_k = -g_k + e^{s_k} ( g_k / D_k + sum_{i: l_i > l_k} g_i / D_i )
// a PREFIX sum ending at the start of k's tie group: O(L) after one register sort.
// Tie groups of a label-descending sorted sequence (element p = lane + 64 r, nv of them):
// gs[p] = first position of p's label group, ge[p] = one past its last.
template <int IPL>
__device__ __forceinline__ void label_tie_groups(const float (&lb)[IPL], int nv, int lane, int (&gs)[IPL], int (&ge)[IPL]) {
    // first-of-group flags need the previous position's label
    float prev_carry = INFINITY;                          // label "before" position 0
    int run_start = 0;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int p = lane + 64 * r;
      float prev = __shfl_up(lb[r], 1, 64);
      if (lane == 0) prev = prev_carry;
      prev_carry = __shfl(lb[r], 63, 64);
      const bool first = (p < nv) && (lb[r] < prev || p == 0);
      // inclusive max-scan of (first ? p : -1)  -> group start
      int v = first ? p : -1;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_up(v, o, 64);
        if (lane >= o) v = v > u ? v : u;
      }
      v = v > run_start ? v : run_start;
      gs[r] = v;
      run_start = __shfl(v, 63, 64);
    }
    // group end = next first-of-group position after p (or nv): reverse min-scan over positions > p
    int next_carry = nv;
    float next_lab_carry = -INFINITY;                     // label "after" the last position
#pragma unroll
    for (int r = IPL - 1; r >= 0; --r) {
      const int p = lane + 64 * r;
      float nxt = __shfl_down(lb[r], 1, 64);
      if (lane == 63) nxt = next_lab_carry;
      next_lab_carry = __shfl(lb[r], 0, 64);
      // position p + 1 starts a new group iff label[p + 1] < label[p]  (or p + 1 >= nv)
      const bool last = (p < nv) && (p + 1 >= nv || nxt < lb[r]);
      int v = last ? p + 1 : 0x7fffffff;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int u = __shfl_down(v, o, 64);
        if (lane + o < 64) v = v < u ? v : u;
      }
      v = v < next_carry ? v : next_carry;
      ge[r] = v;
      next_carry = __shfl(v, 0, 64);
    }
}

template <int IPL>
__global__ __launch_bounds__(64) void unique_softmax_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ list_scale, int L, float temperature, float* __restrict__ loss_out,
    float* __restrict__ dlogits_out, const GridSum sum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* XS = reinterpret_cast<float*>(smem_raw);          // [64 * IPL] logits by original index
  float* LB = XS + 64 * IPL;                               // [64 * IPL] labels by original index
  float* SUF = LB + 64 * IPL;                              // [64 * IPL + 1] suffix sums by sorted position
  float* PRE = SUF + 64 * IPL + 1;                         // [64 * IPL + 1] exclusive prefix of g / D
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;
  constexpr int NP = 64 * IPL;

  uint64_t key[IPL];
  int nv = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    key[r] = 0;
    bool v = false;
    if (i < L) {
      const float lab = labels[base + i];
      v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      XS[i] = v ? logits[base + i] / temperature : 0.0f;
      LB[i] = v ? lab : 0.0f;
      key[r] = make_sort_key(v, v ? lab : 0.0f, 0, i);
      if (dlogits_out && !v) dlogits_out[base + i] = 0.0f;
    }
    nv += __popcll(__ballot(v));
  }
  __syncthreads();
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);       // valid first, label descending, then index

  float xs[IPL], lb[IPL], e[IPL], g[IPL];
  int idx[IPL];
  float mx = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    idx[r] = sort_key_index(key[r]);
    const bool in = p < nv;
    xs[r] = in ? XS[idx[r]] : -INFINITY;
    lb[r] = in ? LB[idx[r]] : -INFINITY;
    g[r] = in ? gain_pow2m1(lb[r]) : 0.0f;
    mx = fmaxf(mx, xs[r]);
  }
  mx = wave_max_u(mx);
#pragma unroll
  for (int r = 0; r < IPL; ++r) e[r] = (lane + 64 * r < nv) ? expf(xs[r] - mx) : 0.0f;
  // suffix sums of e in sorted order -> SUF[p] = sum_{q >= p} e_q, SUF[nv..] = 0
  float carry = 0.f;
#pragma unroll
  for (int r = IPL - 1; r >= 0; --r) {
    float v = e[r];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_down(v, o, 64);
      if (lane + o < 64) v += u;
    }
    SUF[lane + 64 * r] = v + carry;
    carry += __shfl(v, 0, 64);
  }
  if (lane == 0) SUF[NP] = 0.0f;
  int gs[IPL], ge[IPL];
  label_tie_groups<IPL>(lb, nv, lane, gs, ge);
  __syncthreads();                                        // SUF visible
  float D[IPL], term = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    D[r] = 1.0f;
    if (p < nv) {
      D[r] = e[r] + SUF[ge[r] < NP ? ge[r] : NP];
      term += g[r] * (logf(D[r]) - (xs[r] - mx));
    }
  }
  const float loss = wave_sum_u(term);
  if (lane == 0) loss_out[b] = loss;
  if (sum.out) grid_weighted_sum_last(loss_out, b, loss, list_scale, sum.n, sum.out, sum.st, lane);
  if (!dlogits_out) return;
  // exclusive prefix of c = g / D in sorted order -> PRE[p] = sum_{q < p} c_q
  carry = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    const float c = (p < nv) ? g[r] / D[r] : 0.0f;
    float v = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    float ex = __shfl_up(v, 1, 64);                       // exclusive = inclusive of the previous lane
    if (lane == 0) ex = 0.0f;                             // (never `v - c`: cancels when c dominates)
    PRE[p] = ex + carry;
    carry += __shfl(v, 63, 64);
  }
  __syncthreads();
  const float gscale = (list_scale ? list_scale[b] : 1.0f) / temperature;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    if (p < nv) {
      const float higher = PRE[gs[r]];                    // sum over strictly higher labels
      dlogits_out[base + idx[r]] = (-g[r] + e[r] * (g[r] / D[r] + higher)) * gscale;
    }
  }
}


// CircleLoss (losses_impl.py:1036-1116).  Scores are clipped to [0, 1] (get_logits, :1082-1085);
// with alpha_i = relu(1 + margin - s_i), alpha'_j = relu(s_j + margin) held constant (stop_gradient),
//     W = sum_{y_i > y_j} exp(gamma * (a_i + b_j)),  a_i = alpha_i (1 - margin - s_i),  b_j = alpha'_j (s_j - margin)
//     loss = log1p(W);   per-list weight = (#pairs) / (#pairs) -- NaN for a list without a pair, as in
//     the reference (:1109-1111, plain `/`).
// The pair exponent is separable, so the [L, L] matrix of the reference collapses to a sort by label and
// two scans:  W = sum_p e^{ga_p} * sum_{q beyond p's tie group} e^{gb_q}.  Everything is kept in the log
// domain (gamma = 64 puts e^{ga + gb} beyond fp32 range for scores near 1; the reference overflows to inf
// there, this kernel does not).
template <int IPL>
__global__ __launch_bounds__(64) void circle_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ list_scale, int L, float gamma, float margin, int clip,
    float* __restrict__ loss_out, float* __restrict__ weight_out, float* __restrict__ dlogits_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* XS = reinterpret_cast<float*>(smem_raw);          // [64 * IPL] raw logits by original index
  float* LB = XS + 64 * IPL;                               // [64 * IPL] labels by original index
  float* SUF = LB + 64 * IPL;                              // [64 * IPL + 1] suffix sums of F by sorted position
  float* PRE = SUF + 64 * IPL + 1;                         // [64 * IPL + 1] exclusive prefix sums of E
  const int lane = threadIdx.x, b = blockIdx.x;
  const size_t base = (size_t)b * L;
  constexpr int NP = 64 * IPL;

  uint64_t key[IPL];
  int nv = 0;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int i = lane + 64 * r;
    key[r] = 0;
    bool v = false;
    if (i < L) {
      const float lab = labels[base + i];
      v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      XS[i] = logits[base + i];
      LB[i] = lab;
      key[r] = make_sort_key(v, v ? lab : 0.0f, 0, i);
      if (dlogits_out && !v) dlogits_out[base + i] = 0.0f;
    }
    nv += __popcll(__ballot(v));
  }
  __syncthreads();
  wave_bitonic_sort_desc<uint64_t, IPL>(key, lane);       // valid first, label descending, then index

  float lb[IPL], ga[IPL], gb[IPL], al_i[IPL], al_j[IPL];
  int idx[IPL];
  bool inside[IPL];
  float MA = -INFINITY, MB = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    idx[r] = sort_key_index(key[r]);
    const bool in = p < nv;
    const float raw = in ? XS[idx[r]] : 0.0f;
    const float sc = clip ? fminf(fmaxf(raw, 0.0f), 1.0f) : raw;
    inside[r] = in && (!clip || (raw >= 0.0f && raw <= 1.0f));   // clip_by_value passes the gradient on [0, 1]
    lb[r] = in ? LB[idx[r]] : -INFINITY;
    al_i[r] = fmaxf(1.0f - sc + margin, 0.0f);
    al_j[r] = fmaxf(sc + margin, 0.0f);
    ga[r] = in ? gamma * (al_i[r] * (1.0f - sc - margin)) : -INFINITY;
    gb[r] = in ? gamma * (al_j[r] * (sc - margin)) : -INFINITY;
    MA = fmaxf(MA, ga[r]); MB = fmaxf(MB, gb[r]);
  }
  MA = wave_max_u(MA); MB = wave_max_u(MB);
  float E[IPL], F[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const bool in = lane + 64 * r < nv;
    E[r] = in ? expf(ga[r] - MA) : 0.0f;
    F[r] = in ? expf(gb[r] - MB) : 0.0f;
  }
  // SUF[p] = sum_{q >= p} F_q ; PRE[p] = sum_{q < p} E_q
  float carry = 0.f;
#pragma unroll
  for (int r = IPL - 1; r >= 0; --r) {
    float v = F[r];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_down(v, o, 64);
      if (lane + o < 64) v += u;
    }
    SUF[lane + 64 * r] = v + carry;
    carry += __shfl(v, 0, 64);
  }
  if (lane == 0) SUF[NP] = 0.0f;
  carry = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    float v = E[r];
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float u = __shfl_up(v, o, 64);
      if (lane >= o) v += u;
    }
    float ex = __shfl_up(v, 1, 64);                       // exclusive = inclusive of the previous lane
    if (lane == 0) ex = 0.0f;                             // (never `v - E`: cancels when E_p = 1 dominates)
    PRE[lane + 64 * r] = ex + carry;
    carry += __shfl(v, 63, 64);
  }
  int gs[IPL], ge[IPL];
  label_tie_groups<IPL>(lb, nv, lane, gs, ge);
  __syncthreads();

  // t_p = log( e^{ga_p} * sum_{y_q < y_p} e^{gb_q} );  lw = logsumexp_p t_p = log W
  float t[IPL], tm = -INFINITY;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    t[r] = -INFINITY;
    if (p < nv) {
      const float lower = SUF[ge[r] < NP ? ge[r] : NP];
      if (lower > 0.0f) t[r] = ga[r] + (MB + logf(lower));
    }
    tm = fmaxf(tm, t[r]);
  }
  tm = wave_max_u(tm);
  const bool any_pair = tm > -INFINITY;
  float acc = 0.f;
#pragma unroll
  for (int r = 0; r < IPL; ++r) acc += (t[r] > -INFINITY) ? expf(t[r] - tm) : 0.0f;
  acc = wave_sum_u(acc);
  const float lw = any_pair ? tm + logf(acc) : -INFINITY;
  const float loss = any_pair ? (fmaxf(lw, 0.0f) + log1pf(expf(-fabsf(lw)))) : 0.0f;      // log1p(W)
  if (lane == 0) {
    loss_out[b] = loss;
    if (weight_out) weight_out[b] = any_pair ? 1.0f : NAN;
  }
  if (!dlogits_out) return;
  const float sig = any_pair ? 1.0f / (1.0f + expf(-lw)) : 0.0f;                          // W / (1 + W)
  const float gscale = (list_scale ? list_scale[b] : 1.0f) * gamma * sig;
#pragma unroll
  for (int r = 0; r < IPL; ++r) {
    const int p = lane + 64 * r;
    if (p < nv) {
      float gpart = 0.f;
      if (any_pair && inside[r]) {
        const float hi = (t[r] > -INFINITY) ? expf(t[r] - lw) : 0.0f;                    // item as the preferred one
        const float higher = PRE[gs[r]];
        const float lo = (higher > 0.0f) ? expf(gb[r] + (MA + logf(higher)) - lw) : 0.0f; // item as the other one
        gpart = -al_i[r] * hi + al_j[r] * lo;
      }
      dlogits_out[base + idx[r]] = gscale * gpart;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Workgroup forms for 1024 < list_size <= 8192 (the register sort of the wave kernels does not fit beyond 16 keys per
// lane): keys and per-position values in LDS, block_bitonic_sort_desc, Hillis-Steele scans.  Same arithmetic as the
// wave kernels; the scans associate differently, so results agree to rounding, not bit for bit.
// BIG (round 5, list_size > 4096): the per-position arrays (24 / 36 B per item) outgrow the 160 KB of LDS next to the 8 B
// sort keys; they move to a slot of the caller's workspace (global memory, L2-resident: <= 288 KB per slot) -- the sort keys
// and the reduction scratch stay in LDS, the code is the same (__syncthreads orders a workgroup's global accesses as it
// orders its LDS accesses).  A BIG launch has one workgroup per workspace slot and walks the lists with a grid stride.
template <typename T, typename Op>
__device__ __forceinline__ void block_scan_inclusive(T* a, T* scratch, int P, Op op) {    // result ends in `a`
  T* src = a;
  T* dst = scratch;
  for (int o = 1; o < P; o <<= 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += blockDim.x) dst[i] = (i >= o) ? op(src[i - o], src[i]) : src[i];
    T* t = src; src = dst; dst = t;
  }
  __syncthreads();
  if (src != a) {
    for (int i = threadIdx.x; i < P; i += blockDim.x) a[i] = src[i];
    __syncthreads();
  }
}
struct ScanAdd { __device__ float operator()(float a, float b) const { return a + b; } };
struct ScanMaxI { __device__ int operator()(int a, int b) const { return a > b ? a : b; } };
struct ScanMinI { __device__ int operator()(int a, int b) const { return a < b ? a : b; } };

template <bool BIG>
__global__ __launch_bounds__(1024) void list_mle_block_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ pos_weight, const float* __restrict__ list_scale, int B, int L, int P, float temperature,
    float* __restrict__ loss_out, float* __restrict__ dlogits_out, const GridSum sum, float* __restrict__ ws,
    const uint32_t tie_seed) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* XS = BIG ? ws + (size_t)blockIdx.x * 4 * P : reinterpret_cast<float*>(keys + P);   // [P] logits by original index, then scan scratch
  float* XP = XS + P;                                             // [P] logit - max by sorted position
  float* SA = XP + P;                                             // [P] reversed e -> suffix sums
  float* CA = SA + P;                                             // [P] w / S -> prefix sums
  const int T = blockDim.x, tid = threadIdx.x;
 for (int b = blockIdx.x; b < B; b += gridDim.x) {                // (one list per workgroup unless BIG)
  if (BIG) __syncthreads();
  const size_t base = (size_t)b * L;
  for (int i = tid; i < P; i += T) {
    uint64_t k = 0;
    float x = 0.f;
    if (i < L) {
      const float lab = labels[base + i];
      const bool v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      x = v ? logits[base + i] / temperature : kLogEps;
      k = make_sort_key(v, v ? lab : 0.0f, tie_key15(tie_seed, (uint32_t)b, (uint32_t)i), i);   // valid first, label desc, tie key, index
    }
    keys[i] = k; XS[i] = x;
  }
  block_bitonic_sort_desc(keys, P);
  float mx = -INFINITY;
  for (int p = tid; p < L; p += T) mx = fmaxf(mx, XS[sort_key_index(keys[p])]);
  mx = block_max(mx, red);
  for (int p = tid; p < P; p += T) {
    const float x = (p < L) ? XS[sort_key_index(keys[p])] - mx : -INFINITY;
    XP[p] = x;
    SA[P - 1 - p] = (p < L) ? expf(x) : 0.0f;
  }
  __syncthreads();
  block_scan_inclusive(SA, XS, P, ScanAdd());                     // S_p = SA[P - 1 - p] = sum_{q >= p} e_q
  float term = 0.f;
  for (int p = tid; p < P; p += T) {
    float c = 0.f;
    if (p < L) {
      const float w = pos_weight ? pos_weight[p] : 1.0f, S = SA[P - 1 - p];
      term += w * (logf(S) - XP[p]);
      c = w / S;
    }
    CA[p] = c;
  }
  const float loss = block_sum(term, red);
  if (tid == 0) loss_out[b] = loss;
  if (sum.out && tid < 64) grid_weighted_sum_last(loss_out, b, loss, list_scale, sum.n, sum.out, sum.st, tid);
  if (!dlogits_out) continue;
  __syncthreads();
  block_scan_inclusive(CA, XS, P, ScanAdd());                     // C_p = sum_{q <= p} w_q / S_q
  const float gscale = (list_scale ? list_scale[b] : 1.0f) / temperature;
  for (int p = tid; p < L; p += T) {
    const uint64_t k = keys[p];
    const float w = pos_weight ? pos_weight[p] : 1.0f;
    dlogits_out[base + sort_key_index(k)] = (k >> 63) ? (expf(XP[p]) * CA[p] - w) * gscale : 0.0f;
  }
 }
}

template <bool BIG>
__global__ __launch_bounds__(1024) void unique_softmax_block_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ list_scale, int B, int L, int P, float temperature, float* __restrict__ loss_out,
    float* __restrict__ dlogits_out, const GridSum sum, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* XS = BIG ? ws + (size_t)blockIdx.x * 7 * P : reinterpret_cast<float*>(keys + P);   // [P] logits by original index -> suffix sums (SA)
  float* LB = XS + P;                                             // [P] labels by original index -> scan scratch (SB)
  float* XP = LB + P;                                             // [P] logit - max by sorted position
  float* LP = XP + P;                                             // [P] label by sorted position
  int* GS = reinterpret_cast<int*>(LP + P);                       // [P] first position of the label group
  int* GE = GS + P;                                               // [P] one past its last position (reversed index)
  float* CA = reinterpret_cast<float*>(GE + P);                   // [P] g / D -> prefix sums
  const int T = blockDim.x, tid = threadIdx.x;
 for (int b = blockIdx.x; b < B; b += gridDim.x) {                // (one list per workgroup unless BIG)
  if (BIG) __syncthreads();
  const size_t base = (size_t)b * L;
  int nvl = 0;
  for (int i = tid; i < P; i += T) {
    uint64_t k = 0;
    float x = 0.f, lb = 0.f;
    if (i < L) {
      const float lab = labels[base + i];
      const bool v = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      x = v ? logits[base + i] / temperature : 0.0f;
      lb = v ? lab : 0.0f;
      k = make_sort_key(v, lb, 0, i);
      nvl += v ? 1 : 0;
      if (dlogits_out && !v) dlogits_out[base + i] = 0.0f;
    }
    keys[i] = k; XS[i] = x; LB[i] = lb;
  }
  const int nv = (int)(block_sum((float)nvl, red) + 0.5f);       // < 2^24: exact
  block_bitonic_sort_desc(keys, P);                               // valid first, label descending, then index
  float mx = -INFINITY;
  for (int p = tid; p < nv; p += T) mx = fmaxf(mx, XS[sort_key_index(keys[p])]);
  mx = block_max(mx, red);
  for (int p = tid; p < P; p += T) {
    const int idx = sort_key_index(keys[p]);
    XP[p] = (p < nv) ? XS[idx] - mx : -INFINITY;
    LP[p] = (p < nv) ? LB[idx] : -INFINITY;
  }
  __syncthreads();
  float* SA = XS;                                                 // (the by-index arrays are dead now)
  float* SB = LB;
  for (int p = tid; p < P; p += T) {
    SA[P - 1 - p] = (p < nv) ? expf(XP[p]) : 0.0f;
    // tie groups: position p opens a group iff its label is below the previous one; closes one iff the next is below
    const bool first = p < nv && (p == 0 || LP[p] < LP[p - 1]);
    const bool last = p < nv && (p + 1 >= nv || LP[p + 1] < LP[p]);
    GS[p] = first ? p : -1;
    GE[P - 1 - p] = last ? p + 1 : 0x7fffffff;
  }
  __syncthreads();
  block_scan_inclusive(SA, SB, P, ScanAdd());                     // SUF[p] = SA[P - 1 - p] = sum_{q >= p} e_q
  block_scan_inclusive(GS, reinterpret_cast<int*>(SB), P, ScanMaxI());      // group start
  block_scan_inclusive(GE, reinterpret_cast<int*>(SB), P, ScanMinI());      // group end, reversed: GE[P - 1 - p]
  float term = 0.f;
  for (int p = tid; p < P; p += T) {
    float c = 0.f;
    if (p < nv) {
      const int ge = GE[P - 1 - p];
      const float e = expf(XP[p]), D = e + (ge < P ? SA[P - 1 - ge] : 0.0f), g = gain_pow2m1(LP[p]);
      term += g * (logf(D) - XP[p]);
      c = g / D;
    }
    CA[p] = c;
  }
  const float loss = block_sum(term, red);
  if (tid == 0) loss_out[b] = loss;
  if (sum.out && tid < 64) grid_weighted_sum_last(loss_out, b, loss, list_scale, sum.n, sum.out, sum.st, tid);
  if (!dlogits_out) continue;
  __syncthreads();
  block_scan_inclusive(CA, SB, P, ScanAdd());                     // inclusive prefix of g / D
  const float gscale = (list_scale ? list_scale[b] : 1.0f) / temperature;
  for (int p = tid; p < nv; p += T) {
    const int ge = GE[P - 1 - p], gs = GS[p];
    const float e = expf(XP[p]), D = e + (ge < P ? SA[P - 1 - ge] : 0.0f), g = gain_pow2m1(LP[p]);
    const float higher = gs > 0 ? CA[gs - 1] : 0.0f;              // sum over strictly higher labels
    dlogits_out[base + sort_key_index(keys[p])] = (-g + e * (g / D + higher)) * gscale;
  }
 }
}

template <bool BIG>
__global__ __launch_bounds__(1024) void circle_block_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ list_scale, int B, int L, int P, float gamma, float margin, int clip,
    float* __restrict__ loss_out, float* __restrict__ weight_out, float* __restrict__ dlogits_out, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);               // [32]
  uint64_t* keys = reinterpret_cast<uint64_t*>(smem_raw + 128);  // [P]
  float* XS = BIG ? ws + (size_t)blockIdx.x * 7 * P : reinterpret_cast<float*>(keys + P);   // [P] raw logits by original index -> suffix sums of F
  float* LB = XS + P;                                             // [P] labels by original index -> scan scratch
  float* SP = LB + P;                                             // [P] raw score by sorted position
  float* LP = SP + P;                                             // [P] label by sorted position
  int* GS = reinterpret_cast<int*>(LP + P);                       // [P] first position of the label group
  int* GE = GS + P;                                               // [P] one past its last position (reversed index)
  float* EA = reinterpret_cast<float*>(GE + P);                   // [P] E -> inclusive prefix sums
  const int T = blockDim.x, tid = threadIdx.x;
 for (int b = blockIdx.x; b < B; b += gridDim.x) {                // (one list per workgroup unless BIG)
  if (BIG) __syncthreads();
  const size_t base = (size_t)b * L;
  int nvl = 0;
  for (int i = tid; i < P; i += T) {
    uint64_t k = 0;
    float x = 0.f, lb = 0.f;
    if (i < L) {
      lb = labels[base + i];
      const bool v = mask ? (mask[base + i] != 0) : (lb >= 0.0f);
      x = logits[base + i];
      k = make_sort_key(v, v ? lb : 0.0f, 0, i);
      nvl += v ? 1 : 0;
      if (dlogits_out && !v) dlogits_out[base + i] = 0.0f;
    }
    keys[i] = k; XS[i] = x; LB[i] = lb;
  }
  const int nv = (int)(block_sum((float)nvl, red) + 0.5f);
  block_bitonic_sort_desc(keys, P);                               // valid first, label descending, then index
  // per position: a_p = gamma * alpha_i (1 - margin - s), b_p = gamma * alpha'_j (s - margin)
  auto ga_of = [&](float raw) { const float sc = clip ? fminf(fmaxf(raw, 0.0f), 1.0f) : raw;
                                return gamma * (fmaxf(1.0f - sc + margin, 0.0f) * (1.0f - sc - margin)); };
  auto gb_of = [&](float raw) { const float sc = clip ? fminf(fmaxf(raw, 0.0f), 1.0f) : raw;
                                return gamma * (fmaxf(sc + margin, 0.0f) * (sc - margin)); };
  float ma = -INFINITY, mb = -INFINITY;
  for (int p = tid; p < P; p += T) {
    const int idx = sort_key_index(keys[p]);
    const float raw = (p < nv) ? XS[idx] : 0.0f;
    SP[p] = raw;
    LP[p] = (p < nv) ? LB[idx] : -INFINITY;
    if (p < nv) { ma = fmaxf(ma, ga_of(raw)); mb = fmaxf(mb, gb_of(raw)); }
  }
  const float MA = block_max(ma, red), MB = block_max(mb, red);
  __syncthreads();
  float* SA = XS;
  float* SB = LB;
  for (int p = tid; p < P; p += T) {
    const bool in = p < nv;
    SA[P - 1 - p] = in ? expf(gb_of(SP[p]) - MB) : 0.0f;
    EA[p] = in ? expf(ga_of(SP[p]) - MA) : 0.0f;
    const bool first = in && (p == 0 || LP[p] < LP[p - 1]);
    const bool last = in && (p + 1 >= nv || LP[p + 1] < LP[p]);
    GS[p] = first ? p : -1;
    GE[P - 1 - p] = last ? p + 1 : 0x7fffffff;
  }
  __syncthreads();
  block_scan_inclusive(SA, SB, P, ScanAdd());                     // SUF[p] = SA[P - 1 - p] = sum_{q >= p} F_q
  block_scan_inclusive(EA, SB, P, ScanAdd());                     // inclusive prefix of E
  block_scan_inclusive(GS, reinterpret_cast<int*>(SB), P, ScanMaxI());
  block_scan_inclusive(GE, reinterpret_cast<int*>(SB), P, ScanMinI());
  // t_p = log( e^{ga_p} * sum_{y_q < y_p} e^{gb_q} );  lw = logsumexp_p t_p = log W
  auto t_of = [&](int p) {
    const int ge = GE[P - 1 - p];
    const float lower = ge < P ? SA[P - 1 - ge] : 0.0f;
    return lower > 0.0f ? ga_of(SP[p]) + (MB + logf(lower)) : -INFINITY;
  };
  float tm = -INFINITY;
  for (int p = tid; p < nv; p += T) tm = fmaxf(tm, t_of(p));
  tm = block_max(tm, red);
  const bool any_pair = tm > -INFINITY;
  float acc = 0.f;
  for (int p = tid; p < nv; p += T) { const float t = t_of(p); acc += (t > -INFINITY) ? expf(t - tm) : 0.0f; }
  acc = block_sum(acc, red);
  const float lw = any_pair ? tm + logf(acc) : -INFINITY;
  const float loss = any_pair ? (fmaxf(lw, 0.0f) + log1pf(expf(-fabsf(lw)))) : 0.0f;      // log1p(W)
  if (tid == 0) {
    loss_out[b] = loss;
    if (weight_out) weight_out[b] = any_pair ? 1.0f : NAN;
  }
  if (!dlogits_out) continue;
  const float sig = any_pair ? 1.0f / (1.0f + expf(-lw)) : 0.0f;                          // W / (1 + W)
  const float gscale = (list_scale ? list_scale[b] : 1.0f) * gamma * sig;
  for (int p = tid; p < nv; p += T) {
    const float raw = SP[p];
    const float sc = clip ? fminf(fmaxf(raw, 0.0f), 1.0f) : raw;
    const bool inside = !clip || (raw >= 0.0f && raw <= 1.0f);   // clip_by_value passes the gradient on [0, 1]
    float gpart = 0.f;
    if (any_pair && inside) {
      const float t = t_of(p);
      const float hi = (t > -INFINITY) ? expf(t - lw) : 0.0f;                             // item as the preferred one
      const int gs = GS[p];
      const float higher = gs > 0 ? EA[gs - 1] : 0.0f;
      const float lo = (higher > 0.0f) ? expf(gb_of(raw) + (MA + logf(higher)) - lw) : 0.0f;   // item as the other one
      gpart = -fmaxf(1.0f - sc + margin, 0.0f) * hi + fmaxf(sc + margin, 0.0f) * lo;
    }
    dlogits_out[base + sort_key_index(keys[p])] = gscale * gpart;
  }
 }
}

}  // namespace

// Bytes of workspace ONE list in flight needs (include/tfr_hip.h): the per-position arrays of the workgroup kernels once
// they outgrow LDS next to the sort keys.  0: the op runs from LDS at this list size.
extern "C" long tfr_list_workspace_bytes(int op, int L) {
  if (L <= 0 || L > TFR_MAX_LIST) return 0;
  const long P = pow2_ceil(L);
  switch (op) {
    case TFR_WS_LIST_MLE: return L > TFR_LDS_LIST_SIZE_LISTWISE ? 16 * P : 0;
    case TFR_WS_UNIQUE_SOFTMAX: case TFR_WS_CIRCLE: return L > TFR_LDS_LIST_SIZE_LISTWISE ? 28 * P : 0;
    case TFR_WS_RANK_METRIC: return L > TFR_LDS_LIST_SIZE_METRIC ? 24 * P : 0;
    case TFR_WS_DIV_METRIC: return L > TFR_LDS_LIST_SIZE_METRIC ? 20 * P : 0;
    case TFR_WS_NEURAL_SORT_NDCG: return L > TFR_LDS_LIST_SIZE_NEURAL_SORT ? 32 * P : 0;
    case TFR_WS_NEURAL_SORT_CE: return L > TFR_LDS_LIST_SIZE_NEURAL_SORT ? 52 * P : 0;
    default: return 0;
  }
}

static int list_mle_dispatch(const float* logits, const float* labels, const uint8_t* mask,
                             const float* pos_weight, const float* list_scale, int B, int L,
                             float temperature, float* loss_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                             uint32_t tie_seed, void* workspace, long workspace_bytes, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const size_t slot = (size_t)tfr_list_workspace_bytes(TFR_WS_LIST_MLE, L);      // > 0: the per-position arrays outgrow LDS
  if (slot && (!workspace || workspace_bytes < (long)slot)) return TFR_ETOOLARGE;
  if (B == 0) return loss_sum_out ? (int)hipMemsetAsync(loss_sum_out, 0, sizeof(float), (hipStream_t)stream) : TFR_OK;
  const GridSum sum = {loss_sum_out, loss_out, ticket, B};
  hipStream_t st = (hipStream_t)stream;
  if (L > 1024) {
    const int P = pow2_ceil(L);
    const size_t lds = 128 + (size_t)P * (slot ? 8 : 24);
    auto fn = slot ? list_mle_block_kernel<true> : list_mle_block_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(slot ? big_slots(B, (size_t)workspace_bytes, slot) : B), dim3(1024), lds, st, logits, labels, mask,
                       pos_weight, list_scale, B, L, P, temperature, loss_out, dlogits_out, sum, (float*)workspace, tie_seed);
    return (int)hipGetLastError();
  }
#define LM(I) hipLaunchKernelGGL(list_mle_wave_kernel<I>, dim3(B), dim3(64), (size_t)64 * I * sizeof(float), st, logits, labels, mask, pos_weight, list_scale, L, temperature, loss_out, dlogits_out, sum, tie_seed)
  if (L <= 64) LM(1); else if (L <= 128) LM(2); else if (L <= 256) LM(4); else if (L <= 512) LM(8); else LM(16);
#undef LM
  return (int)hipGetLastError();
}

extern "C" int tfr_list_mle_f32(const float* logits, const float* labels, const uint8_t* mask,
                                const float* pos_weight, const float* list_scale, int B, int L,
                                float temperature, float* loss_out, float* dlogits_out, uint32_t tie_seed, void* workspace,
                                long workspace_bytes, void* stream) {
  return list_mle_dispatch(logits, labels, mask, pos_weight, list_scale, B, L, temperature, loss_out, dlogits_out, nullptr,
                           nullptr, tie_seed, workspace, workspace_bytes, stream);
}

extern "C" int tfr_list_mle_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                                    const float* pos_weight, const float* list_scale, int B, int L,
                                    float temperature, float* loss_out, float* dlogits_out, float* loss_sum_out,
                                    uint32_t* ticket, uint32_t tie_seed, void* workspace, long workspace_bytes, void* stream) {
  if (!loss_sum_out || !ticket) return TFR_EINVAL;
  return list_mle_dispatch(logits, labels, mask, pos_weight, list_scale, B, L, temperature, loss_out, dlogits_out,
                           loss_sum_out, ticket, tie_seed, workspace, workspace_bytes, stream);
}

static int unique_softmax_dispatch(const float* logits, const float* labels, const uint8_t* mask,
                                   const float* list_scale, int B, int L, float temperature,
                                   float* loss_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                                   void* workspace, long workspace_bytes, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const size_t slot = (size_t)tfr_list_workspace_bytes(TFR_WS_UNIQUE_SOFTMAX, L);
  if (slot && (!workspace || workspace_bytes < (long)slot)) return TFR_ETOOLARGE;
  if (B == 0) return loss_sum_out ? (int)hipMemsetAsync(loss_sum_out, 0, sizeof(float), (hipStream_t)stream) : TFR_OK;
  const GridSum sum = {loss_sum_out, loss_out, ticket, B};
  hipStream_t st = (hipStream_t)stream;
  if (L > 1024) {
    const int P = pow2_ceil(L);
    const size_t lds = 128 + (size_t)P * (slot ? 8 : 36);
    auto fn = slot ? unique_softmax_block_kernel<true> : unique_softmax_block_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(slot ? big_slots(B, (size_t)workspace_bytes, slot) : B), dim3(1024), lds, st, logits, labels, mask,
                       list_scale, B, L, P, temperature, loss_out, dlogits_out, sum, (float*)workspace);
    return (int)hipGetLastError();
  }
#define US(I) hipLaunchKernelGGL(unique_softmax_wave_kernel<I>, dim3(B), dim3(64), (size_t)(4 * 64 * I + 2) * sizeof(float), st, logits, labels, mask, list_scale, L, temperature, loss_out, dlogits_out, sum)
  if (L <= 64) US(1); else if (L <= 128) US(2); else if (L <= 256) US(4); else if (L <= 512) US(8); else US(16);
#undef US
  return (int)hipGetLastError();
}

extern "C" int tfr_unique_softmax_f32(const float* logits, const float* labels, const uint8_t* mask,
                                      const float* list_scale, int B, int L, float temperature,
                                      float* loss_out, float* dlogits_out, void* workspace, long workspace_bytes,
                                      void* stream) {
  return unique_softmax_dispatch(logits, labels, mask, list_scale, B, L, temperature, loss_out, dlogits_out, nullptr, nullptr,
                                 workspace, workspace_bytes, stream);
}

extern "C" int tfr_unique_softmax_sum_f32(const float* logits, const float* labels, const uint8_t* mask,
                                          const float* list_scale, int B, int L, float temperature,
                                          float* loss_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                                          void* workspace, long workspace_bytes, void* stream) {
  if (!loss_sum_out || !ticket) return TFR_EINVAL;
  return unique_softmax_dispatch(logits, labels, mask, list_scale, B, L, temperature, loss_out, dlogits_out, loss_sum_out,
                                 ticket, workspace, workspace_bytes, stream);
}

extern "C" int tfr_circle_loss_f32(const float* logits, const float* labels, const uint8_t* mask,
                                   const float* list_scale, int B, int L, float gamma, float margin, int clip,
                                   float* loss_out, float* weight_out, float* dlogits_out, void* workspace,
                                   long workspace_bytes, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;
  const size_t slot = (size_t)tfr_list_workspace_bytes(TFR_WS_CIRCLE, L);
  if (slot && (!workspace || workspace_bytes < (long)slot)) return TFR_ETOOLARGE;
  if (B == 0) return TFR_OK;
  hipStream_t st = (hipStream_t)stream;
  if (L > 1024) {
    const int P = pow2_ceil(L);
    const size_t lds = 128 + (size_t)P * (slot ? 8 : 36);
    auto fn = slot ? circle_block_kernel<true> : circle_block_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(fn, dim3(slot ? big_slots(B, (size_t)workspace_bytes, slot) : B), dim3(1024), lds, st, logits, labels, mask,
                       list_scale, B, L, P, gamma, margin, clip, loss_out, weight_out, dlogits_out, (float*)workspace);
    return (int)hipGetLastError();
  }
#define CL(I) hipLaunchKernelGGL(circle_wave_kernel<I>, dim3(B), dim3(64), (size_t)(4 * 64 * I + 2) * sizeof(float), st, logits, labels, mask, list_scale, L, gamma, margin, clip, loss_out, weight_out, dlogits_out)
  if (L <= 64) CL(1); else if (L <= 128) CL(2); else if (L <= 256) CL(4); else if (L <= 512) CL(8); else CL(16);
#undef CL
  return (int)hipGetLastError();
}
