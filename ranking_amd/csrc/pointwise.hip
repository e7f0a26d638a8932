// Pointwise losses, forward + backward fused (gfx950): SigmoidCrossEntropyLoss (losses_impl.py:1425-1446) and
// MeanSquaredLoss (:1449-1469) with _PointwiseLoss's weighting (:1284-1321).
//
// Per item: m = mask (or label >= 0), y = m ? label : 0, x = m ? logit / T : 0,
//   sigmoid CE : l = relu(x) - x y + log1p(exp(-|x|)),  dl/dx = sigma(x) - y
//   MSE        : l = (y - x)^2,                          dl/dx = 2 (x - y)
//   w = (label >= 0 ? item_weight * list_weight : 0) * [m]            (normalize_weights x loss_weights)
// Per list: sum w l, sum w, #(w != 0) -- what the four reductions and compute_per_list need -- and
// d(sum w l)/d logit.  The reference builds ~10 [B, L] tensors forward and as many backward; this is one pass:
// 8 (+4 weights, +1 mask) bytes read and 4 written per item: HBM-bound.  One wavefront per list, any list size.
#include "common.h"
#include "../../include/tfr_hip.h"

using namespace tfr;

namespace {

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;

// IPL > 0: list_size <= 64 * IPL, the loads of a list are all issued before the first use (memory-level
// parallelism is what an HBM-bound kernel of four-iteration waves lacks); IPL == 0: chunked loop, any list size.
// Four lists per 256-thread workgroup.
template <int KIND>
__device__ __forceinline__ void point_item(const float lab, const float logit, const bool m, float w, const float lw,
                                           const float temperature, float& sl, float& sw, float& nz, float& dout) {
  const bool lv = lab >= 0.0f;
  const float y = m ? lab : 0.0f;
  const float x = m ? logit / temperature : 0.0f;
  w = (lv && m) ? w * lw : 0.0f;
  float l, d;
  if (KIND == TFR_POINT_SIGMOID_CE) {
    const float e = __builtin_amdgcn_exp2f(-fabsf(x) * kLog2e);          // exp(-|x|)
    l = fmaxf(x, 0.0f) - x * y + log1pf(e);
    const float q = 1.0f / (1.0f + e);                                    // sigma(|x|)
    d = ((x >= 0.0f) ? q : e * q) - y;
  } else {
    const float t = y - x;
    l = t * t;
    d = -2.0f * t;
  }
  sl = __builtin_fmaf(w, l, sl);
  sw += w;
  nz += (w != 0.0f) ? 1.0f : 0.0f;
  dout = m ? (w * d) / temperature : 0.0f;
}

template <int KIND, int IPL>
__global__ __launch_bounds__(256) void pointwise_wave_kernel(
    const float* __restrict__ logits, const float* __restrict__ labels, const uint8_t* __restrict__ mask,
    const float* __restrict__ item_weights, const float* __restrict__ list_weights, int B, int L, float temperature,
    float* __restrict__ list_loss, float* __restrict__ list_weight, float* __restrict__ list_nnz,
    float* __restrict__ dlogits, const GridSum sum) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  const size_t base = (size_t)b * L;
  const float lw = list_weights ? list_weights[b] : 1.0f;
  float sl = 0.f, sw = 0.f, nz = 0.f;
  if (IPL > 0) {
    constexpr int N = IPL > 0 ? IPL : 1;
    float lab[N], lg[N], w[N];
    bool m[N];
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const int i = lane + 64 * r;
      lab[r] = -1.0f; lg[r] = 0.f; w[r] = 1.0f; m[r] = false;
      if (i < L) {
        lab[r] = labels[base + i];
        lg[r] = logits[base + i];
        if (item_weights) w[r] = item_weights[base + i];
        m[r] = mask ? (mask[base + i] != 0) : (lab[r] >= 0.0f);
      }
    }
#pragma unroll
    for (int r = 0; r < N; ++r) {
      const int i = lane + 64 * r;
      if (i < L) {
        float d;
        point_item<KIND>(lab[r], lg[r], m[r], w[r], lw, temperature, sl, sw, nz, d);
        if (dlogits) dlogits[base + i] = d;
      }
    }
  } else {
    for (int i = lane; i < L; i += 64) {
      const float lab = labels[base + i];
      const bool m = mask ? (mask[base + i] != 0) : (lab >= 0.0f);
      float d;
      point_item<KIND>(lab, logits[base + i], m, item_weights ? item_weights[base + i] : 1.0f, lw, temperature,
                       sl, sw, nz, d);
      if (dlogits) dlogits[base + i] = d;
    }
  }
  sl = wave_sum_u(sl); sw = wave_sum_u(sw); nz = wave_sum_u(nz);
  if (lane == 0) {
    list_loss[b] = sl;
    if (list_weight) list_weight[b] = sw;
    if (list_nnz) list_nnz[b] = nz;
  }
  if (sum.out) grid_sum_contribute(sum, b, sl, lane);        // sum_b list_loss[b] from the same launch (round 5)
}

// out[0] = sum_i x_i * w_i (w nullable: plain sum) over n <= 65536 values: the scalar reduction of a per-list loss
// vector (compute_weighted_loss / Keras reduce, losses_impl.py:787-814).  ONE workgroup, 16-byte loads all issued
// before the adds, a fixed summation order (thread-strided partials, then the block tree): run-to-run identical.
// (The rocBLAS dot it replaces took 4.2 us plus its launch for 16384 values.)
__global__ __launch_bounds__(1024) void list_dot_kernel(const float* __restrict__ x, const float* __restrict__ w, int n,
                                                        float* __restrict__ out) {
  __shared__ float red[32];
  float acc = 0.f;
  const int n4 = n >> 2;
  const float4* x4 = reinterpret_cast<const float4*>(x);
  const float4* w4 = reinterpret_cast<const float4*>(w);
  for (int i = threadIdx.x; i < n4; i += 1024) {
    const float4 a = x4[i];
    if (w) { const float4 b = w4[i]; acc += (a.x * b.x + a.y * b.y) + (a.z * b.z + a.w * b.w); }
    else acc += (a.x + a.y) + (a.z + a.w);
  }
  for (int i = (n4 << 2) + threadIdx.x; i < n; i += 1024) acc += w ? x[i] * w[i] : x[i];
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) out[0] = acc;
}

}  // namespace

static int pointwise_dispatch(int kind, const float* logits, const float* labels, const uint8_t* mask,
                              const float* item_weights, const float* list_weights, int B, int L,
                              float temperature, float* list_loss_out, float* list_weight_out,
                              float* list_nnz_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket, void* stream) {
  if (!logits || !labels || !list_loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (kind != TFR_POINT_SIGMOID_CE && kind != TFR_POINT_MSE) return TFR_EINVAL;
  if (B == 0) return loss_sum_out ? (int)hipMemsetAsync(loss_sum_out, 0, sizeof(float), (hipStream_t)stream) : TFR_OK;
  const GridSum sum = {loss_sum_out, list_loss_out, ticket, B};
  hipStream_t st = (hipStream_t)stream;
#define PW(K, I) hipLaunchKernelGGL((pointwise_wave_kernel<K, I>), dim3((B + 3) / 4), dim3(256), 0, st, logits, labels, mask, item_weights, list_weights, B, L, temperature, list_loss_out, list_weight_out, list_nnz_out, dlogits_out, sum)
#define PW_K(K) do { if (L <= 64) PW(K, 1); else if (L <= 128) PW(K, 2); else if (L <= 256) PW(K, 4); else if (L <= 512) PW(K, 8); else if (L <= 1024) PW(K, 16); else PW(K, 0); } while (0)
  if (kind == TFR_POINT_SIGMOID_CE) PW_K(TFR_POINT_SIGMOID_CE); else PW_K(TFR_POINT_MSE);
#undef PW_K
#undef PW
  return (int)hipGetLastError();
}

extern "C" int tfr_pointwise_loss_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                                      const float* item_weights, const float* list_weights, int B, int L,
                                      float temperature, float* list_loss_out, float* list_weight_out,
                                      float* list_nnz_out, float* dlogits_out, void* stream) {
  return pointwise_dispatch(kind, logits, labels, mask, item_weights, list_weights, B, L, temperature, list_loss_out,
                            list_weight_out, list_nnz_out, dlogits_out, nullptr, nullptr, stream);
}

extern "C" int tfr_pointwise_loss_sum_f32(int kind, const float* logits, const float* labels, const uint8_t* mask,
                                          const float* item_weights, const float* list_weights, int B, int L,
                                          float temperature, float* list_loss_out, float* list_weight_out,
                                          float* list_nnz_out, float* dlogits_out, float* loss_sum_out, uint32_t* ticket,
                                          void* stream) {
  if (!loss_sum_out || !ticket) return TFR_EINVAL;
  return pointwise_dispatch(kind, logits, labels, mask, item_weights, list_weights, B, L, temperature, list_loss_out,
                            list_weight_out, list_nnz_out, dlogits_out, loss_sum_out, ticket, stream);
}

extern "C" int tfr_list_dot_f32(const float* x, const float* w, int n, float* out, void* stream) {
  if (!x || !out || n < 0) return TFR_EINVAL;
  if (n > 65536) return TFR_ETOOLARGE;                       // one workgroup; longer vectors: the caller's library dot
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (w && (reinterpret_cast<uintptr_t>(w) & 15))) return TFR_EINVAL;
  hipLaunchKernelGGL(list_dot_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, x, w, n, out);
  return (int)hipGetLastError();
}
