// ListMLE loss, forward + backward fused, wave-per-list (gfx950).
//
// Reference behaviour restated (losses_impl.py:1541-1576 ListMLELoss, :457-480
// ListMLELambdaWeight): masked labels := 0, masked logits := log(1e-10); items sorted by label
// (descending; the reference shuffles ties with a fixed op seed -- here ties keep index order,
// "parity unpinned" like every tie rule); with s the sorted logits,
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
    float* __restrict__ loss_out, float* __restrict__ dlogits_out) {
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
      key[r] = make_sort_key(v, v ? lab : 0.0f, 0, i);          // valid first, label desc, then index
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

}  // namespace

extern "C" int tfr_list_mle_f32(const float* logits, const float* labels, const uint8_t* mask,
                                const float* pos_weight, const float* list_scale, int B, int L,
                                float temperature, float* loss_out, float* dlogits_out, void* stream) {
  if (!logits || !labels || !loss_out || B < 0 || L <= 0 || !(temperature > 0.0f)) return TFR_EINVAL;
  if (L > 1024) return TFR_ETOOLARGE;            // wave-per-list kernel only
  if (B == 0) return TFR_OK;
  hipStream_t st = (hipStream_t)stream;
#define LM(I) hipLaunchKernelGGL(list_mle_wave_kernel<I>, dim3(B), dim3(64), (size_t)64 * I * sizeof(float), st, logits, labels, mask, pos_weight, list_scale, L, temperature, loss_out, dlogits_out)
  if (L <= 64) LM(1); else if (L <= 128) LM(2); else if (L <= 256) LM(4); else if (L <= 512) LM(8); else LM(16);
#undef LM
  return (int)hipGetLastError();
}
