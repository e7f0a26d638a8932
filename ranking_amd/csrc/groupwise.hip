// Groupwise multi-item scoring around the scorer tower:
//   reference  tensorflow_ranking/python/model.py:164-244 (_rolling_window_indices, _form_group_indices_nd),
//              :341-421 (_compute_logits_impl: gather_nd -> group_score_fn -> 2 x scatter_nd -> div_no_nan),
//              tensorflow_ranking/python/utils.py:203-230 (organize_valid_indices).
//
// Three HBM-bound kernels replace the ~25 index / gather / scatter launches of the op-by-op formulation:
//   group_indices        one wavefront per list: valid-first (optionally key-shuffled) order by a counting rank
//                        (== a stable descending sort of the keys), then the rolling windows -> idx [B, L, gs]
//   group_gather_cast    the [B * G, gs * F] MLP input as bf16, gathered and converted in one pass (it IS the
//                        tower's input cast: nothing fp32 of that size is ever written)
//   group_scatter_avg    logits[b, i] = sum of the scores that landed on item i / their count, one wavefront
//                        per list, contributions added in ENTRY order (deterministic, the order a sequential
//                        scatter uses); its backward is a gather of dlogits / count.
#include "common.h"
#include "../../include/tfr_hip.h"

namespace {

typedef __attribute__((ext_vector_type(2))) float f32x2;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {     // RNE (v_cvt_pk_bf16_f32)
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

// ------------------------------------------------------------------------------------ group indices
// LDS per wave: L floats (keys) + L ints (organized order).
__global__ __launch_bounds__(256) void group_indices_kernel(const uint8_t* __restrict__ is_valid,
                                                            const float* __restrict__ keys, int B, int L, int gs,
                                                            int* __restrict__ idx_out, uint8_t* __restrict__ gmask_out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + wave;          // 4 list-waves per workgroup, fewer for long lists
  if (b >= B) return;
  float* key = reinterpret_cast<float*>(smem) + (long)wave * 2 * L;
  int* organized = reinterpret_cast<int*>(key + L);
  const uint8_t* vd = is_valid + (long)b * L;
  int n = 0;
  for (int p0 = 0; p0 < L; p0 += 64) {
    const int p = p0 + lane;
    const bool ok = p < L && vd[p] != 0;
    // utils.py:219-227: valid entries carry the random value (shuffle) or a value that decreases with the index
    // (no shuffle: index order); invalid entries carry -1e-6, below every valid key.
    if (p < L) key[p] = ok ? (keys ? keys[(long)b * L + p] : (float)(L - 1 - p)) : -1e-6f;
    n += __popcll(__ballot(ok));
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  // position of item i in a STABLE descending sort = #{j : key_j > key_i  or  (key_j == key_i and j < i)}
  for (int i = lane; i < L; i += 64) {
    const float ki = key[i];
    int pos = 0;
    for (int j = 0; j < L; ++j) {
      const float kj = key[j];                                   // wave-uniform address: LDS broadcast
      pos += (kj > ki || (kj == ki && j < i)) ? 1 : 0;
    }
    organized[pos] = i;
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);
  __builtin_amdgcn_wave_barrier();
  const int n1 = n < 1 ? 1 : n;                                  // model.py:196-199
  const long base = (long)b * L;
  for (int g = lane; g < L; g += 64) {
    gmask_out[base + g] = g < n ? 1 : 0;                         // model.py:190-192 (the window's first index)
    for (int k = 0; k < gs; ++k) idx_out[(base + g) * gs + k] = organized[(g + k) % n1];
  }
}

// ------------------------------------------------------------------------------------ gather + cast
// out[(b, g), k * F + f] = bf16(x[b, idx[b, g, k], f]); columns >= gs * F are zero.  One thread per 8 output columns.
template <bool VEC>
__global__ void group_gather_cast_kernel(const float* __restrict__ x, long ldx, const int* __restrict__ idx, int B,
                                         int L, int G, int gs, int F, int Kp, uint16_t* __restrict__ out) {
  const long chunks_per_row = Kp / 8;
  const long total = (long)B * G * chunks_per_row;
  const int width = gs * F;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long row = q / chunks_per_row;                          // = b * G + g
    const int c = (int)(q % chunks_per_row) * 8;
    const long b = row / G;
    float v[8];
    if (VEC) {                                                    // F % 8 == 0: a chunk never straddles two items
      if (c < width) {
        const int k = c / F, f = c - k * F;
        const int item = idx[row * gs + k];
        const float4* src = reinterpret_cast<const float4*>(x + (b * L + item) * ldx + f);
        const float4 a = src[0], d = src[1];
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = d.x; v[5] = d.y; v[6] = d.z; v[7] = d.w;
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = 0.f;
      }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int cc = c + e;
        float t = 0.f;
        if (cc < width) {
          const int k = cc / F, f = cc - k * F;
          t = x[(b * L + idx[row * gs + k]) * ldx + f];
        }
        v[e] = t;
      }
    }
    *reinterpret_cast<uint4*>(out + row * Kp + c) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
  }
}

// ------------------------------------------------------------------------------------ scatter-average
// One wavefront per list.  Entries e = (g, k) in row-major order are staged through LDS in chunks; lane t owns the
// items t, t + 64, ... (<= IPL_MAX per pass) and adds the entries that target them in entry order.
constexpr int SC_CHUNK = 1024;          // entries per LDS chunk per wave: 8 KB
constexpr int SC_IPL = 16;              // items per lane per pass (1024 items per pass)

__global__ __launch_bounds__(256) void group_scatter_avg_kernel(const float* __restrict__ scores,
                                                                const int* __restrict__ idx,
                                                                const uint8_t* __restrict__ gmask, int B, int L, int G,
                                                                int gs, float* __restrict__ logits,
                                                                float* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) int s_item[4][SC_CHUNK];
  __shared__ __attribute__((aligned(16))) float s_val[4][SC_CHUNK];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const long E = (long)G * gs;
  const int* ix = idx + (long)b * E;
  const float* sc = scores + (long)b * E;
  const uint8_t* gm = gmask + (long)b * G;
  for (int i0 = 0; i0 < L; i0 += 64 * SC_IPL) {
    float acc[SC_IPL], cnt[SC_IPL];
#pragma unroll
    for (int u = 0; u < SC_IPL; ++u) { acc[u] = 0.f; cnt[u] = 0.f; }
    for (long e0 = 0; e0 < E; e0 += SC_CHUNK) {
      const int ne = (int)((E - e0 < SC_CHUNK) ? (E - e0) : SC_CHUNK);
      __builtin_amdgcn_wave_barrier();                           // the previous chunk has been consumed
      for (int e = lane; e < ne; e += 64) {
        const long ge = e0 + e;
        const bool on = gm[ge / gs] != 0;                        // model.py:389-396: invalid groups contribute nothing
        s_item[wave][e] = on ? ix[ge] : -1;
        s_val[wave][e] = on ? sc[ge] : 0.f;
      }
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __builtin_amdgcn_wave_barrier();
      for (int e = 0; e < ne; ++e) {
        const int item = s_item[wave][e] - i0 - lane;            // wave-uniform LDS address: broadcast
        const float val = s_val[wave][e];
#pragma unroll
        for (int u = 0; u < SC_IPL; ++u) {
          const bool hit = item == 64 * u;
          acc[u] = hit ? acc[u] + val : acc[u];
          cnt[u] = hit ? cnt[u] + 1.f : cnt[u];
        }
      }
    }
#pragma unroll
    for (int u = 0; u < SC_IPL; ++u) {
      const int i = i0 + 64 * u + lane;
      if (i < L) {
        logits[(long)b * L + i] = cnt[u] != 0.f ? acc[u] / cnt[u] : 0.f;     // div_no_nan, model.py:407
        if (counts) counts[(long)b * L + i] = cnt[u];
      }
    }
  }
}

// Short lists (L <= 64): one item per lane, no unrolled item loop.
__global__ __launch_bounds__(256) void group_scatter_avg_small_kernel(const float* __restrict__ scores,
                                                                      const int* __restrict__ idx,
                                                                      const uint8_t* __restrict__ gmask, int B, int L,
                                                                      int G, int gs, float* __restrict__ logits,
                                                                      float* __restrict__ counts) {
  __shared__ __attribute__((aligned(16))) int s_item[4][SC_CHUNK];
  __shared__ __attribute__((aligned(16))) float s_val[4][SC_CHUNK];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  const long E = (long)G * gs;
  const int* ix = idx + (long)b * E;
  const float* sc = scores + (long)b * E;
  const uint8_t* gm = gmask + (long)b * G;
  float acc = 0.f, cnt = 0.f;
  for (long e0 = 0; e0 < E; e0 += SC_CHUNK) {
    const int ne = (int)((E - e0 < SC_CHUNK) ? (E - e0) : SC_CHUNK);
    __builtin_amdgcn_wave_barrier();
    for (int e = lane; e < ne; e += 64) {
      const long ge = e0 + e;
      const bool on = gm[ge / gs] != 0;
      s_item[wave][e] = on ? ix[ge] : -1;
      s_val[wave][e] = on ? sc[ge] : 0.f;
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    for (int e = 0; e < ne; ++e) {
      const bool hit = s_item[wave][e] == lane;
      const float val = s_val[wave][e];
      acc = hit ? acc + val : acc;
      cnt = hit ? cnt + 1.f : cnt;
    }
  }
  if (lane < L) {
    logits[(long)b * L + lane] = cnt != 0.f ? acc / cnt : 0.f;
    if (counts) counts[(long)b * L + lane] = cnt;
  }
}

// d scores[b, g, k] = gmask[b, g] and count > 0 ? d logits[b, idx] / count[b, idx] : 0
__global__ void group_scatter_avg_bwd_kernel(const float* __restrict__ dlogits, const float* __restrict__ counts,
                                             const int* __restrict__ idx, const uint8_t* __restrict__ gmask, int B,
                                             int L, int G, int gs, float* __restrict__ dscores) {
  const long total = (long)B * G * gs;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long bg = q / gs;
    const long b = bg / G;
    float d = 0.f;
    if (gmask[bg] != 0) {
      const long at = b * L + idx[q];
      const float c = counts[at];
      d = c != 0.f ? dlogits[at] / c : 0.f;
    }
    dscores[q] = d;
  }
}

}  // namespace

extern "C" int tfr_group_indices_i32(const uint8_t* is_valid, const float* keys, int B, int L, int group_size,
                                     int32_t* idx_out, uint8_t* gmask_out, void* stream) {
  if (!is_valid || !idx_out || !gmask_out || B < 0 || L <= 0 || group_size <= 0) return TFR_EINVAL;
  if ((long)B * L * group_size > 0x7fffffffL) return TFR_EINVAL;
  if (L > TFR_MAX_LIST) return TFR_ETOOLARGE;                      // 8 bytes x L of LDS per list-wave, <= 64 KiB
  if (B == 0) return TFR_OK;
  int W = 4;                                                        // list-waves per workgroup
  while (W > 1 && (size_t)L * 8 * W > 64 * 1024) W >>= 1;
  hipLaunchKernelGGL(group_indices_kernel, dim3((B + W - 1) / W), dim3(64 * W), (size_t)L * 8 * W, (hipStream_t)stream,
                     is_valid, keys, B, L, group_size, idx_out, gmask_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_group_gather_cast_f32_bf16(const float* x, long ldx, const int32_t* idx, int B, int L, int G,
                                              int group_size, int F, int Kp, void* out_bf16, void* stream) {
  if (!x || !idx || !out_bf16 || B < 0 || L <= 0 || G < 0 || group_size <= 0 || F <= 0 || ldx < F) return TFR_EINVAL;
  if (Kp < group_size * F || (Kp & 7)) return TFR_EINVAL;
  if (B == 0 || G == 0) return TFR_OK;
  const bool vec = (F % 8 == 0) && (ldx % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
  const int grid = grid_for((long)B * G * (Kp / 8), 256);
  if (vec)
    hipLaunchKernelGGL(group_gather_cast_kernel<true>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, B, L,
                       G, group_size, F, Kp, (uint16_t*)out_bf16);
  else
    hipLaunchKernelGGL(group_gather_cast_kernel<false>, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, idx, B,
                       L, G, group_size, F, Kp, (uint16_t*)out_bf16);
  return (int)hipGetLastError();
}

extern "C" int tfr_group_scatter_avg_f32(const float* scores, const int32_t* idx, const uint8_t* gmask, int B, int L,
                                         int G, int group_size, float* logits_out, float* counts_out, void* stream) {
  if (!scores || !idx || !gmask || !logits_out || B < 0 || L <= 0 || G < 0 || group_size <= 0) return TFR_EINVAL;
  if (B == 0) return TFR_OK;
  if (L <= 64)
    hipLaunchKernelGGL(group_scatter_avg_small_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores,
                       idx, gmask, B, L, G, group_size, logits_out, counts_out);
  else
    hipLaunchKernelGGL(group_scatter_avg_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, scores, idx,
                       gmask, B, L, G, group_size, logits_out, counts_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_group_scatter_avg_bwd_f32(const float* dlogits, const float* counts, const int32_t* idx,
                                             const uint8_t* gmask, int B, int L, int G, int group_size,
                                             float* dscores_out, void* stream) {
  if (!dlogits || !counts || !idx || !gmask || !dscores_out || B < 0 || L <= 0 || G < 0 || group_size <= 0)
    return TFR_EINVAL;
  if (B == 0 || G == 0) return TFR_OK;
  hipLaunchKernelGGL(group_scatter_avg_bwd_kernel, dim3(grid_for((long)B * G * group_size, 256)), dim3(256), 0,
                     (hipStream_t)stream, dlogits, counts, idx, gmask, B, L, G, group_size, dscores_out);
  return (int)hipGetLastError();
}
