// Dense layers of the scorer tower in the REFERENCE's precision (keras/layers.py:26-77 builds fp32 Dense layers):
// C[M, N] = op(A)[M, K] . op(B)[K, N] (+ bias) with fp32 operands and fp32 accumulation on the matrix cores,
// v_mfma_f32_32x32x2_f32 (gfx950: 64 cycles per instruction and SIMD = 64 FLOP/clk/SIMD = 157 TFLOP/s, the fp32 vector
// peak, with the VALU left free; the result is a k-ordered fp32 fma chain, i.e. what an fp32 reference computes).
//
// One kernel serves the three products of a Dense layer through the operands' storage order:
//   forward   y [M, N]  = x [M, K] . W[N, K]^T + b    A k-contiguous, B k-contiguous
//   dgrad     dx[M, K]  = dy[M, N] . W[N, K]          A k-contiguous (k = n), B column-contiguous
//   wgrad     dW[N, K]  = dy[M, N]^T . x[M, K]        both operands row-of-the-contraction major; the contraction runs
//                                                     over the M = batch x list_size rows, so it is cut into `splits`
//                                                     slabs (grid.y) reduced in a fixed order by a second launch
// Any M, N, K and any leading dimensions (widths that are not multiples of 8 are what the bf16 tower does not take):
// 16-byte loads where pointer and pitch allow it, bounds-checked scalar loads otherwise.
//
// Tiling: 128 x 128 x 16 per 256-thread workgroup, four waves of 64 x 64 (2 x 2 MFMA tiles of 32 x 32, 64 accumulator
// registers); operands staged k-major in LDS ([16][128 + 4] floats each: the fragment of a wave is one ds_read_b32 per
// lane over 32 consecutive floats, the +4 keeps the transposing stores of a k-contiguous operand conflict-free), two
// LDS stages with the next tile's global loads in flight during the 32 MFMAs of the current one.  Per k tile a wave
// issues 32 MFMAs (2 048 cycles) against 32 ds_read_b32: MFMA-issue bound.  Bounded to 128 registers so that FOUR
// workgroups share a CU (33 KB of LDS each): with three the counters showed the MFMA pipe 73 % busy and the waves parked
// a quarter of their time on the per-tile barrier / s_waitcnt (profiles/r03_gemm_f32_pmc.txt).  Workgroups are numbered
// so that the tiles of one row block of A (all N) are neighbours on one XCD and find A in that XCD's L2.
#include "common.h"
#include "../../include/tfr_hip.h"

using namespace tfr;

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128, kBN = 128, kBK = 16, kLdp = 132;

struct GemmF32Args {
  const float* A; long lda; int a_kc; int a_vec;
  const float* B; long ldb; int b_kc; int b_vec;
  float* C; long ldc; long slab;
  int M, N, K;
  const float* bias;
  int k_per_split, tiles_m, tiles_n, chunk, S;
};

// One operand tile: 128 rows (m of A, n of B) x 16 steps of the contraction, two float4 per thread.
//   KC: element (r, k) at X[r * ld + k]; thread -> row t / 4 (+ 64), four consecutive k
//   !KC: element (r, k) at X[k * ld + r]; thread -> k = t / 32 (+ 8), four consecutive rows
template <bool KC>
__device__ __forceinline__ void tile_load(const float* __restrict__ X, const long ld, const bool vec, const int r0,
                                          const int R, const int k0, const int kend, const int t, float4 (&v)[2]) {
  if (KC) {
    const int gk = k0 + (t & 3) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gr = r0 + (t >> 2) + 64 * p;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gr < R) {
        const float* src = X + (long)gr * ld + gk;
        if (vec && gk + 3 < kend) {
          x = *reinterpret_cast<const float4*>(src);
        } else {
          if (gk < kend) x.x = src[0];
          if (gk + 1 < kend) x.y = src[1];
          if (gk + 2 < kend) x.z = src[2];
          if (gk + 3 < kend) x.w = src[3];
        }
      }
      v[p] = x;
    }
  } else {
    const int gr = r0 + (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int gk = k0 + (t >> 5) + 8 * p;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gk < kend) {
        const float* src = X + (long)gk * ld + gr;
        if (vec && gr + 3 < R) {
          x = *reinterpret_cast<const float4*>(src);
        } else {
          if (gr < R) x.x = src[0];
          if (gr + 1 < R) x.y = src[1];
          if (gr + 2 < R) x.z = src[2];
          if (gr + 3 < R) x.w = src[3];
        }
      }
      v[p] = x;
    }
  }
}

template <bool KC>
__device__ __forceinline__ void tile_store(float* __restrict__ S, const int t, const float4 (&v)[2]) {
  if (KC) {
    const int kq = (t & 3) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int r = (t >> 2) + 64 * p;
      S[(kq + 0) * kLdp + r] = v[p].x;
      S[(kq + 1) * kLdp + r] = v[p].y;
      S[(kq + 2) * kLdp + r] = v[p].z;
      S[(kq + 3) * kLdp + r] = v[p].w;
    }
  } else {
    const int rq = (t & 31) * 4;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int k = (t >> 5) + 8 * p;
      *reinterpret_cast<float4*>(S + k * kLdp + rq) = v[p];
    }
  }
}

// (Measured and dropped, profiles/r03_gemm_f32.txt: pairing rows j and j + 8 in one MFMA so that the two half-waves read
// banks 32 apart, and reading the next group's fragments ahead of the current MFMAs -- each within +-5 % on every
// product: neither LDS bank conflicts nor fragment latency is what holds the loop at two thirds of the MFMA peak.)
template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 4) void gemm_f32_kernel(const GemmF32Args a) {
  __shared__ __attribute__((aligned(16))) float As[2][kBK * kLdp];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBK * kLdp];
  // Workgroup ids round-robin over the 8 XCDs (each with its own L2).
  //   one slab:  every XCD gets a contiguous run of tiles, n fastest: the tiles of one row block of A are neighbours
  //   S slabs (weight gradient): all tiles of slab z run on XCD z % 8, back to back: the [k slab] x 128 panels of both
  //              operands, which tiles_m + tiles_n tiles share, come from that XCD's L2 and from HBM once
  const int tiles = a.tiles_m * a.tiles_n;
  int logical, z;
  if (a.S == 1) {
    logical = (int)(blockIdx.x & 7) * a.chunk + (int)(blockIdx.x >> 3);
    z = 0;
    if (logical >= tiles) return;
  } else {
    const int q = (int)(blockIdx.x >> 3);
    z = (int)(blockIdx.x & 7) + 8 * (q / tiles);
    logical = q % tiles;
    if (z >= a.S) return;
  }
  const int tn = logical % a.tiles_n, tm = logical / a.tiles_n;
  const int m0 = tm * kBM, n0 = tn * kBN;
  const int kbeg = z * a.k_per_split;
  const int kend = min(a.K, kbeg + a.k_per_split);
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const int i = lane & 31, kh = lane >> 5;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][y][r] = 0.f;

  const int nt = (kend - kbeg + kBK - 1) / kBK;
  float4 ra[2], rb[2];
  if (nt > 0) {
    tile_load<AKC>(a.A, a.lda, a.a_vec != 0, m0, a.M, kbeg, kend, t, ra);
    tile_load<BKC>(a.B, a.ldb, a.b_vec != 0, n0, a.N, kbeg, kend, t, rb);
    tile_store<AKC>(As[0], t, ra);
    tile_store<BKC>(Bs[0], t, rb);
  }
  __syncthreads();
  for (int it = 0; it < nt; ++it) {
    const int cur = it & 1;
    const bool more = it + 1 < nt;
    if (more) {
      tile_load<AKC>(a.A, a.lda, a.a_vec != 0, m0, a.M, kbeg + (it + 1) * kBK, kend, t, ra);
      tile_load<BKC>(a.B, a.ldb, a.b_vec != 0, n0, a.N, kbeg + (it + 1) * kBK, kend, t, rb);
    }
    const float* as = As[cur] + wm + i;
    const float* bs = Bs[cur] + wn + i;
#pragma unroll
    for (int kk = 0; kk < kBK; kk += 2) {
      const int ko = (kk + kh) * kLdp;
      const float a0 = as[ko], a1 = as[ko + 32], b0 = bs[ko], b1 = bs[ko + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    if (more) {
      tile_store<AKC>(As[cur ^ 1], t, ra);
      tile_store<BKC>(Bs[cur ^ 1], t, rb);
    }
    __syncthreads();
  }

  // C/D fragment of the 32x32 MFMA: register r of lane l holds row (r & 3) + 8 (r >> 2) + 4 (l >> 5), column l & 31
  float* __restrict__ C = a.C + (long)z * a.slab;
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) {
      const int gn = n0 + wn + 32 * tj + i;
      if (gn >= a.N) continue;
      const float bv = a.bias ? a.bias[gn] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = m0 + wm + 32 * ti + (r & 3) + 8 * (r >> 2) + 4 * kh;
        if (gm < a.M) C[(long)gm * a.ldc + gn] = acc[ti][tj][r] + bv;
      }
    }
}

// out[m, n] = sum_z slab_z[m, n] (+ bias[n]), z in ascending order: the same bits on every run
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const float* __restrict__ ws, const int S, const long slab,
                                                              const int M, const int N, const float* __restrict__ bias,
                                                              float* __restrict__ C, const long ldc) {
  const long total = (long)M * N;
  for (long e = (long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long)gridDim.x * 256) {
    float s = 0.f;
    for (int z = 0; z < S; ++z) s += ws[(long)z * slab + e];
    const long m = e / N;
    const int n = (int)(e - m * N);
    C[m * ldc + n] = s + (bias ? bias[n] : 0.f);
  }
}

// column sums of X[M, N] (the bias gradient of a Dense layer), two deterministic stages: block b adds its run of rows
// for every column (threads along the columns: coalesced), then one block adds the T partial rows in ascending order
__global__ __launch_bounds__(256) void colsum_f32_stage1_kernel(const float* __restrict__ X, const long ldx, const int M,
                                                                const int N, const int rows_per, float* __restrict__ partial) {
  const int r0 = blockIdx.x * rows_per;
  const int r1 = min(M, r0 + rows_per);
  for (int n = threadIdx.x; n < N; n += 256) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* __restrict__ p = X + n;
    int r = r0;
    for (; r + 3 < r1; r += 4) {
      const float v0 = p[(long)r * ldx], v1 = p[(long)(r + 1) * ldx], v2 = p[(long)(r + 2) * ldx], v3 = p[(long)(r + 3) * ldx];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; r < r1; ++r) s0 += p[(long)r * ldx];
    partial[(long)blockIdx.x * N + n] = (s0 + s1) + (s2 + s3);
  }
}

__global__ __launch_bounds__(256) void colsum_f32_stage2_kernel(const float* __restrict__ partial, const int T, const int N,
                                                                float* __restrict__ out) {
  const int n = blockIdx.x * 256 + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += partial[(long)t * N + n];
  out[n] = s;
}

// ---- output_units <= 4 (the last Dense of the tower): HBM-bound on the [M, K] activations, no matrix cores ----
// y[m, n] = sum_k x[m, k] w[n, k] + b[n]: one wavefront per row, lanes along k
template <int NN>
__global__ __launch_bounds__(256) void dense_thin_fwd_kernel(const float* __restrict__ x, const long ldx,
                                                             const float* __restrict__ w, const long ldw,
                                                             const float* __restrict__ bias, float* __restrict__ y,
                                                             const long ldy, const int M, const int K) {
  const int lane = threadIdx.x & 63;
  const int nw = gridDim.x * 4;
  for (int m = blockIdx.x * 4 + (threadIdx.x >> 6); m < M; m += nw) {
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = 0.f;
    const float* __restrict__ xr = x + (long)m * ldx;
#pragma unroll 4
    for (int k = lane; k < K; k += 64) {
      const float xv = xr[k];
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[n] = fmaf(xv, w[(long)n * ldw + k], acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const float sum = wave_sum(acc[n]);
      if (lane == 0) y[(long)m * ldy + n] = sum + (bias ? bias[n] : 0.f);
    }
  }
}

// dW[n, k] = sum_m dy[m, n] x[m, k]: block b adds its run of rows for every k (threads along k), then the column-sum
// second stage adds the T partial [NN, K] slabs in ascending order
template <int NN>
__global__ __launch_bounds__(256) void dense_thin_wgrad_stage1_kernel(const float* __restrict__ dy, const long ldy,
                                                                      const float* __restrict__ x, const long ldx,
                                                                      const int M, const int K, const int rows_per,
                                                                      float* __restrict__ partial) {
  const int r0 = blockIdx.x * rows_per;
  const int r1 = min(M, r0 + rows_per);
  for (int k = threadIdx.x; k < K; k += 256) {
    float acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = 0.f;
    const float* __restrict__ p = x + k;
#pragma unroll 4
    for (int r = r0; r < r1; ++r) {
      const float xv = p[(long)r * ldx];
#pragma unroll
      for (int n = 0; n < NN; ++n) acc[n] = fmaf(dy[(long)r * ldy + n], xv, acc[n]);
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) partial[((long)blockIdx.x * NN + n) * K + k] = acc[n];
  }
}

// The same two reductions with 16-byte loads (columns a multiple of 4, pitch and pointer 16-byte aligned): the 256
// threads of a block cover C = min(N / 4, 256) column quads x 256 / C row phases, four rows of a phase in flight
// (64 B per lane), the row phases meet in LDS in ascending order.  partial[block][NN][N]; WEIGHTED: out[n, :] =
// sum_r w[r, n] X[r, :] (the weight gradient of output_units <= 4), else NN = 1 and out = sum_r X[r, :].
template <int NN, bool WEIGHTED>
__global__ __launch_bounds__(256) void rowsum_v4_stage1_kernel(const float* __restrict__ X, const long ldx,
                                                               const float* __restrict__ W, const long ldw, const int M,
                                                               const int N, const int rows_per,
                                                               float* __restrict__ partial) {
  __shared__ float4 red[256];
  const int N4 = N >> 2;
  const int C = N4 < 256 ? N4 : 256;
  const int RP = 256 / C;
  const int c = threadIdx.x % C, rp = threadIdx.x / C;
  const int r0 = blockIdx.x * rows_per;
  const int r1 = min(M, r0 + rows_per);
  for (int cb = 0; cb < N4; cb += C) {
    const int cq = cb + c;
    const bool active = rp < RP && cq < N4;
    float4 acc[NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) acc[n] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
      const float* __restrict__ p = X + 4 * cq;
      int r = r0 + rp;
      for (; r + 3 * RP < r1; r += 4 * RP) {
        float4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(r + u * RP) * ldx);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            const float wv = WEIGHTED ? W[(long)(r + u * RP) * ldw + n] : 1.0f;
            if (WEIGHTED) {
              acc[n].x = fmaf(wv, v[u].x, acc[n].x); acc[n].y = fmaf(wv, v[u].y, acc[n].y);
              acc[n].z = fmaf(wv, v[u].z, acc[n].z); acc[n].w = fmaf(wv, v[u].w, acc[n].w);
            } else {
              acc[n].x += v[u].x; acc[n].y += v[u].y; acc[n].z += v[u].z; acc[n].w += v[u].w;
            }
          }
      }
      for (; r < r1; r += RP) {
        const float4 v = *reinterpret_cast<const float4*>(p + (long)r * ldx);
#pragma unroll
        for (int n = 0; n < NN; ++n) {
          const float wv = WEIGHTED ? W[(long)r * ldw + n] : 1.0f;
          if (WEIGHTED) {
            acc[n].x = fmaf(wv, v.x, acc[n].x); acc[n].y = fmaf(wv, v.y, acc[n].y);
            acc[n].z = fmaf(wv, v.z, acc[n].z); acc[n].w = fmaf(wv, v.w, acc[n].w);
          } else {
            acc[n].x += v.x; acc[n].y += v.y; acc[n].z += v.z; acc[n].w += v.w;
          }
        }
      }
    }
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      red[threadIdx.x] = acc[n];
      __syncthreads();
      if (active && rp == 0) {
        float4 s4 = red[c];
        for (int q = 1; q < RP; ++q) {
          const float4 o = red[q * C + c];
          s4.x += o.x; s4.y += o.y; s4.z += o.z; s4.w += o.w;
        }
        *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * NN + n) * N + 4 * cq) = s4;
      }
      __syncthreads();
    }
  }
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

extern "C" int tfr_tower_colsum_rows(int M);

extern "C" int tfr_tower_gemm_f32_splits(int M, int N, int K) {
  // One resident round of workgroups: four 256-thread workgroups per CU (<= 128 registers, 33 KB of LDS each) = 1 024 on
  // the chip; a count that needs a partial second round wastes it (measured with three per CU: 1 024 equal workgroups
  // ran as a round and a third, 75 TFLOP/s against 86-89 with 768).  At least 16 k tiles per slab.
  if (M <= 0 || N <= 0 || K <= 0) return 1;
  if (M <= 4) return tfr_tower_colsum_rows(K);      // the thin weight gradient: one partial slab per run of rows
  const long tiles = (long)((M + kBM - 1) / kBM) * ((N + kBN - 1) / kBN);
  long s = 1024 / tiles;
  const long smax = (K + 16 * kBK - 1) / (16 * kBK);
  if (s > smax) s = smax;
  if (s > 256) s = 256;
  return s < 1 ? 1 : (int)s;
}

extern "C" int tfr_tower_gemm_f32(const float* A, long lda, int a_k_contiguous, const float* B, long ldb,
                                  int b_k_contiguous, float* C, long ldc, int M, int N, int K, const float* bias,
                                  int splits, float* workspace, void* stream) {
  if (M < 0 || N < 0 || K < 0 || splits < 1) return TFR_EINVAL;
  if (M == 0 || N == 0) return TFR_OK;
  if (!C || (K > 0 && (!A || !B)) || ldc < N) return TFR_EINVAL;
  if (K > 0) {
    if (lda < (a_k_contiguous ? K : M) || ldb < (b_k_contiguous ? K : N)) return TFR_EINVAL;
  }
  hipStream_t st = (hipStream_t)stream;
  if (a_k_contiguous && b_k_contiguous && N <= 4 && K > 0) {
    // the tower's last Dense: output_units <= 4
    const unsigned g = (unsigned)((M + 3) / 4 > 8192 ? 8192 : (M + 3) / 4);
#define TFR_THIN_FWD(NN) hipLaunchKernelGGL((dense_thin_fwd_kernel<NN>), dim3(g), dim3(256), 0, st, A, lda, B, ldb, bias, C, ldc, M, K)
    if (N == 1) TFR_THIN_FWD(1); else if (N == 2) TFR_THIN_FWD(2); else if (N == 3) TFR_THIN_FWD(3); else TFR_THIN_FWD(4);
#undef TFR_THIN_FWD
    return (int)hipGetLastError();
  }
  if (!a_k_contiguous && !b_k_contiguous && M <= 4 && K > 0 && workspace && ldc == N && !bias && splits <= 1024) {
    // its weight gradient: dW[M <= 4, N] = dy^T . x over K rows; `splits` partial slabs of [M, N]
    const int rows_per = (K + splits - 1) / splits;
    const int T = (K + rows_per - 1) / rows_per;
    const bool v4 = (N % 4 == 0) && (ldb % 4 == 0) && aligned16(B) && aligned16(workspace);
#define TFR_THIN_WG(NN) do { \
      if (v4) hipLaunchKernelGGL((rowsum_v4_stage1_kernel<NN, true>), dim3(T), dim3(256), 0, st, B, ldb, A, lda, K, N, rows_per, workspace); \
      else hipLaunchKernelGGL((dense_thin_wgrad_stage1_kernel<NN>), dim3(T), dim3(256), 0, st, A, lda, B, ldb, K, N, rows_per, workspace); } while (0)
    if (M == 1) TFR_THIN_WG(1); else if (M == 2) TFR_THIN_WG(2); else if (M == 3) TFR_THIN_WG(3); else TFR_THIN_WG(4);
#undef TFR_THIN_WG
    const int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
    hipLaunchKernelGGL(colsum_f32_stage2_kernel, dim3((M * N + 255) / 256), dim3(256), 0, st, workspace, T, M * N, C);
    return (int)hipGetLastError();
  }
  GemmF32Args a;
  a.A = A; a.lda = lda; a.a_kc = a_k_contiguous != 0; a.a_vec = A && aligned16(A) && (lda % 4 == 0);
  a.B = B; a.ldb = ldb; a.b_kc = b_k_contiguous != 0; a.b_vec = B && aligned16(B) && (ldb % 4 == 0);
  a.M = M; a.N = N; a.K = K;
  a.tiles_m = (M + kBM - 1) / kBM;
  a.tiles_n = (N + kBN - 1) / kBN;
  const long tiles = (long)a.tiles_m * a.tiles_n;
  if (tiles > (1L << 28)) return TFR_ETOOLARGE;
  a.chunk = (int)((tiles + 7) / 8);
  int kps = ((K + splits - 1) / splits + kBK - 1) / kBK * kBK;
  if (kps < kBK) kps = kBK;
  int S = K > 0 ? (K + kps - 1) / kps : 1;           // every slab non-empty
  if (S > 1 && !workspace) return TFR_EINVAL;
  a.k_per_split = kps;
  if (S > 1) { a.C = workspace; a.ldc = N; a.slab = (long)M * N; a.bias = nullptr; }
  else { a.C = C; a.ldc = ldc; a.slab = 0; a.bias = bias; }
  a.S = S;
  const long nblk = S == 1 ? (long)a.chunk * 8 : 8L * tiles * ((S + 7) / 8);
  if (nblk > (1L << 30)) return TFR_ETOOLARGE;
  const dim3 grid((unsigned)nblk), block(256);
  if (a.a_kc && a.b_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, true>), grid, block, 0, st, a);
  else if (a.a_kc) hipLaunchKernelGGL((gemm_f32_kernel<true, false>), grid, block, 0, st, a);
  else if (a.b_kc) hipLaunchKernelGGL((gemm_f32_kernel<false, true>), grid, block, 0, st, a);
  else hipLaunchKernelGGL((gemm_f32_kernel<false, false>), grid, block, 0, st, a);
  int rc = (int)hipGetLastError();
  if (rc != 0 || S == 1) return rc;
  const long total = (long)M * N;
  const unsigned rg = (unsigned)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(gemm_f32_reduce_kernel, dim3(rg), dim3(256), 0, st, workspace, S, (long)M * N, M, N, bias, C, ldc);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_colsum_rows(int M) {
  if (M <= 0) return 1;
  const int T = (M + 255) / 256;
  return T > 1024 ? 1024 : T;
}

extern "C" int tfr_tower_colsum_f32(const float* X, long ldx, int M, int N, float* partial, float* out, void* stream) {
  if (M < 0 || N < 0) return TFR_EINVAL;
  if (N == 0) return TFR_OK;
  if (!out || (M > 0 && (!X || !partial || ldx < N))) return TFR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const int T = tfr_tower_colsum_rows(M);
  const int rows_per = M > 0 ? (M + T - 1) / T : 1;
  const int Tu = M > 0 ? (M + rows_per - 1) / rows_per : 0;
  if (Tu > 0) {
    if ((N % 4 == 0) && (ldx % 4 == 0) && aligned16(X) && aligned16(partial))
      hipLaunchKernelGGL((rowsum_v4_stage1_kernel<1, false>), dim3(Tu), dim3(256), 0, st, X, ldx, (const float*)nullptr, 0L,
                         M, N, rows_per, partial);
    else
      hipLaunchKernelGGL(colsum_f32_stage1_kernel, dim3(Tu), dim3(256), 0, st, X, ldx, M, N, rows_per, partial);
    const int rc = (int)hipGetLastError();
    if (rc != 0) return rc;
  }
  hipLaunchKernelGGL(colsum_f32_stage2_kernel, dim3((N + 255) / 256), dim3(256), 0, st, partial, Tu, N, out);
  return (int)hipGetLastError();
}
