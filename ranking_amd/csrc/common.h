// Device-side helpers shared by the gfx950 ranking kernels.
//
// Execution model used by every kernel in this directory: ONE workgroup per
// ranked list, the list's rows resident in LDS for the whole kernel, 64-wide
// wavefront reductions (DPP/ds_swizzle shuffles) inside a wave and one LDS
// hop across waves.  Nothing O(L^2) ever leaves LDS/registers.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define TFR_WAVE 64
#define TFR_MAX_THREADS 1024
#define TFR_MAX_LIST 8192      /* = TFR_MAX_LIST_SIZE of include/tfr_hip.h (checked by tests/test_host_logic.py) */

#define TFR_OK 0
#define TFR_EINVAL (-1)
#define TFR_ETOOLARGE (-2)

namespace tfr {

__host__ __device__ inline int pow2_ceil(int n) {
  int p = 1;
  while (p < n) p <<= 1;
  return p;
}

// Launches of the workgroup kernels whose per-position arrays live in the caller's workspace (list sizes beyond the LDS
// range): one workgroup per workspace slot, the lists walked with a grid stride.
inline int big_slots(int B, size_t workspace_bytes, size_t slot_bytes) {
  size_t n = workspace_bytes / slot_bytes;
  if (n > (size_t)B) n = (size_t)B;
  if (n > 1024) n = 1024;
  return (int)n;
}

// Monotone map float -> uint32 (a < b  <=>  ord(a) < ord(b)); -0 is folded
// onto +0 so that tied zeros compare equal like tf.math.top_k treats them.
__device__ __forceinline__ uint32_t float_to_ordered(float f) {
  f = f + 0.0f;
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// ---------------------------------------------------------------- wave ops
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Block reductions: every thread gets the result.  `red` = LDS scratch of at
// least 17 floats.  Two barriers; safe to call back-to-back on the same scratch.
template <typename Op>
__device__ __forceinline__ float block_reduce(float v, float identity, float* red, Op op) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = op(v, __shfl_xor(v, o, 64));
  if (nw == 1) return v;
  __syncthreads();                 // previous users of `red` are done
  if (lane == 0) red[wid] = v;
  __syncthreads();
  float r = (lane < nw) ? red[lane] : identity;
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) r = op(r, __shfl_xor(r, o, 64));
  return __shfl(r, 0, 64);
}
struct OpSum { __device__ float operator()(float a, float b) const { return a + b; } };
struct OpMax { __device__ float operator()(float a, float b) const { return fmaxf(a, b); } };
struct OpMin { __device__ float operator()(float a, float b) const { return fminf(a, b); } };

__device__ __forceinline__ float block_sum(float v, float* red) { return block_reduce(v, 0.f, red, OpSum()); }
__device__ __forceinline__ float block_max(float v, float* red) { return block_reduce(v, -INFINITY, red, OpMax()); }
__device__ __forceinline__ float block_min(float v, float* red) { return block_reduce(v, INFINITY, red, OpMin()); }

// Fixed-order fp32 sum shared bit-for-bit with oracle.tfr_ref.tree_sum:
// t[0..P) (P power of two, zero padded) is folded upper half onto lower half.
// On return t[0] holds the sum (all threads must call; ends with a barrier).
__device__ __forceinline__ void block_tree_sum(float* t, int P) {
  for (int h = P >> 1; h >= 1; h >>= 1) {
    __syncthreads();
    for (int i = threadIdx.x; i < h; i += blockDim.x) t[i] = t[i] + t[i + h];
  }
  __syncthreads();
}

// Bitonic sort, DESCENDING, of P (power of two) keys resident in LDS.
// All threads of the block must call.  Ends with a barrier.
template <typename K>
__device__ __forceinline__ void block_bitonic_sort_desc(K* keys, int P) {
  const int half = P >> 1;
  for (int k = 2; k <= P; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (int t = threadIdx.x; t < half; t += blockDim.x) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const K a = keys[i], b = keys[p];
        const bool down = ((i & k) == 0);          // this run sorts descending
        const bool swap = down ? (a < b) : (a > b);
        if (swap) { keys[i] = b; keys[p] = a; }
      }
    }
  }
  __syncthreads();
}

// Sort key for "valid first, score descending, tie-break ascending, then
// index" (utils.py:84-164 with the deterministic tie rule).
//   bit 63      valid
//   bits 62..31 ordered(score)        (0 for invalid entries)
//   bits 30..16 0x7fff - tiebreak     (tiebreak < 32768; default = 0)
//   bits 15..0  0xffff - index        (index < 65536)
// Sorting the packed keys DESCENDING yields the required order and the low 16
// bits recover the item index.
__device__ __forceinline__ uint64_t make_sort_key(bool valid, float score, int tiebreak, int index) {
  const uint64_t s = valid ? (uint64_t)float_to_ordered(score) : 0ull;
  return ((uint64_t)(valid ? 1 : 0) << 63) | (s << 31) |
         ((uint64_t)(0x7fff - (tiebreak & 0x7fff)) << 16) | (uint64_t)(0xffff - (index & 0xffff));
}
__device__ __forceinline__ int sort_key_index(uint64_t key) { return 0xffff - (int)(key & 0xffffull); }

// Random tie order (utils.py:84-112 _get_shuffle_indices: the reference permutes a list at random before its stable sort,
// so equal keys come out in a uniformly random order): a 15-bit counter-based hash of (seed, list, item) for the
// `tiebreak` field of make_sort_key -- the smaller value sorts first among equal scores, the index decides the (2^-15)
// collisions.  seed 0 = no shuffle (index order).  Restated in ranking_amd/_ops.py tie_keys for the tests.
__device__ __forceinline__ int tie_key15(uint32_t seed, uint32_t b, uint32_t i) {
  if (seed == 0u) return 0;
  uint32_t h = b * 0x9E3779B1u + i * 0x85EBCA77u + seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return (int)(h >> 17);
}

// 2^l - 1 (keras/utils.py:79-92): exact for integer grades (the hardware exp2 is
// only 1-ulp accurate, which would break bit-exact NDCG on integer labels).
__device__ __forceinline__ float gain_pow2m1(float l) {
  const float r = rintf(l);
  const float p = (r == l && fabsf(l) < 120.0f) ? ldexpf(1.0f, (int)r) : exp2f(l);
  return p - 1.0f;
}

// exp(t) for a fp32 t given as an exact double-float (t_hi + t_lo), to ~1ulp:
// used once per ITEM (never per pair) so that the factorised sigmoid
// 1/(1 + E_i*F_j) keeps full fp32 accuracy even when |x - m| is large.
__device__ __forceinline__ float exp_df(float t_hi, float t_lo) {
  const float LOG2E_HI = 1.44269502162933349609375f;      // fp32(log2 e)
  const float LOG2E_LO = 1.92596299112661746e-08f;        // log2 e - LOG2E_HI
  const float y_hi = t_hi * LOG2E_HI;
  float y_lo = __builtin_fmaf(t_hi, LOG2E_HI, -y_hi);
  y_lo = __builtin_fmaf(t_hi, LOG2E_LO, y_lo);
  y_lo = __builtin_fmaf(t_lo, LOG2E_HI, y_lo);
  const float n = rintf(y_hi);
  const float f = (y_hi - n) + y_lo;
  return ldexpf(exp2f(f), (int)n);
}


// The same with the single-instruction transcendentals (v_exp_f32 / v_ldexp_f32, 1 ulp): per-item set-up of the
// factorised pair kernels.
__device__ __forceinline__ float exp_df_hw(float t_hi, float t_lo) {
  const float LOG2E_HI = 1.44269502162933349609375f;
  const float LOG2E_LO = 1.92596299112661746e-08f;
  const float y_hi = t_hi * LOG2E_HI;
  float y_lo = __builtin_fmaf(t_hi, LOG2E_HI, -y_hi);
  y_lo = __builtin_fmaf(t_hi, LOG2E_LO, y_lo);
  y_lo = __builtin_fmaf(t_lo, LOG2E_HI, y_lo);
  const float n = rintf(y_hi);
  const float f = (y_hi - n) + y_lo;
  return __builtin_amdgcn_ldexpf(__builtin_amdgcn_exp2f(f), (int)n);
}

// value of lane (lane ^ j), j a power of two < 64.  j = 1, 2, 4, 8 stay inside a 16-lane DPP row: one or two
// v_mov_b32_dpp per dword (quad_perm for 1 and 2, row_ror:8 for 8, row_shr:4 + row_shl:4 on complementary banks for
// 4) instead of a ds_bpermute round trip through the LDS crossbar -- 26 of the 33 cross-lane stages of a 256-key
// bitonic sort.  j = 16, 32 cross rows: ds_bpermute.  (j is a literal after unrolling: the switch folds.)
__device__ __forceinline__ uint32_t wave_xor_exchange(uint32_t v, int j) {
  const int x = (int)v;
  switch (j) {
    case 1: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0xB1, 0xf, 0xf, false);      // quad_perm:[1,0,3,2]
    case 2: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x4E, 0xf, 0xf, false);      // quad_perm:[2,3,0,1]
    case 4: {
      const int t = __builtin_amdgcn_update_dpp(x, x, 0x114, 0xf, 0xa, false);              // row_shr:4 -> banks 1, 3
      return (uint32_t)__builtin_amdgcn_update_dpp(t, x, 0x104, 0xf, 0x5, false);           // row_shl:4 -> banks 0, 2
    }
    case 8: return (uint32_t)__builtin_amdgcn_update_dpp(x, x, 0x128, 0xf, 0xf, false);     // row_ror:8
    default: return (uint32_t)__shfl_xor(x, j, 64);
  }
}
__device__ __forceinline__ uint64_t wave_xor_exchange(uint64_t v, int j) {
  const uint32_t lo = wave_xor_exchange((uint32_t)v, j), hi = wave_xor_exchange((uint32_t)(v >> 32), j);
  return ((uint64_t)hi << 32) | lo;
}

// ---------------------------------------------------------------- wave-per-list helpers
// Bitonic sort, DESCENDING, of 64*IPL keys held in registers: element e = lane + 64*r
// lives in a[r] of `lane`.  No LDS, no barriers (cross-lane exchanges are wave shuffles).
template <typename K, int IPL>
__device__ __forceinline__ void wave_bitonic_sort_desc(K (&a)[IPL], int lane) {
  constexpr int N = 64 * IPL;
#pragma unroll
  for (int kk = 2; kk <= N; kk <<= 1) {
#pragma unroll
    for (int j = kk >> 1; j > 0; j >>= 1) {
      if (j >= 64) {
        const int jr = j >> 6;
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          if ((r & jr) == 0) {
            const int r2 = r | jr;
            const bool desc = (((r << 6) & kk) == 0);
            const K hi = a[r] > a[r2] ? a[r] : a[r2];
            const K lo = a[r] > a[r2] ? a[r2] : a[r];
            a[r] = desc ? hi : lo;
            a[r2] = desc ? lo : hi;
          }
        }
      } else {
#pragma unroll
        for (int r = 0; r < IPL; ++r) {
          const K p = wave_xor_exchange(a[r], j);
          const int e = lane | (r << 6);
          const bool desc = ((e & kk) == 0);
          const bool lower = ((lane & j) == 0);
          const K mx = a[r] > p ? a[r] : p;
          const K mn = a[r] > p ? p : a[r];
          a[r] = (lower == desc) ? mx : mn;
        }
      }
    }
  }
}

// tree_sum (see block_tree_sum) of P = pow2 >= L values laid out e = lane + 64*r, zero padded:
// bit-identical pairing to the LDS version.  Every lane returns the sum.
template <int IPL>
__device__ __forceinline__ float wave_tree_sum(float (&t)[IPL], int P) {
#pragma unroll
  for (int hr = IPL >> 1; hr >= 1; hr >>= 1) {        // h = 64 * hr >= 64: register-level folds
    if (64 * hr * 2 <= P) {
#pragma unroll
      for (int r = 0; r < hr; ++r) t[r] = t[r] + t[r + hr];
    }
  }
  // lane l adds lane l + h for h = 32, 16, 8, 4, 2, 1 (lanes >= h hold don't-care values) -- the same pairs as the
  // __shfl_down form of rounds 1-3, without its seven dependent ds_bpermute round trips through the LDS crossbar
  // (13 tree sums per list in the NDCG metric: ~90 of them in a row): the two cross-row levels are gfx950's
  // v_permlane32_swap / v_permlane16_swap (the "source" result holds lanes [32, 64) in lanes [0, 32), resp. the odd
  // rows in the even ones), the four levels inside a 16-lane row are DPP row_shl, the broadcast is a v_readlane.
  // (inline asm, both operands tied: with the builtin and the same value in both operands hipcc 7.2 picked the
  // DESTINATION result where the source result was asked for -- every level added a lane to itself; tools/permlane_probe.hip.
  // s_nop 1 = the two wait states between a VALU write of an operand and the swap.)
  float v = t[0];
  {
    float a = v, o = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(o));    // o: lanes [0, 32) <- v[32, 64)
    if (64 <= P) v = v + o;
  }
  {
    float a = v, o = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(o));    // o: even rows <- the odd rows of v
    if (32 <= P) v = v + o;
  }
#define TFR_ROW_SHL(x, n) __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, (float)(x)), 0x100 + (n), 0xf, 0xf, true))
  { const float o = TFR_ROW_SHL(v, 8); if (16 <= P) v = v + o; }
  { const float o = TFR_ROW_SHL(v, 4); if (8 <= P) v = v + o; }
  { const float o = TFR_ROW_SHL(v, 2); if (4 <= P) v = v + o; }
  { const float o = TFR_ROW_SHL(v, 1); if (2 <= P) v = v + o; }
#undef TFR_ROW_SHL
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
}

// acc + (acc of lane ^ 1) [+ lanes ^ 2]: the C = 2 / 4 lanes of a row of the pair sweeps, on the DPP network (quad_perm)
// instead of a ds_bpermute round trip per pass.
__device__ __forceinline__ float lanes_sum_c(float acc, int C) {
  if (C >= 2) acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
  if (C >= 4) acc += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, acc), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
  for (int o = 4; o < C; o <<= 1) acc += __shfl_xor(acc, o, 64);
  return acc;
}

// Wave-wide reductions on the DPP network (no LDS traffic, unlike __shfl_xor which lowers to
// ds_bpermute): 4 row_shr steps inside each 16-lane row, row_bcast:15 / row_bcast:31 across
// rows, result read from lane 63 into an SGPR (wave-uniform return value).
#define TFR_DPP_F(old, src, ctrl, rmask, bmask, bc)                                                   \
  __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, (float)(old)),        \
                                                        __builtin_bit_cast(int, (float)(src)), ctrl,  \
                                                        rmask, bmask, bc))
__device__ __forceinline__ float wave_max_u(float v) {
  const float ninf = -INFINITY;
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x111, 0xf, 0xf, false));
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x112, 0xf, 0xf, false));
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x114, 0xf, 0xf, false));
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x118, 0xf, 0xf, false));
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x142, 0xa, 0xf, false));
  v = fmaxf(v, TFR_DPP_F(ninf, v, 0x143, 0xc, 0xf, false));
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
__device__ __forceinline__ float wave_min_u(float v) { return -wave_max_u(-v); }
// wave-wide OR on the DPP network (wave-uniform result, read from lane 63)
__device__ __forceinline__ unsigned wave_or_u(unsigned v) {
  int x = (int)v;
  x |= __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x |= __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return (unsigned)__builtin_amdgcn_readlane(x, 63);
}
__device__ __forceinline__ float wave_sum_u(float v) {
  v += TFR_DPP_F(0.f, v, 0x111, 0xf, 0xf, true);
  v += TFR_DPP_F(0.f, v, 0x112, 0xf, 0xf, true);
  v += TFR_DPP_F(0.f, v, 0x114, 0xf, 0xf, true);
  v += TFR_DPP_F(0.f, v, 0x118, 0xf, 0xf, true);
  v += TFR_DPP_F(0.f, v, 0x142, 0xa, 0xf, false);
  v += TFR_DPP_F(0.f, v, 0x143, 0xc, 0xf, false);
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}

// LDS hand-off inside ONE wavefront (its lanes exchange data through the wave's private LDS
// slice): DS operations of a wave complete in order, so draining the counter is enough.
#define WAVE_LDS_SYNC() do { __builtin_amdgcn_wave_barrier(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); } while (0)

// 0-based rank (score descending, ties by lower compact index) of the n compact scores XS[0 .. n) of ONE wavefront's
// list, for up to IPL row chunks of 64 compact items AT ONCE: every float4 of column scores is read once (not once per
// chunk) and compared against the NC chunk registers, two column groups per trip with both loads issued first -- the
// one-chunk-at-a-time form of round 2 waited one LDS round trip per 8 VALU instructions (17 % of the LambdaRank kernel).
// Items with tied scores end up with the SAME count: an occupancy table (OCC, n ints of scratch) finds them -- rare --
// and only those add the equal scores in front of them.  XS is padded with -inf up to a multiple of 4 (+4).
template <int NC>
__device__ __forceinline__ void wave_rank_chunks(const float* XS, int n, int lane, int* RKS, int* OCC, int p0 = 0) {
  const float4* X4 = reinterpret_cast<const float4*>(XS);
  const int n4 = (n + 3) >> 2;
  float xi[NC];
  int cnt[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) { const int p = p0 + lane + 64 * k; xi[k] = p < n ? XS[p] : INFINITY; cnt[k] = 0; }
  int gq = 0;
  for (; gq + 2 <= n4; gq += 2) {
    const float4 xa = X4[gq], xb = X4[gq + 1];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      cnt[k] += (xa.x > xi[k]) ? 1 : 0; cnt[k] += (xa.y > xi[k]) ? 1 : 0;
      cnt[k] += (xa.z > xi[k]) ? 1 : 0; cnt[k] += (xa.w > xi[k]) ? 1 : 0;
      cnt[k] += (xb.x > xi[k]) ? 1 : 0; cnt[k] += (xb.y > xi[k]) ? 1 : 0;
      cnt[k] += (xb.z > xi[k]) ? 1 : 0; cnt[k] += (xb.w > xi[k]) ? 1 : 0;
    }
  }
  if (gq < n4) {
    const float4 xa = X4[gq];
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      cnt[k] += (xa.x > xi[k]) ? 1 : 0; cnt[k] += (xa.y > xi[k]) ? 1 : 0;
      cnt[k] += (xa.z > xi[k]) ? 1 : 0; cnt[k] += (xa.w > xi[k]) ? 1 : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int p = p0 + lane + 64 * k;
    if (p < n) { RKS[p] = cnt[k]; atomicAdd(&OCC[cnt[k]], 1); }
  }
}

__device__ __forceinline__ void wave_rank_by_count(const float* XS, int n, int lane, int* RKS, int* OCC) {
  for (int p = lane; p < n; p += 64) OCC[p] = 0;
  WAVE_LDS_SYNC();
  if (n <= 64) wave_rank_chunks<1>(XS, n, lane, RKS, OCC);
  else if (n <= 128) wave_rank_chunks<2>(XS, n, lane, RKS, OCC);
  else if (n <= 192) wave_rank_chunks<3>(XS, n, lane, RKS, OCC);
  else if (n <= 256) wave_rank_chunks<4>(XS, n, lane, RKS, OCC);
  else if (n <= 384) wave_rank_chunks<6>(XS, n, lane, RKS, OCC);      // (the NDCG counting kernel serves lists up to 512)
  else if (n <= 512) wave_rank_chunks<8>(XS, n, lane, RKS, OCC);
  else for (int p0 = 0; p0 < n; p0 += 512) wave_rank_chunks<8>(XS, n, lane, RKS, OCC, p0);     // any n: 512 row items per sweep of the columns
  WAVE_LDS_SYNC();
  for (int q0 = 0; q0 < n; q0 += 64) {
    const int p = q0 + lane;
    const bool tie = p < n && OCC[RKS[p]] > 1;
    if (__ballot(tie)) {                                     // wave-uniform: some item of this chunk shares its score
      if (tie) {
        const float xi = XS[p];
        int cnt = RKS[p];
        for (int j = 0; j < p; ++j) cnt += (XS[j] == xi) ? 1 : 0;
        RKS[p] = cnt;                                        // (other lanes read OCC at their OWN first count only)
      }
    }
  }
  WAVE_LDS_SYNC();
}

// inclusive prefix sum over the 64 lanes on the DPP network (the steps of wave_sum_u, every lane keeps its prefix)
__device__ __forceinline__ int wave_scan_incl_i(int x) {
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, true);
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);
  return x;
}

// The ranks of wave_rank_by_count (same result, integer for integer) from a 64-bucket partition of the score range
// instead of all n^2 / 64 compares per lane (round 4): bucket(x) = min(63, int((max - x) * 64 / (max - min))) is monotone
// (a higher score never lands in a later bucket, equal scores share one), so rank = (items in earlier buckets) + (items
// of the OWN bucket with a higher score) -- a histogram (LDS atomics), one wave scan, a scatter into bucket order and a
// scan of at most `hmax` = the fullest bucket's entries per item.  The scan needs no bound check: whatever follows a
// bucket in BX belongs to later buckets (strictly lower scores) or to the -inf padding, and never counts.  Ties leave
// equal counts exactly like the counting sweep, and the same occupancy fix-up orders them by compact index.
// Declines (returns false, wave-uniform; nothing written that the caller reads) when the range is empty / not finite or
// a bucket holds more than kRankBucketMax items (heavy ties, an outlier that squeezes the rest into one bucket): the
// caller then runs wave_rank_by_count.  n <= 64 * NC; BX: n + kRankBucketMax floats, HB: 192 ints, OCC: n ints of scratch.
constexpr int kRankBucketMax = 32;
template <int NC>
__device__ __forceinline__ bool wave_rank_by_bucket(const float* XS, int n, int lane, int* RKS, int* OCC, float* BX,
                                                    int* HB, bool have_range = false, float mn_in = 0.0f,
                                                    float mx_in = 0.0f) {
  // have_range: the caller already holds the (wave-uniform) minimum and maximum of XS[0 .. n) -- two wave reductions less
  float x[NC];
  bool in[NC];
  float mn = INFINITY, mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int p = lane + 64 * k;
    in[k] = p < n;
    x[k] = in[k] ? XS[p] : 0.0f;
    if (!have_range && in[k]) { mn = fminf(mn, x[k]); mx = fmaxf(mx, x[k]); }
  }
  if (have_range) { mn = mn_in; mx = mx_in; }
  else { mn = wave_min_u(mn); mx = wave_max_u(mx); }
  const float range = mx - mn;
  const float scale = 64.0f / range;
  if (!(range > 0.0f) || !(range < INFINITY) || !(scale < INFINITY)) return false;
  int* H = HB; int* FILL = HB + 64; int* START = HB + 128;
  H[lane] = 0; FILL[lane] = 0;
  for (int p = lane; p < n; p += 64) OCC[p] = 0;
  for (int p = n + lane; p < n + kRankBucketMax; p += 64) BX[p] = -INFINITY;
  WAVE_LDS_SYNC();
  int bk[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    int bi = (int)(in[k] ? (mx - x[k]) * scale : 0.0f);      // (mx >= x: in [0, 64])
    bi = bi > 63 ? 63 : bi;
    bk[k] = bi;
    if (in[k]) atomicAdd(&H[bi], 1);
  }
  WAVE_LDS_SYNC();
  const int h = H[lane];
  const int hmax = (int)wave_max_u((float)h);               // (<= 512: exact in fp32)
  if (hmax > kRankBucketMax) return false;
  START[lane] = wave_scan_incl_i(h) - h;
  WAVE_LDS_SYNC();
  int st[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    st[k] = 0;
    if (in[k]) {
      st[k] = START[bk[k]];
      const int slot = atomicAdd(&FILL[bk[k]], 1);           // (any order inside a bucket: the scan below compares values)
      BX[st[k] + slot] = x[k];
    }
  }
  WAVE_LDS_SYNC();
  int cnt[NC];
#pragma unroll
  for (int k = 0; k < NC; ++k) cnt[k] = st[k];
  for (int j = 0; j < hmax; j += 4) {                        // (uniform trip count; reads past a bucket never count)
#pragma unroll
    for (int k = 0; k < NC; ++k) {
      const float* q = BX + st[k] + j;                        // st + j + 3 < n + kRankBucketMax
      const float a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3];
      cnt[k] += (a0 > x[k]) ? 1 : 0; cnt[k] += (a1 > x[k]) ? 1 : 0;
      cnt[k] += (a2 > x[k]) ? 1 : 0; cnt[k] += (a3 > x[k]) ? 1 : 0;
    }
  }
#pragma unroll
  for (int k = 0; k < NC; ++k) {
    const int p = lane + 64 * k;
    if (in[k]) { RKS[p] = cnt[k]; atomicAdd(&OCC[cnt[k]], 1); }
  }
  WAVE_LDS_SYNC();
  for (int q0 = 0; q0 < n; q0 += 64) {                        // ties: the fix-up of wave_rank_by_count
    const int p = q0 + lane;
    const bool tie = p < n && OCC[RKS[p]] > 1;
    if (__ballot(tie)) {
      if (tie) {
        const float xi = XS[p];
        int c = RKS[p];
        for (int j = 0; j < p; ++j) c += (XS[j] == xi) ? 1 : 0;
        RKS[p] = c;
      }
    }
  }
  WAVE_LDS_SYNC();
  return true;
}

// out[0] = sum_i vec[i] * w[i] (w nullable) over the B per-list values of a launch, without a launch of its own: the
// scalar a reduced loss returns.  Called by ONE wavefront of every workgroup; lane 0 stores the workgroup's own entry
// vec[b] = value HERE.  Two levels, both in a FIXED order whoever computes them (the result does not depend on the order
// in which the workgroups finish): entry i belongs to group i % 64; the wave that completes a group (a ticket per
// group) adds the group's entries (lane l takes the l-th, (l + 64)-th, ... entry of the group, then the wave tree) into
// partial[group]; the wave that completes the last group adds the 64 partials.  `st` = kGridSumStateInts uint32 in
// device memory, zero before the first launch, left zero.
// Why two levels: 16 384 fetch-adds on ONE address took 90 us (same-address atomics serialise at the memory side), and
// one wave adding 16 384 entries is 16 dependent round trips.  Coherence without fences: an agent-scope RELEASE fence on
// gfx950 is a write-back of the XCD's whole L2 (buffer_wbl2) with 13 MB of gradient rows dirty in it -- 0.2 ms over a
// launch (measured).  Instead every shared value is stored with an agent-scope atomic store (written through:
// global_store ... sc1), the wave waits for that store (s_waitcnt vmcnt(0)) before it takes its ticket (a device-scope
// atomic, performed at the coherence point), and readers use agent-scope atomic loads (global_load ... sc1).
constexpr int kGridSumGroups = 64;
constexpr int kGridSumStateInts = 2 * kGridSumGroups + 1;    // [0, 64) group tickets, [64] top ticket, [65, 129) partials
__device__ __forceinline__ void grid_weighted_sum_last(float* vec, int b, float value, const float* __restrict__ w, int B,
                                                       float* out, unsigned int* st, int lane) {
  const int j = b & (kGridSumGroups - 1);
  const int cnt = (B - j + kGridSumGroups - 1) / kGridSumGroups;      // entries of group j
  unsigned int t = 0;
  if (lane == 0) {
    __hip_atomic_store(vec + b, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t = __hip_atomic_fetch_add(st + j, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
  if ((int)t != cnt - 1) return;
  // The loads below are ordered after the ticket by (a) the control dependency on its RETURNED value and (b) this
  // compiler barrier (nothing may be hoisted above the RMW).  Hardware assumption, stated: on gfx950 an agent-scope
  // atomic load (global_load ... sc1) is served at the coherence point, so once the ticket RMW -- performed there too,
  // after every contributor's written-through store was acknowledged (its s_waitcnt vmcnt(0)) -- has returned, the
  // entries are visible; `w` is only ever a kernel INPUT (never written by the launch), so its plain loads need no order.
  asm volatile("" ::: "memory");
  float acc = 0.f;
  for (int q = lane; q < cnt; q += 64) {
    const int i = j + kGridSumGroups * q;
    acc = __builtin_fmaf(__hip_atomic_load(vec + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), w ? w[i] : 1.0f, acc);
  }
  acc = wave_sum_u(acc);
  const int groups = B < kGridSumGroups ? B : kGridSumGroups;
  unsigned int t2 = 0;
  if (lane == 0) {
    __hip_atomic_store(reinterpret_cast<float*>(st + kGridSumGroups + 1 + j), acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(st + j, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    t2 = __hip_atomic_fetch_add(st + kGridSumGroups, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  t2 = (unsigned int)__builtin_amdgcn_readfirstlane((int)t2);
  if ((int)t2 != groups - 1) return;
  asm volatile("" ::: "memory");
  float p = (lane < groups) ? __hip_atomic_load(reinterpret_cast<float*>(st + kGridSumGroups + 1 + lane), __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT) : 0.0f;
  p = wave_sum_u(p);
  if (lane == 0) {
    out[0] = p;
    __hip_atomic_store(st + kGridSumGroups, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// The same reduction for any launch (round 5): `vec` is a scratch of >= n floats, entry idx = one contributor's value (a
// list, or a wavefront's own fixed-order partial over the lists it walked), n contributors in total, every one of them
// calls this exactly once with its whole wavefront converged.  out == nullptr: no sum requested.
struct GridSum { float* out; float* vec; unsigned int* st; int n; };
__device__ __forceinline__ void grid_sum_contribute(const GridSum& s, int idx, float value, int lane) {
  grid_weighted_sum_last(s.vec, idx, value, nullptr, s.n, s.out, s.st, lane);
}

// sum_p sorted_desc(g)[p] * table[p] for NON-NEGATIVE g (element e = lane + 64*r, zero beyond
// L) without sorting: the sorted sequence is a few runs of equal values, so repeatedly take the
// largest remaining value v, its multiplicity c, and add v * sum(table[pos .. pos + c)).  Graded
// relevance labels have a handful of distinct values; returns false (result unusable) when
// there are more than `max_runs` of them -- the caller then sorts.
template <int IPL>
__device__ __forceinline__ bool wave_sorted_dot_runs(const float (&g)[IPL], const float (&table)[IPL], int lane,
                                                      int L, int max_runs, float& out) {
  float rem[IPL];
#pragma unroll
  for (int r = 0; r < IPL; ++r) rem[r] = (lane + 64 * r < L) ? g[r] : -1.0f;
  float acc = 0.f;
  int pos = 0;
  for (int it = 0; it < max_runs; ++it) {
    float m = rem[0];
#pragma unroll
    for (int r = 1; r < IPL; ++r) m = fmaxf(m, rem[r]);
    const float v = wave_max_u(m);
    if (v < 0.0f) { out = acc; return true; }
    int c = 0;
    float part = 0.f;
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const bool hit = rem[r] == v;
      c += __popcll(__ballot(hit));
      rem[r] = hit ? -1.0f : rem[r];
    }
#pragma unroll
    for (int r = 0; r < IPL; ++r) {
      const int e = lane + 64 * r;
      part += (e >= pos && e < pos + c) ? table[r] : 0.0f;
    }
    acc = __builtin_fmaf(v, wave_sum_u(part), acc);
    pos += c;
  }
  float m = rem[0];
#pragma unroll
  for (int r = 1; r < IPL; ++r) m = fmaxf(m, rem[r]);
  if (wave_max_u(m) < 0.0f) { out = acc; return true; }
  return false;
}

}  // namespace tfr
