// Feed-forward scorer tower (Dense -> BatchNorm -> activation blocks) for gfx950:
// bf16 MFMA GEMMs with the BatchNorm / ReLU / bias / cast work fused into the
// GEMM prologues and epilogues, forward and backward.
//
// Reference behaviour restated: keras/layers.py:26-77 (create_tower: Dense,
// BatchNormalization(momentum), Activation, Dropout per hidden layer, then
// Dense(output_units)) driven by keras/model.py:780-817 (DNNScorer over the
// flattened [B*L, F] matrix).  The reference runs these as separate fp32 TF ops
// (MatMul, BiasAdd, FusedBatchNormV3, Relu); SURVEY.md 8a row a21.
//
// Design (MI355X).  A hidden layer is ONE launch:
//     Z_l = act_{l-1}(Z_{l-1}) . W_l^T + b_l            (bf16 operands, fp32 accumulate)
// where act_{l-1}(z) = relu(z * scale + shift) is the previous layer's BatchNorm +
// ReLU applied WHILE the A tile is staged into LDS (the normalised activation is
// never written to HBM), and the epilogue adds the bias, writes Z_l as bf16 and
// emits the per-column partial sums (sum z, sum z^2) the NEXT BatchNorm needs, so a
// [M, 512] layer costs one read + one write of an [M, 512] bf16 matrix.  At
// M = 409600, N = K = 512 that is 0.84 GB against 0.215 TFLOP: 256 FLOP/B, below
// the machine balance (2.5 PFLOP/s / ~6 TB/s = ~420 FLOP/B), i.e. the layer is
// HBM-bound once the MFMA pipe runs above ~55 % -- which is why nothing else is
// allowed to touch HBM.
//
// GEMM kernel: 128x128 output tile, BK = 64, 4 wavefronts (2x2, 64x64 each as 4x4
// v_mfma_f32_16x16x32_bf16 fragments), LDS double buffer (64 KB -> 2 workgroups per
// CU), register-staged global->LDS copy (the prologue transform needs the registers
// anyway) issued one tile ahead, XOR-swizzled 16-byte chunks so that the
// ds_read_b128 fragment reads are (nearly) conflict free, and an XCD-aware
// workgroup -> tile map that keeps the N-tiles of one M-tile on one XCD (its A tile
// is then served by that XCD's L2).  The MFMA operands are swapped (weights as
// "A", activations as "B") so that a lane ends up with 4 consecutive output
// COLUMNS of one row: 8-byte bf16 stores and a 16-lane DPP reduction for the
// column statistics.
#include "common.h"
#include "../../include/tfr_hip.h"

#include <stdio.h>
#include <type_traits>
#include <stdlib.h>

using namespace tfr;

namespace {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;          // 16 KB per operand tile

__device__ __forceinline__ float bf16_lo(uint32_t u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(uint32_t u) { return __uint_as_float(u & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack_bf16(float a, float b) {     // RNE (v_cvt_pk_bf16_f32)
  f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

// 16-byte global -> LDS copy without VGPR staging (global_load_lds_dwordx4): the LDS destination is the
// wave-uniform `lds_wave_base` + 16 * lane, the global source is per lane.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Dropout (keras/layers.py:72-73: tf.keras.layers.Dropout after every hidden activation).  The
// keep mask is a counter-based hash of (layer seed, row, column word) -- the same function in the
// forward prologues and in the backward kernels, so nothing is stored.  A 32-bit hash word serves 2^lge
// consecutive columns with a field of 32 >> lge bits each: an element is kept iff its field >= thr (the rate in
// field units) and is then scaled by 1 / (1 - rate).  Round 3: the field is as narrow as the rate allows EXACTLY,
// (rate 0.5 -- the reference default -- needs ONE bit: a word serves 32 columns; 0.25 two bits; multiples of 1 / 256
// eight).  Round 5: any other rate (0.1, 0.3 ...) takes 16-bit fields, i.e. the rate to 1 / 65 536 with the scale following
// the threshold, instead of being rounded to a multiple of 1 / 256 (0.1 was 0.1016): twice the hashes of the 8-bit form.  Round 2 hashed once per column PAIR (16-bit fields): the two 32-bit multiplies of the hash are
// quarter-rate instructions and sat in every GEMM prologue -- the hidden-layer forward GEMM of BASELINE config 5
// took 41.8 us with Dropout against 27.7 us without.  The callers walk aligned runs of 4 / 8 columns: one hash per
// run (two for an 8-run at 8-bit fields).
struct Drop { uint32_t seed; uint32_t thr; float scale; uint32_t lge; const uint32_t* sp; };     // lge = 1 .. 5
// The seed a kernel hashes with: the host's layer seed + (device step counter) * golden ratio.  The counter lives in
// device memory (tfr_tower_dropout.step, nullable) so that a hipGraph replay of a training step draws a NEW mask: a
// seed passed by value is baked into the captured launch and every replay would reuse the first step's mask.
__device__ __forceinline__ Drop drop_resolve(const Drop d) {
  Drop r = d;
  if (d.thr != 0u && d.sp) r.seed = d.seed + (uint32_t)__builtin_amdgcn_readfirstlane((int)*d.sp) * 0x9E3779B9u;
  r.sp = nullptr;
  return r;
}
__device__ __forceinline__ uint32_t drop_hash(uint32_t seed, uint32_t m, uint32_t word) {
  uint32_t h = m * 0x9E3779B1u + word * 0x85EBCA77u + seed;
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
// keep factors of the N (4 or 8) consecutive columns c .. c + N - 1 of row m, c a multiple of N
// W16: the caller also serves 16-bit fields (compiled out of the persistent GEMM forms the fixed-rate towers run: the extra
// path cost them 60-90 bytes of scratch per lane)
template <int N, bool W16 = true>
__device__ __forceinline__ void drop_run(const Drop d, uint32_t m, uint32_t c, float (&f)[N]) {
  if (W16 && d.lge == 1u) {                                  // 16-bit fields: a word serves two columns (wave-uniform branch)
#pragma unroll
    for (int w = 0; w < N / 2; ++w) {
      const uint32_t h = drop_hash(d.seed, m, (c >> 1) + (uint32_t)w);
      f[2 * w] = ((h & 0xffffu) >= d.thr) ? d.scale : 0.0f;
      f[2 * w + 1] = ((h >> 16) >= d.thr) ? d.scale : 0.0f;
    }
    return;
  }
  const uint32_t fb = 32u >> d.lge;                          // (wave-uniform: scalar registers)
  const uint32_t off = (c & ((1u << d.lge) - 1u)) << (5u - d.lge);
  uint32_t x0 = drop_hash(d.seed, m, c >> d.lge) >> off;
  uint32_t x1 = d.lge == 2u ? 0u : x0 >> (4u * fb);          // (fb = 8: the shift would be by 32 -- x1 comes from the next word below)
  if (N == 8 && d.lge == 2u) x1 = drop_hash(d.seed, m, (c + 4u) >> 2);     // 8-bit fields: four columns per word
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const uint32_t fld = __builtin_amdgcn_ubfe(j < 4 ? x0 : x1, (uint32_t)(j & 3) * fb, fb);
    f[j] = (fld >= d.thr) ? d.scale : 0.0f;
  }
}
// keep factors of columns (2 * kpair, 2 * kpair + 1) of row m (the non-persistent kernels' chunk loops)
__device__ __forceinline__ void drop_pair(const Drop d, uint32_t m, uint32_t kpair, float& f0, float& f1) {
  const uint32_t c0 = 2u * kpair;
  const uint32_t fb = 32u >> d.lge;
  const uint32_t h = drop_hash(d.seed, m, c0 >> d.lge);
  const uint32_t off = (c0 & ((1u << d.lge) - 1u)) << (5u - d.lge);
  f0 = (__builtin_amdgcn_ubfe(h, off, fb) >= d.thr) ? d.scale : 0.0f;
  f1 = (__builtin_amdgcn_ubfe(h, off + fb, fb) >= d.thr) ? d.scale : 0.0f;
}

// Stochastic rounding of the BatchNorm-backward output dz = p * dy + q * z + r to bf16 (round 4).  dy is stored as bf16
// (an 8-bit lattice k * 2^e), p is one fp32 factor per column, and s = q * z + r -- the terms that make sum_m dz = 0 and
// sum_m dz * zhat = 0 -- is ~ 1 / sqrt(M) of |dy|: below half a bf16 ulp once M reaches 10^5.  Round-to-nearest of
// p * k * 2^e + s then returns round(p * k * 2^e) unless p * k sits within |s| of a rounding boundary, and p * k takes only
// 128 positions per column: the share of s that survives is a per-column accident, the column sums of dz drift away from
// zero, and every gradient that sees the MEAN of the layer input (weights / BatchNorm parameters below a ReLU) picks up an
// error that grows like sqrt(M) against the signal (measured at M = 819 200: 3.4 % on a weight matrix, 5.4 % on a beta,
// against 0.6 % at M = 51 200; tools/tower_error_probe.py).  Adding uniform bits below the kept mantissa before the
// truncation makes E[bf16(x)] = x whatever the lattice: the drift becomes zero-mean noise that averages out over M
// (0.5-0.6 % on every gradient at M = 819 200 afterwards).  The dither is a fixed function of (row, column): the same
// bits on every run and in both backward paths.
// Cost: the dither is 8 bits per element (centred: byte * 256 + 128 below the kept mantissa, i.e. the round-up probability
// is the dropped fraction to 1 / 256 ulp), 64 bits per 8-column chunk from ONE two-multiply seed and two xorshift32 rounds
// (full-rate shifts / xors) -- the first version hashed per column pair with the dropout hash (four quarter-rate multiplies
// each) and made the bandwidth-bound apply pass issue-bound: 0.24 -> 0.31 ms at M = 512 000 (profiles/r04_all_workloads.txt).
constexpr uint32_t kDitherSeed = 0x5bd1e995u;
__device__ __forceinline__ uint32_t xorshift32(uint32_t x) { x ^= x << 13; x ^= x >> 17; x ^= x << 5; return x; }
// dither words of the 8-column chunk (row m, columns 8 * k8 ..): byte e of {a, b} belongs to column e
__device__ __forceinline__ void sr_dither8(uint32_t m, uint32_t k8, uint32_t& a, uint32_t& b) {
  a = xorshift32(m * 0x9E3779B1u + k8 * 0x85EBCA77u + kDitherSeed);
  b = xorshift32(a);
}
// bf16 pair (columns e, e + 1 of the chunk, e = 0, 2, 4, 6) by stochastic rounding with the chunk's dither words
__device__ __forceinline__ uint32_t pack_bf16_sr(float lo, float hi, uint32_t a, uint32_t b, int e) {
  const uint32_t w = e < 4 ? a : b;
  const uint32_t d0 = (__builtin_amdgcn_ubfe(w, (uint32_t)(8 * (e & 3)), 8u) << 8) | 0x80u;
  const uint32_t d1 = (__builtin_amdgcn_ubfe(w, (uint32_t)(8 * (e & 3) + 8), 8u) << 8) | 0x80u;
  const uint32_t ua = __float_as_uint(lo) + d0, ub = __float_as_uint(hi) + d1;
  return (ub & 0xffff0000u) | (ua >> 16);
}

// byte offset of 16-byte chunk c (0..7) of row r inside a swizzled [128][64] bf16 tile
__device__ __forceinline__ int swz(int r, int c) { return r * 128 + ((c ^ ((r >> 1) & 7)) << 4); }

struct GemmArgs {
  const uint16_t* A; long lda;       // [M, K] activations (bf16; pre-BN z when PRO != 0)
  const uint16_t* B; long ldb;       // [N, K] weights (bf16)
  uint16_t* C; long ldc;             // [M, N] bf16
  const float* bias;                 // [N] nullable
  const float* a_scale;              // [K] (PRO != 0)
  const float* a_shift;              // [K]
  float* stats;                      // [2 tiles_m][2][N] partials per 64-row half tile  (EPI != 0)
  const uint16_t* Zp; long ldz;      // [M, N] pre-BN z of the layer below               (EPI == 2)
  const float* e_scale;              // [N] its BN scale / shift (relu mask) ...
  const float* e_shift;
  const float* e_mean;               // ... and mean / rstd (z_hat for d gamma)
  const float* e_rstd;
  Drop pro_drop;                     // dropout of the layer below, applied in the A prologue (thr = 0: off)
  Drop epi_drop;                     // ... and in the EPI_RELU_BWD epilogue (the layer whose Zp is given)
  int M, N, K, tiles_m, tiles_n;
  int flags;                           // persistent 256 x 256 kernel: PF_* bits
  int act;                             // ACT_* of PRO_AFFINE_ACT / EPI_ACT_BWD
  int row0;                            // global index of row 0 of A / C (dropout hash) when a launch covers a row range
  uint16_t* Aout; long ldao;           // nullable [M, K] bf16: the persistent kernel writes pro(A) -- the operand it forms
};                                     // in registers -- for the weight-gradient kernel to read back (round 4)

enum { PRO_NONE = 0, PRO_AFFINE = 1, PRO_AFFINE_RELU = 2, PRO_AFFINE_ACT = 3 };
enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_RELU_BWD = 2, EPI_ACT_BWD = 3 };
// Activations other than ReLU (keras/layers.py:66-70: tf.keras.layers.Activation(activation) after the BatchNorm): the
// PRO_AFFINE_ACT prologues apply act(z * scale + shift), the EPI_ACT_BWD epilogue multiplies by act'(.) of the
// recomputed pre-activation.  The code travels in bits 8.. of the C ABI's `prologue` / `epilogue` arguments.
enum { ACT_TANH = 1, ACT_SIGMOID = 2, ACT_ELU = 3, ACT_SOFTPLUS = 4, ACT_SWISH = 5, ACT_LAST = 5 };
__device__ __forceinline__ float act_fwd(int act, float y) {
  switch (act) {
    case ACT_TANH: return tanhf(y);
    case ACT_SIGMOID: return 1.0f / (1.0f + expf(-y));
    case ACT_ELU: return y > 0.f ? y : expm1f(y);
    case ACT_SOFTPLUS: return fmaxf(y, 0.f) + log1pf(expf(-fabsf(y)));
    case ACT_SWISH: return y / (1.0f + expf(-y));
  }
  return y;
}
__device__ __forceinline__ float act_grad(int act, float y) {
  switch (act) {
    case ACT_TANH: { const float t = tanhf(y); return 1.0f - t * t; }
    case ACT_SIGMOID: { const float t = 1.0f / (1.0f + expf(-y)); return t * (1.0f - t); }
    case ACT_ELU: return y > 0.f ? 1.0f : expf(y);
    case ACT_SOFTPLUS: return 1.0f / (1.0f + expf(-y));
    case ACT_SWISH: { const float t = 1.0f / (1.0f + expf(-y)); return t * (1.0f + y * (1.0f - t)); }
  }
  return 1.0f;
}

template <int PRO>
__device__ __forceinline__ uint4 transform_chunk(uint4 v, const float* sc, const float* sh, const Drop drop = Drop{0, 0, 1.f, 2, nullptr},
                                                 uint32_t m = 0, uint32_t k = 0, int act = 0) {
  if (PRO == PRO_NONE) return v;
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  float kf[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
  if (drop.thr) drop_run<8>(drop, m, k, kf);            // wave-uniform
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    float a = __builtin_fmaf(bf16_lo(w[i]), sc[2 * i], sh[2 * i]);
    float b = __builtin_fmaf(bf16_hi(w[i]), sc[2 * i + 1], sh[2 * i + 1]);
    if (PRO == PRO_AFFINE_RELU) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
    if (PRO == PRO_AFFINE_ACT) { a = act_fwd(act, a); b = act_fwd(act, b); }
    if (drop.thr) { a *= kf[2 * i]; b *= kf[2 * i + 1]; }
    w[i] = pack_bf16(a, b);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

// 16-lane row sum with DPP (result in lane 15 of every 16-lane row).
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true));
  return v;
}

constexpr int WPITCH = 144;                      // bytes per row of a wave's 64 x 64 bf16 epilogue tile (128 + 16 pad)

// C[M, N] = pro(A)[M, K] . B[N, K]^T (+ bias), bf16 out.
// GL (K % 64 == 0): the weight tile goes global -> LDS by LDS-DMA (no VGPR staging, no ds_write_b128 pass --
// the slowest LDS instruction, 79 B/clk/CU, was half of this kernel's LDS time); with PRO_NONE (dgrad, layer 1)
// the activation tile does too.  The XOR swizzle is applied on the SOURCE address (the LDS image of an
// LDS-DMA is lane-linear).
template <int PRO, int EPI, bool GL>
__global__ __launch_bounds__(256, 2) void tower_gemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Drop pdrop = drop_resolve(g.pro_drop), edrop = drop_resolve(g.epi_drop);
  // [buf][A tile | B tile], then the per-column prologue coefficients.
  unsigned char* tiles = smem;
  float* s_scale = reinterpret_cast<float*>(smem + 4 * TILE_BYTES);
  float* s_shift = s_scale + ((g.K + 63) & ~63);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave & 1, wn = wave >> 1;

  // XCD-aware tile map: workgroup id -> (m tile, n tile); the n tiles of one m tile
  // run back to back on one XCD (ids congruent mod 8 share an XCD).
  const int id = blockIdx.x;
  const int xcd = id & 7, j = id >> 3;
  const int tn = j % g.tiles_n;
  const int tm = (j / g.tiles_n) * 8 + xcd;
  if (tm >= g.tiles_m) return;
  const int m0 = tm * BM, n0 = tn * BN;
  const int nk = (g.K + BK - 1) / BK;

  // staging: thread owns chunk column c = tid & 7 of rows (tid >> 3) + 32 * i.
  const int c = tid & 7, r0 = tid >> 3;
  struct RegsA { uint4 a[4]; };
  struct RegsB { uint4 b[4]; };
  auto load_a = [&](int kt, RegsA& R) {
    const int k = kt * BK + c * 8;
    const bool kin = k < g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long am = m0 + r0 + 32 * i;
      R.a[i] = (kin && am < g.M) ? *reinterpret_cast<const uint4*>(g.A + am * g.lda + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto load_b = [&](int kt, RegsB& R) {
    const int k = kt * BK + c * 8;
    const bool kin = k < g.K;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long bn = n0 + r0 + 32 * i;
      R.b[i] = (kin && bn < g.N) ? *reinterpret_cast<const uint4*>(g.B + bn * g.ldb + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_a = [&](int kt, int buf, const RegsA& RA) {
    unsigned char* ta = tiles + buf * 2 * TILE_BYTES;
    const int k = kt * BK + c * 8;
    float sc[8], sh[8];
    if (PRO != PRO_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = s_scale[k + e]; sh[e] = s_shift[k + e]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      uint4 va = RA.a[i];
      // columns beyond K have scale = shift = 0 -> transform(0) = 0 for both modes
      if (PRO != PRO_NONE) va = transform_chunk<PRO>(va, sc, sh, pdrop, (uint32_t)(g.row0 + m0 + row), (uint32_t)k, g.act);
      *reinterpret_cast<uint4*>(ta + swz(row, c)) = va;
    }
  };
  auto store_tile = [&](int kt, int buf, const RegsA& RA, const RegsB& RB) {
    unsigned char* ta = tiles + buf * 2 * TILE_BYTES;
    unsigned char* tb = ta + TILE_BYTES;
    const int k = kt * BK + c * 8;
    float sc[8], sh[8];
    if (PRO != PRO_NONE) {
#pragma unroll
      for (int e = 0; e < 8; ++e) { sc[e] = s_scale[k + e]; sh[e] = s_shift[k + e]; }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 32 * i;
      uint4 va = RA.a[i];
      if (PRO != PRO_NONE) va = transform_chunk<PRO>(va, sc, sh, pdrop, (uint32_t)(g.row0 + m0 + row), (uint32_t)k, g.act);
      *reinterpret_cast<uint4*>(ta + swz(row, c)) = va;
      *reinterpret_cast<uint4*>(tb + swz(row, c)) = RB.b[i];
    }
  };
  // LDS-DMA of a [128][64] bf16 tile (GL): wave w copies rows [32 w + 8 i, + 8), i < 4, one KiB per instruction;
  // lane -> (row, physical 16-byte slot); the slot holds logical chunk slot ^ ((row >> 1) & 7) (= swz()).
  // Rows past `nrows` are clamped to the last one: they only reach output rows / columns that are never stored.
  auto glds_tile = [&](const uint16_t* G, long ld, long row0, long nrows, int kt, unsigned char* tdst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rl = wave * 32 + i * 8 + (lane >> 3);
      const int cl = (lane & 7) ^ ((rl >> 1) & 7);
      long gr = row0 + rl;
      gr = gr < nrows ? gr : nrows - 1;
      glds16(G + gr * ld + kt * BK + cl * 8, tdst + (wave * 32 + i * 8) * 128);
    }
  };

  f32x4 acc[4][4];      // [fn][fm]: D[n][m]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](int buf) {
    const unsigned char* ta = tiles + buf * 2 * TILE_BYTES;
    const unsigned char* tb = ta + TILE_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        fb[f] = *reinterpret_cast<const bf16x8*>(tb + swz(wn * 64 + f * 16 + fr, kk * 4 + fq));   // weights
        fa[f] = *reinterpret_cast<const bf16x8*>(ta + swz(wm * 64 + f * 16 + fr, kk * 4 + fq));   // activations
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fm = 0; fm < 4; ++fm)
          acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[fn], fa[fm], acc[fn][fm], 0, 0, 0);
    }
  };

  if constexpr (!GL) {
  // A (streamed from HBM) is requested two tiles ahead in two register sets, B (the weights,
  // L2 resident) one tile ahead: an A load has two compute phases to land.
  RegsA A0, A1;
  RegsB Bx;
  load_a(0, A0);
  load_b(0, Bx);
  if (nk > 1) load_a(1, A1);
  if (PRO != PRO_NONE) {
    for (int k = tid; k < nk * BK; k += 256) {
      s_scale[k] = (k < g.K) ? g.a_scale[k] : 0.f;
      s_shift[k] = (k < g.K) ? g.a_shift[k] : 0.f;
    }
    __syncthreads();
  }
  store_tile(0, 0, A0, Bx);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) load_b(kt + 1, Bx);
    if (kt + 2 < nk) load_a(kt + 2, A0);
    compute(0);
    if (kt + 1 < nk) store_tile(kt + 1, 1, A1, Bx);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 2 < nk) load_b(kt + 2, Bx);
    if (kt + 3 < nk) load_a(kt + 3, A1);
    compute(1);
    if (kt + 2 < nk) store_tile(kt + 2, 0, A0, Bx);
    __syncthreads();
  }
  } else {
  // LDS-DMA pipeline, two LDS buffers: the DMA of tile kt + 1 is issued right after the barrier that
  // retired the reads of its buffer, lands during compute(kt), and is retired by the vmcnt(0) that
  // __syncthreads() carries.  PRO != NONE keeps the activation tile on the register path (it needs the
  // BatchNorm / ReLU / Dropout transform), two tiles ahead as before.
  constexpr bool GLA = (PRO == PRO_NONE);
  unsigned char* const tA0 = tiles;
  unsigned char* const tB0 = tiles + TILE_BYTES;
  unsigned char* const tA1 = tiles + 2 * TILE_BYTES;
  unsigned char* const tB1 = tiles + 3 * TILE_BYTES;
  RegsA A0, A1;
  if (!GLA) { load_a(0, A0); if (nk > 1) load_a(1, A1); }
  glds_tile(g.B, g.ldb, n0, g.N, 0, tB0);
  if (GLA) glds_tile(g.A, g.lda, m0, g.M, 0, tA0);
  if (PRO != PRO_NONE) {
    for (int k = tid; k < nk * BK; k += 256) {
      s_scale[k] = (k < g.K) ? g.a_scale[k] : 0.f;
      s_shift[k] = (k < g.K) ? g.a_shift[k] : 0.f;
    }
    __syncthreads();
  }
  if (!GLA) store_a(0, 0, A0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 2) {
    if (kt + 1 < nk) {
      glds_tile(g.B, g.ldb, n0, g.N, kt + 1, tB1);
      if (GLA) glds_tile(g.A, g.lda, m0, g.M, kt + 1, tA1);
    }
    if (!GLA && kt + 2 < nk) load_a(kt + 2, A0);
    compute(0);
    if (!GLA && kt + 1 < nk) store_a(kt + 1, 1, A1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    if (kt + 2 < nk) {
      glds_tile(g.B, g.ldb, n0, g.N, kt + 2, tB0);
      if (GLA) glds_tile(g.A, g.lda, m0, g.M, kt + 2, tA0);
    }
    if (!GLA && kt + 3 < nk) load_a(kt + 3, A1);
    compute(1);
    if (!GLA && kt + 2 < nk) store_a(kt + 2, 0, A0);
    __syncthreads();
  }
  }

  // ---- epilogue.  Lane holds D[n = nb + 4*fq + r][m = mb + fr], r = 0..3.  The staging
  // buffers are dead: every wave transposes its 64 x 64 bf16 result through a private LDS
  // region so that the global stores (and the Zp loads of EPI_RELU_BWD) are 16 bytes per
  // lane and 128 contiguous bytes per row.  No workgroup barrier from here on.
  unsigned char* wl = smem + wave * (64 * WPITCH);
  const long mb = m0 + wm * 64;
  const int nb = n0 + wn * 64;
  if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = lane + 64 * i, row = q >> 3, cc = q & 7;
      uint4 zz = make_uint4(0, 0, 0, 0);
      if (mb + row < g.M && nb + cc * 8 < g.N)
        zz = *reinterpret_cast<const uint4*>(g.Zp + (mb + row) * g.ldz + nb + cc * 8);
      *reinterpret_cast<uint4*>(wl + row * WPITCH + cc * 16) = zz;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  float s1[4][4], s2[4][4];
#pragma unroll
  for (int fn = 0; fn < 4; ++fn) {
    const int n = nb + fn * 16 + fq * 4;
    float bias4[4] = {0.f, 0.f, 0.f, 0.f};
    if (g.bias) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bias4[r] = (n + r < g.N) ? g.bias[n + r] : 0.f;
    }
    float em[4], er[4], es[4], eh[4];
    if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bool in = n + r < g.N;
        es[r] = in ? g.e_scale[n + r] : 0.f; eh[r] = in ? g.e_shift[n + r] : 0.f;
        em[r] = in ? g.e_mean[n + r] : 0.f;  er[r] = in ? g.e_rstd[n + r] : 0.f;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) { s1[fn][r] = 0.f; s2[fn][r] = 0.f; }
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      const int row = fm * 16 + fr;
      const bool min = mb + row < g.M;
      unsigned char* slot = wl + row * WPITCH + (fn * 16 + fq * 4) * 2;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[fn][fm][r] + bias4[r];
      if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
        // dy = da * 1[y > 0], y = z * scale + shift;  column partials: sum dy, sum dy * z_hat
        const uint2 zz = *reinterpret_cast<const uint2*>(slot);
        const float z[4] = {bf16_lo(zz.x), bf16_hi(zz.x), bf16_lo(zz.y), bf16_hi(zz.y)};
        float kf[4] = {1.f, 1.f, 1.f, 1.f};
        if (edrop.thr) {                           // d a / d relu = keep / (1 - rate): same hash as the forward
          drop_run<4>(edrop, (uint32_t)(g.row0 + mb + row), (uint32_t)(nb + fn * 16 + fq * 4), kf);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float y = __builtin_fmaf(z[r], es[r], eh[r]);
          if (EPI == EPI_ACT_BWD) v[r] = min ? v[r] * kf[r] * act_grad(g.act, y) : 0.f;
          else v[r] = (y > 0.f && min) ? v[r] * kf[r] : 0.f;
          s1[fn][r] += v[r];
          s2[fn][r] = __builtin_fmaf(v[r], (z[r] - em[r]) * er[r], s2[fn][r]);
        }
      } else if (EPI == EPI_STATS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float t = min ? v[r] : 0.f;
          s1[fn][r] += t;
          s2[fn][r] = __builtin_fmaf(t, t, s2[fn][r]);
        }
      }
      *reinterpret_cast<uint2*>(slot) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
    }
  }
  if (EPI != EPI_PLAIN && mb < g.M) {
    // column partials of this wave's 64 rows: stats[(2 tm + wm)][which][n]  (one row per 64-row slab below M)
    float* st = g.stats + ((long)(tm * 2 + wm) * 2) * g.N;
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = row16_sum(s1[fn][r]), b = row16_sum(s2[fn][r]);
        const int n = nb + fn * 16 + fq * 4 + r;
        if (fr == 15 && n < g.N) { st[n] = a; st[g.N + n] = b; }
      }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int q = lane + 64 * i, row = q >> 3, cc = q & 7;
    if (mb + row < g.M && nb + cc * 8 < g.N)      // N % 8 == 0 is checked on the host
      *reinterpret_cast<uint4*>(g.C + (mb + row) * g.ldc + nb + cc * 8) =
          *reinterpret_cast<const uint4*>(wl + row * WPITCH + cc * 16);
  }
}

// ------------------------------------------------------------------------------------
// 256 x 256 tile variant (K % 64 == 0, N >= 256): 512 threads = 8 waves as 4 (m) x 2 (n), each wave a
// 64 x 128 sub-tile (acc[8][4]); two LDS buffers of (32 KB activations + 32 KB weights), weights -- and
// activations when there is no prologue -- by LDS-DMA.  Against the 128 x 128 kernel every byte that enters
// the CU feeds twice the MFMA work (the k-loop there is bound by the L2 -> LDS path, not by MFMA issue),
// and an A row block is fetched by N / 256 instead of N / 128 workgroups.
constexpr int BM2 = 256, BN2 = 256;
constexpr int TILE2_BYTES = BM2 * BK * 2;         // 32 KB per operand tile
// Developer ablations (tools/gemm_ablate.py builds tower.hip with -DTFR_GEMM_ABLATE=mask; the product build has 0):
// 1 = no MFMA block, 2 = no global -> LDS staging inside the k loop, 4 = no epilogue, 8 = MFMAs on registers (no ds_read);
// (the persistent kernel takes stamps only: 16).
#ifndef TFR_GEMM_ABLATE
#define TFR_GEMM_ABLATE 0
#endif
constexpr int kAb = TFR_GEMM_ABLATE;
// mask 16: thread 0 of every workgroup records s_memtime at the phase boundaries (+ HW_ID, XCC_ID) into a buffer
// set with tfr_prof_set_buffer_gemm() -- [workgroup][8] u64.
#if (TFR_GEMM_ABLATE & 16)
__device__ unsigned long long* g_prof_gemm = nullptr;
#define GEMM_STAMP(i) do { if (tid == 0) prof_t[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define GEMM_STAMP(i) do { } while (0)
#endif

template <int PRO, int EPI>
__global__ __launch_bounds__(512, 1) void tower_gemm256_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Drop pdrop = drop_resolve(g.pro_drop), edrop = drop_resolve(g.epi_drop);
  unsigned char* tiles = smem;                    // [buf][A tile | B tile]
  float* s_scale = reinterpret_cast<float*>(smem + 4 * TILE2_BYTES);
  float* s_shift = s_scale + g.K;                 // K % 64 == 0

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);       // wave-uniform: LDS bases / tile offsets in SGPRs
  const int wm = wave & 3, wn = wave >> 2;
  const int id = blockIdx.x;
#if (TFR_GEMM_ABLATE & 16)
  unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
  GEMM_STAMP(0);
  const int xcd = id & 7, j = id >> 3;
  const int tn = j % g.tiles_n;
  const int tm = (j / g.tiles_n) * 8 + xcd;
  if (tm >= g.tiles_m) return;
  const int m0 = tm * BM2, n0 = tn * BN2;
  const int nk = g.K / BK;
  constexpr bool GLA = (PRO == PRO_NONE);

  const int c = tid & 7, r0 = tid >> 3;           // staging: chunk column c of rows r0 + 64 i
  struct RegsA { uint4 a[4]; };
  auto load_a = [&](int kt, RegsA& R) {
    const int k = kt * BK + c * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int ar = r0 + 64 * i;
      asm volatile("" : "+v"(ar));
      const long am = m0 + ar;
      R.a[i] = (am < g.M) ? *reinterpret_cast<const uint4*>(g.A + am * g.lda + k) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_a = [&](int kt, unsigned char* ta, const RegsA& RA) {
    const int k = kt * BK + c * 8;
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = s_scale[k + e]; sh[e] = s_shift[k + e]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 64 * i;
      const uint4 va = transform_chunk<PRO>(RA.a[i], sc, sh, pdrop, (uint32_t)(g.row0 + m0 + row), (uint32_t)k, g.act);
      *reinterpret_cast<uint4*>(ta + swz(row, c)) = va;
    }
  };
  auto glds_tile = [&](const uint16_t* G, long ld, long row0, long nrows, int kt, unsigned char* tdst) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int rl = wave * 32 + i * 8 + (lane >> 3);
      asm volatile("" : "+v"(rl));                  // keep the address arithmetic in the loop (hoisted copies spill)
      const int cl = (lane & 7) ^ ((rl >> 1) & 7);
      long gr = row0 + rl;
      gr = gr < nrows ? gr : nrows - 1;
      glds16(G + gr * ld + kt * BK + cl * 8, tdst + (wave * 32 + i * 8) * 128);
    }
  };

  f32x4 acc[8][4];      // [fn][fm]: D[n][m]
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  auto compute = [&](const unsigned char* ta, const unsigned char* tb) {
#pragma unroll 1
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        if (kAb & 8) asm volatile("" : "=v"(fa[f]));
        else fa[f] = *reinterpret_cast<const bf16x8*>(ta + swz(wm * 64 + f * 16 + fr, kk * 4 + fq));    // activations
      }
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        bf16x8 fb[4];
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          if (kAb & 8) asm volatile("" : "=v"(fb[f]));
          else fb[f] = *reinterpret_cast<const bf16x8*>(tb + swz(wn * 128 + hn * 64 + f * 16 + fr, kk * 4 + fq));   // weights
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
          for (int fm = 0; fm < 4; ++fm)
            acc[hn * 4 + fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[fn], fa[fm], acc[hn * 4 + fn][fm], 0, 0, 0);
      }
    }
  };

  // One code path for both LDS buffers (a second inlined copy of the MFMA block makes the register allocator
  // carry two sets of the 128 accumulators across the loop).  The staging of tile kt + 1 is issued at the top of
  // iteration kt and has the 64 MFMAs of compute(kt) to land: LDS-DMA for the weights (and for the activations
  // when there is no prologue), otherwise one register set that is transformed and written behind the MFMAs.
  RegsA R;
  if (!GLA) load_a(0, R);
  glds_tile(g.B, g.ldb, n0, g.N, 0, tiles + TILE2_BYTES);
  if (GLA) glds_tile(g.A, g.lda, m0, g.M, 0, tiles);
  if (PRO != PRO_NONE) {
    for (int k = tid; k < g.K; k += 512) { s_scale[k] = g.a_scale[k]; s_shift[k] = g.a_shift[k]; }
    __syncthreads();
  }
  if (!GLA) store_a(0, tiles, R);
  __syncthreads();
  GEMM_STAMP(1);
  // EPI_RELU_BWD: the epilogue needs the wave's 64 x 128 piece of Zp.  With one workgroup per CU nothing else
  // would cover those loads, so they are software-pipelined: half 0 is requested during the last k step (its
  // 64 MFMAs hide the HBM round trip), half 1 while half 0 is being processed.
  const long mb = m0 + wm * 64;
  uint4 zq[8];
  auto zp_load = [&](int h) {
    const int nbz = n0 + wn * 128 + h * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = lane + 64 * i, row = q >> 3, cc = q & 7;
      zq[i] = (mb + row < g.M && nbz + cc * 8 < g.N)
                  ? *reinterpret_cast<const uint4*>(g.Zp + (mb + row) * g.ldz + nbz + cc * 8) : make_uint4(0, 0, 0, 0);
    }
  };
#pragma unroll 1
  for (int kt = 0; kt < nk; ++kt) {
    unsigned char* const tAc = tiles + (kt & 1) * (2 * TILE2_BYTES);
    unsigned char* const tAn = tiles + ((kt + 1) & 1) * (2 * TILE2_BYTES);
    const bool more = kt + 1 < nk;
    if (more && !(kAb & 2)) {
      glds_tile(g.B, g.ldb, n0, g.N, kt + 1, tAn + TILE2_BYTES);
      if (GLA) glds_tile(g.A, g.lda, m0, g.M, kt + 1, tAn);
      else load_a(kt + 1, R);
    }
    if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && !more) zp_load(0);
    if (!(kAb & 1)) compute(tAc, tAc + TILE2_BYTES);
    if (!GLA && more && !(kAb & 2)) store_a(kt + 1, tAn, R);
    __syncthreads();
  }
  GEMM_STAMP(2);
  if (kAb & 4) {
    float s = 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) s += acc[a][b][0] + acc[a][b][1] + acc[a][b][2] + acc[a][b][3];
    if (s == 12345.678f) g.C[tid] = 1;
    return;
  }

  // ---- epilogue: the wave's 64 x 128 result as two 64 x 64 halves through its private LDS region
  // (same code shape as the 128 x 128 kernel; DS operations of a wave complete in order).
  unsigned char* wl = smem + wave * (64 * WPITCH);
  const bool slab_live = mb < g.M;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int nb = n0 + wn * 128 + h * 64;
    if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int q = lane + 64 * i, row = q >> 3, cc = q & 7;
        *reinterpret_cast<uint4*>(wl + row * WPITCH + cc * 16) = zq[i];
      }
      if (h == 0) zp_load(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    if (h == 1) GEMM_STAMP(3);
    float* const st = g.stats + ((long)(tm * 4 + wm) * 2) * g.N;      // one row of partials per 64-row slab
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      float s1[4], s2[4];
      const int n = nb + fn * 16 + fq * 4;
      float bias4[4] = {0.f, 0.f, 0.f, 0.f};
      if (g.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r) bias4[r] = (n + r < g.N) ? g.bias[n + r] : 0.f;
      }
      float em[4], er[4], es[4], eh[4];
      if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const bool in = n + r < g.N;
          es[r] = in ? g.e_scale[n + r] : 0.f; eh[r] = in ? g.e_shift[n + r] : 0.f;
          em[r] = in ? g.e_mean[n + r] : 0.f;  er[r] = in ? g.e_rstd[n + r] : 0.f;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) { s1[r] = 0.f; s2[r] = 0.f; }
#pragma unroll
      for (int fm = 0; fm < 4; ++fm) {
        const int row = fm * 16 + fr;
        const bool min = mb + row < g.M;
        unsigned char* slot = wl + row * WPITCH + (fn * 16 + fq * 4) * 2;
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = acc[h * 4 + fn][fm][r] + bias4[r];
        if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
          const uint2 zz = *reinterpret_cast<const uint2*>(slot);
          const float z[4] = {bf16_lo(zz.x), bf16_hi(zz.x), bf16_lo(zz.y), bf16_hi(zz.y)};
          float kf[4] = {1.f, 1.f, 1.f, 1.f};
          if (edrop.thr) {
            drop_run<4>(edrop, (uint32_t)(g.row0 + mb + row), (uint32_t)(nb + fn * 16 + fq * 4), kf);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float y = __builtin_fmaf(z[r], es[r], eh[r]);
            if (EPI == EPI_ACT_BWD) v[r] = min ? v[r] * kf[r] * act_grad(g.act, y) : 0.f;
            else v[r] = (y > 0.f && min) ? v[r] * kf[r] : 0.f;
            s1[r] += v[r];
            s2[r] = __builtin_fmaf(v[r], (z[r] - em[r]) * er[r], s2[r]);
          }
        } else if (EPI == EPI_STATS) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float t = min ? v[r] : 0.f;
            s1[r] += t;
            s2[r] = __builtin_fmaf(t, t, s2[r]);
          }
        }
        *reinterpret_cast<uint2*>(slot) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      }
      if (EPI != EPI_PLAIN && slab_live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float a = row16_sum(s1[r]), b = row16_sum(s2[r]);
          if (fr == 15 && n + r < g.N) { st[n + r] = a; st[g.N + n + r] = b; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int q = lane + 64 * i, row = q >> 3, cc = q & 7;
      if (mb + row < g.M && nb + cc * 8 < g.N)
        *reinterpret_cast<uint4*>(g.C + (mb + row) * g.ldc + nb + cc * 8) =
            *reinterpret_cast<const uint4*>(wl + row * WPITCH + cc * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  GEMM_STAMP(4);
#if (TFR_GEMM_ABLATE & 16)
  if (kAb & 32) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // ... and when the stores have been acknowledged
  GEMM_STAMP(5);
  if (tid == 0 && g_prof_gemm) {
    prof_t[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
    prof_t[7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
    for (int i = 0; i < 8; ++i) g_prof_gemm[(size_t)id * 8 + i] = prof_t[i];
  }
#endif
}

#if (TFR_GEMM_ABLATE & 16)
extern "C" int tfr_prof_set_buffer_gemm(void* device_u64_buffer) {
  unsigned long long* p = (unsigned long long*)device_u64_buffer;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_gemm), &p, sizeof(p));
}
#endif

// ------------------------------------------------------------------------------------
// Persistent form of the 256 x 256 kernel: full tiles only (M % 256 == 0, N % 256 == 0, K % 64 == 0, 128 <= K <= 1024;
// the launcher sends a ragged last M-tile through the kernel above).  Why (tools/gemm_timeline.py, config-2 hidden
// layer, round 2): with one workgroup per tile a CU spent 26 / 34 / 40 us per tile (plain / forward / dgrad) of which
// the MFMA block needs 8.4: 1.3-2.2 us between workgroups, 2.6-3.7 us until the first tile is staged, a k step that
// lasts as long as an HBM round trip (2.2 us, the stage is requested at the top of the step that precedes its use),
// and an epilogue of 4.3 / 7.3 / 16.4 us that nothing overlaps and that was instruction-bound (per-element bounds
// checks, 64-bit address arithmetic per access, per-lane global loads of the per-column coefficients).  Here
//  * one workgroup per CU walks its XCD's tiles (the two n-tiles of an M-tile still run side by side on one XCD);
//  * every LDS-DMA is issued by hand (inline asm, SGPR base + a per-lane 32-bit offset computed once per kernel) and
//    waited for by hand, so the next tile's first TWO stages are in flight before the epilogue starts (no-prologue
//    form) and the epilogue's stores drain behind the next tile's first k steps;
//  * waves 0-3 touch the A lines of stage kt + 2 (one dword per 128-byte line, LDS-DMA'd into a junk slot) so that the
//    stage requested one step ahead is an L2 hit; waves 4-7 do the same for the Zp tile of the dgrad epilogue;
//  * the epilogue works on 16-row x 64-column chunks through a 2 KB per-wave staging slot outside the stage buffers,
//    keeps the per-column coefficients in registers per 64-column half (they come from LDS, loaded once per tile),
//    has no bounds checks, and folds (z - mean) * rstd out of the per-element work:
//    sum dy * zhat = rstd * sum dy * z - mean * rstd * sum dy.
constexpr int P_SCALE = 4 * TILE2_BYTES;              // [2][K] floats (K <= 1024)
constexpr int P_EPI = P_SCALE + 2 * 1024 * 4;         // [4][256] floats: the tile's per-column epilogue coefficients
constexpr int P_STAGE = P_EPI + 4 * 256 * 4;          // 8 waves x 2 KB
constexpr int P_JUNK = P_STAGE + 8 * 2048;            // 8 waves x 256 B (touch destinations, never read)
constexpr int P_LDS = P_JUNK + 8 * 256;               // 157 696 B
// the keep bits of a stage's A tile (rate 1/2), [256 rows][2 words] = 2 KB per buffer, double buffered in space the forward
// forms do not use: the touch slots and the three epilogue-coefficient rows of the dgrad forms (161 792 B of dynamic LDS
// was refused by the runtime)
constexpr int P_MASK0 = P_JUNK, P_MASK1 = P_EPI + 256 * 4;

__device__ __forceinline__ void dma16_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  // (s_nop 4: an SGPR fresh from v_readfirstlane must not be read by a VMEM instruction within 5 states)
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ void touch4_s(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  asm volatile("s_nop 4\n\ts_mov_b32 m0, %2\n\tglobal_load_lds_dword %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
__device__ __forceinline__ f32x4 row16_sum4(f32x4 v) {
  f32x4 o;
#pragma unroll
  for (int r = 0; r < 4; ++r) o[r] = row16_sum(v[r]);
  return o;
}

enum { PF_TOUCH_A = 1, PF_TOUCH_Z = 2, PF_IDLE_DMA = 4, PF_NO_BIT_TABLE = 8 };   // PF_IDLE_DMA (TFR_GEMM_FLAGS=4): rounds 2-3, a step with nothing to request still issued its pieces (into an idle slot)

typedef short i16x2 __attribute__((ext_vector_type(2)));

// The previous layer's BatchNorm (+ ReLU, + dropout) on one MFMA operand fragment (8 consecutive k of one row) as it
// comes out of LDS: relu on the packed bf16 pair is a signed 16-bit max with 0 (v_pk_max_i16), the affine a v_pk_fma_f32.
// DROP == 2: the keep bits of the fragment's 8 columns come from the stage's bit table in LDS (`bits8`: bit j = column j
// kept) and the factor 1 / (1 - rate) = 2 is already folded into sc / sh -- see the table's comment in
// tower_gemm256p_kernel; DROP == 1: one hash per fragment (drop_run).
template <int PRO, int DROP>       // DROP: 0 none, 1 one hash per fragment (drop_run), 2 the stage's keep-bit table, 3 = 1 + 16-bit fields
__device__ __forceinline__ bf16x8 transform_frag(bf16x8 raw, const f32x4 sc0, const f32x4 sc1, const f32x4 sh0, const f32x4 sh1,
                                                 const Drop d, uint32_t m, uint32_t k, int act, uint32_t bits8 = 0u) {
  constexpr bool bt = DROP == 2;
  if (PRO == PRO_NONE) return raw;
  const uint4 v = __builtin_bit_cast(uint4, raw);
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
  const float sc[8] = {sc0[0], sc0[1], sc0[2], sc0[3], sc1[0], sc1[1], sc1[2], sc1[3]};
  const float sh[8] = {sh0[0], sh0[1], sh0[2], sh0[3], sh1[0], sh1[1], sh1[2], sh1[3]};
  uint32_t o[4];
  float kf[8];
  if (DROP == 1 || DROP == 3) drop_run<8, DROP == 3>(d, m, k, kf);     // one hash for the fragment's 8 columns when the rate allows
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    f32x2 x = {bf16_lo(w[i]), bf16_hi(w[i])};
    const f32x2 s = {sc[2 * i], sc[2 * i + 1]}, h = {sh[2 * i], sh[2 * i + 1]};
    x = __builtin_elementwise_fma(x, s, h);           // one v_pk_fma_f32 (-ffp-contract=off splits x * s + h into v_pk_mul + v_pk_add)
    if (PRO == PRO_AFFINE_ACT) { x[0] = act_fwd(act, x[0]); x[1] = act_fwd(act, x[1]); }
    if (DROP == 1 || DROP == 3) x = x * f32x2{kf[2 * i], kf[2 * i + 1]};    // relu(y) * f == relu(y * f) for f >= 0
    uint32_t pk = pack_bf16(x[0], x[1]);
    if (PRO == PRO_AFFINE_RELU)
      pk = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), i16x2{0, 0}));
    if (bt) {                                        // zero the dropped halves of the packed pair: 0 / -1 per bit, merged
      const uint32_t m0 = (uint32_t)__builtin_amdgcn_sbfe((int)bits8, 2 * i, 1), m1 = (uint32_t)__builtin_amdgcn_sbfe((int)bits8, 2 * i + 1, 1);
      pk &= (m0 & 0xffffu) | (m1 & 0xffff0000u);
    }
    o[i] = pk;
  }
  return __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
}

// PP (round 5, TFR_GEMM_PP): the k loop as a two-group PING-PONG.  The single-phase loop below it costs 4 440 cycles per k
// step against 2 048 of MFMA because the two wavefronts of a SIMD do the same thing at the same time: both wait for their
// fragment reads, both queue their MFMAs, and MFMA issue, LDS-read return and LDS-DMA issue add up.  Here the
// waves of n-half 1 (waves 4-7: the OTHER wavefront of every SIMD) run one phase behind those of n-half 0, and a k step is
// four phases separated by workgroup barriers:   LOAD(kk) = the 12 fragment reads of a 32-wide k half (+ the prologue's
// BatchNorm / ReLU / Dropout on them, + this group's LDS-DMA pieces of the next stage)   |   MMA(kk) = its 32 MFMAs under
// s_setprio(1), from registers only.  While one group of a SIMD is in MMA the other is in LOAD: the matrix pipe is fed by
// one wave while the other wave's LDS reads and VALU prologue run beside it.
//   group 0:        L00 | M00 | L01 | M01 | L10 | ...  | M(n-1,1) | epilogue
//   group 1:  (b) |     L00 | M00 | L01 | M01 | ...  | L(n-1,1) | M(n-1,1), epilogue      (b = one extra barrier per tile, the
//                                                                                           last barrier of the tile left out)
// Hazards, by barrier count (group 1's j-th barrier of a k step is the workgroup's (4 s + j + 1)-th, group 0's the (4 s + j)-th):
//  * the stage of step s + 1 is requested during step s into the other buffer, whose last reader (group 1's L(s-1,1)) finished
//    before barrier 4 s; group 0 asks for its pieces in L(s,0) / L(s,1), group 1 for all of its pieces in L(s,0) (three phases
//    ahead of its wait), and both wait vmcnt(0) before barrier 4 s + 4 -- group 0 after M(s,1), group 1 after L(s,1) -- which is
//    the barrier group 0's L(s+1,0) starts behind;
//  * every LOAD phase drains its LDS reads / writes (lgkmcnt(0)) before its closing barrier: no buffer is re-staged under a read;
//  * the keep-bit table of step s + 1 is written in L(s,0) by both groups, i.e. before barrier 4 s + 2, and read after 4 s + 4;
//  * at the end of a tile both groups pass barrier 4 n together and run their (barrier-free) epilogues side by side.
template <int PRO, int EPI, int DROP, int PP = 0>      // PP: 0 single-phase loop, 1 ping-pong with the LDS-DMA pieces in the LOAD phases, 2 with the pieces among the MFMAs of MMA(kt, 0); DROP: 0 none, 1 hashed per fragment / epilogue run (fields <= 8 bits), 2 = 1 + the prologue's keep-bit table, 3 = 1 + 16-bit fields (rates that are not a multiple of 1 / 256)
__global__ __launch_bounds__(512, 1) void tower_gemm256p_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Drop pdrop = drop_resolve(g.pro_drop), edrop = drop_resolve(g.epi_drop);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave & 3, wn = wave >> 2;
  const int nk = g.K / BK;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float* s_scale = reinterpret_cast<float*>(smem + P_SCALE);
  float* s_shift = s_scale + g.K;
  float* s_epi = reinterpret_cast<float*>(smem + P_EPI);
  unsigned char* sw = smem + P_STAGE + wave * 2048;
  const uint32_t junk = lds0 + P_JUNK + wave * 256;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
  const int nq = ((g.tiles_m - xcd + 7) >> 3) * g.tiles_n;       // this XCD's tiles: M-tiles xcd, xcd + 8, ...
  if (slot >= nq) return;
  const bool touch_z = (EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && (g.flags & PF_TOUCH_Z) != 0 && wave >= 4;
  // Keep-bit table (round 4; rate 1/2 = the reference default, 1-bit fields: a hash word serves 32 columns of a row).
  // Per fragment the prologue hashed once and expanded 8 keep factors (42 VALU on top of the 20 of BatchNorm + ReLU; the
  // forward GEMM took 0.53 ms against 0.42 without Dropout at M = 512 000).  The A tile of a stage is 256 rows x 64
  // columns = 512 hash words: every thread computes ONE per k step, a step ahead, into LDS; a fragment then costs one
  // ds_read_b32 + a shift for its byte and 4 x (2 v_bfe_i32 + merge + and) on the packed pairs, with the factor 2 folded
  // into the per-column scale / shift (exact).  Same bits as drop_run (the hash and its word / bit layout are unchanged).
  constexpr bool bt = DROP == 2;                    // (the launcher: rate 1/2, a homogeneous activation)
  auto mask_buf = [&](int mbuf) __attribute__((always_inline)) { return reinterpret_cast<uint32_t*>(smem + (mbuf ? P_MASK1 : P_MASK0)); };   // [256][2]
  auto fill_mask = [&](int mbuf, int tm_, int kt_) __attribute__((always_inline)) {
    mask_buf(mbuf)[tid] = drop_hash(pdrop.seed, (uint32_t)(g.row0 + tm_ * BM2 + (tid >> 1)), (uint32_t)(2 * kt_ + (tid & 1)));
  };
  int mcur = 0;                                      // the table buffer of the step being computed

  // per-lane byte offsets of the staging pieces (tile independent): piece i = rows wave * 32 + 8 i .. + 8
  uint32_t offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int rl = wave * 32 + i * 8 + (lane >> 3);
    const int cl = (lane & 7) ^ ((rl >> 1) & 7);
    offA[i] = (uint32_t)((rl * g.lda + cl * 8) * 2);
    offB[i] = (uint32_t)((rl * g.ldb + cl * 8) * 2);
  }
  const uint32_t offTZ = (uint32_t)(((wave & 3) * 64 + lane) * g.ldz * 2);
  const int fr = lane & 15, fq = lane >> 4;

  auto tile_of = [&](int q, int& tm, int& tn) __attribute__((always_inline)) { tm = (q / g.tiles_n) * 8 + xcd; tn = q % g.tiles_n; };
  auto a_base = [&](int tm, int kt) __attribute__((always_inline)) { return reinterpret_cast<const char*>(g.A) + ((long)tm * BM2 * g.lda + kt * BK) * 2; };
  auto b_base = [&](int tn, int kt) __attribute__((always_inline)) { return reinterpret_cast<const char*>(g.B) + ((long)tn * BN2 * g.ldb + kt * BK) * 2; };
  auto issue_b = [&](int tn, int kt, int buf) __attribute__((always_inline)) {
    const char* bb = b_base(tn, kt);
    const uint32_t dst = lds0 + buf * (2 * TILE2_BYTES) + TILE2_BYTES + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16_s(offB[i], bb, dst + i * 1024);
  };
  auto issue_a = [&](int tm, int kt, int buf) __attribute__((always_inline)) {
    const char* ab = a_base(tm, kt);
    const uint32_t dst = lds0 + buf * (2 * TILE2_BYTES) + wave * 4096;
#pragma unroll
    for (int i = 0; i < 4; ++i) dma16_s(offA[i], ab, dst + i * 1024);
  };
  f32x4 acc[8][4];      // [fn][fm]: D[n][m]
  // With a prologue the raw z tile is staged like any A tile and the BatchNorm / ReLU / dropout is applied to the
  // operand fragments in registers (each fragment is read by the two waves of its row block: twice the arithmetic of
  // a staged transform, but no register-staged copy of the tile, no ds_write_b128 pass and the same two-stage
  // run-ahead at tile boundaries as the plain form).
  // One k step = 4 blocks of 16 MFMAs; the 8 LDS-DMA pieces of the next stage go out two per block (A first).  A step
  // that has nothing to request (the first of every tile: its stage 1 has been in flight since before the previous
  // epilogue) skips them with a uniform branch around the two instructions -- round 4; rounds 2-3 sent them to the wave's
  // idle epilogue slot (one 64 KB stage of L2 -> LDS traffic in eight for nothing, on the LDS pipe that bounds the loop)
  // because two inlined COPIES of this block -- with / without the pieces -- make the register allocator carry two sets of
  // accumulators; a branch inside the one copy does not.
  // What was measured on the way (tools/gemm_timeline.py, config-2 hidden layer, cycles per k step and SIMD): MFMAs
  // alone 2360 (128 x 16 = 2048 is the floor), + fragment reads 3130, + LDS-DMA pieces 4440 -- the three add up
  // whatever the order: all pieces at the top of the step, one or two per block, alternating between the two waves of
  // a SIMD; fragment reads one or two half blocks ahead of their MFMAs; accumulators in VGPRs or (hand-assigned) in
  // AGPRs; L2-warming touches two stages ahead; sched_group_barrier patterns (3 MFMAs : 1 LDS read); the same flops as
  // half as many v_mfma_f32_32x32x16_bf16.  None of these moved the step by more than 5 %.
  auto compute = [&](const unsigned char* ta, const unsigned char* tb, bool issue, const char* ab, const char* bb,
                     uint32_t dstbuf, int kt, int m0_, bool store_a, int store_f) __attribute__((always_inline)) {
    const uint32_t dst_a = issue ? dstbuf + wave * 4096 : lds0 + P_STAGE + wave * 2048;
    const uint32_t dst_b = issue ? dstbuf + TILE2_BYTES + wave * 4096 : lds0 + P_STAGE + wave * 2048;
    const uint32_t dst_step = issue ? 1024u : 0u;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      bf16x8 fa[4];
#pragma unroll
      for (int f = 0; f < 4; ++f)
        fa[f] = *reinterpret_cast<const bf16x8*>(ta + swz(wm * 64 + f * 16 + fr, kk * 4 + fq));    // activations
      if (PRO != PRO_NONE) {
        const int k = kt * BK + kk * 32 + fq * 8;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_scale + k), sc1 = *reinterpret_cast<const f32x4*>(s_scale + k + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_shift + k), sh1 = *reinterpret_cast<const f32x4*>(s_shift + k + 4);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          uint32_t bits8 = 0u;
          if (bt) bits8 = mask_buf(mcur)[(wm * 64 + f * 16 + fr) * 2 + kk] >> (fq * 8);
          fa[f] = transform_frag<PRO, DROP>(fa[f], sc0, sc1, sh0, sh1, pdrop,
                                            (uint32_t)(g.row0 + m0_ + wm * 64 + f * 16 + fr), (uint32_t)k, g.act, bits8);
        }
        // The transformed operand act(BN(z)) (* keep mask) exists only here, in registers.  The weight gradient of THIS
        // layer needs exactly it (dW = dz^T . pro(A)): written out once (a lane holds 8 consecutive k of one row = one
        // 16-byte store), the weight-gradient kernel reads it
        // back WITHOUT a prologue -- with Dropout its prologue would be a hash per element of the transposed fragments,
        // which is why that kernel fell back to the 128 x 128 register-staged form (0.56 vs 0.38 ms at M = 512 000).
        // Who stores: the SAME fragments exist in both waves of a row block (wn = 0, 1) and in every n-tile of the
        // M-tile, and the eight waves of the workgroup meet at a barrier every k step -- eight stores on one wave in four
        // made that wave the step's straggler.  The work is dealt out instead: wave wn takes the k half kk = wn, n-tile
        // tn takes the fragment rows f with f mod min(tiles_n, 4) = tn: two stores per wave and k step at N = 512.
        if (store_a && kk == wn) {
#pragma unroll
          for (int f = 0; f < 4; ++f)
            if ((store_f >> f) & 1)
              *reinterpret_cast<bf16x8*>(g.Aout + (long)(m0_ + wm * 64 + f * 16 + fr) * g.ldao + k) = fa[f];
        }
      }
#pragma unroll
      for (int hn = 0; hn < 2; ++hn) {
        const int blk = kk * 2 + hn;
        if (issue || (g.flags & PF_IDLE_DMA)) {      // (uniform; a branch around two instructions, no second copy of the block)
          if (blk < 2) {
            dma16_s(offA[2 * blk], ab, dst_a + (2 * blk) * dst_step);
            dma16_s(offA[2 * blk + 1], ab, dst_a + (2 * blk + 1) * dst_step);
          } else {
            dma16_s(offB[2 * blk - 4], bb, dst_b + (2 * blk - 4) * dst_step);
            dma16_s(offB[2 * blk - 3], bb, dst_b + (2 * blk - 3) * dst_step);
          }
        }
#pragma unroll
        for (int fp = 0; fp < 2; ++fp) {            // weight fragments two at a time (8 registers instead of 16)
          bf16x8 fb[2];
#pragma unroll
          for (int f = 0; f < 2; ++f)
            fb[f] = *reinterpret_cast<const bf16x8*>(tb + swz(wn * 128 + hn * 64 + (fp * 2 + f) * 16 + fr, kk * 4 + fq));
#pragma unroll
          for (int f = 0; f < 2; ++f)
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
              acc[hn * 4 + fp * 2 + f][fm] =
                  __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[f], fa[fm], acc[hn * 4 + fp * 2 + f][fm], 0, 0, 0);
        }
      }
    }
  };

  // the per-column epilogue coefficients of n-tile tn -> LDS (with tiles_n | nslots a workgroup keeps its n-tile)
  auto fill_epi = [&](int tn_) __attribute__((always_inline)) {
    if (tid < 256) {
      const int n = tn_ * BN2 + tid;
      if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
        const float rs = g.e_rstd[n];
        s_epi[tid] = g.e_scale[n]; s_epi[256 + tid] = g.e_shift[n];
        s_epi[512 + tid] = rs; s_epi[768 + tid] = -g.e_mean[n] * rs;
      } else {
        s_epi[tid] = g.bias ? g.bias[n] : 0.f;
      }
    }
  };

  // ---- first tile: stage 0 (and stage 1 when both operands go by LDS-DMA)
  int q = slot, tm, tn;
  tile_of(q, tm, tn);
  bool refill = false;
  fill_epi(tn);
  int p = 0;                                       // the buffer that holds stage 0 of the current tile
  issue_b(tn, 0, 0); issue_a(tm, 0, 0); issue_b(tn, 1, 1); issue_a(tm, 1, 1);
  if (PRO != PRO_NONE) {
    const float fold = bt ? pdrop.scale : 1.0f;     // (x s + h) * 2 == x (2 s) + 2 h exactly
    for (int k = tid; k < g.K; k += 512) { s_scale[k] = g.a_scale[k] * fold; s_shift[k] = g.a_shift[k] * fold; }
  }
  if (bt) fill_mask(0, tm, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  while (true) {
    const int qn = q + nslots;                     // this workgroup's next tile
    const bool have_next = qn < nq;
    int tmn = tm, tnn = tn;
    if (have_next) tile_of(qn, tmn, tnn);
    const int m0 = tm * BM2, n0 = tn * BN2;
    const long mb = (long)m0 + wm * 64;
    int store_f = 0;                               // bit f: this n-tile writes fragment row f of the transformed operand
    {
      const int tdiv = g.tiles_n < 4 ? g.tiles_n : 4;
#pragma unroll
      for (int f = 0; f < 4; ++f) store_f |= ((f % tdiv) == tn) ? (1 << f) : 0;
    }
#if (TFR_GEMM_ABLATE & 16)
    unsigned long long prof_t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    GEMM_STAMP(0);

#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    uint4 zq0 = make_uint4(0, 0, 0, 0), zq1 = zq0;
    if constexpr (PP != 0) {
      bf16x8 fa[4], fb[8];
      if (wn == 1) __builtin_amdgcn_s_barrier();      // the stagger of this tile (see the kernel's comment)
#pragma unroll 1
      for (int kt = 0; kt < nk; ++kt) {
        const int cur = p ^ (kt & 1), oth = cur ^ 1;
        const bool last = kt + 1 == nk;
        const bool issued = kt != 0 && (!last || have_next);
        const int stm = last ? tmn : tm, stn = last ? tnn : tn, skt = last ? 0 : kt + 1;
        if (touch_z && kt < 4)
          touch4_s(offTZ, reinterpret_cast<const char*>(g.Zp) + ((long)m0 * g.ldz + n0 + kt * 64) * 2, junk);
        if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && last) {            // first Zp chunk of the epilogue
          const char* zb = reinterpret_cast<const char*>(g.Zp) + (mb * g.ldz + n0 + wn * 128) * 2;
          zq0 = *reinterpret_cast<const uint4*>(zb + (uint32_t)(((lane >> 3) * g.ldz + (lane & 7) * 8) * 2));
          zq1 = *reinterpret_cast<const uint4*>(zb + (uint32_t)(((8 + (lane >> 3)) * g.ldz + (lane & 7) * 8) * 2));
        }
        const unsigned char* ta = smem + cur * (2 * TILE2_BYTES);
        const unsigned char* tb = ta + TILE2_BYTES;
        const char* ab = a_base(stm, skt);
        const char* bb = b_base(stn, skt);
        const uint32_t dst_a = lds0 + oth * (2 * TILE2_BYTES) + wave * 4096, dst_b = dst_a + TILE2_BYTES;
        const bool store_a = PRO != PRO_NONE && g.Aout != nullptr && tn < 4;
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          // ---------------- LOAD(kt, kk)
          if (PP == 1 && issued) {                      // (uniform) this wave's eight pieces of the next stage
            if (wn == 1) {
              if (kk == 0) {
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16_s(offA[i], ab, dst_a + i * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i) dma16_s(offB[i], bb, dst_b + i * 1024);
              }
            } else if (kk == 0) {
#pragma unroll
              for (int i = 0; i < 4; ++i) dma16_s(offA[i], ab, dst_a + i * 1024);
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i) dma16_s(offB[i], bb, dst_b + i * 1024);
            }
          }
#pragma unroll
          for (int f = 0; f < 4; ++f)
            fa[f] = *reinterpret_cast<const bf16x8*>(ta + swz(wm * 64 + f * 16 + fr, kk * 4 + fq));    // activations
#pragma unroll
          for (int j = 0; j < 8; ++j)
            fb[j] = *reinterpret_cast<const bf16x8*>(tb + swz(wn * 128 + j * 16 + fr, kk * 4 + fq));   // weights
          if (PRO != PRO_NONE) {
            const int k = kt * BK + kk * 32 + fq * 8;
            const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_scale + k), sc1 = *reinterpret_cast<const f32x4*>(s_scale + k + 4);
            const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_shift + k), sh1 = *reinterpret_cast<const f32x4*>(s_shift + k + 4);
#pragma unroll
            for (int f = 0; f < 4; ++f) {
              uint32_t bits8 = 0u;
              if (bt) bits8 = mask_buf(mcur)[(wm * 64 + f * 16 + fr) * 2 + kk] >> (fq * 8);
              fa[f] = transform_frag<PRO, DROP>(fa[f], sc0, sc1, sh0, sh1, pdrop,
                                                (uint32_t)(g.row0 + m0 + wm * 64 + f * 16 + fr), (uint32_t)k, g.act, bits8);
            }
            if (store_a && kk == wn) {                // (who stores what: see `compute` above)
#pragma unroll
              for (int f = 0; f < 4; ++f)
                if ((store_f >> f) & 1)
                  *reinterpret_cast<bf16x8*>(g.Aout + (long)(m0 + wm * 64 + f * 16 + fr) * g.ldao + k) = fa[f];
            }
          }
          if (bt && kk == 0 && (!last || have_next)) fill_mask(mcur ^ 1, stm, skt);   // the table of the NEXT step
          if (kk == 0 && kt == 1 && refill) fill_epi(tn);      // (rare: the n-tile changed; every wave is past the old epilogue)
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 1 && wn == 1) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
          // ---------------- MMA(kt, kk): registers only
          __builtin_amdgcn_s_setprio(1);
#pragma unroll
          for (int j = 0; j < 8; ++j) {
#pragma unroll
            for (int fm = 0; fm < 4; ++fm)
              acc[j][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[fm], acc[j][fm], 0, 0, 0);
            // PP == 2: the eight pieces of the next stage go out among the MFMAs of MMA(kt, 0), one per four MFMAs -- where
            // an LDS-DMA instruction is cheapest (MI355X_MICROARCH.md: ~60 cycles among bare MFMAs against 100-185 inside a
            // phase that also carries fragment reads); both groups wait for them before barrier 4 s + 4 (group 1 has its
            // L(s,1) phase in between, group 0 its L(s,1) and M(s,1))
            if (PP == 2 && kk == 0 && issued) {
              if (j < 4) dma16_s(offA[j], ab, dst_a + j * 1024);
              else dma16_s(offB[j - 4], bb, dst_b + (j - 4) * 1024);
            }
          }
          __builtin_amdgcn_s_setprio(0);
          __builtin_amdgcn_sched_barrier(0);
          if (kk == 1 && wn == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          if (!(kk == 1 && wn == 1 && last)) __builtin_amdgcn_s_barrier();
          __builtin_amdgcn_sched_barrier(0);
        }
        mcur ^= 1;
      }
    } else {
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = p ^ (kt & 1), oth = cur ^ 1;
      const bool last = kt + 1 == nk;
      // stage kt + 1, or the next tile's stage 0, goes to the other buffer during this step; the tile's stage 1 has
      // been in flight since before the previous epilogue, so step 0 requests nothing
      const bool issued = kt != 0 && (!last || have_next);
      const int stm = last ? tmn : tm, stn = last ? tnn : tn, skt = last ? 0 : kt + 1;
      if (touch_z && kt < 4)                        // the tile's Zp lines: 4 lines (256 columns) per row
        touch4_s(offTZ, reinterpret_cast<const char*>(g.Zp) + ((long)m0 * g.ldz + n0 + kt * 64) * 2, junk);
      if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && last) {            // first Zp chunk of the epilogue
        const char* zb = reinterpret_cast<const char*>(g.Zp) + (mb * g.ldz + n0 + wn * 128) * 2;
        zq0 = *reinterpret_cast<const uint4*>(zb + (uint32_t)(((lane >> 3) * g.ldz + (lane & 7) * 8) * 2));
        zq1 = *reinterpret_cast<const uint4*>(zb + (uint32_t)(((8 + (lane >> 3)) * g.ldz + (lane & 7) * 8) * 2));
      }
      compute(smem + cur * (2 * TILE2_BYTES), smem + cur * (2 * TILE2_BYTES) + TILE2_BYTES, issued, a_base(stm, skt),
              b_base(stn, skt), lds0 + oth * (2 * TILE2_BYTES), kt, m0, PRO != PRO_NONE && g.Aout != nullptr && tn < 4, store_f);
      if (kt == 1 && refill) {                     // (rare: the n-tile changed) every wave is past the old epilogue here
        asm volatile("" ::: "memory");
        fill_epi(tn);
      }
      if (bt && (!last || have_next)) fill_mask(mcur ^ 1, stm, skt);   // the table of the NEXT step (this or the next tile)
      // the pieces requested in this step must have landed (every wave's) before anyone reads them
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      mcur ^= 1;
    }
    }
    GEMM_STAMP(2);
    const int freeb = p ^ ((nk - 1) & 1);           // the buffer of the last step: free now
    const int pn = freeb ^ 1;                       // ... and the next tile's stage 0 sits in the other one
    if (have_next) { issue_b(tnn, 1, freeb); issue_a(tmn, 1, freeb); }

    // ---- epilogue: 2 halves (64 columns) x 4 chunks (16 rows) per wave through the wave's 2 KB slot
    // (per-lane offsets recomputed per tile: they would cost ten registers through the k loop)
    uint32_t offC[2], offZ[2], stg_rm[2], stg_acc[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                  // row-major 16-byte pieces of a [16][64] chunk, two per lane
      const int qq = lane + 64 * i, row = qq >> 3, cc = qq & 7;
      offC[i] = (uint32_t)((row * g.ldc + cc * 8) * 2);
      offZ[i] = (uint32_t)((row * g.ldz + cc * 8) * 2);
      stg_rm[i] = (uint32_t)(row * 128 + ((cc ^ (row & 7)) << 4));
    }
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)                 // the lane's 8-byte slot of fragment column fn in the chunk
      stg_acc[fn] = (uint32_t)(fr * 128 + (((fn * 2 + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8);
    bool first_store = true;
#pragma clang loop unroll(full)
    for (int h = 0; h < 2; ++h) {
      const int nl = wn * 128 + h * 64;             // column of the half inside the tile
      f32x4 pb[4], pe[4];                           // bias | BN scale, shift of the layer below (relu mask)
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        pb[fn] = *reinterpret_cast<const f32x4*>(s_epi + nl + fn * 16 + fq * 4);
        if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) pe[fn] = *reinterpret_cast<const f32x4*>(s_epi + 256 + nl + fn * 16 + fq * 4);
      }
      f32x4 s1[4], s2[4];
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) { s1[fn] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[fn] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      char* cb = reinterpret_cast<char*>(g.C) + (mb * g.ldc + n0 + nl) * 2;
      const char* zb = reinterpret_cast<const char*>(g.Zp) + (mb * g.ldz + n0 + nl) * 2;
#pragma unroll
      for (int fm = 0; fm < 4; ++fm) {
        if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
          *reinterpret_cast<uint4*>(sw + stg_rm[0]) = zq0;
          *reinterpret_cast<uint4*>(sw + stg_rm[1]) = zq1;
          if (!(h == 1 && fm == 3)) {               // next chunk (next 16 rows, or the other half's first rows)
            const char* zn = (fm < 3) ? zb + (long)(fm + 1) * 16 * g.ldz * 2 : zb + 64 * 2;
            zq0 = *reinterpret_cast<const uint4*>(zn + offZ[0]);
            zq1 = *reinterpret_cast<const uint4*>(zn + offZ[1]);
          }
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          f32x4 v = acc[h * 4 + fn][fm];
          if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {
            const uint2 zz = *reinterpret_cast<const uint2*>(sw + stg_acc[fn]);
            const f32x4 z = {bf16_lo(zz.x), bf16_hi(zz.x), bf16_lo(zz.y), bf16_hi(zz.y)};
            const f32x4 y = z * pb[fn] + pe[fn];
            if (DROP) {
              float kf[4];
              drop_run<4, DROP == 3>(edrop, (uint32_t)(g.row0 + mb + fm * 16 + fr), (uint32_t)(n0 + nl + fn * 16 + fq * 4), kf);
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] *= kf[r];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = (EPI == EPI_ACT_BWD) ? v[r] * act_grad(g.act, y[r]) : (y[r] > 0.f ? v[r] : 0.f);
            s1[fn] += v;
            s2[fn] += v * z;
          } else {
            v += pb[fn];
            if (EPI == EPI_STATS) { s1[fn] += v; s2[fn] += v * v; }
          }
          *reinterpret_cast<uint2*>(sw + stg_acc[fn]) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
        }
        if (first_store) {                          // the next tile's two stages have had the first chunk's time
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          first_store = false;
        }
        const uint4 o0 = *reinterpret_cast<const uint4*>(sw + stg_rm[0]);
        const uint4 o1 = *reinterpret_cast<const uint4*>(sw + stg_rm[1]);
        char* cc = cb + (long)fm * 16 * g.ldc * 2;
        *reinterpret_cast<uint4*>(cc + offC[0]) = o0;
        *reinterpret_cast<uint4*>(cc + offC[1]) = o1;
      }
      if (h == 0) GEMM_STAMP(3);
      if (h == 1) GEMM_STAMP(4);
      if (EPI != EPI_PLAIN) {
        float* const st = g.stats + ((long)(tm * 4 + wm) * 2) * g.N + n0 + nl;      // one row of partials per 64-row slab
#pragma unroll
        for (int fn = 0; fn < 4; ++fn) {
          f32x4 a = row16_sum4(s1[fn]), b = row16_sum4(s2[fn]);
          if ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD)) {                // sum dy * zhat = rstd * sum dy z - mean rstd * sum dy
            const f32x4 er = *reinterpret_cast<const f32x4*>(s_epi + 512 + nl + fn * 16 + fq * 4);
            const f32x4 c2 = *reinterpret_cast<const f32x4*>(s_epi + 768 + nl + fn * 16 + fq * 4);
            b = b * er + a * c2;
          }
          if (fr == 15) {
            *reinterpret_cast<f32x4*>(st + fn * 16 + fq * 4) = a;
            *reinterpret_cast<f32x4*>(st + g.N + fn * 16 + fq * 4) = b;
          }
        }
      }
    }
    GEMM_STAMP(5);
#if (TFR_GEMM_ABLATE & 16)
    if (tid == 0 && g_prof_gemm) {
      prof_t[1] = prof_t[0];
      prof_t[6] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);
      prof_t[7] = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);
      for (int i = 0; i < 8; ++i) g_prof_gemm[(size_t)(tm * g.tiles_n + tn) * 8 + i] = prof_t[i];
    }
#endif
    if (!have_next) break;
    refill = tnn != tn;
    q = qn; tm = tmn; tn = tnn; p = pn;
  }
}

#include "tower_gemm_rp.h"
#include "tower_gemm_bs.h"

#ifdef TFR_BS_STAMPS
}  // namespace
extern "C" int tfr_prof_set_buffer_bs(void* device_u64_buffer) {
  unsigned long long* p = (unsigned long long*)device_u64_buffer;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_bs), &p, sizeof(p));
}
namespace {
#endif

#if (TFR_RP_ABLATE & 16)
}  // namespace
extern "C" int tfr_prof_set_buffer_rp(void* device_u64_buffer) {
  unsigned long long* p = (unsigned long long*)device_u64_buffer;
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_prof_rp), &p, sizeof(p));
}
namespace {
#endif

#ifdef TFR_RP_DEV
// developer aid: `hipcc -DTFR_RP_DEV=P,E,D,NK -S tower.hip` compiles ONE instantiation of the resident-panel kernel (20 s instead
// of the 2.5 min of the whole translation unit) for reading its ISA / register use
#ifdef TFR_BS_DEV
template __global__ void tower_gemm_bs_kernel<TFR_BS_DEV>(const GemmArgs);
#else
template __global__ void tower_gemm_rp_kernel<TFR_RP_DEV>(const GemmArgs);
#endif
}  // namespace
#else

// ------------------------------------------------------------------------------------
// fp32 [M, F] -> bf16 [M, Kp] (zero padded columns), optional per-column affine (input BN).  TIn = uint16_t: the
// features arrive as bf16 already (the host parser's bf16 ingest) -- the same gather, padding and
// affine, the element widened exactly; without an affine the values pass through bit for bit.
__device__ __forceinline__ float feat_f32(float v) { return v; }
__device__ __forceinline__ float feat_f32(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
template <typename TIn>
__global__ void tower_cast_kernel(const TIn* __restrict__ x, long ldx, int M, int F, int Kp,
                                  const float* __restrict__ scale, const float* __restrict__ shift,
                                  const int* __restrict__ row_index, uint16_t* __restrict__ out) {
  const long chunks_per_row = Kp / 8;
  const long total = (long)M * chunks_per_row;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long m = q / chunks_per_row;
    const long ms = row_index ? (long)row_index[m] : m;        // FlattenList's circular padding: a row gather
    const int k = (int)(q % chunks_per_row) * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float t = (k + e < F) ? feat_f32(x[ms * ldx + k + e]) : 0.f;
      if (scale && k + e < F) t = __builtin_fmaf(t, scale[k + e], shift[k + e]);
      v[e] = t;
    }
    *reinterpret_cast<uint4*>(out + m * Kp + k) =
        make_uint4(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7]));
  }
}

// fp32 [R, C] -> bf16 [R, Cp] row-major, or its transpose bf16 [C, Rp] (weights: once per step).
__global__ void tower_weight_cast_kernel(const float* __restrict__ w, int R, int C, int transpose,
                                         int pitch, uint16_t* __restrict__ out) {
  const int total = transpose ? C * pitch : R * pitch;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const int orow = q / pitch, ocol = q % pitch;
    float v = 0.f;
    if (!transpose) { if (ocol < C) v = w[(long)orow * C + ocol]; }
    else            { if (ocol < R) v = w[(long)ocol * C + orow]; }
    out[q] = (uint16_t)(pack_bf16(v, 0.f) & 0xffffu);
  }
}

// The same for up to 8 matrices in one launch (blockIdx.y = matrix): every weight cast of a training step
// (forward operands and the transposed dgrad operands) together.
struct WCastBatch { const float* w[8]; uint16_t* out[8]; int R[8], C[8], transpose[8], pitch[8]; int* step; int* step_copy; };
__global__ void tower_weight_cast_batch_kernel(const WCastBatch a) {
  // the training-step counter of the Dropout masks advances HERE (round 6: it was an add_ and a clone launch of their own):
  // nothing earlier in the step reads it, everything later reads the copy
  if (a.step && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { const int c = *a.step + 1; *a.step = c; *a.step_copy = c; }
  const int j = blockIdx.y;
  const float* __restrict__ w = a.w[j];
  uint16_t* __restrict__ out = a.out[j];
  const int R = a.R[j], C = a.C[j], transpose = a.transpose[j], pitch = a.pitch[j];
  const int total = transpose ? C * pitch : R * pitch;
  for (int q = blockIdx.x * blockDim.x + threadIdx.x; q < total; q += gridDim.x * blockDim.x) {
    const int orow = q / pitch, ocol = q % pitch;
    float v = 0.f;
    if (!transpose) { if (ocol < C) v = w[(long)orow * C + ocol]; }
    else            { if (ocol < R) v = w[(long)ocol * C + orow]; }
    out[q] = (uint16_t)(pack_bf16(v, 0.f) & 0xffffu);
  }
}

// Column sums of per-workgroup partials, stage 1: out[c][i] = sum_{t in chunk c} partial[t][i].
// Grid (ceil(W / 256), ceil(T / 64)); fully coalesced, deterministic (fixed order), so that the
// finishing kernels below only walk ceil(T / 64) rows.
constexpr int kReduceChunk = 64;
__global__ void tower_reduce_rows_kernel(const float* __restrict__ partial, int T, int W, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W) return;
  const int t0 = blockIdx.y * kReduceChunk;
  const int t1 = (t0 + kReduceChunk < T) ? t0 + kReduceChunk : T;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int t = t0;
  for (; t + 15 < t1; t += 16) {                               // 16 loads in flight (the walk is latency-bound)
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = partial[(long)(t + u) * W + i];
#pragma unroll
    for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
  }
  for (; t + 3 < t1; t += 4) {
    s0 += partial[(long)t * W + i];       s1 += partial[(long)(t + 1) * W + i];
    s2 += partial[(long)(t + 2) * W + i]; s3 += partial[(long)(t + 3) * W + i];
  }
  for (; t < t1; ++t) s0 += partial[(long)t * W + i];
  out[(long)blockIdx.y * W + i] = (s0 + s1) + (s2 + s3);
}

// BatchNorm statistics: partial [T][2][N] (fp32) -> mean, biased variance (fp64 combine),
// scale = gamma * rsqrt(var + eps), shift = beta - mean * scale; moving averages updated in
// place (Keras: moving = moving * momentum + batch * (1 - momentum); biased variance).
__global__ void tower_bn_finalize_kernel(const float* __restrict__ partial, int T, int N, long M,
                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                         float eps, float momentum, float* __restrict__ moving_mean,
                                         float* __restrict__ moving_var, float* __restrict__ scale,
                                         float* __restrict__ shift, float* __restrict__ mean_out,
                                         float* __restrict__ rstd_out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  double s = 0.0, ss = 0.0;
  int t = 0;
  for (; t + 7 < T; t += 8) {                                  // 16 loads in flight, same summation order
    float a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = partial[((long)(t + u) * 2 + 0) * N + n];
      b[u] = partial[((long)(t + u) * 2 + 1) * N + n];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += (double)a[u]; ss += (double)b[u]; }
  }
  for (; t < T; ++t) {
    s += (double)partial[((long)t * 2 + 0) * N + n];
    ss += (double)partial[((long)t * 2 + 1) * N + n];
  }
  const double mean = s / (double)M;
  double var = ss / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma ? gamma[n] : 1.f, be = beta ? beta[n] : 0.f;
  const float sc = ga * rstd;
  scale[n] = sc;
  shift[n] = be - (float)mean * sc;
  mean_out[n] = (float)mean;
  rstd_out[n] = rstd;
  if (moving_mean) moving_mean[n] = moving_mean[n] * momentum + (float)mean * (1.f - momentum);
  if (moving_var) moving_var[n] = moving_var[n] * momentum + (float)var * (1.f - momentum);
}

// Finishes per-workgroup column partials: out[i] = sum_t partial[t][i], i < W (fp64 combine).
__global__ void tower_reduce_partials_kernel(const float* __restrict__ partial, int T, int W,
                                             float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= W) return;
  double s = 0.0;
  int t = 0;
  for (; t + 15 < T; t += 16) {
    float v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = partial[(long)(t + u) * W + i];
#pragma unroll
    for (int u = 0; u < 16; ++u) s += (double)v[u];
  }
  for (; t < T; ++t) s += (double)partial[(long)t * W + i];
  out[i] = (float)s;
}

// Short partial matrices (T <= kFusedRows rows: the launch-bound steps, M = 25600 in BASELINE config 5) finish in ONE
// launch instead of reduce_rows + the finishing kernel: a 1024-thread workgroup owns 64 columns, its 16 waves take the
// rows round-robin (row t goes to wave t mod 16: a wave walks <= 64 rows, 16 loads in flight), the 16 partial sums of a
// column meet in LDS and wave 0 adds them in wave order in fp64.  Fixed association, run-to-run identical; it differs
// from the two-launch path (chunks of 64 rows) in the last bits only.  (A first version that kept the 64-row chunks
// -- 4 waves, <= 16 chunks -- took 10.5 us per launch at T = 400: as long as the two launches it replaced.)
constexpr int kFusedRows = 1024;
constexpr int kFusedWaves = 16;

// sums[j] (double) of column n = blockIdx.x * 64 + lane of the J stacked [N]-wide rows of partial [T][J][N]; valid in
// wave 0 after the call (all 1024 threads must call).
template <int JMAX>
__device__ __forceinline__ void fused_column_sums(const float* __restrict__ partial, int T, int J, int N, int n,
                                                  float (*part)[JMAX][64], double (&sums)[JMAX]) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long W = (long)J * N;
  const bool live = n < N;
  for (int j = 0; j < J; ++j) {
    const long col = (long)j * N + n;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int t = wave;
    if (live) {
      for (; t + 15 * kFusedWaves < T; t += 16 * kFusedWaves) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = partial[(long)(t + u * kFusedWaves) * W + col];
#pragma unroll
        for (int u = 0; u < 16; u += 4) { s0 += v[u]; s1 += v[u + 1]; s2 += v[u + 2]; s3 += v[u + 3]; }
      }
      if (t < T) {                                            // the remaining (< 16) rows: all loads first (round 6 -- they were
        float w[16];                                          // up to 15 DEPENDENT loads, ~0.3 us each: T = 400 has nine)
#pragma unroll
        for (int u = 0; u < 16; ++u) { const int tt = t + u * kFusedWaves; w[u] = tt < T ? partial[(long)tt * W + col] : 0.0f; }
#pragma unroll
        for (int u = 0; u < 16; ++u) s0 += w[u];              // same order as the one-at-a-time loop (+ 0.0f is exact)
      }
    }
    part[wave][j][lane] = (s0 + s1) + (s2 + s3);
  }
  __syncthreads();
  if (wave == 0) {
    for (int j = 0; j < J; ++j) {
      double s = 0.0;
#pragma unroll
      for (int w = 0; w < kFusedWaves; ++w) s += (double)part[w][j][lane];
      sums[j] = s;
    }
  }
}

__global__ __launch_bounds__(1024) void tower_bn_finalize_fused_kernel(
    const float* __restrict__ partial, int T, int N, long M, const float* __restrict__ gamma,
    const float* __restrict__ beta, float eps, float momentum, float* __restrict__ moving_mean,
    float* __restrict__ moving_var, float* __restrict__ scale, float* __restrict__ shift,
    float* __restrict__ mean_out, float* __restrict__ rstd_out) {
  __shared__ float part[kFusedWaves][2][64];
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  double sums[2];
  fused_column_sums<2>(partial, T, 2, N, n, part, sums);
  if ((threadIdx.x >> 6) != 0 || n >= N) return;
  const double mean = sums[0] / (double)M;
  double var = sums[1] / (double)M - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float ga = gamma ? gamma[n] : 1.f, be = beta ? beta[n] : 0.f;
  const float sc = ga * rstd;
  scale[n] = sc;
  shift[n] = be - (float)mean * sc;
  mean_out[n] = (float)mean;
  rstd_out[n] = rstd;
  if (moving_mean) moving_mean[n] = moving_mean[n] * momentum + (float)mean * (1.f - momentum);
  if (moving_var) moving_var[n] = moving_var[n] * momentum + (float)var * (1.f - momentum);
}

// out[j][n] = sum_t partial[t][j][n] for the J <= 6 stacked rows, and -- when gamma is given -- the BatchNorm-backward
// coefficients pqr[3][N] of tower_bn_bwd_coeffs_kernel from rows 0 / 1 (sum dy, sum dy * zhat) in the same launch.
// `dl` (optional, round 6): one more workgroup adds up the columns of dl[Mr][O] (the output layer's bias gradient, O <= 4) --
// it was a torch reduction launch of its own (12.8 us of the 0.49 ms config-5 step).  Fixed order: thread t takes rows t,
// t + 1024, ... in fp64, the 64 lanes of a wave meet in a shuffle tree, wave 0 adds the 16 wave sums in wave order.
__global__ __launch_bounds__(1024) void tower_reduce_partials_fused_kernel(
    const float* __restrict__ partial, int T, int J, int N, float* __restrict__ out, const float* __restrict__ gamma,
    const float* __restrict__ rstd, const float* __restrict__ mean, float inv_m, float* __restrict__ pqr,
    const float* __restrict__ dl, long Mr, int O, float* __restrict__ db) {
  __shared__ float part[kFusedWaves][6][64];
  if (dl && blockIdx.x == gridDim.x - 1) {
    double (*ws)[4] = reinterpret_cast<double (*)[4]>(&part[0][0][0]);      // [16 waves][4] doubles
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const long nel = Mr * O;
    if ((O == 1 || O == 2 || O == 4) && (nel & 3) == 0 && (reinterpret_cast<uintptr_t>(dl) & 15) == 0) {
      // the matrix as a flat run of float4: element e belongs to column e % O, a fixed pattern inside a float4; eight
      // 16-byte loads of a thread in flight -- [25 600, 2] is two memory round trips (25 dependent loads took 11 us; sixteen in
      // flight pushed the kernel to 128 registers and into scratch)
      const float4* p4 = reinterpret_cast<const float4*>(dl);
      const long n4 = nel >> 2;
      for (long k0 = threadIdx.x; k0 < n4; k0 += 8 * 1024) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const long k = k0 + (long)u * 1024; v[u] = k < n4 ? p4[k] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (O == 1) { acc[0] += (double)v[u].x; acc[0] += (double)v[u].y; acc[0] += (double)v[u].z; acc[0] += (double)v[u].w; }
          else if (O == 2) { acc[0] += (double)v[u].x; acc[1] += (double)v[u].y; acc[0] += (double)v[u].z; acc[1] += (double)v[u].w; }
          else { acc[0] += (double)v[u].x; acc[1] += (double)v[u].y; acc[2] += (double)v[u].z; acc[3] += (double)v[u].w; }
        }
      }
    } else {
      for (long m0 = threadIdx.x; m0 < Mr; m0 += 8 * 1024) {
        float v[8][4];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const long m = m0 + (long)u * 1024;
#pragma unroll
          for (int o = 0; o < 4; ++o) v[u][o] = (m < Mr && o < O) ? dl[m * O + o] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
          for (int o = 0; o < 4; ++o) acc[o] += (double)v[u][o];
      }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 0; o < O; ++o) {
      double v = acc[o];
      for (int d = 32; d >= 1; d >>= 1) v += __shfl_down(v, d, 64);
      if (lane == 0) ws[wave][o] = v;
    }
    __syncthreads();
    if (threadIdx.x < O) {
      double t = 0.0;
      for (int w = 0; w < kFusedWaves; ++w) t += ws[w][threadIdx.x];
      db[threadIdx.x] = (float)t;
    }
    return;
  }
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  double sums[6];
  fused_column_sums<6>(partial, T, J, N, n, part, sums);
  if ((threadIdx.x >> 6) != 0 || n >= N) return;
  for (int j = 0; j < J; ++j) out[(long)j * N + n] = (float)sums[j];
  if (gamma) {
    const float s = gamma[n] * rstd[n];
    const float c1 = (float)sums[0], c2 = (float)sums[1];
    pqr[n] = s;
    pqr[N + n] = -s * rstd[n] * c2 * inv_m;
    pqr[2 * N + n] = s * (rstd[n] * c2 * mean[n] - c1) * inv_m;
  }
}

// Output layer: logits[m, o] = sum_k act(z[m, k]) * w[o, k] + b[o], O <= 4 (GEMV class:
// one read of z).  A row is owned by 16 lanes, 8 bf16 (16 B) per lane per trip; the
// per-column coefficients live in LDS as [K/8][(2 + O)][8] floats (one float4 pair each).
template <int PRO>
__global__ __launch_bounds__(256) void tower_out_kernel(const uint16_t* __restrict__ z, long ldz, int M, int K,
                                                        const float* __restrict__ scale,
                                                        const float* __restrict__ shift,
                                                        const float* __restrict__ w, const float* __restrict__ b,
                                                        int O, float* __restrict__ out, const Drop drop_in, const int act) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Drop drop = drop_resolve(drop_in);
  float* coef = reinterpret_cast<float*>(smem);          // [K/8][2 + O][8]
  const int stride = (2 + O) * 8;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    float* cc = coef + (k >> 3) * stride + (k & 7);
    cc[0] = (PRO != PRO_NONE) ? scale[k] : 1.f;
    cc[8] = (PRO != PRO_NONE) ? shift[k] : 0.f;
    for (int o = 0; o < O; ++o) cc[16 + 8 * o] = w[(long)o * K + k];
  }
  __syncthreads();
  const int lane16 = threadIdx.x & 15;
  const long row_in_grid = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const long rows_per_grid = ((long)gridDim.x * blockDim.x) >> 4;
  for (long m = row_in_grid; m < M; m += rows_per_grid) {   // the 16 lanes of a row share m
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k = lane16 * 8; k < K; k += 128) {
      const uint4 v = *reinterpret_cast<const uint4*>(z + m * ldz + k);
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
      const float* cc = coef + (k >> 3) * stride;
      float a[8];
#pragma unroll
      for (int i = 0; i < 4; ++i) { a[2 * i] = bf16_lo(u[i]); a[2 * i + 1] = bf16_hi(u[i]); }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float t = a[e];
        if (PRO != PRO_NONE) t = __builtin_fmaf(t, cc[e], cc[8 + e]);
        if (PRO == PRO_AFFINE_RELU) t = fmaxf(t, 0.f);
        if (PRO == PRO_AFFINE_ACT) t = act_fwd(act, t);
        a[e] = t;
      }
      if (PRO != PRO_NONE && drop.thr) {
        float kf[8];
        drop_run<8>(drop, (uint32_t)m, (uint32_t)k, kf);
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= kf[e];
      }
      for (int o = 0; o < O; ++o) {
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[o] = __builtin_fmaf(a[e], cc[16 + 8 * o + e], acc[o]);
      }
    }
#pragma unroll
    for (int o = 0; o < 4; ++o)
#pragma unroll
      for (int s = 8; s > 0; s >>= 1) acc[o] += __shfl_xor(acc[o], s, 64);
    if (lane16 == 0)
      for (int o = 0; o < O; ++o) out[m * O + o] = acc[o] + (b ? b[o] : 0.f);
  }
}

// ====================================================================================
// Backward.
//
// Per hidden layer l (z_l = a_{l-1} W_l^T + b_l, y_l = BN(z_l), a_l = relu(y_l)):
//   dy_l  = da_l * 1[y_l > 0]                       (fused: output-layer backward / dgrad epilogue)
//   c1 = sum_m dy_l, c2 = sum_m dy_l * zhat_l       (column partials from the same epilogue)
//   dz_l  = scale * (dy_l - c1/M - zhat_l * c2/M) = p*dy + q*z + r  per column  (bn_bwd_apply)
//   dW_l  = dz_l^T a_{l-1}  (wgrad: transposed-operand MFMA GEMM, split over M)
//   da_{l-1} = dz_l W_l     (dgrad: the forward GEMM kernel on W_l^T, EPI_RELU_BWD)
//   d gamma = c2, d beta = c1, d b_l = sum_m dz_l (= 0 under BatchNorm).

// Output layer backward.  logits[m, o] = sum_k a[m, k] w[o, k] + b[o], a = act(z):
//   dy[m, k] = (sum_o dlogits[m, o] w[o, k]) * 1[y > 0]   -> bf16 [M, K]
//   partial[blk][0][k] = sum_m dy, [1][k] = sum_m dy * zhat, [2 + o][k] = sum_m dlogits[m, o] a[m, k]
// MODE 0: dy and partial (the plain backward).  MODE 1: partial only -- first pass of the BatchNorm backward,
// nothing of size [M, K] is written.  MODE 2: second pass, d is recomputed and dz = p * bf16(d) + q * z + r
// goes straight out (instead of: write dy, read dy + z, write dz).
// LPR lanes share one row (8 columns each): 64 = a wave reads 1 KB of a row (K >= 512), 16 for narrower layers.
// OT = O when 1 (the usual single logit), else 4 (loops run to O).
template <int PRO, int OT, int LPR, int MODE>
__global__ __launch_bounds__(256) void tower_out_bwd_kernel(
    const uint16_t* __restrict__ z, long ldz, int M, int K, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ w, const float* __restrict__ dlogits, int O, uint16_t* __restrict__ dy, long lddy,
    float* __restrict__ partial, int rows_per_block, const Drop drop_in, const float* __restrict__ pqr, const int act,
    const int sr) {
  const Drop drop = drop_resolve(drop_in);
  constexpr int G = 256 / LPR, CW = LPR * 8;                   // row groups per block, columns per pass
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* red = reinterpret_cast<float*>(smem);                 // [G][(2 + O) * CW]
  const int lane = threadIdx.x % LPR;
  const int grp = (LPR == 64) ? __builtin_amdgcn_readfirstlane(threadIdx.x >> 6) : (int)(threadIdx.x / LPR);
  const long mb = (long)blockIdx.x * rows_per_block;
  const long me = (mb + rows_per_block < M) ? mb + rows_per_block : M;
  const int J = 2 + O;
  for (int kp = 0; kp < K; kp += CW) {
    const int k = kp + lane * 8;
    const bool kin = k < K;
    float sc[8], sh[8], mu[8], rs[8], ww[OT][8], cp[8], cq[8], cr[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      cp[e] = (kin && MODE == 2) ? pqr[k + e] : 1.f;
      cq[e] = (kin && MODE == 2) ? pqr[K + k + e] : 0.f;
      cr[e] = (kin && MODE == 2) ? pqr[2 * K + k + e] : 0.f;
      sc[e] = (kin && PRO != PRO_NONE) ? scale[k + e] : 1.f;
      sh[e] = (kin && PRO != PRO_NONE) ? shift[k + e] : 0.f;
      mu[e] = (kin && mean && MODE != 2) ? mean[k + e] : 0.f;
      rs[e] = (kin && rstd && MODE != 2) ? rstd[k + e] : 1.f;
#pragma unroll
      for (int o = 0; o < OT; ++o) ww[o][e] = (kin && o < O) ? w[(long)o * K + k + e] : 0.f;
    }
    float s1[8], s2[8], dw[OT][8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      s1[e] = 0.f; s2[e] = 0.f;
#pragma unroll
      for (int o = 0; o < OT; ++o) dw[o][e] = 0.f;
    }
    auto row = [&](long m, const uint4 v, const float (&dl)[OT]) {
      const uint32_t u[4] = {v.x, v.y, v.z, v.w};
      float out[8], kf[8] = {1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f, 1.f};
      if (PRO != PRO_NONE && drop.thr) drop_run<8>(drop, (uint32_t)m, (uint32_t)k, kf);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float zz = (e & 1) ? bf16_hi(u[e >> 1]) : bf16_lo(u[e >> 1]);
        const float y = __builtin_fmaf(zz, sc[e], sh[e]);
        const float a = ((PRO == PRO_AFFINE_RELU) ? fmaxf(y, 0.f) : (PRO == PRO_AFFINE_ACT) ? act_fwd(act, y) : y) * kf[e];
        float da = 0.f;
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          da = __builtin_fmaf(dl[o], ww[o][e], da);
          if (MODE != 2) dw[o][e] = __builtin_fmaf(dl[o], a, dw[o][e]);
        }
        const float d = (PRO == PRO_AFFINE_RELU && !(y > 0.f)) ? 0.f : (PRO == PRO_AFFINE_ACT) ? da * kf[e] * act_grad(act, y) : da * kf[e];
        if (MODE != 2) {
          s1[e] += d;
          s2[e] = __builtin_fmaf(d, (zz - mu[e]) * rs[e], s2[e]);
          out[e] = d;
        } else {                                               // the value the two-kernel path produces, bit for bit
          const float db = bf16_lo(pack_bf16(d, 0.f));
          out[e] = __builtin_fmaf(cp[e], db, __builtin_fmaf(cq[e], zz, cr[e]));
        }
      }
      if (MODE == 2 && sr) {                                   // dz: unbiased rounding (see pack_bf16_sr)
        uint32_t da, db;
        sr_dither8((uint32_t)m, (uint32_t)k >> 3, da, db);
        *reinterpret_cast<uint4*>(dy + m * lddy + k) =
            make_uint4(pack_bf16_sr(out[0], out[1], da, db, 0), pack_bf16_sr(out[2], out[3], da, db, 2),
                       pack_bf16_sr(out[4], out[5], da, db, 4), pack_bf16_sr(out[6], out[7], da, db, 6));
      } else if (MODE != 1)
        *reinterpret_cast<uint4*>(dy + m * lddy + k) =
            make_uint4(pack_bf16(out[0], out[1]), pack_bf16(out[2], out[3]), pack_bf16(out[4], out[5]), pack_bf16(out[6], out[7]));
    };
    if (kin) {
      long m = mb + grp;
      for (; m + G < me; m += 2 * G) {                         // two rows in flight
        const uint4 v0 = *reinterpret_cast<const uint4*>(z + m * ldz + k);
        const uint4 v1 = *reinterpret_cast<const uint4*>(z + (m + G) * ldz + k);
        float d0[OT], d1[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) {
          d0[o] = (o < O) ? dlogits[m * O + o] : 0.f;
          d1[o] = (o < O) ? dlogits[(m + G) * O + o] : 0.f;
        }
        row(m, v0, d0);
        row(m + G, v1, d1);
      }
      if (m < me) {
        const uint4 v0 = *reinterpret_cast<const uint4*>(z + m * ldz + k);
        float d0[OT];
#pragma unroll
        for (int o = 0; o < OT; ++o) d0[o] = (o < O) ? dlogits[m * O + o] : 0.f;
        row(m, v0, d0);
      }
    }
    if (MODE == 2) continue;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float* rr = red + grp * (J * CW) + lane * 8 + e;
      rr[0] = s1[e]; rr[CW] = s2[e];
#pragma unroll
      for (int o = 0; o < OT; ++o) if (o < O) rr[(2 + o) * CW] = dw[o][e];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < J * CW; i += 256) {
      const int jj = i / CW, kk = kp + (i % CW);
      if (kk < K) {
        float t = 0.f;
#pragma unroll
        for (int gq = 0; gq < G; ++gq) t += red[gq * (J * CW) + i];
        partial[((long)blockIdx.x * J + jj) * K + kk] = t;
      }
    }
  }
}

// dz = p[k] * dy + q[k] * z + r[k], in place over dy (bf16 [M, K]).
__global__ __launch_bounds__(256) void tower_bn_bwd_apply_kernel(uint16_t* __restrict__ dy, long lddy,
                                                                 const uint16_t* __restrict__ z, long ldz,
                                                                 int M, int K, const float* __restrict__ pqr,
                                                                 const int sr) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* c = reinterpret_cast<float*>(smem);                   // [3][K]
  for (int i = threadIdx.x; i < 3 * K; i += blockDim.x) c[i] = pqr[i];
  __syncthreads();
  const int cpr = K / 8;
  const long total = (long)M * cpr;
  for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (long)gridDim.x * blockDim.x) {
    const long m = q / cpr;
    const int k = (int)(q % cpr) * 8;
    const uint4 a = *reinterpret_cast<const uint4*>(dy + m * lddy + k);
    const uint4 b = *reinterpret_cast<const uint4*>(z + m * ldz + k);
    const uint32_t ua[4] = {a.x, a.y, a.z, a.w}, ub[4] = {b.x, b.y, b.z, b.w};
    uint32_t o[4];
    uint32_t da = 0, db = 0;
    if (sr) sr_dither8((uint32_t)m, (uint32_t)k >> 3, da, db);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int k0 = k + 2 * i, k1 = k0 + 1;
      const float lo = __builtin_fmaf(c[k0], bf16_lo(ua[i]), __builtin_fmaf(c[K + k0], bf16_lo(ub[i]), c[2 * K + k0]));
      const float hi = __builtin_fmaf(c[k1], bf16_hi(ua[i]), __builtin_fmaf(c[K + k1], bf16_hi(ub[i]), c[2 * K + k1]));
      o[i] = sr ? pack_bf16_sr(lo, hi, da, db, 2 * i) : pack_bf16(lo, hi);
    }
    *reinterpret_cast<uint4*>(dy + m * lddy + k) = make_uint4(o[0], o[1], o[2], o[3]);
  }
}

// Weight gradient: dW[n, k] = sum_m dz[m, n] * pro(A)[m, k], both operands row-major in m
// (the contraction index is the ROW of both), so the MFMA fragments are read from LDS with
// the gfx950 transpose load ds_read_b64_tr_b16: the 16 lanes of a group each supply the
// address of an 8-byte piece of a [4 rows][16 cols] block and receive one COLUMN of it.
// Tile 128 (n) x 128 (k), 64 rows of m per step, double-buffered LDS, split over M
// (blockIdx.z) into fp32 slabs [split][N][ldw] that tower_slab_reduce sums.
struct WgradArgs {
  const uint16_t* DZ; long lddz;     // [M, N]
  const uint16_t* A; long lda;       // [M, K] (pre-BN z of the layer below when PRO != 0)
  const float* a_scale; const float* a_shift;
  float* slab; long ldw;             // [splits][N][ldw]
  Drop drop;                         // dropout of the layer that produced A (prologue)
  int M, N, K, rows_per_split, splits, tiles_n, tiles_k;
  int act;                           // ACT_* of PRO_AFFINE_ACT
};

// byte offset of 16-byte chunk cc (0..15) of row r in a swizzled [64][128] bf16 tile
__device__ __forceinline__ int swz_t(int r, int cc) { return r * 256 + ((cc ^ (((r & 3) | ((r >> 1) & 4)) << 1)) << 4); }
// byte offset of the 8-byte slot s (0..31) of row r in the same tile
__device__ __forceinline__ int swz_t8(int r, int s) { return r * 256 + ((s ^ (((r & 3) | ((r >> 1) & 4)) << 2)) << 3); }

__device__ __forceinline__ bf16x4 lds_tr16(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
      (__attribute__((address_space(3))) bf16x4*)(uintptr_t)(uint32_t)(uintptr_t)p);
}

template <int PRO>
__global__ __launch_bounds__(256, 2) void tower_wgrad_kernel(const WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const Drop pdrop = drop_resolve(g.drop);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wn = wave & 1, wk = wave >> 1;
  // XCD-aware map: the tiles_n * tiles_k output tiles of one M-slice run on ONE XCD at about
  // the same time, so each dz / A row block is fetched from HBM once and re-read from that L2.
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3, tiles = g.tiles_n * g.tiles_k;
  const int split = (jj / tiles) * 8 + xcd, tile = jj % tiles;
  if (split >= g.splits) return;
  const int n0 = (tile % g.tiles_n) * 128, k0 = (tile / g.tiles_n) * 128;
  const long ms = (long)split * g.rows_per_split;
  const long me = (ms + g.rows_per_split < g.M) ? ms + g.rows_per_split : g.M;
  const int steps = (int)((me - ms + 63) / 64);

  const int cc = tid & 15, r0 = tid >> 4;         // chunk column, first row (rows r0 + 16 i)
  float sc[8], sh[8];
  if (PRO != PRO_NONE) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = k0 + cc * 8 + e;
      sc[e] = k < g.K ? g.a_scale[k] : 0.f;
      sh[e] = k < g.K ? g.a_shift[k] : 0.f;
    }
  }
  const bool nin = n0 + cc * 8 < g.N, kin = k0 + cc * 8 < g.K;
  uint4 rd[4], ra[4];
  auto load_tile = [&](int st) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long m = ms + (long)st * 64 + r0 + 16 * i;
      const bool min = m < me;
      rd[i] = (min && nin) ? *reinterpret_cast<const uint4*>(g.DZ + m * g.lddz + n0 + cc * 8) : make_uint4(0, 0, 0, 0);
      ra[i] = (min && kin) ? *reinterpret_cast<const uint4*>(g.A + m * g.lda + k0 + cc * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto store_tile = [&](int buf, int cur_step) {
    unsigned char* td = smem + buf * 32768;
    unsigned char* ta = td + 16384;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = r0 + 16 * i;
      uint4 va = ra[i];
      if (PRO != PRO_NONE) va = transform_chunk<PRO>(va, sc, sh, pdrop, (uint32_t)(ms + (long)cur_step * 64 + row), (uint32_t)(k0 + cc * 8), g.act);
      *reinterpret_cast<uint4*>(td + swz_t(row, cc)) = rd[i];
      *reinterpret_cast<uint4*>(ta + swz_t(row, cc)) = va;
    }
  };

  f32x4 acc[4][4];      // [fn][fk]: D[n][k]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (steps > 0) {
    load_tile(0);
    store_tile(0, 0);
    __syncthreads();
    if (steps > 1) load_tile(1);
  }
  const int fr = lane & 15, fq = lane >> 4;
  const int trow = fr >> 2, tslot = fr & 3;       // this lane's piece of the [4][16] block
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    const unsigned char* td = smem + cur * 32768;
    const unsigned char* ta = td + 16384;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {              // 32 rows of m per MFMA
      bf16x8 fd[4], fa[4];
      const int rb = kk * 32 + fq * 8 + trow;
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const int sd = (wn * 64 + f * 16) / 4 + tslot, sa = (wk * 64 + f * 16) / 4 + tslot;
        const bf16x4 d0 = lds_tr16(td + swz_t8(rb, sd)), d1 = lds_tr16(td + swz_t8(rb + 4, sd));
        const bf16x4 a0 = lds_tr16(ta + swz_t8(rb, sa)), a1 = lds_tr16(ta + swz_t8(rb + 4, sa));
        fd[f] = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
        fa[f] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn)
#pragma unroll
        for (int fk = 0; fk < 4; ++fk)
          acc[fn][fk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[fn], fa[fk], acc[fn][fk], 0, 0, 0);
    }
    if (st + 1 < steps) store_tile(cur ^ 1, st + 1);
    __syncthreads();
    if (st + 2 < steps) load_tile(st + 2);
  }
  // D[n = nb + 4 fq + r][k = kb + fr]
  float* out = g.slab + (long)split * g.N * g.ldw;
#pragma unroll
  for (int fn = 0; fn < 4; ++fn)
#pragma unroll
    for (int fk = 0; fk < 4; ++fk) {
      const int k = k0 + wk * 64 + fk * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + fn * 16 + fq * 4 + r;
        if (n < g.N && k < g.K) out[(long)n * g.ldw + k] = acc[fn][fk][r];
      }
    }
}

// ------------------------------------------------------------------------------------
// 256 (n) x 256 (k) weight-gradient tile, 8 waves (4 over n x 2 over k: 64 x 128 each, acc[4][8]), both operands
// staged by hand-issued LDS-DMA in 64-row steps, double buffered (2 x 64 KB), one workgroup per (M split, tile).
// Needs N % 256 == 0, K % 64 == 0 (K >= 128), M % (64 * splits) == 0 and no dropout (with the transposed fragments a lane holds
// 8 rows of ONE column, so the keep mask would cost a hash per element); everything else takes the kernel above.
// Against the 128 x 128 kernel: half the bytes through the CU per MAC (each dz / A row block is read by 2 instead of 4
// workgroups), no register staging and no ds_write_b128 pass, the prologue applied to the operand fragments with the
// column's scale / shift held in registers for the whole launch (a lane's fragments are always the same 8 columns).
// LDS image of an operand stage: two [64][128] bf16 half tiles in the swizzle of the kernel above (swz_t / swz_t8).
template <int PRO>
__global__ __launch_bounds__(512, 1) void tower_wgrad256_kernel(const WgradArgs g) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 3, wk = wave >> 2;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const int id = blockIdx.x, xcd = id & 7, jj = id >> 3, tiles = g.tiles_n * g.tiles_k;
  const int split = (jj / tiles) * 8 + xcd, tile = jj % tiles;          // the tiles of one M slice on one XCD
  if (split >= g.splits) return;
  const int n0 = (tile % g.tiles_n) * 256, k0 = (tile / g.tiles_n) * 256;
  const long ms = (long)split * g.rows_per_split;
  const int steps = g.rows_per_split / 64;

  // staging pieces: a half tile is 16 pieces of 4 rows x 256 B; wave w carries rows 8 w .. 8 w + 7 of all four
  // half tiles (dz 0/1, A 0/1): 8 pieces per step
  // (K may end inside the tile -- layer 1 stages 136 features as 192 columns: the pieces beyond column K re-read the
  // row's last 8 columns, their products land in output columns >= K that are never stored)
  uint32_t offD[2], offA[2][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 8 + j * 4 + (lane >> 4);
    const int cc = (lane & 15) ^ ((((r & 3) | ((r >> 1) & 4)) << 1));   // logical chunk that lands in physical slot lane & 15
    offD[j] = (uint32_t)((r * g.lddz + cc * 8) * 2);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      int col = h * 128 + cc * 8;
      if (k0 + col >= g.K) col = g.K - 8 - k0;
      offA[j][h] = (uint32_t)((r * g.lda + col) * 2);
    }
  }
  auto issue = [&](int st, int buf) __attribute__((always_inline)) {
    const char* db = reinterpret_cast<const char*>(g.DZ) + ((ms + (long)st * 64) * g.lddz + n0) * 2;
    const char* ab = reinterpret_cast<const char*>(g.A) + ((ms + (long)st * 64) * g.lda + k0) * 2;
    const uint32_t dst = lds0 + buf * 65536 + wave * 2048;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        dma16_s(offD[j], db + h * 256, dst + h * 16384 + j * 1024);
        dma16_s(offA[j][h], ab, dst + 32768 + h * 16384 + j * 1024);
      }
  };

  const int fr = lane & 15, fq = lane >> 4;
  const int trow = fr >> 2, tslot = fr & 3;       // this lane's piece of the [4][16] block of a transpose read
  // byte offsets of this lane's transpose-read pieces inside a half tile, row block 0 (the swizzle term is the same for
  // rows rb, rb + 4 and rb + 32: they are immediates on top of these)
  const int rb0 = fq * 8 + trow;
  uint32_t adD[4], adA[8];
#pragma unroll
  for (int f = 0; f < 4; ++f) adD[f] = (uint32_t)swz_t8(rb0, ((wn & 1) * 64 + f * 16) / 4 + tslot);
#pragma unroll
  for (int f = 0; f < 8; ++f) adA[f] = (uint32_t)swz_t8(rb0, (f * 16) / 4 + tslot);
  float scf[8], shf[8];                           // the prologue coefficients of this lane's 8 fragment columns
  if (PRO != PRO_NONE) {
#pragma unroll
    for (int f = 0; f < 8; ++f) {
      const int k = k0 + wk * 128 + f * 16 + fr;
      scf[f] = k < g.K ? g.a_scale[k] : 0.f; shf[f] = k < g.K ? g.a_shift[k] : 0.f;
    }
  }

  f32x4 acc[4][8];      // [fn][fk]: D[n][k]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  issue(0, 0);
  if (steps > 1) issue(1, 1);
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll 1
  for (int st = 0; st < steps; ++st) {
    const int cur = st & 1;
    const bool more = st >= 1 && st + 1 < steps;   // stage st + 1 -> the other buffer (stage 1 went out up front)
    const unsigned char* td = smem + cur * 65536 + (wn >> 1) * 16384;            // dz half tile of this wave's 64 columns
    const unsigned char* ta = smem + cur * 65536 + 32768 + wk * 16384;           // A half tile of its 128 columns
    const char* db = reinterpret_cast<const char*>(g.DZ) + ((ms + (long)(st + 1) * 64) * g.lddz + n0) * 2;
    const char* ab = reinterpret_cast<const char*>(g.A) + ((ms + (long)(st + 1) * 64) * g.lda + k0) * 2;
    const uint32_t dst = more ? lds0 + (cur ^ 1) * 65536 + wave * 2048 : lds0 + cur * 65536 + wave * 2048;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {              // 32 rows of m per MFMA
      bf16x8 fd[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) {
        const bf16x4 d0 = lds_tr16(td + adD[f] + kk * 8192), d1 = lds_tr16(td + adD[f] + kk * 8192 + 1024);
        fd[f] = __builtin_shufflevector(d0, d1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int hk = 0; hk < 4; ++hk) {            // A fragments two at a time (register budget)
        if (more && !(hk & 1)) {                  // 2 of the 8 pieces per two half blocks (uniform branch)
          const int blk = kk * 2 + (hk >> 1), h = blk >> 1, j = blk & 1;
          dma16_s(offD[j], db + h * 256, dst + h * 16384 + j * 1024);
          dma16_s(h ? offA[j][1] : offA[j][0], ab, dst + 32768 + h * 16384 + j * 1024);
        }
        bf16x8 fa[2];
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          const bf16x4 a0 = lds_tr16(ta + adA[hk * 2 + f] + kk * 8192), a1 = lds_tr16(ta + adA[hk * 2 + f] + kk * 8192 + 1024);
          bf16x8 v = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
          if (PRO != PRO_NONE) {
            const uint4 u = __builtin_bit_cast(uint4, v);
            const uint32_t w[4] = {u.x, u.y, u.z, u.w};
            uint32_t o[4];
            const float sc = scf[hk * 2 + f], sh = shf[hk * 2 + f];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float x0 = __builtin_fmaf(bf16_lo(w[i]), sc, sh), x1 = __builtin_fmaf(bf16_hi(w[i]), sc, sh);
              if (PRO == PRO_AFFINE_ACT) { x0 = act_fwd(g.act, x0); x1 = act_fwd(g.act, x1); }
              uint32_t pk = pack_bf16(x0, x1);
              if (PRO == PRO_AFFINE_RELU)
                pk = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2, pk), i16x2{0, 0}));
              o[i] = pk;
            }
            v = __builtin_bit_cast(bf16x8, make_uint4(o[0], o[1], o[2], o[3]));
          }
          fa[f] = v;
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
          for (int f = 0; f < 2; ++f)
            acc[fn][hk * 2 + f] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fd[fn], fa[f], acc[fn][hk * 2 + f], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  // D[n = nb + 4 fq + r][k = kb + fr]
  float* out = g.slab + (long)split * g.N * g.ldw;
#pragma unroll
  for (int fn = 0; fn < 4; ++fn)
#pragma unroll
    for (int fk = 0; fk < 8; ++fk) {
      const int k = k0 + wk * 128 + fk * 16 + fr;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = n0 + wn * 64 + fn * 16 + fq * 4 + r;
        if (k < g.K) out[(long)n * g.ldw + k] = acc[fn][fk][r];
      }
    }
}

// out[r][c] (+)= sum_s slab[s][r][c] for c < Cout, slab rows Cs >= Cout wide (the weight gradient of a layer whose input is
// staged wider than the weight matrix: the k-step padding of the first layer) -- the reduction writes the gradient's own
// layout, so it can accumulate in place there too (round 6: it was a slice + copy launch, and the in-place mode fell back)
__global__ void tower_slab_reduce_cols_kernel(const float* __restrict__ slab, int S, int R, int Cs, int Cout,
                                              float* __restrict__ out, int accumulate) {
  const long n = (long)R * Cout, sn = (long)R * Cs;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / Cout, c = i - r * Cout;
    const float* p = slab + r * Cs + c;
    float t = 0.f;
    int s = 0;
    for (; s + 15 < S; s += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = p[(long)(s + u) * sn];
#pragma unroll
      for (int u = 0; u < 16; ++u) t += v[u];
    }
    if (s < S) {
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = (s + u < S) ? p[(long)(s + u) * sn] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += w[u];
    }
    out[i] = accumulate ? out[i] + t : t;
  }
}

// out[i] (+)= sum_s slab[s][i]
__global__ void tower_slab_reduce_kernel(const float* __restrict__ slab, int S, long n, float* __restrict__ out,
                                         int accumulate) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float t = 0.f;
    int s = 0;
    for (; s + 15 < S; s += 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = slab[(long)(s + u) * n + i];
#pragma unroll
      for (int u = 0; u < 16; ++u) t += v[u];
    }
    if (s < S) {                                              // the remaining (< 16) slabs: all loads first, then the same adds
      float w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = (s + u < S) ? slab[(long)(s + u) * n + i] : 0.0f;
#pragma unroll
      for (int u = 0; u < 16; ++u) t += w[u];
    }
    out[i] = accumulate ? out[i] + t : t;
  }
}

// BatchNorm backward coefficients (one launch instead of a dozen elementwise ops): with s = gamma * rstd,
//   dz = p * dy + q * z + r,   p = s,   q = -s * rstd * c2 / M,   r = s * (rstd * c2 * mean - c1) / M
// c = [sum dy ; sum dy * zhat] per column (= d beta ; d gamma).
__global__ void tower_bn_bwd_coeffs_kernel(const float* __restrict__ gamma, const float* __restrict__ rstd,
                                           const float* __restrict__ mean, const float* __restrict__ c, int N,
                                           float inv_m, float* __restrict__ pqr) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  const float s = gamma[n] * rstd[n];
  const float c1 = c[n], c2 = c[N + n];
  pqr[n] = s;
  pqr[N + n] = -s * rstd[n] * c2 * inv_m;
  pqr[2 * N + n] = s * (rstd[n] * c2 * mean[n] - c1) * inv_m;
}

// `prologue` / `epilogue` arguments of the C ABI: mode in bits 0-7, ACT_* code in bits 8.. (modes 3 only)
static inline bool split_mode(int arg, int& mode, int& act) {
  mode = arg & 0xff; act = arg >> 8;
  if (arg < 0 || mode > 3) return false;
  if (mode == 3) return act >= 1 && act <= ACT_LAST;
  return act == 0;
}

Drop to_drop(const tfr_tower_dropout* d) {
  if (!d || d->threshold16 == 0) return Drop{0u, 0u, 1.0f, 2u, nullptr};
  const uint32_t t16 = d->threshold16 > 65535u ? 65535u : d->threshold16;
  // the narrowest field that represents the rate exactly: lge = 5 (1 bit) ... 2 (8 bits); the caller's scale is the
  // exact 1 / (1 - rate) in those cases
  for (uint32_t lge = 5; lge >= 2; --lge) {
    const uint32_t fb = 32u >> lge;
    if ((t16 & ((1u << (16u - fb)) - 1u)) == 0u) return Drop{d->seed, t16 >> (16u - fb), d->scale, lge, d->step};
  }
  // otherwise 16-bit fields (two columns per hash word): the rate to 1 / 65 536, the scale following the threshold so that
  // the mask is unbiased whatever the caller passed
  return Drop{d->seed, t16, 65536.0f / (65536.0f - (float)t16), 1u, d->step};
}

int grid_for(long work_items, int block) {
  long g = (work_items + block - 1) / block;
  if (g > 256 * 16) g = 256 * 16;
  if (g < 1) g = 1;
  return (int)g;
}

template <int PRO, int EPI, bool GL>
int launch_gemm_v(const GemmArgs& g, hipStream_t st) {
  const int kpad = (g.K + 63) & ~63;
  const size_t lds = 4 * TILE_BYTES + 2 * (size_t)kpad * sizeof(float);
  auto fn = tower_gemm_kernel<PRO, EPI, GL>;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  const int groups = (g.tiles_m + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(groups * 8 * g.tiles_n), dim3(256), lds, st, g);
  return (int)hipGetLastError();
}

template <int PRO, int EPI>
int launch_gemm256(const GemmArgs& g0, hipStream_t st) {
  GemmArgs g = g0;
  g.tiles_m = (g.M + BM2 - 1) / BM2; g.tiles_n = (g.N + BN2 - 1) / BN2;
  const size_t lds = 4 * TILE2_BYTES + 2 * (size_t)g.K * sizeof(float);
  auto fn = tower_gemm256_kernel<PRO, EPI>;
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)lds);
  if (e != hipSuccess) return (int)e;
  const int groups = (g.tiles_m + 7) / 8;
  hipLaunchKernelGGL(fn, dim3(groups * 8 * g.tiles_n), dim3(512), lds, st, g);
  return (int)hipGetLastError();
}

// Persistent kernel over the full 256-row tiles, the kernel above over a ragged rest.
template <int PRO, int EPI>
int launch_gemm256p(const GemmArgs& g0, hipStream_t st) {
  static const int flags = [] { const char* e = getenv("TFR_GEMM_FLAGS"); return (e && *e) ? atoi(e) : 0; }();   // PF_TOUCH_Z: measured, no gain, +47 % fetch
  GemmArgs g = g0;
  const int m_full = g0.M & ~(BM2 - 1);
  g.M = m_full;
  g.tiles_m = m_full / BM2; g.tiles_n = g.N / BN2;
  g.flags = flags;
  const bool drop = (PRO != PRO_NONE && g.pro_drop.thr) || ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && g.epi_drop.thr);
  const int nq0 = ((g.tiles_m + 7) >> 3) * g.tiles_n;
  const int nslots = nq0 < 32 ? nq0 : 32;
  // the prologue's keep-bit table: rate 1/2 (1-bit fields), an activation that commutes with the factor 2
  const bool table = PRO != PRO_NONE && PRO != PRO_AFFINE_ACT && g.pro_drop.thr != 0u && g.pro_drop.lge == 5u &&
                     !(flags & PF_NO_BIT_TABLE);
  auto fn = !drop ? tower_gemm256p_kernel<PRO, EPI, 0>
                  : ((table && EPI <= EPI_STATS) ? tower_gemm256p_kernel<PRO, EPI, ((PRO == PRO_AFFINE || PRO == PRO_AFFINE_RELU) && EPI <= EPI_STATS) ? 2 : 1>
                                                                            : tower_gemm256p_kernel<PRO, EPI, 1>);
  // 16-bit keep fields (a rate that is not a multiple of 1 / 256) in either mask: the form that carries that path
  const bool drop16 = (PRO != PRO_NONE && g.pro_drop.thr && g.pro_drop.lge == 1u) ||
                      ((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && g.epi_drop.thr && g.epi_drop.lge == 1u);
  if (drop && drop16) fn = tower_gemm256p_kernel<PRO, EPI, 3>;
  // the two-group ping-pong k loop (round 5; TFR_GEMM_PP=1 / 2; default 0 = the single-phase loop of rounds 2-4: measured
  // equal within 2 % -- profiles/r05_gemm_pp.txt), compiled for the forms a
  // BatchNorm + ReLU tower runs: hidden-layer forward (2, 1), its dgrad (0, 2), layer 1 / plain products (0, 1), (0, 0)
  static const int env_pp = [] { const char* e = getenv("TFR_GEMM_PP"); return (e && *e) ? atoi(e) : 0; }();
  constexpr bool pp_form = (PRO == PRO_AFFINE_RELU && EPI == EPI_STATS) || (PRO == PRO_NONE && EPI <= EPI_RELU_BWD);
  if constexpr (pp_form) if (!drop16) {
    if (env_pp == 1)
      fn = !drop ? tower_gemm256p_kernel<PRO, EPI, 0, 1>
                 : ((table && EPI <= EPI_STATS) ? tower_gemm256p_kernel<PRO, EPI, (PRO == PRO_AFFINE_RELU && EPI <= EPI_STATS) ? 2 : 1, 1>
                                                : tower_gemm256p_kernel<PRO, EPI, 1, 1>);
    else if (env_pp == 2)
      fn = !drop ? tower_gemm256p_kernel<PRO, EPI, 0, 2>
                 : ((table && EPI <= EPI_STATS) ? tower_gemm256p_kernel<PRO, EPI, (PRO == PRO_AFFINE_RELU && EPI <= EPI_STATS) ? 2 : 1, 2>
                                                : tower_gemm256p_kernel<PRO, EPI, 1, 2>);
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, P_LDS);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(fn, dim3(8 * nslots), dim3(512), P_LDS, st, g);
  int rc = (int)hipGetLastError();
  if (rc != 0 || m_full == g0.M) return rc;
  GemmArgs t = g0;                                  // the last M - m_full (< 256) rows
  t.A = g0.A + (long)m_full * g0.lda; t.C = g0.C + (long)m_full * g0.ldc;
  if (g0.Zp) t.Zp = g0.Zp + (long)m_full * g0.ldz;
  if (g0.stats) t.stats = g0.stats + (long)(m_full / 64) * 2 * g0.N;
  t.M = g0.M - m_full; t.row0 = g0.row0 + m_full;
  t.tiles_m = (t.M + BM - 1) / BM; t.tiles_n = (t.N + BN - 1) / BN;
  return launch_gemm_v<PRO, EPI, true>(t, st);
}

// Round 6: the resident-panel kernel (tower_gemm_rp.h) over the full 512-row tiles of the shapes it serves; the last
// M % 512 rows and every other shape stay with the kernels above.  TFR_GEMM_RP=0 restores round 5's dispatch.
static bool gemm_rp_shape_ok(int M, int N, int K, long lda, long ldb, long ldc, long ldz) {
  // (read on every call, not cached: tests and A/B runs flip them inside one process; two getenv calls per GEMM launch)
  const char* e_rp = getenv("TFR_GEMM_RP");
  const char* e_mt = getenv("TFR_GEMM_RP_MIN_TILES");
  const int env_rp = (e_rp && *e_rp) ? atoi(e_rp) : 0;      // (off by default until it beats the round-5 kernels on every form)
  const int min_tiles = (e_mt && *e_mt) ? atoi(e_mt) : 512;
  if (!env_rp || (N % RP_BN) != 0 || K != 512) return false;
  const int tiles_n = N / RP_BN;
  if (tiles_n > 32 || (32 % tiles_n) != 0) return false;
  if ((long)(M / RP_BM) * tiles_n < min_tiles) return false;          // fewer than ~two tiles per CU: the 256 x 256 kernel spreads better
  return lda < (1L << 21) && ldb < (1L << 21) && ldc < (1L << 21) && ldz < (1L << 21);
}

template <int PRO, int EPI> int launch_gemm_rest(const GemmArgs& g, hipStream_t st);

// Round 6: the weight-stationary kernel (tower_gemm_bs.h) for the forms without a prologue.  TFR_GEMM_BS=0 / 1.
static bool gemm_bs_shape_ok(int M, int N, int K, long lda, long ldb, long ldc, long ldz) {
  const char* e_bs = getenv("TFR_GEMM_BS");
  const char* e_mt = getenv("TFR_GEMM_BS_MIN_TILES");
  const int env_bs = (e_bs && *e_bs) ? atoi(e_bs) : 0;
  const int min_tiles = (e_mt && *e_mt) ? atoi(e_mt) : 2048;         // (64 x 256 tiles: eight per CU)
  if (!env_bs || (N % BS_BN) != 0 || K != 512) return false;
  const int tiles_n = N / BS_BN;
  if (tiles_n > 32 || (32 % tiles_n) != 0) return false;
  if ((long)(M / BS_BM) * tiles_n < min_tiles) return false;
  return lda < (1L << 21) && ldb < (1L << 21) && ldc < (1L << 21) && ldz < (1L << 21);
}

template <int EPI>
int launch_gemm_bs(const GemmArgs& g0, hipStream_t st) {
  constexpr bool BWD = EPI == EPI_RELU_BWD;
  GemmArgs g = g0;
  const int m_full = (g0.M / BS_BM) * BS_BM;
  g.M = m_full; g.tiles_m = m_full / BS_BM; g.tiles_n = g.N / BS_BN; g.flags = 0;
  const bool drop = BWD && g.epi_drop.thr;
  const bool drop16 = drop && g.epi_drop.lge == 1u;
  constexpr int D = BWD ? 5 : 7;                    // stages in flight (16 KB each); the dgrad's epilogue operations limit the counted waits to 5
  auto fn = !drop ? tower_gemm_bs_kernel<EPI, 0, D> : (drop16 ? tower_gemm_bs_kernel<EPI, 3, D> : tower_gemm_bs_kernel<EPI, 1, D>);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, BS_LDS);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(fn, dim3(256), dim3(256), BS_LDS, st, g);
  int rc = (int)hipGetLastError();
  if (rc != 0 || m_full == g0.M) return rc;
  GemmArgs t = g0;                                  // the last M - m_full (< 64) rows
  t.A = g0.A + (long)m_full * g0.lda; t.C = g0.C + (long)m_full * g0.ldc;
  if (g0.Zp) t.Zp = g0.Zp + (long)m_full * g0.ldz;
  if (g0.stats) t.stats = g0.stats + (long)(m_full / 64) * 2 * g0.N;
  t.M = g0.M - m_full; t.row0 = g0.row0 + m_full;
  t.tiles_m = (t.M + BM - 1) / BM; t.tiles_n = (t.N + BN - 1) / BN;
  return launch_gemm_rest<PRO_NONE, EPI>(t, st);
}

template <int PRO, int EPI>
int launch_gemm_rp(const GemmArgs& g0, hipStream_t st) {
  constexpr bool BWD = EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD;
  GemmArgs g = g0;
  const int m_full = (g0.M / RP_BM) * RP_BM;
  g.M = m_full; g.tiles_m = m_full / RP_BM; g.tiles_n = g.N / RP_BN;
  { const char* e_rot = getenv("TFR_GEMM_RP_ROT"); g.flags = (e_rot && *e_rot && atoi(e_rot) == 0) ? 1 : 0;
    const char* e_st = getenv("TFR_GEMM_RP_STAGGER"); g.flags |= (((e_st && *e_st) ? atoi(e_st) : 4) & 0xff) << 8; }   // bit 0: every n-tile in the same k order (bit-identical to the round-5 kernels)
  // Dropout: the prologue form is the keep-bit-table one (rate 1/2, the reference default; DROP = 2) -- the per-fragment hash
  // forms of other rates need more registers than this kernel has left (19-90 spilled) and stay with the round-5 kernels;
  // the epilogue mask of the dgrad forms (DROP = 1: 8-bit fields, 3: 16-bit fields) fits.
  void (*fn)(const GemmArgs) = nullptr;
  if constexpr (PRO != PRO_NONE) {
    const bool table = PRO != PRO_AFFINE_ACT && g.pro_drop.lge == 5u && EPI <= EPI_STATS;
    if (g.pro_drop.thr != 0u && !table) return launch_gemm_rest<PRO, EPI>(g0, st);
    if (g.Aout != nullptr) fn = g.pro_drop.thr ? tower_gemm_rp_kernel<PRO, EPI, 2, 8, true> : tower_gemm_rp_kernel<PRO, EPI, 0, 8, true>;
    else fn = g.pro_drop.thr ? tower_gemm_rp_kernel<PRO, EPI, 2, 8> : tower_gemm_rp_kernel<PRO, EPI, 0, 8>;
  } else {
    const bool drop = BWD && g.epi_drop.thr;
    const bool drop16 = drop && g.epi_drop.lge == 1u;
    fn = !drop ? tower_gemm_rp_kernel<PRO, EPI, 0, 8> : (drop16 ? tower_gemm_rp_kernel<PRO, EPI, 3, 8> : tower_gemm_rp_kernel<PRO, EPI, 1, 8>);
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, RP_LDS);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(fn, dim3(256), dim3(512), RP_LDS, st, g);
  int rc = (int)hipGetLastError();
  if (rc != 0 || m_full == g0.M) return rc;
  GemmArgs t = g0;                                  // the last M - m_full (< 512) rows
  t.A = g0.A + (long)m_full * g0.lda; t.C = g0.C + (long)m_full * g0.ldc;
  if (g0.Zp) t.Zp = g0.Zp + (long)m_full * g0.ldz;
  if (g0.stats) t.stats = g0.stats + (long)(m_full / 64) * 2 * g0.N;
  t.M = g0.M - m_full; t.row0 = g0.row0 + m_full;
  t.tiles_m = (t.M + BM - 1) / BM; t.tiles_n = (t.N + BN - 1) / BN;
  return launch_gemm_rest<PRO, EPI>(t, st);
}

// The shapes the persistent kernel serves (tfr_tower_gemm_persistent: the host asks before it passes `a_out`).
static bool gemm_persistent_ok(int M, int N, int K, long lda, long ldb, long ldc, long ldz) {
  static const bool persist = [] { const char* e = getenv("TFR_TOWER_PERSIST"); return !(e && *e) || atoi(e) != 0; }();
  return persist && M >= BM2 && (N % BN2) == 0 && (K % BK) == 0 && K >= 2 * BK && K <= 1024 && lda < (1L << 21) &&
         ldb < (1L << 21) && ldc < (1L << 21) && ldz < (1L << 21);
}

template <int PRO, int EPI>
int launch_gemm(const GemmArgs& g, hipStream_t st) {
  // the forms a BatchNorm + ReLU tower runs: hidden-layer forward (2, 1), its dgrad (0, 2), plain products (0, 0), (0, 1)
  constexpr bool rp_form = (PRO == PRO_AFFINE_RELU && EPI == EPI_STATS) || (PRO == PRO_NONE && EPI <= EPI_RELU_BWD);
  if constexpr (PRO == PRO_NONE && (EPI == EPI_PLAIN || EPI == EPI_RELU_BWD)) {
    if (g.Aout == nullptr && gemm_bs_shape_ok(g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldz) && !(EPI == EPI_RELU_BWD && g.bias))
      return launch_gemm_bs<EPI>(g, st);
  }
  if constexpr (rp_form) {
    if (gemm_rp_shape_ok(g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldz) && !(EPI == EPI_RELU_BWD && g.bias) &&
        (g.Aout == nullptr || ((g.M % RP_BM) == 0 && PRO != PRO_NONE)))
      return launch_gemm_rp<PRO, EPI>(g, st);
  }
  return launch_gemm_rest<PRO, EPI>(g, st);
}

template <int PRO, int EPI>
int launch_gemm_rest(const GemmArgs& g, hipStream_t st) {
  const bool pers = gemm_persistent_ok(g.M, g.N, g.K, g.lda, g.ldb, g.ldc, g.ldz) &&
                    !((EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD) && g.bias);
  if (g.Aout && !(pers && (g.M % BM2) == 0 && PRO != PRO_NONE && EPI <= EPI_STATS)) return TFR_EINVAL;
  if (pers) return launch_gemm256p<PRO, EPI>(g, st);
  static const bool no_gl = [] { const char* e = getenv("TFR_TOWER_NO_LDSDMA"); return e && *e && atoi(e) != 0; }();
  static const int tile = [] { const char* e = getenv("TFR_TOWER_TILE"); return (e && *e) ? atoi(e) : 256; }();
  if ((g.K % BK) == 0 && !no_gl && tile == 256 && g.N >= BN2 && g.M >= BM2 &&
      4 * TILE2_BYTES + 2 * (size_t)g.K * sizeof(float) <= 160 * 1024)
    return launch_gemm256<PRO, EPI>(g, st);
  if ((g.K % BK) == 0 && !no_gl) return launch_gemm_v<PRO, EPI, true>(g, st);
  return launch_gemm_v<PRO, EPI, false>(g, st);
}

// FlattenList's gather index in one launch (utils.py:203-230 organize_valid_indices(shuffle=False) + :308-356
// padded_nd_indices): position p of list b reads row b * L + v[p mod max(n, 1)], v = the valid positions of the
// list in index order (then the invalid ones -- reachable only when n = 0, where v[0] = 0).  One wave per list,
// ballot + popcount compaction into LDS; replaces a stable sort and a dozen elementwise launches.
__global__ __launch_bounds__(256) void flatten_row_index_kernel(const uint8_t* __restrict__ mask, int B, int L,
                                                                int* __restrict__ rows) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  int* v = reinterpret_cast<int*>(smem) + (long)wave * L;
  const uint8_t* mk = mask + (long)b * L;
  int n = 0;
  for (int p0 = 0; p0 < L; p0 += 64) {
    const int p = p0 + lane;
    const bool ok = p < L && mk[p] != 0;
    const unsigned long long bal = __ballot(ok);
    if (ok) v[n + __popcll(bal & ((1ull << lane) - 1ull))] = p;
    n += __popcll(bal);
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);                          // lgkmcnt(0): the wave's own LDS writes have landed
  __builtin_amdgcn_wave_barrier();
  const long base = (long)b * L;
  for (int p = lane; p < L; p += 64) rows[base + p] = (int)base + (n > 0 ? v[p % n] : 0);
}

// dst[j][i] += src[j][i] for up to 16 small vectors in one launch (blockIdx.y = j); the pointers travel in
// the kernel arguments.
struct MultiAdd { float* dst[16]; const float* src[16]; int n[16]; };
__global__ void tower_multi_add_kernel(const MultiAdd a) {
  const int j = blockIdx.y;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < a.n[j]; i += gridDim.x * blockDim.x) a.dst[j][i] += a.src[j][i];
}

}  // namespace

extern "C" int tfr_tower_multi_add(float* const* dst, const float* const* src, const int* n, int count, void* stream) {
  if (count < 0 || (count > 0 && (!dst || !src || !n))) return TFR_EINVAL;
  if (count == 0) return TFR_OK;
  for (int c0 = 0; c0 < count; c0 += 16) {
    MultiAdd a;
    const int c = (count - c0 < 16) ? count - c0 : 16;
    int nmax = 1;
    for (int j = 0; j < 16; ++j) {
      a.dst[j] = j < c ? dst[c0 + j] : nullptr; a.src[j] = j < c ? src[c0 + j] : nullptr; a.n[j] = j < c ? n[c0 + j] : 0;
      if (j < c && (!a.dst[j] || !a.src[j] || a.n[j] < 0)) return TFR_EINVAL;
      if (a.n[j] > nmax) nmax = a.n[j];
    }
    const int gx = (nmax + 255) / 256 > 64 ? 64 : (nmax + 255) / 256;
    hipLaunchKernelGGL(tower_multi_add_kernel, dim3(gx, c), dim3(256), 0, (hipStream_t)stream, a);
  }
  return (int)hipGetLastError();
}

// rows[b * L + p] for tfr_tower_cast_gather_f32_bf16 (keras/layers.py:122-183 FlattenList, utils.py:308-356).
extern "C" int tfr_flatten_row_index(const unsigned char* mask, int B, int L, int* rows, void* stream) {
  if (!mask || !rows || B < 0 || L <= 0 || (long)B * L > 0x7fffffffL) return TFR_EINVAL;
  if (L > TFR_MAX_LIST_SIZE) return TFR_ETOOLARGE;           // 4 B of LDS per item and list-wave, 4 list-waves: <= 128 KiB
  if (B == 0) return TFR_OK;
  const size_t lds = (size_t)L * 16;
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&flatten_row_index_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(flatten_row_index_kernel, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, mask, B, L, rows);
  return (int)hipGetLastError();
}

// input_batch_norm (keras/layers.py:57-60: BatchNormalization on the raw features): per-column partial sums
// (sum x, sum x^2) of the fp32 input, rows gathered like the cast does; partial[T][2][F] feeds tfr_tower_bn_finalize,
// whose scale / shift the cast kernel then applies.
template <typename TIn>
__global__ __launch_bounds__(256) void tower_input_stats_kernel(const TIn* __restrict__ x, long ldx, int M, int F,
                                                                const int* __restrict__ row_index,
                                                                float* __restrict__ partial, int rows_per_block,
                                                                const float* __restrict__ pivot) {
  // `pivot` (nullable, [F]): the sums are taken of x - pivot.  var = E[x^2] - mean^2 from fp32 partial sums loses
  // every digit of a column with |mean| >> std (raw features are exactly that case: ADVICE r2); shifted by a sample
  // of the column (the caller passes the first row) both sums stay at the scale of the spread.
  const long mb = (long)blockIdx.x * rows_per_block;
  const long me = (mb + rows_per_block < M) ? mb + rows_per_block : M;
  for (int c = threadIdx.x; c < F; c += blockDim.x) {
    float s1 = 0.f, s2 = 0.f;
    const float pv = pivot ? pivot[c] : 0.0f;
    for (long m = mb; m < me; ++m) {
      const long ms = row_index ? (long)row_index[m] : m;
      const float v = feat_f32(x[ms * ldx + c]) - pv;
      s1 += v; s2 = __builtin_fmaf(v, v, s2);
    }
    partial[((long)blockIdx.x * 2) * F + c] = s1;
    partial[((long)blockIdx.x * 2 + 1) * F + c] = s2;
  }
}

extern "C" int tfr_tower_input_stats_f32(const float* x, long ldx, int M, int F, const int* row_index, float* partial,
                                         int n_blocks, const float* pivot, void* stream) {
  if (!x || !partial || M <= 0 || F <= 0 || n_blocks < 1 || ldx < F) return TFR_EINVAL;
  const int rows = (int)(((long)M + n_blocks - 1) / n_blocks);
  hipLaunchKernelGGL(tower_input_stats_kernel<float>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream, x, ldx, M, F,
                     row_index, partial, rows, pivot);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_input_stats_bf16(const void* x_bf16, long ldx, int M, int F, const int* row_index,
                                          float* partial, int n_blocks, const float* pivot, void* stream) {
  if (!x_bf16 || !partial || M <= 0 || F <= 0 || n_blocks < 1 || ldx < F) return TFR_EINVAL;
  const int rows = (int)(((long)M + n_blocks - 1) / n_blocks);
  hipLaunchKernelGGL(tower_input_stats_kernel<uint16_t>, dim3(n_blocks), dim3(256), 0, (hipStream_t)stream,
                     (const uint16_t*)x_bf16, ldx, M, F, row_index, partial, rows, pivot);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_cast_gather_f32_bf16(const float* x, long ldx, int M, int F, int Kp, const float* scale,
                                              const float* shift, const int* row_index, void* out_bf16,
                                              void* stream) {
  if (!x || !out_bf16 || M < 0 || F <= 0 || Kp < F || (Kp & 7)) return TFR_EINVAL;
  if (M == 0) return TFR_OK;
  hipLaunchKernelGGL(tower_cast_kernel<float>, dim3(grid_for((long)M * (Kp / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, x, ldx, M, F, Kp, scale, shift, row_index, (uint16_t*)out_bf16);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_cast_gather_bf16_bf16(const void* x_bf16, long ldx, int M, int F, int Kp, const float* scale,
                                               const float* shift, const int* row_index, void* out_bf16,
                                               void* stream) {
  if (!x_bf16 || !out_bf16 || M < 0 || F <= 0 || Kp < F || (Kp & 7) || ldx < F) return TFR_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return TFR_EINVAL;
  if (M == 0) return TFR_OK;
  hipLaunchKernelGGL(tower_cast_kernel<uint16_t>, dim3(grid_for((long)M * (Kp / 8), 256)), dim3(256), 0,
                     (hipStream_t)stream, (const uint16_t*)x_bf16, ldx, M, F, Kp, scale, shift, row_index,
                     (uint16_t*)out_bf16);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_cast_f32_bf16(const float* x, long ldx, int M, int F, int Kp, const float* scale,
                                       const float* shift, void* out_bf16, void* stream) {
  return tfr_tower_cast_gather_f32_bf16(x, ldx, M, F, Kp, scale, shift, nullptr, out_bf16, stream);
}

extern "C" int tfr_tower_weight_cast(const float* w, int R, int C, int transpose, int pitch, void* out_bf16,
                                     void* stream) {
  if (!w || !out_bf16 || R <= 0 || C <= 0 || pitch < (transpose ? R : C)) return TFR_EINVAL;
  const int total = transpose ? C * pitch : R * pitch;
  hipLaunchKernelGGL(tower_weight_cast_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                     w, R, C, transpose, pitch, (uint16_t*)out_bf16);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_weight_cast_batch_step(const float* const* w, const int* R, const int* C, const int* transpose,
                                                const int* pitch, void* const* out_bf16, int count, int32_t* step,
                                                int32_t* step_copy, void* stream);
extern "C" int tfr_tower_weight_cast_batch(const float* const* w, const int* R, const int* C, const int* transpose,
                                           const int* pitch, void* const* out_bf16, int count, void* stream) {
  return tfr_tower_weight_cast_batch_step(w, R, C, transpose, pitch, out_bf16, count, nullptr, nullptr, stream);
}

extern "C" int tfr_tower_weight_cast_batch_step(const float* const* w, const int* R, const int* C, const int* transpose,
                                                const int* pitch, void* const* out_bf16, int count, int32_t* step,
                                                int32_t* step_copy, void* stream) {
  if (count < 0 || (count > 0 && (!w || !R || !C || !transpose || !pitch || !out_bf16))) return TFR_EINVAL;
  if ((step != nullptr) != (step_copy != nullptr) || (step && count == 0)) return TFR_EINVAL;
  for (int j = 0; j < count; ++j)
    if (!w[j] || !out_bf16[j] || R[j] <= 0 || C[j] <= 0 || pitch[j] < (transpose[j] ? R[j] : C[j])) return TFR_EINVAL;
  for (int c0 = 0; c0 < count; c0 += 8) {
    WCastBatch a;
    const int c = (count - c0 < 8) ? count - c0 : 8;
    int tmax = 1;
    for (int j = 0; j < 8; ++j) {
      const int s = c0 + (j < c ? j : 0);                      // unused slots repeat a valid one (never launched)
      a.w[j] = w[s]; a.out[j] = (uint16_t*)out_bf16[s]; a.R[j] = R[s]; a.C[j] = C[s];
      a.transpose[j] = transpose[s]; a.pitch[j] = pitch[s];
      const int total = (transpose[s] ? C[s] : R[s]) * pitch[s];
      if (j < c && total > tmax) tmax = total;
    }
    a.step = c0 == 0 ? step : nullptr; a.step_copy = step_copy;
    hipLaunchKernelGGL(tower_weight_cast_batch_kernel, dim3(grid_for(tmax, 256), c), dim3(256), 0,
                       (hipStream_t)stream, a);
  }
  return count == 0 ? TFR_OK : (int)hipGetLastError();
}

extern "C" int tfr_tower_gemm_bf16_aout(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                        int M, int N, int K, int prologue, const float* a_scale,
                                        const float* a_shift, const float* bias, int epilogue, float* stats,
                                        const void* Zp, long ldz, const float* e_scale, const float* e_shift,
                                        const float* e_mean, const float* e_rstd,
                                        const tfr_tower_dropout* pro_dropout, const tfr_tower_dropout* epi_dropout,
                                        void* a_out, long ldao, void* stream);

extern "C" int tfr_tower_gemm_bf16(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                   int M, int N, int K, int prologue, const float* a_scale,
                                   const float* a_shift, const float* bias, int epilogue, float* stats,
                                   const void* Zp, long ldz, const float* e_scale, const float* e_shift,
                                   const float* e_mean, const float* e_rstd,
                                   const tfr_tower_dropout* pro_dropout, const tfr_tower_dropout* epi_dropout,
                                   void* stream) {
  return tfr_tower_gemm_bf16_aout(A, lda, B, ldb, C, ldc, M, N, K, prologue, a_scale, a_shift, bias, epilogue, stats, Zp,
                                  ldz, e_scale, e_shift, e_mean, e_rstd, pro_dropout, epi_dropout, nullptr, 0, stream);
}

// 1 when tfr_tower_gemm_bf16_aout accepts `a_out` for this shape (the persistent 256 x 256 kernel over full tiles)
extern "C" int tfr_tower_gemm_writes_operand(int M, int N, int K) {
  return (gemm_persistent_ok(M, N, K, K, K, N, N) && (M % BM2) == 0) ? 1 : 0;
}

extern "C" int tfr_tower_gemm_bf16_aout(const void* A, long lda, const void* B, long ldb, void* C, long ldc,
                                        int M, int N, int K, int prologue, const float* a_scale,
                                        const float* a_shift, const float* bias, int epilogue, float* stats,
                                        const void* Zp, long ldz, const float* e_scale, const float* e_shift,
                                        const float* e_mean, const float* e_rstd,
                                        const tfr_tower_dropout* pro_dropout, const tfr_tower_dropout* epi_dropout,
                                        void* a_out, long ldao, void* stream) {
  if (!A || !B || !C || M < 0 || N <= 0 || K <= 0) return TFR_EINVAL;
  if (a_out && ((ldao & 7) || ldao < K)) return TFR_EINVAL;
  if ((lda & 7) || (ldb & 7) || (ldc & 7) || (N & 7) || (K & 7) || lda < K || ldb < K || ldc < N) return TFR_EINVAL;
  int pact = 0, eact = 0;
  if (!split_mode(prologue, prologue, pact) || !split_mode(epilogue, epilogue, eact)) return TFR_EINVAL;
  if ((prologue == PRO_AFFINE_ACT && epilogue >= EPI_RELU_BWD) || (epilogue == EPI_ACT_BWD && prologue != PRO_NONE))
    return TFR_EINVAL;                                                                // no model runs those pairs
  if (prologue != PRO_NONE && (!a_scale || !a_shift)) return TFR_EINVAL;
  if (epilogue != EPI_PLAIN && !stats) return TFR_EINVAL;
  if (epilogue >= EPI_RELU_BWD && (!Zp || !e_scale || !e_shift || !e_mean || !e_rstd || (ldz & 7))) return TFR_EINVAL;
  if (K > 4096) return TFR_ETOOLARGE;
  if (M == 0) return TFR_OK;
  GemmArgs g;
  g.A = (const uint16_t*)A; g.lda = lda; g.B = (const uint16_t*)B; g.ldb = ldb;
  g.C = (uint16_t*)C; g.ldc = ldc; g.bias = bias; g.a_scale = a_scale; g.a_shift = a_shift;
  g.stats = stats; g.Zp = (const uint16_t*)Zp; g.ldz = ldz; g.e_scale = e_scale; g.e_shift = e_shift;
  g.e_mean = e_mean; g.e_rstd = e_rstd; g.M = M; g.N = N; g.K = K;
  g.tiles_m = (M + BM - 1) / BM; g.tiles_n = (N + BN - 1) / BN;
  g.pro_drop = to_drop(pro_dropout); g.epi_drop = to_drop(epi_dropout);
  g.flags = 0; g.row0 = 0; g.act = pact ? pact : eact;
  g.Aout = (uint16_t*)a_out; g.ldao = ldao;
  hipStream_t st = (hipStream_t)stream;
#define TG(P, E) if (prologue == P && epilogue == E) return launch_gemm<P, E>(g, st)
  TG(0, 0); TG(0, 1); TG(0, 2); TG(1, 0); TG(1, 1); TG(1, 2); TG(2, 0); TG(2, 1); TG(2, 2);
  TG(3, 0); TG(3, 1); TG(0, 3);
#undef TG
  return TFR_EINVAL;
}

extern "C" int tfr_tower_gemm_stats_rows(int M) { return (M + 63) / 64; }   // one row per 64-row slab

extern "C" int tfr_tower_reduce_scratch_rows(int T) { return (T + kReduceChunk - 1) / kReduceChunk; }

extern "C" int tfr_tower_bn_finalize(const float* partial, int T, int N, long M, const float* gamma,
                                     const float* beta, float eps, float momentum, float* moving_mean,
                                     float* moving_var, float* scale, float* shift, float* mean_out,
                                     float* rstd_out, float* scratch, void* stream) {
  if (!partial || T <= 0 || N <= 0 || M <= 0 || !scale || !shift || !mean_out || !rstd_out) return TFR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (T <= kFusedRows) {                       // short partial matrix: one launch (same arithmetic, same order)
    hipLaunchKernelGGL(tower_bn_finalize_fused_kernel, dim3((N + 63) / 64), dim3(1024), 0, st, partial, T, N, M, gamma,
                       beta, eps, momentum, moving_mean, moving_var, scale, shift, mean_out, rstd_out);
    return (int)hipGetLastError();
  }
  const float* src = partial;
  int rows = T;
  if (T > kReduceChunk && scratch) {          // stage 1: [T][2N] -> [ceil(T/64)][2N]
    rows = (T + kReduceChunk - 1) / kReduceChunk;
    hipLaunchKernelGGL(tower_reduce_rows_kernel, dim3((2 * N + 255) / 256, rows), dim3(256), 0, st, partial, T,
                       2 * N, scratch);
    src = scratch;
  }
  hipLaunchKernelGGL(tower_bn_finalize_kernel, dim3((N + 63) / 64), dim3(64), 0, st, src, rows, N, M, gamma, beta,
                     eps, momentum, moving_mean, moving_var, scale, shift, mean_out, rstd_out);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_reduce_partials(const float* partial, int T, int W, float* out, float* scratch,
                                         void* stream) {
  if (!partial || T <= 0 || W <= 0 || !out) return TFR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  const float* src = partial;
  int rows = T;
  if (T > kReduceChunk && scratch) {
    rows = (T + kReduceChunk - 1) / kReduceChunk;
    hipLaunchKernelGGL(tower_reduce_rows_kernel, dim3((W + 255) / 256, rows), dim3(256), 0, st, partial, T, W, scratch);
    src = scratch;
  }
  hipLaunchKernelGGL(tower_reduce_partials_kernel, dim3((W + 63) / 64), dim3(64), 0, st, src, rows, W, out);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_out_f32(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                                 const float* shift, const float* w, const float* b, int O, float* out,
                                 const tfr_tower_dropout* dropout, void* stream) {
  if (!z || !w || !out || M < 0 || K <= 0 || (K & 7) || (ldz & 7) || O < 1 || O > 4) return TFR_EINVAL;
  int act = 0;
  if (!split_mode(prologue, prologue, act)) return TFR_EINVAL;
  if (prologue != PRO_NONE && (!scale || !shift)) return TFR_EINVAL;
  if (M == 0) return TFR_OK;
  const int grid = grid_for((long)M * 16, 256) > 2048 ? 2048 : grid_for((long)M * 16, 256);
  const size_t lds = (size_t)(K / 8) * (2 + O) * 8 * sizeof(float);
  if (lds > 64 * 1024) return TFR_ETOOLARGE;
  hipStream_t st = (hipStream_t)stream;
  const uint16_t* zz = (const uint16_t*)z;
  const Drop dr = to_drop(dropout);
  if (prologue == PRO_NONE) hipLaunchKernelGGL(tower_out_kernel<PRO_NONE>, dim3(grid), dim3(256), lds, st, zz, ldz, M, K, scale, shift, w, b, O, out, dr, act);
  else if (prologue == PRO_AFFINE) hipLaunchKernelGGL(tower_out_kernel<PRO_AFFINE>, dim3(grid), dim3(256), lds, st, zz, ldz, M, K, scale, shift, w, b, O, out, dr, act);
  else if (prologue == PRO_AFFINE_RELU) hipLaunchKernelGGL(tower_out_kernel<PRO_AFFINE_RELU>, dim3(grid), dim3(256), lds, st, zz, ldz, M, K, scale, shift, w, b, O, out, dr, act);
  else if (prologue == PRO_AFFINE_ACT) hipLaunchKernelGGL(tower_out_kernel<PRO_AFFINE_ACT>, dim3(grid), dim3(256), lds, st, zz, ldz, M, K, scale, shift, w, b, O, out, dr, act);
  else return TFR_EINVAL;
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_out_bwd2(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* w,
                                  const float* dlogits, int O, void* dy_bf16, long lddy, float* partial,
                                  int n_blocks, const tfr_tower_dropout* dropout, const float* pqr, void* stream);

extern "C" int tfr_tower_out_bwd(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                                 const float* shift, const float* mean, const float* rstd, const float* w,
                                 const float* dlogits, int O, void* dy_bf16, long lddy, float* partial,
                                 int n_blocks, const tfr_tower_dropout* dropout, void* stream) {
  return tfr_tower_out_bwd2(z, ldz, M, K, prologue, scale, shift, mean, rstd, w, dlogits, O, dy_bf16, lddy, partial,
                            n_blocks, dropout, nullptr, stream);
}

// TFR_TOWER_DZ_SR=0: round-to-nearest dz (rounds 1-3) instead of the stochastic rounding of pack_bf16_sr
static int tower_dz_sr() {
  static const int v = [] { const char* e = getenv("TFR_TOWER_DZ_SR"); return (e && *e) ? atoi(e) : 1; }();
  return v;
}

extern "C" int tfr_tower_out_bwd2(const void* z, long ldz, int M, int K, int prologue, const float* scale,
                                  const float* shift, const float* mean, const float* rstd, const float* w,
                                  const float* dlogits, int O, void* dy_bf16, long lddy, float* partial,
                                  int n_blocks, const tfr_tower_dropout* dropout, const float* pqr, void* stream) {
  if (!z || !w || !dlogits || (!dy_bf16 && !partial) || M <= 0 || K <= 0 || (K & 7) || (ldz & 7) || (lddy & 7) ||
      O < 1 || O > 4 || n_blocks < 1) return TFR_EINVAL;
  if (pqr && (!dy_bf16 || partial)) return TFR_EINVAL;       // the apply pass writes dz and nothing else
  if (!pqr && !partial) return TFR_EINVAL;                   // without coefficients the column sums are the point
  int act = 0;
  if (!split_mode(prologue, prologue, act)) return TFR_EINVAL;
  if (prologue != PRO_NONE && (!scale || !shift)) return TFR_EINVAL;
  int rows = (int)(((long)M + n_blocks - 1) / n_blocks);
  rows = (rows + 15) / 16 * 16;
  const size_t lds = (size_t)2048 * (2 + O) * sizeof(float);   // G * CW = 2048 either way
  hipStream_t st = (hipStream_t)stream;
  const Drop dr = to_drop(dropout);
  const int mode = pqr ? 2 : (dy_bf16 ? 0 : 1);
  const bool wide = K >= 512;
#define OB4(P, OT, LPR, MD) hipLaunchKernelGGL((tower_out_bwd_kernel<P, OT, LPR, MD>), dim3(n_blocks), dim3(256), MD == 2 ? 0 : lds, st, (const uint16_t*)z, ldz, M, K, scale, shift, mean, rstd, w, dlogits, O, (uint16_t*)dy_bf16, lddy, partial, rows, dr, pqr, act, tower_dz_sr())
#define OB3(P, OT, LPR) do { if (mode == 0) OB4(P, OT, LPR, 0); else if (mode == 1) OB4(P, OT, LPR, 1); else OB4(P, OT, LPR, 2); } while (0)
#define OB(P) do { if (O == 1) { if (wide) OB3(P, 1, 64); else OB3(P, 1, 16); } else { if (wide) OB3(P, 4, 64); else OB3(P, 4, 16); } } while (0)
  if (prologue == PRO_NONE) OB(PRO_NONE); else if (prologue == PRO_AFFINE) OB(PRO_AFFINE);
  else if (prologue == PRO_AFFINE_RELU) OB(PRO_AFFINE_RELU); else if (prologue == PRO_AFFINE_ACT) OB(PRO_AFFINE_ACT);
  else return TFR_EINVAL;
#undef OB
#undef OB3
#undef OB4
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_bn_bwd_apply(void* dy_bf16, long lddy, const void* z, long ldz, int M, int K,
                                      const float* pqr, void* stream) {
  if (!dy_bf16 || !z || !pqr || M < 0 || K <= 0 || (K & 7) || (lddy & 7) || (ldz & 7)) return TFR_EINVAL;
  if (3 * (size_t)K * 4 > 64 * 1024) return TFR_ETOOLARGE;
  if (M == 0) return TFR_OK;
  hipLaunchKernelGGL(tower_bn_bwd_apply_kernel, dim3(grid_for((long)M * (K / 8), 256) > 2048 ? 2048 : grid_for((long)M * (K / 8), 256)),
                     dim3(256), 3 * (size_t)K * 4, (hipStream_t)stream, (uint16_t*)dy_bf16, lddy,
                     (const uint16_t*)z, ldz, M, K, pqr, tower_dz_sr());
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_wgrad_bf16(const void* DZ, long lddz, const void* A, long lda, int M, int N, int K,
                                    int prologue, const float* a_scale, const float* a_shift, float* slab,
                                    long ldw, int splits, const tfr_tower_dropout* dropout, void* stream) {
  if (!DZ || !A || !slab || M <= 0 || N <= 0 || K <= 0 || (lddz & 7) || (lda & 7) || (N & 7) || (K & 7) ||
      ldw < K || splits < 1 || splits > 65535) return TFR_EINVAL;
  int act = 0;
  if (!split_mode(prologue, prologue, act) || (prologue != PRO_NONE && (!a_scale || !a_shift))) return TFR_EINVAL;
  WgradArgs g;
  g.act = act;
  g.DZ = (const uint16_t*)DZ; g.lddz = lddz; g.A = (const uint16_t*)A; g.lda = lda; g.a_scale = a_scale;
  g.a_shift = a_shift; g.slab = slab; g.ldw = ldw; g.M = M; g.N = N; g.K = K; g.drop = to_drop(dropout);
  hipStream_t st = (hipStream_t)stream;
  static const bool big = [] { const char* e = getenv("TFR_WGRAD_256"); return !(e && *e) || atoi(e) != 0; }();
  if (big && (N % 256) == 0 && (K % 64) == 0 && K >= 128 && g.drop.thr == 0 && ((long)M % (64L * splits)) == 0 &&
      lddz < (1L << 21) && lda < (1L << 21)) {
    g.rows_per_split = (int)((long)M / splits);
    g.splits = splits; g.tiles_n = N / 256; g.tiles_k = (K + 255) / 256;
    const dim3 grid256((splits + 7) / 8 * 8 * g.tiles_n * g.tiles_k);
#define WG2(P) do { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_wgrad256_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072); if (e != hipSuccess) return (int)e; hipLaunchKernelGGL(tower_wgrad256_kernel<P>, grid256, dim3(512), 131072, st, g); } while (0)
    if (prologue == PRO_NONE) WG2(PRO_NONE); else if (prologue == PRO_AFFINE) WG2(PRO_AFFINE);
    else if (prologue == PRO_AFFINE_RELU) WG2(PRO_AFFINE_RELU); else WG2(PRO_AFFINE_ACT);
#undef WG2
    return (int)hipGetLastError();
  }
  int rows = (int)(((long)M + splits - 1) / splits);
  g.rows_per_split = (rows + 63) / 64 * 64;
  g.splits = splits; g.tiles_n = (N + 127) / 128; g.tiles_k = (K + 127) / 128;
  const dim3 grid((splits + 7) / 8 * 8 * g.tiles_n * g.tiles_k);
#define WG(P) do { hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&tower_wgrad_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536); if (e != hipSuccess) return (int)e; hipLaunchKernelGGL(tower_wgrad_kernel<P>, grid, dim3(256), 65536, st, g); } while (0)
  if (prologue == PRO_NONE) WG(PRO_NONE); else if (prologue == PRO_AFFINE) WG(PRO_AFFINE);
  else if (prologue == PRO_AFFINE_RELU) WG(PRO_AFFINE_RELU); else WG(PRO_AFFINE_ACT);
#undef WG
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_slab_reduce(const float* slab, int S, long n, float* out, int accumulate, void* stream) {
  if (!slab || !out || S < 1 || n <= 0) return TFR_EINVAL;
  hipLaunchKernelGGL(tower_slab_reduce_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, slab, S,
                     n, out, accumulate);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_slab_reduce_cols(const float* slab, int S, int R, int Cs, int Cout, float* out, int accumulate,
                                          void* stream) {
  if (!slab || !out || S < 1 || R <= 0 || Cout <= 0 || Cs < Cout) return TFR_EINVAL;
  hipLaunchKernelGGL(tower_slab_reduce_cols_kernel, dim3(grid_for((long)R * Cout, 256)), dim3(256), 0, (hipStream_t)stream,
                     slab, S, R, Cs, Cout, out, accumulate);
  return (int)hipGetLastError();
}

extern "C" int tfr_tower_reduce_partials_coeffs_db(const float* partial, int T, int J, int N, float* out, float* scratch,
                                                   const float* gamma, const float* rstd, const float* mean, long M,
                                                   float* pqr, const float* dl, long Mr, int O, float* db, void* stream);
extern "C" int tfr_tower_reduce_partials_coeffs(const float* partial, int T, int J, int N, float* out, float* scratch,
                                                const float* gamma, const float* rstd, const float* mean, long M,
                                                float* pqr, void* stream) {
  return tfr_tower_reduce_partials_coeffs_db(partial, T, J, N, out, scratch, gamma, rstd, mean, M, pqr, nullptr, 0, 0, nullptr,
                                             stream);
}

// 1 when tfr_tower_reduce_partials_coeffs_db also serves `db` for this shape (the one-launch form)
extern "C" int tfr_tower_reduce_partials_serves_db(int T, int J) { return (T <= kFusedRows && J <= 6) ? 1 : 0; }

extern "C" int tfr_tower_reduce_partials_coeffs_db(const float* partial, int T, int J, int N, float* out, float* scratch,
                                                   const float* gamma, const float* rstd, const float* mean, long M,
                                                   float* pqr, const float* dl, long Mr, int O, float* db, void* stream) {
  if (!partial || T <= 0 || J <= 0 || N <= 0 || !out) return TFR_EINVAL;
  if (gamma && (!rstd || !mean || !pqr || M <= 0 || J < 2)) return TFR_EINVAL;
  if (dl && (!db || Mr <= 0 || O < 1 || O > 4 || !(T <= kFusedRows && J <= 6))) return TFR_EINVAL;
  hipStream_t st = (hipStream_t)stream;
  if (T <= kFusedRows && J <= 6) {
    hipLaunchKernelGGL(tower_reduce_partials_fused_kernel, dim3((N + 63) / 64 + (dl ? 1 : 0)), dim3(1024), 0, st, partial, T, J,
                       N, out, gamma, rstd, mean, gamma ? 1.0f / (float)M : 0.0f, pqr, dl, Mr, O, db);
    return (int)hipGetLastError();
  }
  const int rc = tfr_tower_reduce_partials(partial, T, J * N, out, scratch, stream);
  if (rc != TFR_OK || !gamma) return rc;
  return tfr_tower_bn_bwd_coeffs(gamma, rstd, mean, out, N, M, pqr, stream);
}

extern "C" int tfr_tower_bn_bwd_coeffs(const float* gamma, const float* rstd, const float* mean, const float* c,
                                       int N, long M, float* pqr, void* stream) {
  if (!gamma || !rstd || !mean || !c || !pqr || N <= 0 || M <= 0) return TFR_EINVAL;
  hipLaunchKernelGGL(tower_bn_bwd_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, gamma,
                     rstd, mean, c, N, 1.0f / (float)M, pqr);
  return (int)hipGetLastError();
}
#endif  // TFR_RP_DEV
