// tower_gemm_bs_kernel: WEIGHT-STATIONARY tower GEMM for the forms without a prologue -- plain products and the dgrad with its
// ReLU-backward epilogue (round 6).  Included by tower.hip inside its anonymous namespace.
//
// What rounds 2-6 measured about a bf16 GEMM with K = N = 512 on gfx950 (profiles/r06_gemm_findings.txt):
//  * operand FILL costs matrix-pipe time: every KB that enters a SIMD's registers from LDS (ds_read_b128) holds that SIMD's
//    MFMA issue for ~16 cycles -- "MFMA + fragment reads" is the SUM of the two, in the 256 x 256 kernel (round 2 stamps: 2360
//    -> 3130 cycles per k step) as in the resident-panel kernel (ablations: 108 -> 190 us), whatever the read-ahead depth.
//    For a wave tile Mw x Nw the fill per 16-cycle MFMA is 16 (1 / Mw + 1 / Nw) cycles: 64 x 128 tiles top out at 73 % of the
//    MFMA rate, 32 x 128 at 62 %;
//  * a per-step `vmcnt(0) + barrier` with ONE stage in flight makes a k step last one memory round trip (the 256 x 256 kernel).
// Here the WEIGHTS never move: a workgroup of four wavefronts (one per SIMD, 512 registers each) keeps a 256-column n-tile for
// its whole life, wave w holding W[n0 + 64 w .. + 64][0 .. 512) as 64 MFMA fragments in 256 registers.  Only the activations
// are filled: fill per MFMA = 16 / Nw = a quarter of its time, whatever the tile height.  So the M-tile is small (64 rows: 64
// accumulator registers) and the LDS holds nothing but an 8-stage ring of activation stages (64 rows x 128 k = 16 KB, LDS-DMA,
// source-swizzled): while stage ks of a tile is multiplied, stage ks of the NEXT tile is requested -- four stages (64 KB per
// CU) in flight across tile boundaries, one workgroup barrier per 64 MFMAs, no prologue arithmetic anywhere near the loop.
// Shapes: K = 512, N % 256 == 0 with N / 256 dividing 32, full 64-row tiles (the rest goes through the older kernels).
constexpr int BS_BM = 64, BS_BN = 256, BS_BK = 128, BS_NK = 4;       // rows and columns of a workgroup tile, k per stage, stages per tile (K = 512)
constexpr int BS_STAGE = BS_BM * BS_BK * 2;                          // 16 KB
constexpr int BS_EPI = 8 * BS_STAGE;                                 // [4][256] floats: bias | e_scale, e_shift, rstd, -mean rstd
constexpr int BS_STG = BS_EPI + 4 * 256 * 4;                         // 4 waves x 2 KB: one [16 rows][64 columns] bf16 chunk
constexpr int BS_LDS = BS_STG + 4 * 2048;                            // 143 360 B

__device__ __forceinline__ void bs_dma16(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}
template <int N> __device__ __forceinline__ void bs_wait_barrier() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(N) : "memory");     // (lgkmcnt(0): this wave's reads of the previous stage are done -- its slot may be re-filled after the barrier)
}

// Counted waits.  All loads of this kernel are hand-issued (LDS-DMA stages, the dgrad's Zp rows) and waited for by COUNT: VMEM
// operations return in order, so "stage s + 1 has landed" = "at most N operations are outstanding", N = everything issued after
// that stage's four pieces.  With stages requested D steps ahead (step s requests stage s + D) that is, at the wait of step
// s = 4 ti + ks: the D - 1 stages s + 2 .. s + D, the epilogues (E operations each) that ran after step s - (D - 1) and the Zp
// batches (ZP each, issued at the top of a tile's first step) of steps s - (D - 2) .. s.
constexpr int bs_count(int ks, int lo_back, int hi_back, int residue, int tiles_so_far_max) {
  // how many j in [s - lo_back, s - hi_back] have j % 4 == residue, for s % 4 == ks; at most `tiles_so_far_max` of them exist yet
  int n = 0;
  for (int b = hi_back; b <= lo_back; ++b) n += (((ks - b) % 4 + 4) % 4 == residue) ? 1 : 0;
  return n < tiles_so_far_max ? n : tiles_so_far_max;
}

// developer aid (tools/gemm_bs_timeline.py): -DTFR_BS_STAMPS -- lane 0 of wave 0 records s_memtime per tile (the first 48 tiles
// of a workgroup): [2 ks] before the stage wait, [2 ks + 1] after its barrier, [8] start of the epilogue, [9] its end
#ifdef TFR_BS_STAMPS
__device__ unsigned long long* g_prof_bs = nullptr;
#define BS_STAMP(i) do { if (tid == 0 && ti < 48) g_prof_bs[((size_t)blockIdx.x * 48 + ti) * 10 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define BS_STAMP(i) do { } while (0)
#endif

template <int EPI, int DROP, int D>
__global__ __launch_bounds__(256, 1) void tower_gemm_bs_kernel(const GemmArgs g) {
  static_assert(D >= 2 && D <= 7, "8 ring slots: the slot of stage s - 1 is free once every wave has passed the barrier of stage s");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool BWD = EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD;
  // VMEM operations of one tile's epilogue (the counted waits below step over them): 8 row-major stores of C (+ 8 statistic stores)
  constexpr int E = BWD ? 16 : 8;
  constexpr int ZP = BWD ? 8 : 0;                    // Zp loads of a tile, issued at the top of its first stage
  const Drop edrop = drop_resolve(g.epi_drop);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  float* s_epi = reinterpret_cast<float*>(smem + BS_EPI);
  unsigned char* sw = smem + BS_STG + wave * 2048;

  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;           // 32 slots per XCD; the n-tile of a workgroup never changes
  const int nq = ((g.tiles_m - xcd + 7) >> 3) * g.tiles_n;          // this XCD's tiles (M-tiles xcd, xcd + 8, ...)
  if (slot >= nq) return;                                           // (uniform, before any barrier)
  const int tn = slot % g.tiles_n;
  const int ntile = (nq - slot + 31) >> 5;                          // q = slot, slot + 32, ...
  const int nw0 = tn * BS_BN + wave * 64;                           // the wave's 64 columns
  auto tm_of = [&](int ti) __attribute__((always_inline)) { return ((slot + 32 * ti) / g.tiles_n) * 8 + xcd; };

  // ---- the wave's weights -> registers, once: W[nw0 .. + 64][0 .. 512) through the (still empty) ring, a k half at a time,
  // chunk-major ([32 chunks of 8 k][256 n][16 B]: copied in with 8 lanes = 8 rows of a chunk column, read back as fragments
  // -- 16 lanes x 256 contiguous bytes -- without a bank conflict)
  bf16x8 breg[4][16];                                               // [fn][k / 32]
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const uint16_t* bp = g.B + (long)(tn * BS_BN) * g.ldb + half * 256;
    for (int it = wave; it < 32 * 4; it += 4) {                      // 32 row groups x 4 chunk groups
      const int nb = it & 31, cb = it >> 5;
      const int n = nb * 8 + (lane & 7), c = cb * 8 + (lane >> 3);
      const uint4 v = *reinterpret_cast<const uint4*>(bp + (long)n * g.ldb + c * 8);
      *reinterpret_cast<uint4*>(smem + c * 4096 + n * 16) = v;
    }
    __syncthreads();
#pragma unroll
    for (int fn = 0; fn < 4; ++fn)
#pragma unroll
      for (int cc = 0; cc < 8; ++cc)
        breg[fn][half * 8 + cc] = *reinterpret_cast<const bf16x8*>(smem + (cc * 4 + fq) * 4096 + (wave * 64 + fn * 16 + fr) * 16);
    __syncthreads();
  }
  if (tid < BS_BN) {
    const int n = tn * BS_BN + tid;
    if (BWD) {
      const float rs = g.e_rstd[n];
      s_epi[tid] = g.e_scale[n]; s_epi[256 + tid] = g.e_shift[n];
      s_epi[512 + tid] = rs; s_epi[768 + tid] = -g.e_mean[n] * rs;
    } else {
      s_epi[tid] = g.bias ? g.bias[n] : 0.f;
    }
  }

  // ---- activation stages: [64 rows][16 chunks of 16 B], chunk position = chunk ^ (row & 15) (applied to the SOURCE address: the
  // LDS image of an LDS-DMA is lane-linear); a wave requests rows 16 w .. + 16 as four 1 KB pieces of 4 rows
  uint32_t offA[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = wave * 16 + i * 4 + (lane >> 4);
    offA[i] = (uint32_t)((row * g.lda + (((lane & 15) ^ (row & 15)) << 3)) * 2);
  }
  auto issue_stage = [&](int ti, int ks) __attribute__((always_inline)) {          // (ti past the last tile: the last tile again -- keeps the counts uniform)
    const int tc = ti < ntile ? ti : ntile - 1;
    const char* sb = reinterpret_cast<const char*>(g.A) + ((long)tm_of(tc) * BS_BM * g.lda + ks * BS_BK) * 2;
    const uint32_t dst = lds0 + (((ti & 1) * BS_NK + ks) * BS_STAGE) + wave * 4096;       // stage s = 4 ti + ks lives in slot s % 8
#pragma unroll
    for (int i = 0; i < 4; ++i) bs_dma16(offA[i], sb, dst + i * 1024);
  };
  // fragment (fm, kk) of a stage: row fm * 16 + fr, chunk kk * 4 + fq at position (kk * 4 + fq) ^ fr
  uint32_t loff[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) loff[kk] = (uint32_t)(fr * 256 + ((((kk << 2) | fq) ^ fr) << 4));
  auto read_frags = [&](int ti, int ks, int kk, bf16x8 (&fa)[4]) __attribute__((always_inline)) {
    const unsigned char* st = smem + ((ti & 1) * BS_NK + ks) * BS_STAGE + loff[kk];       // slot (4 ti + ks) % 8
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) fa[fm] = *reinterpret_cast<const bf16x8*>(st + fm * 4096);
  };

  // epilogue addressing (as in the 256 x 256 kernels): row-major 16-byte pieces of a [16][64] chunk, two per lane; the lane's
  // 8-byte slot of fragment column fn
  uint32_t offC[2], offZ[2], stg_rm[2], stg_acc[4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int qq = lane + 64 * i, row = qq >> 3, cc = qq & 7;
    offC[i] = (uint32_t)((row * g.ldc + cc * 8) * 2);
    offZ[i] = (uint32_t)((row * g.ldz + cc * 8) * 2);
    stg_rm[i] = (uint32_t)(row * 128 + ((cc ^ (row & 7)) << 4));
  }
#pragma unroll
  for (int fn = 0; fn < 4; ++fn) stg_acc[fn] = (uint32_t)(fr * 128 + (((fn * 2 + (fq >> 1)) ^ (fr & 7)) << 4) + (fq & 1) * 8);

  // ---- the first D stages, then the first fragments
#pragma unroll
  for (int s0 = 0; s0 < D; ++s0) issue_stage(s0 >> 2, s0 & 3);
  bf16x8 fa[2][4];
  rp_i32x4 zq[8];                                     // BWD: the tile's Zp rows (row-major pieces: chunk fm = zq[2 fm], zq[2 fm + 1])
  bs_wait_barrier<4 * (D - 1)>();                     // stage 0 of every wave has landed (and s_epi is written)
  read_frags(0, 0, 0, fa[0]);

  for (int ti = 0; ti < ntile; ++ti) {
    const int tm = tm_of(ti);
    const long m0 = (long)tm * BS_BM;
    f32x4 acc[4][4];                                  // [fn][fm]
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto stage = [&](auto ks_c) __attribute__((always_inline)) {        // (a generic lambda: ks is a compile-time constant for the counted waits)
      constexpr int ks = decltype(ks_c)::value;
      if (BWD && ks == 0) {                           // the tile's Zp rows, AHEAD of the next tile's stages in the (in-order) VMEM queue
        const char* zb = reinterpret_cast<const char*>(g.Zp) + (m0 * g.ldz + nw0) * 2;
#pragma unroll
        for (int c = 0; c < 8; ++c)
          asm volatile("global_load_dwordx4 %0, %1, %2" : "=&v"(zq[c]) : "v"(offZ[c & 1] + (uint32_t)((c >> 1) * 16 * g.ldz * 2)), "s"(zb) : "memory");
      }
      issue_stage(ti + ((ks + D) >> 2), (ks + D) & 3);          // stage s + D: its slot (s + D) % 8 was last read in step s + D - 8 <= s - 1
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        const int cur = kk & 1;
        if (kk < 3) {
          read_frags(ti, ks, kk + 1, fa[cur ^ 1]);
        } else {
          // the next stage has landed in every wave (counts: see bs_count)
          constexpr int base = 4 * (D - 1);
          constexpr int n2 = base + E * bs_count(ks, D - 1, 1, 3, 2) + ZP * bs_count(ks, D - 2, 0, 0, 2);
          constexpr int n1 = base + E * bs_count(ks, D - 1, 1, 3, 1) + ZP * bs_count(ks, D - 2, 0, 0, 2);
          constexpr int n0 = base + ZP * bs_count(ks, D - 2, 0, 0, 1);
          static_assert(n2 <= 63, "vmcnt is a 6-bit field");
          BS_STAMP(2 * ks);
          if (ti >= 2) bs_wait_barrier<n2>(); else if (ti == 1) bs_wait_barrier<n1>(); else bs_wait_barrier<n0>();
          BS_STAMP(2 * ks + 1);
          if (ks < 3) read_frags(ti, ks + 1, 0, fa[cur ^ 1]); else read_frags(ti + 1, 0, 0, fa[cur ^ 1]);
        }
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
#pragma unroll
          for (int fm = 0; fm < 4; ++fm)
            acc[fn][fm] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(breg[fn][ks * 4 + kk], fa[cur][fm], acc[fn][fm], 0, 0, 0);
      }
    };
    stage(std::integral_constant<int, 0>{}); stage(std::integral_constant<int, 1>{});
    stage(std::integral_constant<int, 2>{}); stage(std::integral_constant<int, 3>{});
    // (fa[0] now holds the next tile's first fragments: 4 kk per stage, an even number of swaps)

    BS_STAMP(8);
    // ---- epilogue: 4 chunks of [16 rows][64 columns] through the wave's 2 KB slot
    if (BWD) asm volatile("s_waitcnt vmcnt(16)" : "+v"(zq[0]), "+v"(zq[1]), "+v"(zq[2]), "+v"(zq[3]), "+v"(zq[4]), "+v"(zq[5]), "+v"(zq[6]), "+v"(zq[7]) :: "memory");   // behind the Zp loads: the 16 pieces of stages (ti + 1, 0 .. 3)
    f32x4 pb[4], pe[4], s1[4], s2[4];
#pragma unroll
    for (int fn = 0; fn < 4; ++fn) {
      pb[fn] = *reinterpret_cast<const f32x4*>(s_epi + wave * 64 + fn * 16 + fq * 4);
      if (BWD) pe[fn] = *reinterpret_cast<const f32x4*>(s_epi + 256 + wave * 64 + fn * 16 + fq * 4);
      s1[fn] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[fn] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    char* cb = reinterpret_cast<char*>(g.C) + (m0 * g.ldc + nw0) * 2;
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      if (BWD) {
        *reinterpret_cast<rp_i32x4*>(sw + stg_rm[0]) = zq[2 * fm];
        *reinterpret_cast<rp_i32x4*>(sw + stg_rm[1]) = zq[2 * fm + 1];
      }
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        f32x4 v = acc[fn][fm];
        if (BWD) {
          const uint2 zz = *reinterpret_cast<const uint2*>(sw + stg_acc[fn]);
          const f32x4 z = {bf16_lo(zz.x), bf16_hi(zz.x), bf16_lo(zz.y), bf16_hi(zz.y)};
          const f32x4 y = z * pb[fn] + pe[fn];
          if (DROP) {
            float kf[4];
            drop_run<4, DROP == 3>(edrop, (uint32_t)(g.row0 + m0 + fm * 16 + fr), (uint32_t)(nw0 + fn * 16 + fq * 4), kf);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] *= kf[r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = (EPI == EPI_ACT_BWD) ? v[r] * act_grad(g.act, y[r]) : (y[r] > 0.f ? v[r] : 0.f);
          s1[fn] += v;
          s2[fn] += v * z;
        } else {
          v += pb[fn];
        }
        *reinterpret_cast<uint2*>(sw + stg_acc[fn]) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
      }
      const uint4 o0 = *reinterpret_cast<const uint4*>(sw + stg_rm[0]);
      const uint4 o1 = *reinterpret_cast<const uint4*>(sw + stg_rm[1]);
      char* cc = cb + (long)fm * 16 * g.ldc * 2;
      *reinterpret_cast<uint4*>(cc + offC[0]) = o0;
      *reinterpret_cast<uint4*>(cc + offC[1]) = o1;
    }
    if (BWD) {                                        // one row of partials per 64-row slab = per tile
      float* const st = g.stats + ((long)tm * 2) * g.N + nw0;
#pragma unroll
      for (int fn = 0; fn < 4; ++fn) {
        f32x4 a = row16_sum4(s1[fn]), b = row16_sum4(s2[fn]);
        const f32x4 er = *reinterpret_cast<const f32x4*>(s_epi + 512 + wave * 64 + fn * 16 + fq * 4);
        const f32x4 c2 = *reinterpret_cast<const f32x4*>(s_epi + 768 + wave * 64 + fn * 16 + fq * 4);
        b = b * er + a * c2;                          // sum dy * zhat = rstd * sum dy z - mean rstd * sum dy
        // (every lane issues the two stores -- a fixed number of VMEM operations per tile for the counted waits above --, the
        // lanes that do not hold the row sums under a zeroed exec mask)
        if (fr == 15) {
          *reinterpret_cast<f32x4*>(st + fn * 16 + fq * 4) = a;
          *reinterpret_cast<f32x4*>(st + g.N + fn * 16 + fq * 4) = b;
        }
      }
    }
    BS_STAMP(9);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the dummy stages past the last tile land before the workgroup leaves
}
