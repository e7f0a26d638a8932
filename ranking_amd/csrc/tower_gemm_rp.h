// tower_gemm_rp_kernel: the hidden-layer GEMM of the scorer tower with a RESIDENT weight panel and the activation operand
// loaded straight into MFMA-layout registers (round 6).  Included by tower.hip inside its anonymous namespace.
//
// Why (profiles/r06_fill_bench.txt, tools/fill_bench.hip): rounds 2-5 read the k loop of tower_gemm256p_kernel as bound by
// "LDS-DMA delivery, 25 GB/s per CU".  The fill micro-benchmark says otherwise: an LDS-DMA stream reaches 9.9 B/clk/CU from
// HBM -- the chip's HBM rate, whatever the path (LDS-DMA, global_load -> VGPR) -- and 47-57 B/clk/CU from L2.  The old loop
// asks for stage kt + 1 at the top of step kt and waits for it (vmcnt(0) + barrier) at the bottom: one memory round trip per
// k step with at most one 64 KB stage in flight per CU, half of it L2 hits.  Under full-chip load that round trip is
// ~2.2-2.6 us, the step's MFMAs need 0.9: the loop is LATENCY bound, and a 160 KB LDS cannot hold a third 64 KB stage.
//
// Here, per workgroup (512 threads = 8 wavefronts, one workgroup per CU, persistent):
//  * the workgroup keeps ONE 128-column n-tile for its whole life: its weight panel [128][K <= 512] bf16 (128 KB) is loaded
//    into LDS once, chunk-major ([K / 8][128 rows][16 B]: a fragment read is 16 lanes x 256 contiguous bytes, conflict
//    free without a swizzle).  No operand staging, no LDS-DMA and NO BARRIER in the steady state: the eight wavefronts run
//    free, so one wave's memory wait is covered by the MFMAs of the others;
//  * a wavefront owns 64 rows of a 512-row M-tile as two 32-row passes (2 x 8 accumulator tiles of 16 x 16 = 64 registers)
//    and loads ITS rows of the activation matrix directly from global memory into registers (quad-contiguous: four adjacent
//    lanes take one 64-byte segment; a ds_bpermute_b32 per register turns a block into MFMA operand layout, see a_off0):
//    a ring of four k steps in registers, the loads of step s + 3 issued before step s is consumed -- three k steps (12 KB
//    per wave, 96 KB per CU) in flight across pass and tile boundaries;
//  * the BatchNorm / activation / Dropout prologue is applied to those registers by the one wave that owns them (the old
//    4 x 2 wave layout transformed every fragment in two waves); the written operand (`Aout`) comes from that wave;
//  * the four n-tiles of an M-tile run on four CUs of one XCD at the same time (workgroup -> (XCD, group, n-tile)), so the
//    activation rows come from HBM once and from that XCD's L2 three times;
//  * BatchNorm statistics keep the layout of the other kernels (one row of partials per 64-row slab): a wave's two passes
//    are the two halves of ONE slab, the first pass parks its column sums in a private 1 KB LDS slot.
// Shapes: N % 128 == 0 with N / 128 dividing 32, K in {256, 512} (NK = K / 64 a multiple of the ring depth), full 512-row
// tiles; the launcher sends the last M % 512 rows through the older kernels.
constexpr int RP_BM = 512, RP_BN = 128, RP_RING = 4;
#ifndef TFR_RP_QA
#define TFR_RP_QA 1
#endif
constexpr int RP_QA = TFR_RP_QA;
// developer aid (tools/gemm_rp_ablate.py; results are garbage, timing only): 1 no activation loads inside the k loop,
// 2 no epilogue (nothing written), 4 no MFMAs, 8 no panel fragment reads
#ifndef TFR_RP_ABLATE
#define TFR_RP_ABLATE 0
#endif
constexpr int kRpAb = TFR_RP_ABLATE;
// mask 16: lane 0 of every wave records s_memtime at the start of a pass, after its k loop and after its epilogue (the first
// 64 passes) into a buffer set with tfr_prof_set_buffer_rp(): [workgroup][wave][pass][4] u64 (tools/gemm_rp_ablate.py timeline)
#if (TFR_RP_ABLATE & 16)
__device__ unsigned long long* g_prof_rp = nullptr;
#define RP_STAMP(i) do { if (lane == 0 && npass < 64) g_prof_rp[(((size_t)blockIdx.x * 8 + wave) * 64 + npass) * 4 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define RP_STAMP(i) do { } while (0)
#endif
constexpr int RP_PANEL = 0;                            // [K / 8][128][16 B]
constexpr int RP_SCALE = 128 * 1024;                   // [2][512] floats: prologue scale | shift
constexpr int RP_EPI = RP_SCALE + 2 * 512 * 4;         // [4][128] floats: bias | e_scale, e_shift, rstd, -mean rstd
constexpr int RP_STAGE = RP_EPI + 4 * 128 * 4;         // 8 waves x 1 KB: one [16 rows][32 columns] bf16 sub-chunk
constexpr int RP_STAT = RP_STAGE + 8 * 1024;           // 8 waves x 1 KB: pass 0's [2][128] column sums
constexpr int RP_LDS = RP_STAT + 8 * 1024;             // 153 600 B

// o[j] = w of lane (lane & 15) + 16 j: the four 16-lane rows of a wavefront exchange one word each (gfx950 lane swaps;
// inline asm with both operands tied -- see common.h wave_tree_sum for why not the builtin)
__device__ __forceinline__ void rows_allgather4(uint32_t w, uint32_t (&o)[4]) {
  uint32_t x = w, y = w;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(x), "+v"(y));      // x = rows (0, 1, 0, 1), y = rows (2, 3, 2, 3)
  uint32_t a = x, b = x, c = y, d = y;
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));      // a = row 0 everywhere, b = row 1
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(c), "+v"(d));      // c = row 2, d = row 3
  o[0] = a; o[1] = b; o[2] = c; o[3] = d;
}

typedef int rp_i32x4 __attribute__((ext_vector_type(4)));

template <int PRO, int EPI, int DROP, int NK, bool AOUT = false>
__global__ __launch_bounds__(512, 1) void tower_gemm_rp_kernel(const GemmArgs g) {
  static_assert(NK % RP_RING == 0, "the ring slot of a k step must be a compile-time constant");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr bool BWD = EPI == EPI_RELU_BWD || EPI == EPI_ACT_BWD;
  constexpr bool bt = DROP == 2;
  constexpr int ZEARLY = 8;                          // BWD: fragment columns of Zp requested ahead of the next pass's ring loads (8 = all: 32 registers)
  const Drop pdrop = drop_resolve(g.pro_drop), edrop = drop_resolve(g.epi_drop);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int fr = lane & 15, fq = lane >> 4;
  float* s_scale = reinterpret_cast<float*>(smem + RP_SCALE);
  float* s_shift = s_scale + 512;
  float* s_epi = reinterpret_cast<float*>(smem + RP_EPI);
  unsigned char* sw = smem + RP_STAGE + wave * 1024;
  float* s_stat = reinterpret_cast<float*>(smem + RP_STAT + wave * 1024);

  // workgroup -> (XCD, group of tiles_n CUs, n-tile); the groups of an XCD deal its M-tiles (tm = xcd, xcd + 8, ...) round robin
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tn = slot % g.tiles_n, grp = slot / g.tiles_n, ngrp = 32 / g.tiles_n;
  const int nx = (g.tiles_m - xcd + 7) >> 3;
  int j = grp;
  if (j >= nx) return;                              // (uniform per workgroup, before the only barrier)
  const int n0 = tn * RP_BN;
  int tm = xcd + 8 * j;

  // Activation loads: a lane CANNOT take its MFMA fragment straight from global memory at speed -- in operand layout adjacent
  // lanes are adjacent ROWS (1 KB apart) and the texture addresser serves a quad of lanes per cycle only when the quad's 64
  // bytes are contiguous: measured 15.9 B/clk/CU against 58 for quad-contiguous addresses (profiles/r06_fill_bench.txt,
  // PATH 5 / 7; the first version of this kernel was bound by exactly that: 2 060 cycles per k step).  So lane l LOADS row
  // (l >> 2), 16-byte piece (l & 3) of a (16-row, 32-column) block -- four adjacent lanes = one 64-byte segment -- and the
  // block is turned into operand layout (lane l <- lane 4 (l & 15) + (l >> 4)) by ds_bpermute_b32, the LDS crossbar (no LDS
  // memory), one (kt, kk) ahead of its MFMAs.
  const uint32_t a_off0 = (uint32_t)(((lane >> 2) * g.lda + (lane & 3) * 8) * 2), a_off1 = a_off0 + (uint32_t)(16 * g.lda * 2);
  const int bperm_src = (4 * fr + fq) * 4;
  const uint32_t z_off[2] = {(uint32_t)((fr * g.ldz + fq * 4) * 2), (uint32_t)(((16 + fr) * g.ldz + fq * 4) * 2)};           // BWD: Zp in accumulator layout
  const int ao_off[2] = {(int)((fr * g.ldao + fq * 8) * 2), (int)(((16 + fr) * g.ldao + fq * 8) * 2)};                           // AOUT: the written operand
  const float* epi_l = s_epi + fq * 4;               // the lane's four columns of fragment column fn: + fn * 16 (+ 128 j)
  float* stat_l = s_stat + fq * 4;
  auto a_rows = [&](int tm_, int pass_) __attribute__((always_inline)) {
    return reinterpret_cast<const char*>(g.A) + ((long)(tm_ * RP_BM + wave * 64 + pass_ * 32) * g.lda) * 2;
  };
  // The k ORDER is rotated per n-tile: position kt of the loop works on k block ka[kt] = (kt + tn * NK / tiles_n) mod NK.
  // The tiles_n workgroups of a group read the same activation rows at the same time; in the same k order all of them ask for
  // the same lines at once, every request waits a full HBM round trip and the bytes in flight that are UNIQUE are a quarter
  // of the registers spent on them (measured: tools/fill_bench.hip PATH 9 -- 3.7 TB/s instead of 5.9).  Rotated, each
  // workgroup misses on NK / tiles_n blocks of a pass and finds the others in L2, fetched by its neighbours two steps
  // earlier; the four requests for one 1 KB row are in flight together (one DRAM page).  fp32 accumulation order differs
  // from the round-5 kernels for tn > 0: results agree to fp32 rounding, not bit for bit.
  int ka[NK];
  {
    const int rot = (g.flags & 1) ? 0 : (g.tiles_n <= NK ? tn * (NK / g.tiles_n) : tn % NK);
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) ka[kt] = (kt + rot) & (NK - 1);
  }
  uint4 ar[RP_RING][4];                              // [ring slot][fm * 2 + kk]
#define RP_LOAD(SLOT, BASE, KT)                                                                  \
  do {                                                                                           \
    ar[SLOT][0] = *reinterpret_cast<const uint4*>((BASE) + a_off0 + ka[KT] * 128);                  \
    ar[SLOT][1] = *reinterpret_cast<const uint4*>((BASE) + a_off0 + ka[KT] * 128 + 64);             \
    ar[SLOT][2] = *reinterpret_cast<const uint4*>((BASE) + a_off1 + ka[KT] * 128);                  \
    ar[SLOT][3] = *reinterpret_cast<const uint4*>((BASE) + a_off1 + ka[KT] * 128 + 64);             \
  } while (0)

  // ---- the first three k steps of the first pass go out before anything else
  const char* abase = a_rows(tm, 0);
  RP_LOAD(0, abase, 0); RP_LOAD(1, abase, 1); RP_LOAD(2, abase, 2);

  // ---- one-time set-up: weight panel (8 lanes = 8 rows of one chunk column: 128 contiguous LDS bytes; the 64 lanes of an
  // instruction cover 8 full 128-byte lines), prologue / epilogue coefficients
  {
    const uint16_t* bp = g.B + (long)n0 * g.ldb;
    const int nchunk8 = g.K / 64;                    // groups of 8 chunks along k
    for (int it = wave; it < 16 * nchunk8; it += 8) {
      const int nb = it % 16, cb = it / 16;
      const int n = nb * 8 + (lane & 7), c = cb * 8 + (lane >> 3);
      const uint4 v = *reinterpret_cast<const uint4*>(bp + (long)n * g.ldb + c * 8);
      *reinterpret_cast<uint4*>(smem + RP_PANEL + c * 2048 + n * 16) = v;
    }
    if (PRO != PRO_NONE) {
      const float fold = bt ? pdrop.scale : 1.0f;   // (x s + h) * 2 == x (2 s) + 2 h exactly
      for (int k = tid; k < g.K; k += 512) { s_scale[k] = g.a_scale[k] * fold; s_shift[k] = g.a_shift[k] * fold; }
    }
    if (tid < RP_BN) {
      const int n = n0 + tid;
      if (BWD) {
        const float rs = g.e_rstd[n];
        s_epi[tid] = g.e_scale[n]; s_epi[128 + tid] = g.e_shift[n];
        s_epi[256 + tid] = rs; s_epi[384 + tid] = -g.e_mean[n] * rs;
      } else {
        s_epi[tid] = g.bias ? g.bias[n] : 0.f;
      }
    }
  }
  __syncthreads();
  // De-phase the two wavefronts of every SIMD (waves w and w + 4).  All eight waves run the same instruction stream on equal
  // work: started together they STAY together -- all in the k loop, then all in the epilogue -- and the phases of the launch
  // add up instead of overlapping (ablations, tools/gemm_rp_ablate.py: MFMA 108 + panel reads 82 + activation loads 67 +
  // epilogue 114 = the 371 us of the full kernel).  Half a pass of delay for waves 4-7, once, puts one wave of a SIMD in its
  // epilogue / memory waits while the other feeds the matrix pipe.  flags bits 8-15: the delay in units of 2 048 cycles.
  if (wave >= 4) {
    const int units = (g.flags >> 8) & 0xff;
    for (int i = 0; i < units; ++i) __builtin_amdgcn_s_sleep(32);
  }

  const unsigned char* pb_lane = smem + RP_PANEL + fq * 2048 + fr * 16;
  // the operand a wave forms in its prologue is written out for the weight gradient by ONE of the tiles_n workgroups that
  // form it: n-tile tn takes the k steps kt with kt mod min(tiles_n, NK) == tn
  int store_mask = 0;
  if (AOUT) {
    const int tdiv = g.tiles_n < NK ? g.tiles_n : NK;
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) store_mask |= ((kt % tdiv) == tn) ? (1 << kt) : 0;      // (kt = loop position: any partition of the k blocks does)
  }

  // epilogue addressing (tile independent): the lane's 8-byte slot of fragment column f2 (0 / 1) in a [16][32] sub-chunk,
  // and its 16-byte row-major piece
  uint32_t stg_acc[2];
#pragma unroll
  for (int f2 = 0; f2 < 2; ++f2) {
    const int P = f2 * 2 + (fq >> 1);               // 16-byte piece of the lane's four columns
    stg_acc[f2] = (uint32_t)(fr * 64 + ((P ^ ((fr >> 1) & 3)) << 4) + (fq & 1) * 8);
  }
  const int rm_row = lane >> 2, rm_p = lane & 3;
  const uint32_t stg_rm = (uint32_t)(rm_row * 64 + ((rm_p ^ ((rm_row >> 1) & 3)) << 4));
  const uint32_t offC = (uint32_t)((rm_row * g.ldc + rm_p * 8) * 2);

  int pass = 0;
  int npass = 0; (void)npass;
  while (true) {
    const int jn = j + ngrp;
    const bool have_next = jn < nx;
    const int tmn = xcd + 8 * jn;
    const bool more = pass == 0 || have_next;         // is there a pass after this one?
    const char* nbase = pass == 0 ? a_rows(tm, 1) : a_rows(have_next ? tmn : tm, 0);      // (no next pass: valid rows, loaded and dropped)
    const int m0 = tm * RP_BM + wave * 64 + pass * 32;

    f32x4 acc[8][2];
#pragma unroll
    for (int a = 0; a < 8; ++a) { acc[a][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[a][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    uint2 zreg[2][8];                                  // BWD: the pass's Zp tile in accumulator layout

    // The k loop, fully unrolled and software pipelined BY HAND -- left alone, the scheduler sinks the ring loads to just
    // before their use (register pressure) and reads the panel two fragments at a time behind a wait each:
    //  * the global loads of step kt + 3 sit between two sched_barriers at the top of step kt (nothing moves across);
    //  * the A operand of the NEXT (kt, kk) -- prologue transform, written operand -- is formed beside the MFMAs of the current one;
    //  * the panel fragments of the NEXT quad (four fragments = eight MFMAs) are read before the MFMAs of the current quad.
    uint32_t wk[4] = {0u, 0u, 0u, 0u};
    auto prep = [&](int kt, int kk, bf16x8 (&fa)[2]) __attribute__((always_inline)) {
      const int sl = kt % RP_RING;
#pragma unroll
      for (int fm = 0; fm < 2; ++fm) {
        const uint4 v = ar[sl][fm * 2 + kk];
        rp_i32x4 t;
        t[0] = __builtin_amdgcn_ds_bpermute(bperm_src, (int)v.x); t[1] = __builtin_amdgcn_ds_bpermute(bperm_src, (int)v.y);
        t[2] = __builtin_amdgcn_ds_bpermute(bperm_src, (int)v.z); t[3] = __builtin_amdgcn_ds_bpermute(bperm_src, (int)v.w);
        fa[fm] = __builtin_bit_cast(bf16x8, t);
      }
      if (PRO != PRO_NONE) {
        uint32_t bits[2] = {0u, 0u};
        if (bt) {      // the k step's keep words (rate 1/2: a hash word serves 32 columns of a row): ONE hash per lane -- lane row fq takes
                       // (fm, kk) = (fq >> 1, fq & 1) -- exchanged across the four 16-lane rows; same words and bits as drop_run
          if (kk == 0) {
            const uint32_t own = drop_hash(pdrop.seed, (uint32_t)(g.row0 + m0 + (fq >> 1) * 16 + fr), (uint32_t)(2 * ka[kt] + (fq & 1)));
            rows_allgather4(own, wk);
          }
          bits[0] = wk[kk] >> (fq * 8); bits[1] = wk[2 + kk] >> (fq * 8);
        }
        const int k = ka[kt] * BK + kk * 32 + fq * 8;
        const f32x4 sc0 = *reinterpret_cast<const f32x4*>(s_scale + k), sc1 = *reinterpret_cast<const f32x4*>(s_scale + k + 4);
        const f32x4 sh0 = *reinterpret_cast<const f32x4*>(s_shift + k), sh1 = *reinterpret_cast<const f32x4*>(s_shift + k + 4);
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
          fa[fm] = transform_frag<PRO, DROP>(fa[fm], sc0, sc1, sh0, sh1, pdrop, (uint32_t)(g.row0 + m0 + fm * 16 + fr), (uint32_t)k,
                                             g.act, bits[fm]);
        // the operand this wave formed, for the weight gradient.  Whether THIS workgroup stores k step kt is uniform, but a
        // branch here (also the exec-mask form: the compiler skips a masked store with s_cbranch_execz) would cut the k step
        // into scheduling regions and the MFMAs could no longer interleave with the next operand's VALU.  The stores are
        // buffer stores through a resource whose size is 0 when they are not this workgroup's: the hardware drops them.
        if (AOUT) {
          const uint32_t span = ((store_mask >> kt) & 1) ? (uint32_t)(32 * g.ldao * 2) : 0u;
          const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(g.Aout + (long)m0 * g.ldao, 0, (int)span, 0x00020000);
#pragma unroll
          for (int fm = 0; fm < 2; ++fm)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(rp_i32x4, fa[fm]), rs, ao_off[fm], ka[kt] * 128 + kk * 64, 0);
        }
      }
    };
    auto read_quad = [&](int qi, bf16x8 (&fb)[4]) __attribute__((always_inline)) {       // qi = (kt * 2 + kk) * 2 + half
      const unsigned char* pk = pb_lane + (ka[qi >> 2] * 8 + ((qi >> 1) & 1) * 4) * 2048 + (qi & 1) * 4 * 256;
#pragma unroll
      for (int f = 0; f < 4; ++f) fb[f] = *reinterpret_cast<const bf16x8*>(pk + f * 256);
    };
    RP_STAMP(0);
    // panel fragments: RP_QA quads (of four fragments = eight MFMAs) requested ahead of the one being multiplied
    bf16x8 fa_cur[2], fa_nxt[2], fbq[RP_QA + 1][4];
    prep(0, 0, fa_cur);
#pragma unroll
    for (int q0 = 0; q0 < RP_QA; ++q0) read_quad(q0, fbq[q0]);
#pragma unroll
    for (int kt = 0; kt < NK; ++kt) {
      __builtin_amdgcn_sched_barrier(0);
      if (BWD && kt == (NK >= 4 ? NK - 4 : 0)) {       // ahead of the next pass's ring loads: loads return in order
        const char* zb = reinterpret_cast<const char*>(g.Zp) + ((long)m0 * g.ldz + n0) * 2;
#pragma unroll
        for (int fm = 0; fm < 2; ++fm)
#pragma unroll
          for (int fn = 0; fn < ZEARLY; ++fn)       // (two per-lane offsets + immediates: sixteen precomputed 64-bit offsets were spilled)
            zreg[fm][fn] = *reinterpret_cast<const uint2*>(zb + z_off[fm] + fn * 32);
      }
      // the loads of step kt + 3: this pass, or the first steps of the next one (after the last pass: of this one again --
      // 12 KB per wave once per launch instead of a branch in every step)
      if (!(kRpAb & 1)) {
        if (kt + 3 < NK) RP_LOAD((kt + 3) % RP_RING, abase, kt + 3);
        else RP_LOAD((kt + 3) % RP_RING, nbase, kt + 3 - NK);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int kn = kt * 2 + kk + 1;                // the next (kt, kk)
        if (kn < NK * 2) prep(kn >> 1, kn & 1, fa_nxt);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int qi = (kt * 2 + kk) * 2 + half, cur = qi % (RP_QA + 1);
          if (qi + RP_QA < NK * 4 && !(kRpAb & 8)) read_quad(qi + RP_QA, fbq[(qi + RP_QA) % (RP_QA + 1)]);
#pragma unroll
          for (int f = 0; f < 4; ++f) {
            if (kRpAb & 4) {                           // (keeps the operands live without the matrix pipe)
              acc[half * 4 + f][0][0] += __builtin_bit_cast(f32x4, fbq[cur][f])[0] * __builtin_bit_cast(f32x4, fa_cur[0])[0];
              acc[half * 4 + f][1][0] += __builtin_bit_cast(f32x4, fbq[cur][f])[1] * __builtin_bit_cast(f32x4, fa_cur[1])[1];
              continue;
            }
            acc[half * 4 + f][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbq[cur][f], fa_cur[0], acc[half * 4 + f][0], 0, 0, 0);
            acc[half * 4 + f][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fbq[cur][f], fa_cur[1], acc[half * 4 + f][1], 0, 0, 0);
          }
          // the order inside the quad: the LDS reads first (the next quad's four fragments; with a prologue, in the first quad
          // of a (kt, kk), also the four scale / shift reads of the next operand), then the eight MFMAs with the next
          // operand's VALU dealt out between them (VPM per MFMA: a 16-cycle MFMA covers ~3-4 independent VALU issues)
          constexpr int VPM = PRO == PRO_NONE ? 0 : (DROP == 0 ? 3 : 6);
          // (DS group of the first quad of a (kt, kk): 4 panel fragments + the 8 ds_bpermute of the next operand [+ 4 scale / shift reads])
          if (half == 0) __builtin_amdgcn_sched_group_barrier(0x080, PRO != PRO_NONE ? 16 : 12, 0);
          else __builtin_amdgcn_sched_group_barrier(0x080, 4, 0);
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            if (VPM) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
          }
        }
        fa_cur[0] = fa_nxt[0]; fa_cur[1] = fa_nxt[1];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    RP_STAMP(1);
    if (BWD && ZEARLY < 8) {                           // the second column half of Zp: behind the ring loads, used after the first half's chunks
      const char* zb = reinterpret_cast<const char*>(g.Zp) + ((long)m0 * g.ldz + n0) * 2;
#pragma unroll
      for (int fm = 0; fm < 2; ++fm)
#pragma unroll
        for (int fn = ZEARLY; fn < 8; ++fn)
          zreg[fm][fn] = *reinterpret_cast<const uint2*>(zb + z_off[fm] + fn * 32);
    }

    // ---- epilogue of the pass: 32 rows x 128 columns as 2 (fm) x 4 (column pairs of fragments) sub-chunks of [16][32] through
    // the wave's 1 KB slot (LDS operations of one wave execute in order: no wait between a sub-chunk's read and the next write)
    char* cb = reinterpret_cast<char*>(g.C) + ((long)m0 * g.ldc + n0) * 2;
    if (kRpAb & 2) {                                   // (one store per wave and pass keeps the accumulators live)
      f32x4 t = acc[0][0];
#pragma unroll
      for (int a = 0; a < 8; ++a) { t += acc[a][0]; t += acc[a][1]; }
      if (t[0] + t[1] + t[2] + t[3] == 12345.678f) *reinterpret_cast<f32x4*>(cb + offC) = t;
    } else
#pragma clang loop unroll(full)
    for (int h = 0; h < 2; ++h) {                      // 64-column halves: the statistics of a half live in 32 registers
      f32x4 s1[4], s2[4];
#pragma unroll
      for (int f = 0; f < 4; ++f) { s1[f] = f32x4{0.f, 0.f, 0.f, 0.f}; s2[f] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int p2 = 0; p2 < 2; ++p2) {
#pragma unroll
        for (int fm = 0; fm < 2; ++fm) {
#pragma unroll
          for (int f2 = 0; f2 < 2; ++f2) {
            const int f = p2 * 2 + f2, fn = h * 4 + f;
            const int col = fn * 16 + fq * 4;
            f32x4 v = acc[fn][fm];
            const f32x4 pbv = *reinterpret_cast<const f32x4*>(epi_l + fn * 16);
            if (BWD) {
              const f32x4 pev = *reinterpret_cast<const f32x4*>(epi_l + 128 + fn * 16);
              const uint2 zz = zreg[fm][fn];
              const f32x4 z = {bf16_lo(zz.x), bf16_hi(zz.x), bf16_lo(zz.y), bf16_hi(zz.y)};
              const f32x4 y = z * pbv + pev;
              if (DROP) {
                float kf[4];
                drop_run<4, DROP == 3>(edrop, (uint32_t)(g.row0 + m0 + fm * 16 + fr), (uint32_t)(n0 + col), kf);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] *= kf[r];
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) v[r] = (EPI == EPI_ACT_BWD) ? v[r] * act_grad(g.act, y[r]) : (y[r] > 0.f ? v[r] : 0.f);
              s1[f] += v;
              s2[f] += v * z;
            } else {
              v += pbv;
              if (EPI == EPI_STATS) { s1[f] += v; s2[f] += v * v; }
            }
            *reinterpret_cast<uint2*>(sw + stg_acc[f2]) = make_uint2(pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]));
          }
          const uint4 o = *reinterpret_cast<const uint4*>(sw + stg_rm);
          *reinterpret_cast<uint4*>(cb + (long)(fm * 16) * g.ldc * 2 + (h * 64 + p2 * 32) * 2 + offC) = o;
          __builtin_amdgcn_sched_barrier(0);           // (keeps the coefficient reads of later sub-chunks from being hoisted: registers)
        }
      }
      if (EPI != EPI_PLAIN) {
        // column sums of the pass's 32 rows (lane 15 of every 16-lane row); pass 0 parks them, pass 1 adds and writes the slab's row
        float* const st = g.stats + ((long)(tm * 8 + wave) * 2) * g.N + n0 + h * 64;
#pragma unroll
        for (int f = 0; f < 4; ++f) {
          f32x4 a = row16_sum4(s1[f]), b = row16_sum4(s2[f]);
          const int cf = h * 64 + f * 16;              // (+ fq * 4: in epi_l / stat_l)
          if (fr == 15) {
            if (pass == 0) {
              *reinterpret_cast<f32x4*>(stat_l + cf) = a;
              *reinterpret_cast<f32x4*>(stat_l + 128 + cf) = b;
            } else {
              a += *reinterpret_cast<const f32x4*>(stat_l + cf);
              b += *reinterpret_cast<const f32x4*>(stat_l + 128 + cf);
              if (BWD) {                               // sum dy * zhat = rstd * sum dy z - mean rstd * sum dy
                const f32x4 er = *reinterpret_cast<const f32x4*>(epi_l + 256 + cf);
                const f32x4 c2 = *reinterpret_cast<const f32x4*>(epi_l + 384 + cf);
                b = b * er + a * c2;
              }
              *reinterpret_cast<f32x4*>(st + f * 16 + fq * 4) = a;
              *reinterpret_cast<f32x4*>(st + g.N + f * 16 + fq * 4) = b;
            }
          }
        }
      }
    }

    RP_STAMP(2);
    ++npass;
    if (!more) break;
    abase = nbase;
    if (pass == 1) { j = jn; tm = tmn; }
    pass ^= 1;
  }
#undef RP_LOAD
}
