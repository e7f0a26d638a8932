// Per-CU FILL-rate micro-benchmark for the tower GEMM's operand staging ("is 6.4 TB/s the limit of the
// LDS-DMA path or of the chip?").  One 512-thread workgroup per CU streams 64 KB "k steps" through one of the paths:
//   PATH 0  global_load_lds_dwordx4 (LDS-DMA), 8 pieces per wave and step           -- what tower_gemm256p does for A and B
//   PATH 1  global_load_dwordx4 -> VGPR (consumed by an xor)                         -- "A fragments straight into registers"
//   PATH 2  global_load_dwordx4 -> VGPR -> ds_write_b128                             -- the register-staged path of round 1
//   PATH 3  half the bytes by PATH 1 from the streamed source, half by PATH 0 from a 512 KB panel (L2 resident)
//           -- the proposed split: activations direct to registers, the weight panel by LDS-DMA
//   PATH 4  PATH 3 + every wave ds_read_b128's the whole 32 KB panel stage (256 KB of LDS reads per step and CU)
//   PATH 5  global_load_dwordx4 -> VGPR in MFMA OPERAND LAYOUT: lane l takes 16 B of row (l & 15) at chunk (l >> 4) of a
//           [256 rows][1024 B] tile -- adjacent lanes are 1 KB apart, four NON-adjacent lanes cover a 64-byte segment
//           (what tower_gemm_rp.h's first version did for its activation operand); 32 KB per step and workgroup
//   PATH 6  the same bytes with lane l taking row (l >> 3), 16-byte piece (l & 7): eight adjacent lanes = one 128-byte line
//   PATH 7  lane l takes row (l >> 2), piece (l & 3): four adjacent lanes = one 64-byte segment (16 rows per instruction)
//   PATH 8  lane l takes row (l >> 1), piece (l & 1): two adjacent lanes = 32 bytes (32 rows per instruction)
//   PATH 9  HBM stream, 32 KB per step and workgroup with three steps in flight (the register ring of tower_gemm_rp.h), but the
//           FOUR workgroups of a group (same XCD: slots 4 g .. 4 g + 3) read the SAME slice -- the four n-tile CUs of an M-tile:
//           the unique bytes in flight are a quarter of the requested ones.  DEPTH here = the variant: 1 = as described,
//           2 = every workgroup also touches (one dword per 128-byte line) its quarter of the lines 8 steps ahead,
//           3 = no sharing (every workgroup its own slice): the reference
// SRC 0: every workgroup streams its own slice of a 2 GiB buffer (HBM); SRC 1: every workgroup re-reads one 512 KB panel
// (L2 resident after the first pass).  DEPTH = steps in flight (vmcnt-counted, 1..3).
//   hipcc --offload-arch=gfx950 -O3 -o tools/fill_bench tools/fill_bench.hip && tools/fill_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int STEP = 65536;                 // bytes per workgroup and step
constexpr int PIECE = 1024;                 // one wave-instruction of 16 B per lane

__device__ __forceinline__ void dma16(uint32_t voff, const void* sbase, uint32_t lds_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory");
}

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int PATH, int DEPTH>
__global__ __launch_bounds__(512, 1) void fill_kernel(const unsigned char* __restrict__ src, long wg_stride, int wrap, int n_steps,
                                                      const unsigned char* __restrict__ panel, uint32_t* __restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // DEPTH+1 stage buffers of 64 KB (<= 2 used at once per path)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
  const unsigned char* base = src + (long)blockIdx.x * wg_stride;
  const uint32_t voff = (uint32_t)(lane * 16);
  uint4 acc = make_uint4(0, 0, 0, 0);
  constexpr int NB = 2;                       // LDS stage buffers (64 KB each)
  // pieces per wave and step: 8 (64 KB / 8 waves / 1 KB)
  auto issue = [&](int s) __attribute__((always_inline)) {
    const unsigned char* sb = base + (long)(s % wrap) * STEP + wave * 8 * PIECE;
    const uint32_t dst = lds0 + (s % NB) * STEP + wave * 8 * PIECE;
#pragma unroll
    for (int i = 0; i < 8; ++i) dma16(voff, sb + i * PIECE, dst + i * PIECE);
  };
  if (PATH == 0) {
    for (int s = 0; s < DEPTH; ++s) issue(s);
    for (int s = 0; s < n_steps; ++s) {
      // wait for step s (the oldest of the DEPTH in flight)
      if (DEPTH == 1) wait_vm<0>(); else if (DEPTH == 2) wait_vm<8>(); else wait_vm<16>();
      __builtin_amdgcn_s_barrier();
      // "consume": one ds_read per lane so the stage is touched
      acc.x ^= *reinterpret_cast<const uint32_t*>(smem + (s % NB) * STEP + tid * 16);
      __builtin_amdgcn_s_barrier();
      if (NB > DEPTH || true) issue(s + DEPTH);          // (re-reads wrap inside the slice: same bytes per step)
    }
    wait_vm<0>();
  } else if (PATH == 1 || PATH == 2) {
    uint4 r[3][8];
    auto load = [&](int s, uint4 (&d)[8]) __attribute__((always_inline)) {
      const unsigned char* sb = base + (long)(s % wrap) * STEP + wave * 8 * PIECE + lane * 16;
#pragma unroll
      for (int i = 0; i < 8; ++i) d[i] = *reinterpret_cast<const uint4*>(sb + i * PIECE);
    };
    auto use = [&](int s, uint4 (&d)[8]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (PATH == 2) *reinterpret_cast<uint4*>(smem + (s % NB) * STEP + wave * 8 * PIECE + i * PIECE + lane * 16) = d[i];
        else { acc.x ^= d[i].x; acc.y ^= d[i].y; acc.z ^= d[i].z; acc.w ^= d[i].w; }
      }
    };
    load(0, r[0]);
    if (DEPTH >= 2) load(1, r[1]);
    if (DEPTH >= 3) load(2, r[2]);
    for (int s = 0; s < n_steps; s += DEPTH) {          // unrolled by DEPTH so that the register sets stay static
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        use(s + d, r[d]);
        load(s + d + DEPTH, r[d]);
        if (PATH == 2) { __builtin_amdgcn_s_barrier(); }
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) use(d, r[d]);
  } else if (PATH >= 5 && PATH <= 8) {
    uint4 r[3][4];
    auto load = [&](int s, uint4 (&d)[4]) __attribute__((always_inline)) {
      // tile: [256 rows][1024 B] (L2 resident: src = panel-sized region); wave w owns rows 32 w .., step s the 128-byte column block s % 8
      const unsigned char* sb = base + (long)(wave * 32) * 1024 + (s % 8) * 128;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int off = PATH == 5 ? (((i >> 1) * 16 + (lane & 15)) * 1024 + (i & 1) * 64 + (lane >> 4) * 16)
                      : PATH == 6 ? ((i * 8 + (lane >> 3)) * 1024 + (lane & 7) * 16)
                      : PATH == 7 ? (((i >> 1) * 16 + (lane >> 2)) * 1024 + (i & 1) * 64 + (lane & 3) * 16)
                                  : ((lane >> 1) * 1024 + i * 32 + (lane & 1) * 16);
        d[i] = *reinterpret_cast<const uint4*>(sb + off);
      }
    };
    auto use = [&](uint4 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x ^= d[i].x; acc.y ^= d[i].y; acc.z ^= d[i].z; acc.w ^= d[i].w; }
    };
    load(0, r[0]);
    if (DEPTH >= 2) load(1, r[1]);
    if (DEPTH >= 3) load(2, r[2]);
    for (int s = 0; s < n_steps; s += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) { use(r[d]); load(s + d + DEPTH, r[d]); }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) use(r[d]);
  } else if (PATH == 9) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, grp = slot >> 2, tn = slot & 3;
    const unsigned char* gb = src + (long)((DEPTH == 3 ? (int)blockIdx.x : (xcd * 8 + grp) * 4)) * wg_stride;   // the slice (shared by the group unless DEPTH == 3)
    uint4 r[3][4];
    uint32_t tacc = 0;
    auto load = [&](int s, uint4 (&d)[4]) __attribute__((always_inline)) {
      const unsigned char* sb = gb + (long)(s % wrap) * (STEP / 2) + wave * 4 * PIECE + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const uint4*>(sb + i * PIECE);
    };
    auto use = [&](uint4 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x ^= d[i].x; acc.y ^= d[i].y; acc.z ^= d[i].z; acc.w ^= d[i].w; }
    };
    load(0, r[0]); load(1, r[1]); load(2, r[2]);
    for (int s = 0; s < n_steps; s += 3) {
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (DEPTH == 2 && ((s + d) & 7) == 0) {
          // 8 steps = 256 KB of the slice = 2048 lines; this workgroup's quarter = 512 lines = 8 waves x 64 lanes: one dword each
          const int line = (wave * 64 + lane) * 4 + tn;
          tacc ^= *reinterpret_cast<const uint32_t*>(gb + (long)(((s + d) / 8 + 1) * 8 % wrap) * (STEP / 2) + (long)line * 128);
        }
        use(r[d]); load(s + d + 3, r[d]);
      }
    }
    use(r[0]); use(r[1]); use(r[2]);
    acc.x ^= tacc;
  } else {                                    // PATH 3 / 4: 32 KB direct (4 loads per lane) + 32 KB panel by LDS-DMA (4 pieces per wave)
    uint4 r[3][4];
    auto load = [&](int s, uint4 (&d)[4]) __attribute__((always_inline)) {
      const unsigned char* sb = base + (long)(s % wrap) * (STEP / 2) + wave * 4 * PIECE + lane * 16;
#pragma unroll
      for (int i = 0; i < 4; ++i) d[i] = *reinterpret_cast<const uint4*>(sb + i * PIECE);
      const unsigned char* pb = panel + (long)(s % 16) * (STEP / 2) + wave * 4 * PIECE;
      const uint32_t dst = lds0 + (s % NB) * (STEP / 2) + wave * 4 * PIECE;
#pragma unroll
      for (int i = 0; i < 4; ++i) dma16(voff, pb + i * PIECE, dst + i * PIECE);
    };
    auto use = [&](int s, uint4 (&d)[4]) __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < 4; ++i) { acc.x ^= d[i].x; acc.y ^= d[i].y; acc.z ^= d[i].z; acc.w ^= d[i].w; }
      if (PATH == 4) {                        // every wave reads the whole 32 KB panel stage
        const unsigned char* st = smem + (s % NB) * (STEP / 2) + lane * 16;
#pragma unroll
        for (int i = 0; i < 32; ++i) { const uint4 v = *reinterpret_cast<const uint4*>(st + i * PIECE); acc.x ^= v.x; acc.y ^= v.w; }
      }
    };
    load(0, r[0]);
    if (DEPTH >= 2) load(1, r[1]);
    if (DEPTH >= 3) load(2, r[2]);
    for (int s = 0; s < n_steps; s += DEPTH) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        // the 8 VMEM ops of step s + d are the oldest: wait until only the younger (DEPTH - 1) * 8 remain
        if (DEPTH == 1) wait_vm<0>(); else if (DEPTH == 2) wait_vm<8>(); else wait_vm<16>();
        __builtin_amdgcn_s_barrier();
        use(s + d, r[d]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        load(s + d + DEPTH, r[d]);
      }
    }
    wait_vm<0>();
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[blockIdx.x * 512 + tid] = acc.x;
}

template <typename Fn>
double time_ms(Fn launch, int reps = 4) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch(); launch();
  CHECK(hipDeviceSynchronize());
  double best = 1e30;
  for (int r = 0; r < reps; ++r) {
    CHECK(hipEventRecord(a));
    launch();
    CHECK(hipEventRecord(b));
    CHECK(hipEventSynchronize(b));
    float ms; CHECK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  const int WGS = 256;
  const long total = 2L << 30;                       // 2 GiB streamed source (beyond the 256 MB Infinity Cache)
  unsigned char *src, *panel; uint32_t* out;
  CHECK(hipMalloc(&src, total)); CHECK(hipMalloc(&panel, 512 * 1024)); CHECK(hipMalloc(&out, WGS * 512 * 4));
  CHECK(hipMemset(src, 1, total)); CHECK(hipMemset(panel, 2, 512 * 1024));
  const char* pn[] = {"LDS-DMA (global_load_lds_dwordx4)", "global_load_dwordx4 -> VGPR", "global_load_dwordx4 -> VGPR -> ds_write_b128",
                      "half direct (stream) + half LDS-DMA (L2 panel)", "  + every wave reads the 32 KB panel stage",
                      "dwordx4 -> VGPR, MFMA operand layout (32 KB / step)", "dwordx4 -> VGPR, 8 lanes per 128-B line (32 KB / step)",
                      "dwordx4 -> VGPR, 4 lanes per 64-B segment (32 KB / step)", "dwordx4 -> VGPR, 2 lanes per 32 B (32 KB / step)",
                      "ring of 3 x 32 KB, 4 CUs share a slice (1) + touch ahead (2) / own slice (3)"};
  printf("%-52s %-4s %5s %9s %10s %12s %10s\n", "path", "src", "depth", "ms", "TB/s chip", "GB/s per CU", "B/clk/CU@2.4");
#define RUN(P, D, SRCK) { \
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&fill_kernel<P, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STEP)); \
    const bool hbm = (SRCK) == 0; \
    const long stride = hbm ? total / WGS : 0;                  /* L2: every workgroup re-reads the same 512 KB */ \
    const int per_step = (P >= 3) ? STEP / 2 : STEP;            /* streamed bytes per step */ \
    const int p9 = (P == 9); \
    const int wrap = hbm ? (int)(stride / per_step) : (512 * 1024) / per_step; \
    const int n_steps = hbm ? (p9 ? wrap - 24 : ((P >= 3) ? wrap / 2 : wrap)) / 6 * 6 : 6000; \
    double ms = time_ms([&] { hipLaunchKernelGGL((fill_kernel<P, D>), dim3(WGS), dim3(512), 2 * STEP, 0, hbm ? src : panel, stride, wrap, n_steps, panel, out); }); \
    const double bytes = (double)WGS * n_steps * ((P >= 5) ? STEP / 2 : STEP) / ((P == 9 && D != 3) ? 4 : 1);   /* PATH 9: UNIQUE bytes */ \
    printf("%-52s %-4s %5d %9.4f %10.3f %12.1f %10.1f\n", pn[P], hbm ? "HBM" : "L2", D, ms, bytes / ms / 1e9, bytes / ms / 1e6 / WGS, bytes / (ms * 1e-3) / WGS / 2.4e9); }
  RUN(0, 1, 0) RUN(0, 2, 0) RUN(0, 1, 1) RUN(0, 2, 1)
  RUN(1, 1, 0) RUN(1, 2, 0) RUN(1, 3, 0) RUN(1, 1, 1) RUN(1, 2, 1) RUN(1, 3, 1)
  RUN(2, 1, 0) RUN(2, 2, 0) RUN(2, 1, 1) RUN(2, 2, 1)
  RUN(3, 1, 0) RUN(3, 2, 0) RUN(3, 3, 0) RUN(3, 2, 1) RUN(3, 3, 1)
  RUN(4, 2, 0) RUN(4, 3, 0) RUN(4, 3, 1)
  RUN(5, 1, 1) RUN(5, 2, 1) RUN(5, 3, 1) RUN(6, 1, 1) RUN(6, 2, 1) RUN(6, 3, 1)
  RUN(7, 1, 1) RUN(7, 2, 1) RUN(8, 1, 1) RUN(8, 2, 1)
  RUN(9, 3, 0) RUN(9, 1, 0) RUN(9, 2, 0) RUN(9, 3, 0) RUN(9, 1, 0) RUN(9, 2, 0)
  return 0;
}
