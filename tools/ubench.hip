// Micro-benchmarks that size the VALU / transcendental roofline of the pair
// kernels on gfx950: wave-instruction issue cost of v_fma/v_pk_fma/v_rcp/v_exp/
// v_log and of candidate inner loops of the ApproxNDCG pair sweep.
//   hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip && ./ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef float float2v __attribute__((ext_vector_type(2)));

constexpr int ITERS = 4096;

template <int OP>
__global__ void k_inst(float* out, float seed) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
  const float b = seed * 0.5f + 1.0f, c = seed + 0.25f;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) a[i] = __builtin_fmaf(a[i], b, c);
      if (OP == 1) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == 2) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (OP == 3) a[i] = __builtin_amdgcn_logf(a[i]);
      if (OP == 4) a[i] = a[i] + b;
      if (OP == 5) a[i] = __builtin_amdgcn_sqrtf(a[i]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}


// Issue cost of the non-FMA instructions the pair kernels lean on (round 6): one instruction per asm statement, eight
// independent chains.
template <int OP>
__global__ void k_iop(unsigned* out, unsigned seed) {
  unsigned a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 3u + i;
  const unsigned b = seed * 5u + 1u, c = seed + 77u;
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == 0) asm volatile("v_sad_u16 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 2) asm volatile("v_and_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 3) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 5) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]));
      if (OP == 6) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(b));
      if (OP == 7) asm volatile("v_sad_u32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      if (OP == 8) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 9) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(a[i]), "v"(b) : "vcc");
      if (OP == 11) asm volatile("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
      if (OP == 12) asm volatile("v_bcnt_u32_b32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 13) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
      if (OP == 14) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(a[i]) : "s20");
      if (OP == 15) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b));
    }
  }
  unsigned sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) sum += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

__global__ void k_pkfma(float* out, float seed) {
  float2v a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = float2v{seed + threadIdx.x * 1e-3f + i, seed + i};
  const float2v b = {seed * 0.5f + 1.0f, seed * 0.25f + 1.0f}, c = {seed + 0.25f, seed};
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], b, c);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// Candidate inner loops.  LDS holds N floats F (and A); each lane sweeps all of it.
// VAR 0: fwd scalar   acc += rcp(fma(E, F, 1))
// VAR 1: fwd packed   pk_fma + 2 rcp + pk_add
// VAR 2: bwd scalar   s = rcp(fma(E,F,1)); acc = fma(A - ak, fma(-s,s,s), acc)
// VAR 3: bwd packed
// VAR 4: fwd scalar with exp path  acc += rcp(1 + exp2(x - F))
template <int VAR>
__global__ void k_pair(float* out, float seed, int N) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* F = lds;
  float* A = lds + N;
  for (int i = threadIdx.x; i < N; i += blockDim.x) { F[i] = seed + i * 1e-3f; A[i] = seed * i; }
  __syncthreads();
  const float E = seed + threadIdx.x * 1e-4f, ak = seed * 3.f;
  const float4* F4 = reinterpret_cast<const float4*>(F);
  const float4* A4 = reinterpret_cast<const float4*>(A);
  float a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  float2v p0 = {0, 0}, p1 = {0, 0};
  for (int rep = 0; rep < 16; ++rep) {
    for (int g = 0; g < N / 4; ++g) {
      const float4 f = F4[g];
      if (VAR == 0) {
        a0 += __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.x, 1.0f));
        a1 += __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.y, 1.0f));
        a2 += __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.z, 1.0f));
        a3 += __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.w, 1.0f));
      } else if (VAR == 1) {
        const float2v e2 = {E, E}, one = {1.0f, 1.0f};
        float2v u0 = __builtin_elementwise_fma(e2, float2v{f.x, f.y}, one);
        float2v u1 = __builtin_elementwise_fma(e2, float2v{f.z, f.w}, one);
        u0 = float2v{__builtin_amdgcn_rcpf(u0.x), __builtin_amdgcn_rcpf(u0.y)};
        u1 = float2v{__builtin_amdgcn_rcpf(u1.x), __builtin_amdgcn_rcpf(u1.y)};
        p0 += u0; p1 += u1;
      } else if (VAR == 2) {
        const float4 aj = A4[g];
        const float s0 = __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.x, 1.0f));
        const float s1 = __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.y, 1.0f));
        const float s2 = __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.z, 1.0f));
        const float s3 = __builtin_amdgcn_rcpf(__builtin_fmaf(E, f.w, 1.0f));
        a0 = __builtin_fmaf(aj.x - ak, __builtin_fmaf(-s0, s0, s0), a0);
        a1 = __builtin_fmaf(aj.y - ak, __builtin_fmaf(-s1, s1, s1), a1);
        a2 = __builtin_fmaf(aj.z - ak, __builtin_fmaf(-s2, s2, s2), a2);
        a3 = __builtin_fmaf(aj.w - ak, __builtin_fmaf(-s3, s3, s3), a3);
      } else if (VAR == 3) {
        const float4 aj = A4[g];
        const float2v e2 = {E, E}, one = {1.0f, 1.0f}, ak2 = {ak, ak};
        float2v u0 = __builtin_elementwise_fma(e2, float2v{f.x, f.y}, one);
        float2v u1 = __builtin_elementwise_fma(e2, float2v{f.z, f.w}, one);
        u0 = float2v{__builtin_amdgcn_rcpf(u0.x), __builtin_amdgcn_rcpf(u0.y)};
        u1 = float2v{__builtin_amdgcn_rcpf(u1.x), __builtin_amdgcn_rcpf(u1.y)};
        const float2v d0 = __builtin_elementwise_fma(-u0, u0, u0);
        const float2v d1 = __builtin_elementwise_fma(-u1, u1, u1);
        p0 = __builtin_elementwise_fma(float2v{aj.x, aj.y} - ak2, d0, p0);
        p1 = __builtin_elementwise_fma(float2v{aj.z, aj.w} - ak2, d1, p1);
      } else if (VAR == 4) {
        a0 += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(E - f.x));
        a1 += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(E - f.y));
        a2 += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(E - f.z));
        a3 += __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(E - f.w));
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y;
}


// LambdaRank group-kernel trips (ranking_amd/csrc/lambdarank_group.h): records (B, rank) as float2 in LDS, the
// rank-difference table replicated per bank, v_sad_u16 address, two columns per lane and trip.
// VAR 0: hi trip (fma, rcp, log, 2 fma per column)   VAR 1: lo trip (fma, rcp, fma per column)
// DEPTH: trips in flight (1 = load, wait, compute; 2 / 3 = software pipelined like the kernel)
typedef const __attribute__((address_space(3))) float lds_cf;
__device__ __forceinline__ float gat(unsigned ri, float rj, unsigned ubase) {
  return *(lds_cf*)(uintptr_t)__builtin_amdgcn_sad_u16(ri, (unsigned)__float_as_int(rj), ubase);
}
template <int VAR, int DEPTH>
__global__ void k_trip(float* out, float seed, int NREC, int L) {
  extern __shared__ __attribute__((aligned(128))) float lds[];
  float* U = lds;                      // [L * 32]
  float2* rec = reinterpret_cast<float2*>(lds + L * 32);   // [NREC + 16]
  for (int i = threadIdx.x; i < L * 32; i += blockDim.x) U[i] = 1.0f / (1.0f + (i >> 5));
  for (int i = threadIdx.x; i < NREC + 16; i += blockDim.x) rec[i] = make_float2(0.5f + (i % 7) * 0.1f, __int_as_float(((i * 37) % L) * 128));
  __syncthreads();
  const int lane = threadIdx.x & 63, c = lane & 1;
  const unsigned ubase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)U + 4u * (lane & 31);
  const unsigned ri = (unsigned)(((lane * 13) % L) * 128);
  const float Ai = seed + lane * 1e-3f;
  const float4* p = reinterpret_cast<const float4*>(rec + 2 * c);
  const int trips = NREC / 4;
  float al = 0.f, ag = 0.f;
  for (int rep = 0; rep < 64; ++rep) {
#define TRIP(CR, U0, U1)                                                             \
    do {                                                                             \
      const float w0 = __builtin_fmaf(Ai, CR.x, 1.0f), w1 = __builtin_fmaf(Ai, CR.z, 1.0f);   \
      const float q0 = __builtin_amdgcn_rcpf(w0), q1 = __builtin_amdgcn_rcpf(w1);     \
      if (VAR == 0) {                                                                \
        const float l0 = __builtin_amdgcn_logf(w0), l1 = __builtin_amdgcn_logf(w1);   \
        al = __builtin_fmaf(U0, l0, al); al = __builtin_fmaf(U1, l1, al);             \
      }                                                                              \
      ag = __builtin_fmaf(U0, 1.0f - q0, ag); ag = __builtin_fmaf(U1, 1.0f - q1, ag); \
    } while (0)
    if (DEPTH == 1) {
      for (int t = 0; t < trips; ++t) {
        const float4 cr = p[2 * t];
        const float u0 = gat(ri, cr.y, ubase), u1 = gat(ri, cr.w, ubase);
        TRIP(cr, u0, u1);
      }
    } else {
      float4 ca = p[0], cb = p[2], cc = p[4];
      float ua0 = gat(ri, ca.y, ubase), ua1 = gat(ri, ca.w, ubase);
      float ub0 = gat(ri, cb.y, ubase), ub1 = gat(ri, cb.w, ubase);
      for (int t = 0; t + 3 <= trips; t += 3) {
        float4 cd = p[2 * t + 6];
        float uc0 = gat(ri, cc.y, ubase), uc1 = gat(ri, cc.w, ubase);
        TRIP(ca, ua0, ua1);
        float4 ce = p[2 * t + 8];
        float ud0 = gat(ri, cd.y, ubase), ud1 = gat(ri, cd.w, ubase);
        TRIP(cb, ub0, ub1);
        float4 cf = p[2 * t + 10];
        float ue0 = gat(ri, ce.y, ubase), ue1 = gat(ri, ce.w, ubase);
        TRIP(cc, uc0, uc1);
        ca = cd; ua0 = ud0; ua1 = ud1; cb = ce; ub0 = ue0; ub1 = ue1; cc = cf;
      }
    }
#undef TRIP
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = al + ag;
}


// Round 6 candidate: the same trips with TWO ROWS PER LANE and four lanes per row pair -- lane c of a quad takes ONE
// column of a 4-column trip (ds_read_b64: 4 LDS cycles per wave instruction instead of the 8 of a b128) for both of
// its rows: per 128 pairs the LDS port serves 4 + 2 * 2 = 8 cycles instead of 8 + 2 * 2 = 12.
template <int VAR>
__global__ void k_trip2(float* out, float seed, int NREC, int L) {
  extern __shared__ __attribute__((aligned(128))) float lds[];
  float* U = lds;
  float2* rec = reinterpret_cast<float2*>(lds + L * 32);
  for (int i = threadIdx.x; i < L * 32; i += blockDim.x) U[i] = 1.0f / (1.0f + (i >> 5));
  for (int i = threadIdx.x; i < NREC + 16; i += blockDim.x) rec[i] = make_float2(0.5f + (i % 7) * 0.1f, __int_as_float(((i * 37) % L) * 128));
  __syncthreads();
  const int lane = threadIdx.x & 63, c = lane & 3;
  const unsigned ubase = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void*)U + 4u * (lane & 31);
  const unsigned ri0 = (unsigned)(((lane * 13) % L) * 128), ri1 = (unsigned)(((lane * 29 + 7) % L) * 128);
  const float Ai0 = seed + lane * 1e-3f, Ai1 = seed * 0.9f + lane * 2e-3f;
  const float2* p = rec + c;
  const int trips = NREC / 4;
  float al0 = 0.f, ag0 = 0.f, al1 = 0.f, ag1 = 0.f;
  for (int rep = 0; rep < 64; ++rep) {
#define TRIP2(CR, U0, U1)                                                            \
    do {                                                                             \
      const float w0 = __builtin_fmaf(Ai0, CR.x, 1.0f), w1 = __builtin_fmaf(Ai1, CR.x, 1.0f);   \
      const float q0 = __builtin_amdgcn_rcpf(w0), q1 = __builtin_amdgcn_rcpf(w1);     \
      if (VAR == 0) {                                                                \
        const float l0 = __builtin_amdgcn_logf(w0), l1 = __builtin_amdgcn_logf(w1);   \
        al0 = __builtin_fmaf(U0, l0, al0); al1 = __builtin_fmaf(U1, l1, al1);         \
      }                                                                              \
      ag0 = __builtin_fmaf(U0, 1.0f - q0, ag0); ag1 = __builtin_fmaf(U1, 1.0f - q1, ag1); \
    } while (0)
    float2 ca = p[0], cb = p[4], cc = p[8];
    float ua0 = gat(ri0, ca.y, ubase), ua1 = gat(ri1, ca.y, ubase);
    float ub0 = gat(ri0, cb.y, ubase), ub1 = gat(ri1, cb.y, ubase);
    for (int t = 0; t + 3 <= trips; t += 3) {
      float2 cd = p[4 * t + 12];
      float uc0 = gat(ri0, cc.y, ubase), uc1 = gat(ri1, cc.y, ubase);
      TRIP2(ca, ua0, ua1);
      float2 ce = p[4 * t + 16];
      float ud0 = gat(ri0, cd.y, ubase), ud1 = gat(ri1, cd.y, ubase);
      TRIP2(cb, ub0, ub1);
      float2 cf = p[4 * t + 20];
      float ue0 = gat(ri0, ce.y, ubase), ue1 = gat(ri1, ce.y, ubase);
      TRIP2(cc, uc0, uc1);
      ca = cd; ua0 = ud0; ua1 = ud1; cb = ce; ub0 = ue0; ub1 = ue1; cc = cf;
    }
#undef TRIP2
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = al0 + ag0 + al1 + ag1;
}

template <typename Fn>
double time_ms(Fn launch, int reps = 5) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch();
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
  const int blocks = 256 * 8, threads = 256;
  float* out; CHECK(hipMalloc(&out, blocks * threads * sizeof(float)));
  const double waves = (double)blocks * threads / 64.0;
  const char* names[] = {"v_fma_f32", "v_rcp_f32", "v_exp_f32", "v_log_f32", "v_add_f32", "v_sqrt_f32"};
  printf("%-14s %10s %14s %22s\n", "inst", "ms", "Gwaveinst/s", "cyc/waveinst/SIMD@2.4GHz");
#define RUN_INST(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_inst<OP>, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); }); \
    double wi = waves * ITERS * 8; printf("%-14s %10.4f %14.2f %22.2f\n", names[OP], ms, wi / ms / 1e6, 1024.0 * 2.4e9 * ms * 1e-3 / wi); }
  RUN_INST(0) RUN_INST(1) RUN_INST(2) RUN_INST(3) RUN_INST(4) RUN_INST(5)
  { double ms = time_ms([&] { hipLaunchKernelGGL(k_pkfma, dim3(blocks), dim3(threads), 0, 0, out, 1.5f); });
    double wi = waves * ITERS * 8; printf("%-14s %10.4f %14.2f %22.2f\n", "v_pk_fma_f32", ms, wi / ms / 1e6, 1024.0 * 2.4e9 * ms * 1e-3 / wi); }
  {
    const char* in[] = {"v_sad_u16", "v_add_u32", "v_and_b32", "v_lshl_add_u32", "v_mad_u32_u24", "v_cvt_u32_f32", "v_cndmask_b32", "v_sad_u32",
                        "v_sub_f32", "v_max_f32", "v_cmp_gt_f32", "v_mov_b32_dpp", "v_bcnt_u32_b32", "v_mul_f32", "v_readlane_b32", "v_sub_u32"};
#define RUN_IOP(OP) { double ms = time_ms([&] { hipLaunchKernelGGL(k_iop<OP>, dim3(blocks), dim3(threads), 0, 0, (unsigned*)out, 3u); }); \
    double wi = waves * ITERS * 8; printf("%-14s %10.4f %14.2f %22.2f\n", in[OP], ms, wi / ms / 1e6, 1024.0 * 2.4e9 * ms * 1e-3 / wi); }
    RUN_IOP(0) RUN_IOP(1) RUN_IOP(2) RUN_IOP(3) RUN_IOP(4) RUN_IOP(5) RUN_IOP(6) RUN_IOP(7) RUN_IOP(8) RUN_IOP(9) RUN_IOP(10) RUN_IOP(11)
    RUN_IOP(12) RUN_IOP(13) RUN_IOP(14) RUN_IOP(15)
  }
  const int N = 256;
  const char* vn[] = {"fwd scalar", "fwd packed", "bwd scalar", "bwd packed", "fwd exp path"};
  printf("%-14s %10s %16s %22s\n", "pair loop", "ms", "Gpair-evals/s", "cyc/64pairs/SIMD@2.4GHz");
#define RUN_PAIR(V) { double ms = time_ms([&] { hipLaunchKernelGGL(k_pair<V>, dim3(blocks), dim3(threads), 2 * N * sizeof(float), 0, out, 1.5f, N); }); \
    double pe = (double)blocks * threads * 16.0 * N; printf("%-14s %10.4f %16.2f %22.2f\n", vn[V], ms, pe / ms / 1e6, 1024.0 * 2.4e9 * ms * 1e-3 / (pe / 64.0)); }
  RUN_PAIR(0) RUN_PAIR(1) RUN_PAIR(2) RUN_PAIR(3) RUN_PAIR(4)
  {
    const int NREC = 96, L = 200;                       // 24 trips per lane pair and repetition
    const size_t lds_b = (size_t)L * 32 * 4 + (NREC + 16) * 8;
    printf("%-34s %10s %22s\n", "lambdarank trip (2 columns/lane)", "ms", "cyc/trip/SIMD@2.4GHz");
#define RUN_TRIP(V, D, WPS) { const int wg = 256 * WPS; CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trip<V, D>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
      double ms = time_ms([&] { hipLaunchKernelGGL((k_trip<V, D>), dim3(wg), dim3(256), lds_b, 0, out, 0.7f, NREC, L); }); \
      double tr = (double)wg * 4 * 64.0 * (NREC / 4); printf("%s depth %d, %d waves/SIMD            %10.4f %22.2f\n", V ? "lo" : "hi", D, WPS, ms, 1024.0 * 2.4e9 * ms * 1e-3 / tr); }
    RUN_TRIP(0, 1, 1) RUN_TRIP(0, 1, 2) RUN_TRIP(0, 1, 4) RUN_TRIP(0, 3, 1) RUN_TRIP(0, 3, 2) RUN_TRIP(0, 3, 4)
    RUN_TRIP(1, 1, 1) RUN_TRIP(1, 1, 2) RUN_TRIP(1, 1, 4) RUN_TRIP(1, 3, 1) RUN_TRIP(1, 3, 2) RUN_TRIP(1, 3, 4)
    RUN_TRIP(0, 3, 6) RUN_TRIP(1, 3, 6)
    const size_t lds_b2 = (size_t)L * 32 * 4 + (NREC + 32) * 8;
    printf("%-34s %10s %22s\n", "two rows per lane, 4 lanes per row pair (round 6)", "ms", "cyc/trip/SIMD@2.4GHz");
#define RUN_TRIP2(V, WPS) { const int wg = 256 * WPS; CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_trip2<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024)); \
      double ms = time_ms([&] { hipLaunchKernelGGL((k_trip2<V>), dim3(wg), dim3(256), lds_b2, 0, out, 0.7f, NREC, L); }); \
      double tr = (double)wg * 4 * 64.0 * (NREC / 4); printf("%s 2-row, %d waves/SIMD                 %10.4f %22.2f\n", V ? "lo" : "hi", WPS, ms, 1024.0 * 2.4e9 * ms * 1e-3 / tr); }
    RUN_TRIP2(0, 1) RUN_TRIP2(0, 2) RUN_TRIP2(0, 4) RUN_TRIP2(0, 6) RUN_TRIP2(1, 1) RUN_TRIP2(1, 2) RUN_TRIP2(1, 4) RUN_TRIP2(1, 6)
  }
  return 0;
}
