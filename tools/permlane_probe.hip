// Developer probe (round 4): what v_permlane32_swap / v_permlane16_swap return on gfx950, and whether the DPP / permlane
// form of wave_tree_sum (csrc/common.h) pairs the same elements as the __shfl_down form.  hipcc --offload-arch=gfx950
// tools/permlane_probe.hip -o tools/permlane_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include "../ranking_amd/csrc/common.h"

__global__ void probe(int* o32a, int* o32b, int* o16a, int* o16b, int* shl8) {
  const unsigned v = threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  o32a[threadIdx.x] = r[0]; o32b[threadIdx.x] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  o16a[threadIdx.x] = q[0]; o16b[threadIdx.x] = q[1];
  shl8[threadIdx.x] = __builtin_amdgcn_update_dpp(-1, (int)v, 0x108, 0xf, 0xf, true);
}

template <int IPL>
__device__ float tree_ref(float (&t)[IPL], int P) {
  for (int hr = IPL >> 1; hr >= 1; hr >>= 1)
    if (64 * hr * 2 <= P) for (int r = 0; r < hr; ++r) t[r] = t[r] + t[r + hr];
  float v = t[0];
  for (int h = 32; h >= 1; h >>= 1) { const float o = __shfl_down(v, h, 64); if (2 * h <= P) v = v + o; }
  return __shfl(v, 0, 64);
}

__global__ void sums(const float* x, int P, float* a, float* b) {
  float t1[4], t2[4];
  for (int r = 0; r < 4; ++r) { const int e = threadIdx.x + 64 * r; t1[r] = t2[r] = e < P ? x[blockIdx.x * 256 + e] : 0.f; }
  const float s1 = tree_ref<4>(t1, P), s2 = tfr::wave_tree_sum<4>(t2, P);
  if (threadIdx.x == 0) { a[blockIdx.x] = s1; b[blockIdx.x] = s2; }
}

int main() {
  int *d; hipMalloc(&d, 5 * 64 * 4);
  probe<<<1, 64>>>(d, d + 64, d + 128, d + 192, d + 256);
  int h[320]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  const char* names[5] = {"permlane32_swap r[0]", "permlane32_swap r[1]", "permlane16_swap r[0]", "permlane16_swap r[1]", "dpp row_shl:8"};
  for (int k = 0; k < 5; ++k) { printf("%-22s", names[k]); for (int i = 0; i < 64; ++i) printf(" %d", h[64 * k + i]); printf("\n"); }
  const int NB = 64;
  float* hx = (float*)malloc(NB * 256 * 4);
  srand(1); for (int i = 0; i < NB * 256; ++i) hx[i] = (float)rand() / RAND_MAX * 3.0f;
  float *dx, *da, *db; hipMalloc(&dx, NB * 256 * 4); hipMalloc(&da, NB * 4); hipMalloc(&db, NB * 4);
  hipMemcpy(dx, hx, NB * 256 * 4, hipMemcpyHostToDevice);
  for (int P = 2; P <= 256; P *= 2) {
    sums<<<NB, 64>>>(dx, P, da, db);
    float ha[NB], hb[NB]; hipMemcpy(ha, da, NB * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, NB * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < NB; ++i) bad += ha[i] != hb[i];
    printf("P = %3d: %d of %d sums differ (e.g. %.9g vs %.9g)\n", P, bad, NB, ha[0], hb[0]);
  }
  return 0;
}
