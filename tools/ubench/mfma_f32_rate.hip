// Peak issue rate of the fp32 MFMA shapes on gfx950 (no memory traffic): 4 independent accumulator chains per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

template <int SHAPE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
  if (SHAPE == 0) {
    f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c3, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  } else {
    f4v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c3, 0, 0, 0);
      }
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
  }
}

template <int SHAPE>
int run(const char* name, int wgs_per_cu, double flop_per_mfma) {
  float* out; CK(hipMalloc(&out, 4 * 256 * 256 * 8));
  const int iters = 2000, grid = 256 * wgs_per_cu;
  k<SHAPE><<<grid, 256>>>(out, 10); CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0)); k<SHAPE><<<grid, 256>>>(out, iters); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double mfmas = (double)grid * 4 * iters * 32;
  printf("%-22s %d wave(s)/SIMD: %.1f TFLOP/s\n", name, wgs_per_cu, mfmas * flop_per_mfma / ms / 1e9);
  CK(hipFree(out)); return 0;
}
int main() {
  for (int w : {1, 2, 4}) { run<0>("v_mfma_f32_32x32x2", w, 4096.0); run<1>("v_mfma_f32_16x16x4", w, 2048.0); }
  return 0;
}
