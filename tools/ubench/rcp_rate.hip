// Calibration: issue cost of the K1 inner triple (fma, rcp, fma) on gfx950, per waves/SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  float ea[8], acc[8];
  for (int i = 0; i < 8; ++i) { ea[i] = seed + threadIdx.x * 1e-3f + i; acc[i] = 0.f; }
  float es = seed * 0.5f, w = 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) {            // fma, rcp, fma
          acc[i] = fmaf(w, __builtin_amdgcn_rcpf(fmaf(ea[i], es, 1.f)), acc[i]);
        } else if (MODE == 1) {     // fma, fma only (no transcendental)
          acc[i] = fmaf(w, fmaf(ea[i], es, 1.f), acc[i]);
        } else if (MODE == 2) {     // rcp only
          acc[i] = __builtin_amdgcn_rcpf(acc[i] + ea[i]);
        } else if (MODE == 3) {     // exp2 only
          acc[i] = __builtin_amdgcn_exp2f(acc[i] * 0.001f + ea[i]);
        }
      }
      es += 1e-6f;
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int threads, int blocks_per_cu, int per_elem_ops) {
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 8 * 1024));
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * blocks_per_cu;
  k<MODE><<<grid, threads>>>(out, 10, 1.f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  k<MODE><<<grid, threads>>>(out, iters, 1.f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double elems = (double)grid * threads * iters * 32;      // inner-body executions
  const double waves_per_simd = (double)threads / 64 * blocks_per_cu / 4;
  // cycles per wave-level body (64 lanes) per SIMD at 2.4 GHz
  const double cyc = ms * 1e-3 * 2.4e9 / (elems / 64 / 1024);
  printf("%-14s waves/SIMD=%.0f  %.3f ms  %.1f Gelem/s  %.1f cyc@2.4GHz per wave-body\n", name, waves_per_simd, ms, elems / ms / 1e6, cyc);
  CK(hipFree(out));
  return 0;
}

int main() {
  for (int bpc : {1, 2, 4, 8}) {
    run<0>("fma+rcp+fma", 256, bpc, 3);
    run<1>("fma+fma", 256, bpc, 2);
    run<2>("add+rcp", 256, bpc, 2);
    run<3>("fma+exp2", 256, bpc, 2);
  }
  return 0;
}
