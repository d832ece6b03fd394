// Calibration: packed-f32 issue cost and the shared-reciprocal pair step on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float* out, int iters, float seed) {
  v2f ea[8], acc[8];
  for (int i = 0; i < 8; ++i) { ea[i] = (v2f){seed + threadIdx.x * 1e-3f + i, seed + i * 0.5f}; acc[i] = (v2f){0.f, 0.f}; }
  float es = seed * 0.5f, w = 0.25f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if (MODE == 0) {            // pair step, shared rcp
          const v2f x = __builtin_elementwise_fma(ea[i], (v2f){es, es}, (v2f){1.f, 1.f});
          const float r = __builtin_amdgcn_rcpf(x.x * x.y);
          const v2f inv = (v2f){r, r} * (v2f){x.y, x.x};
          acc[i] = __builtin_elementwise_fma((v2f){w, w}, inv, acc[i]);
        } else if (MODE == 1) {     // pair step, two rcp, packed fma
          const v2f x = __builtin_elementwise_fma(ea[i], (v2f){es, es}, (v2f){1.f, 1.f});
          const v2f inv = (v2f){__builtin_amdgcn_rcpf(x.x), __builtin_amdgcn_rcpf(x.y)};
          acc[i] = __builtin_elementwise_fma((v2f){w, w}, inv, acc[i]);
        } else if (MODE == 2) {     // 2 packed fma only
          const v2f x = __builtin_elementwise_fma(ea[i], (v2f){es, es}, (v2f){1.f, 1.f});
          acc[i] = __builtin_elementwise_fma((v2f){w, w}, x, acc[i]);
        } else if (MODE == 3) {     // 4 scalar fma (same flops as MODE 2)
          const float x0 = fmaf(ea[i].x, es, 1.f), x1 = fmaf(ea[i].y, es, 1.f);
          float a0 = acc[i].x, a1 = acc[i].y;
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w), "v"(x0));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w), "v"(x1));
          acc[i] = (v2f){a0, a1};
        } else if (MODE == 4) {     // scalar pair step, shared rcp (no packed ops)
          const float x0 = fmaf(ea[i].x, es, 1.f), x1 = fmaf(ea[i].y, es, 1.f);
          const float r = __builtin_amdgcn_rcpf(x0 * x1);
          float a0 = acc[i].x, a1 = acc[i].y;
          const float i0 = r * x1, i1 = r * x0;
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w), "v"(i0));
          asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w), "v"(i1));
          acc[i] = (v2f){a0, a1};
        }
      }
      es += 1e-6f;
    }
  }
  float s = 0; for (int i = 0; i < 8; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int threads, int blocks_per_cu) {
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
  const double pairs = (double)grid * threads * iters * 32;      // pair steps (2 elements each)
  const double cyc = ms * 1e-3 * 2.4e9 / (pairs / 64 / 1024);
  printf("%-26s waves/SIMD=%.0f  %.3f ms  %.1f Gelem/s  %.1f cyc@2.4GHz per wave pair-step\n", name,
         (double)threads / 64 * blocks_per_cu / 4, ms, 2 * pairs / ms / 1e6, cyc);
  CK(hipFree(out));
  return 0;
}

int main() {
  for (int bpc : {2, 4}) {
    run<0>("pk pair, shared rcp", 256, bpc);
    run<1>("pk pair, 2 rcp", 256, bpc);
    run<2>("2 pk_fma", 256, bpc);
    run<3>("4 fma", 256, bpc);
    run<4>("scalar pair, shared rcp", 256, bpc);
  }
  return 0;
}
