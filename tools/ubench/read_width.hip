// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE per access width on gfx950: streaming read + write of N bytes with
// 4 / 8 / 16 bytes per lane (the K1 backward reads 8 B per lane; MI355X_MICROARCH.md calibrates 16 B per lane only).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <typename T> __global__ void copy_k(const T* __restrict__ in, T* __restrict__ out, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
  const size_t bytes = 512ull << 20;                       // 512 MiB each way: past the 256 MiB Infinity Cache
  void *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMemset(a, 1, bytes));
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(copy_k<float>, dim3(4096), dim3(256), 0, 0, (const float*)a, (float*)b, bytes / 4);
    hipLaunchKernelGGL(copy_k<float2>, dim3(4096), dim3(256), 0, 0, (const float2*)a, (float2*)b, bytes / 8);
    hipLaunchKernelGGL(copy_k<float4>, dim3(4096), dim3(256), 0, 0, (const float4*)a, (float4*)b, bytes / 16);
  }
  CK(hipDeviceSynchronize());
  printf("copied %zu bytes per kernel (read %zu, written %zu)\n", bytes, bytes, bytes);
  return 0;
}
