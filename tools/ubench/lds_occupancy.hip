// How many 256-thread workgroups with N bytes of dynamic LDS does the runtime place on one CU?  (hipOccupancyMaxActiveBlocksPerMultiprocessor
// + a measurement: 2 x 256 workgroups that each spin ~50 us; co-resident pairs finish in one round.)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void spin(unsigned long long cycles, unsigned* out) {
  extern __shared__ unsigned lds[];
  lds[threadIdx.x] = threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  while (__builtin_amdgcn_s_memtime() - t0 < cycles) {}
  if (threadIdx.x == 0) out[blockIdx.x] = lds[0];
}
int main() {
  unsigned* out; hipMalloc(&out, 4096 * 4);
  for (size_t lds : {65536ul, 78848ul, 80384ul, 81920ul, 83456ul}) {
    hipFuncSetAttribute((const void*)spin, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    int n = -1;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, spin, 256, lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int rep = 0; rep < 2; ++rep) {
      const int grid = rep == 0 ? 256 : 512;
      hipLaunchKernelGGL(spin, dim3(grid), dim3(256), lds, 0, 100000ull, out);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(spin, dim3(grid), dim3(256), lds, 0, 100000ull, out);
      hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[rep], e0, e1);
    }
    printf("LDS %6zu B: occupancy query %d blocks/CU; 256 WGs %.1f us, 512 WGs %.1f us -> %s\n", lds, n, ms[0] * 1e3, ms[1] * 1e3,
           ms[1] < 1.5f * ms[0] ? "2 per CU" : "1 per CU");
  }
  return 0;
}
