// Which XCD does workgroup i of a 1-D grid land on?  (HW_REG_XCC_ID; developer probe for the persistent LSTM's exchange groups)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(512) void k(unsigned* o) {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) o[blockIdx.x] = x;
}
int main() {
  const int n = 256;
  unsigned* d; hipMalloc(&d, n * 4);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k, dim3(n), dim3(512), 0, 0, d);
    unsigned h[n]; hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
    int rr = 0;
    for (int i = 0; i < n; ++i) rr += ((h[i] & 0xf) == (unsigned)(i % 8));
    printf("raw[0..15]:"); for (int i = 0; i < 16; ++i) printf(" %x", h[i]); printf("\nblocks with (xcc_id & 15) == blockIdx %% 8: %d of %d\n", rr, n);
  }
  return 0;
}
