// Round-trip latency of the persistent LSTM's hand-off primitive (developer probe): two workgroups bounce a sequence number
// through global memory -- plain store + agent-scope (sc1) polling load on ONE XCD (the data stays in that XCD's L2), and
// write-through (sc1) store + sc1 load across two XCDs.  Prints microseconds per ONE-WAY hop.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ unsigned ld_sc1(const unsigned* p) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void st_plain(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_sc1(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory"); }
// blocks `a` and `b` play; everyone else exits.  flags[0]: a -> b, flags[32]: b -> a (separate 128-byte lines)
__global__ void pingpong(unsigned* flags, int a, int b, int n, int wt, unsigned* xcc) {
  const int me = blockIdx.x;
  if (me != a && me != b) return;
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  if (threadIdx.x == 0) xcc[me == a ? 0 : 1] = x & 15;
  if (threadIdx.x != 0) return;
  unsigned* mine = flags + (me == a ? 0 : 32);
  const unsigned* theirs = flags + (me == a ? 32 : 0);
  for (int i = 1; i <= n; ++i) {
    if (me == a) {
      if (wt) st_sc1(mine, i); else st_plain(mine, i);
      int spins = 0;
      while (ld_sc1(theirs) != (unsigned)i && ++spins < (1 << 22)) {}
    } else {
      int spins = 0;
      while (ld_sc1(theirs) != (unsigned)i && ++spins < (1 << 22)) {}
      if (wt) st_sc1(mine, i); else st_plain(mine, i);
    }
  }
}
int main() {
  unsigned *flags, *xcc;
  hipMalloc(&flags, 1024); hipMalloc(&xcc, 8);
  const int n = 20000;
  struct { const char* name; int a, b, wt; } cases[] = {{"same XCD, plain store + sc1 load", 0, 8, 0}, {"same XCD, write-through store + sc1 load", 0, 8, 1},
                                                        {"two XCDs, write-through store + sc1 load", 0, 1, 1}};
  for (auto& c : cases) {
    hipMemset(flags, 0, 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(pingpong, dim3(16), dim3(64), 0, 0, flags, c.a, c.b, n, c.wt, xcc);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    unsigned h[2], f[64]; hipMemcpy(h, xcc, 8, hipMemcpyDeviceToHost); hipMemcpy(f, flags, 256, hipMemcpyDeviceToHost);
    printf("%-44s XCDs %u/%u  %.3f us per one-way hop (%s)\n", c.name, h[0], h[1], ms * 1e3 / (2.0 * n), f[32] == (unsigned)n ? "completed" : "TIMED OUT");
  }
  return 0;
}
