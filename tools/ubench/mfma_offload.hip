// Can the K1 score loop push its two FMAs onto the matrix pipe?  v_mfma_f32_4x4x1_16b_f32 as a per-lane-block outer
// product: X[v] = Ea[row v] * Es[word j] + 1 for 16 h-blocks at once; r = rcp(X[v]) on the VALU; the accumulation
// acc[row][word] += w*r again as a 4x4x1 MFMA with a one-hot*w B operand.  Part A checks the lane layout, part B the rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void layout_kernel(const float* a, const float* b, const float* c, float* d) {
  const int l = threadIdx.x;
  f4 cc = {c[l * 4 + 0], c[l * 4 + 1], c[l * 4 + 2], c[l * 4 + 3]};
  f4 r = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], cc, 0, 0, 0);
  for (int v = 0; v < 4; ++v) d[l * 4 + v] = r[v];
}

template <int MODE>
__global__ __launch_bounds__(512) void rate_kernel(float* out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  float ea[4], es[5][4], w[4];
  for (int i = 0; i < 4; ++i) { ea[i] = seed + lane * 1e-3f + i; w[i] = 0.25f + i; }
  for (int g = 0; g < 5; ++g) for (int i = 0; i < 4; ++i) es[g][i] = seed * 0.5f + g + 0.1f * i;
  const f4 ones = {1.f, 1.f, 1.f, 1.f};
  float oh[4];
  for (int v = 0; v < 4; ++v) oh[v] = (lane & 3) == v ? 1.f : 0.f;
  f4 acc[5]; float accv[5]; float accs[5][4];
  for (int g = 0; g < 5; ++g) { acc[g] = (f4){0.f, 0.f, 0.f, 0.f}; accv[g] = 0.f; for (int i = 0; i < 4; ++i) accs[g][i] = 0.f; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {                 // 4 h-steps (one float4 of Ea / Es / w)
      float bw[4];
      if (MODE >= 2) {
#pragma unroll
        for (int v = 0; v < 4; ++v) bw[v] = oh[v] * w[s];
      }
#pragma unroll
      for (int g = 0; g < 5; ++g) {
        if (MODE == 0) {                          // baseline: 4 rows x (fma, rcp, fma) per lane = 4 elem-instr
#pragma unroll
          for (int v = 0; v < 4; ++v)
            accs[g][v] = fmaf(w[s], __builtin_amdgcn_rcpf(fmaf(ea[s] + v, es[g][s], 1.f)), accs[g][v]);
        } else {
          const f4 x = __builtin_amdgcn_mfma_f32_4x4x1f32(ea[s], es[g][s], ones, 0, 0, 0);
          float r[4];
#pragma unroll
          for (int v = 0; v < 4; ++v) r[v] = __builtin_amdgcn_rcpf(x[v]);
          if (MODE == 1) {
#pragma unroll
            for (int v = 0; v < 4; ++v) accs[g][v] = fmaf(w[s], r[v], accs[g][v]);
          } else if (MODE == 2) {
            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r[0], bw[0], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r[1], bw[1], acc[g], 0, 0, 0);
            acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r[2], bw[2], acc[g], 0, 0, 0);
            accv[g] = fmaf(w[s], r[3], accv[g]);
          } else {
#pragma unroll
            for (int v = 0; v < 4; ++v) acc[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(r[v], bw[v], acc[g], 0, 0, 0);
          }
        }
      }
    }
    for (int i = 0; i < 4; ++i) ea[i] += 1e-6f;
  }
  float s = 0;
  for (int g = 0; g < 5; ++g) { s += accv[g]; for (int i = 0; i < 4; ++i) s += acc[g][i] + accs[g][i]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
int run(const char* name, int threads, int blocks_per_cu) {
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 8 * 1024));
  const int iters = 1000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int grid = 256 * blocks_per_cu;
  rate_kernel<MODE><<<grid, threads>>>(out, 10, 1.f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  rate_kernel<MODE><<<grid, threads>>>(out, iters, 1.f);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double einstr = (double)grid * (threads / 64) * iters * 4 * 5 * 4;   // wave-level (64-element) tanh evaluations
  const double cyc = ms * 1e-3 * 2.4e9 / (einstr / 1024);
  printf("%-34s waves/SIMD=%d  %.3f ms  %.2f cyc@2.4GHz per 64-element evaluation per SIMD\n", name, threads / 64 * blocks_per_cu / 4, ms, cyc);
  CK(hipFree(out));
  return 0;
}

int main() {
  // ---- part A: layout ----
  std::vector<float> a(64), b(64), c(256), d(256);
  for (int l = 0; l < 64; ++l) { a[l] = 1.f + l; b[l] = 100.f + 3 * l; }
  for (int i = 0; i < 256; ++i) c[i] = 0.001f * i;
  float *da, *db, *dc, *dd;
  CK(hipMalloc(&da, 256)); CK(hipMalloc(&db, 256)); CK(hipMalloc(&dc, 1024)); CK(hipMalloc(&dd, 1024));
  CK(hipMemcpy(da, a.data(), 256, hipMemcpyHostToDevice)); CK(hipMemcpy(db, b.data(), 256, hipMemcpyHostToDevice));
  CK(hipMemcpy(dc, c.data(), 1024, hipMemcpyHostToDevice));
  layout_kernel<<<1, 64>>>(da, db, dc, dd);
  CK(hipMemcpy(d.data(), dd, 1024, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v) {
      const int blk = l / 4;
      const float want = a[blk * 4 + v] * b[l] + c[l * 4 + v];      // D[v] on lane (blk, j) = A[(blk, v)] * B[(blk, j)] + C
      if (fabsf(want - d[l * 4 + v]) > 1e-3f * fabsf(want)) { if (bad < 5) printf("layout mismatch lane %d v %d: got %f want %f\n", l, v, d[l * 4 + v], want); ++bad; }
    }
  printf("4x4x1 layout  D[v][lane(b,j)] = A[lane(b,v)] * B[lane(b,j)] + C[v][lane]: %s\n", bad ? "MISMATCH" : "confirmed");
  // ---- part B: rate, 2 waves per SIMD (512-thread workgroup, 1 per CU) and 1 wave per SIMD ----
  for (int thr : {512, 256}) {
    run<0>("VALU fma+rcp+fma", thr, 1);
    run<1>("MFMA x, VALU rcp+fma", thr, 1);
    run<2>("MFMA x, rcp, 3 MFMA acc + 1 fma", thr, 1);
    run<3>("MFMA x, rcp, 4 MFMA acc", thr, 1);
  }
  return 0;
}
