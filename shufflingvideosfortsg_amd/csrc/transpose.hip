// Batched fp32 matrix transpose: dst[b][c][r] = src[b][r][c]; src rows ld floats apart (a column slice of a wider matrix is read in place).
// Where the path needs it: the input gradients dX = dY W run on the NT GEMM (tsg_gemm_f32s) with the weight passed transposed, and the
// LSTM backward reads W_hh transposed (csrc/lstm.hip); the reference has no counterpart (autograd / cuDNN transpose internally:
// networks/RNN.py:31,42, networks/attention.py:104-106).  torch's `.t().contiguous()` is a generic strided copy: 25 us for
// [4096 x 1024] (1.3 TB/s) and 11 launches per train step; this is a 64 x 64 tile through LDS with 16-byte accesses on both sides.
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kTT = 64;                       // tile edge
// 256 threads: thread (ty = tid >> 4, tx = tid & 15) reads float4 tx of rows ty, ty + 16, .. and writes float4 tx of the transposed rows
__global__ __launch_bounds__(256) void transpose_f32_kernel(const float* __restrict__ src, long ld, float* __restrict__ dst, int rows, int cols,
                                                            int tiles_c) {
  __shared__ float tile[kTT][kTT + 1];
  const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
  const int tr = blockIdx.x / tiles_c, tc = blockIdx.x % tiles_c;
  const float* s = src + blockIdx.y * (size_t)rows * ld;
  float* d = dst + blockIdx.y * (size_t)rows * cols;
  const int r0 = tr * kTT, c0 = tc * kTT;
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = r0 + ty + 16 * p, c = c0 + 4 * tx;
    if (r < rows && c < cols) {                                  // cols % 4 == 0: a float4 is inside or outside as a whole
      const float4 v = *reinterpret_cast<const float4*>(s + (size_t)r * ld + c);
      tile[ty + 16 * p][4 * tx + 0] = v.x; tile[ty + 16 * p][4 * tx + 1] = v.y;
      tile[ty + 16 * p][4 * tx + 2] = v.z; tile[ty + 16 * p][4 * tx + 3] = v.w;
    }
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int c = c0 + ty + 16 * p, r = r0 + 4 * tx;              // output row c, output columns r .. r + 3
    if (c < cols && r < rows) {
      const float4 v = make_float4(tile[4 * tx + 0][ty + 16 * p], tile[4 * tx + 1][ty + 16 * p], tile[4 * tx + 2][ty + 16 * p],
                                   tile[4 * tx + 3][ty + 16 * p]);
      *reinterpret_cast<float4*>(d + (size_t)c * rows + r) = v;
    }
  }
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_transpose_f32(const void* src, long long ld, void* dst, int batch, int rows, int cols, void* stream) {
  const char* fn = "tsg_transpose_f32";
  if (!src || !dst) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(src) || !aligned16(dst)) return set_error(TSG_E_ALIGN, "%s: operands must be 16-byte aligned", fn);
  if (batch <= 0 || rows <= 0 || cols <= 0 || rows % 4 || cols % 4 || batch > 65535 || ld < cols || ld % 4)
    return set_error(TSG_E_SHAPE, "%s: batch=%d rows=%d cols=%d ld=%lld (rows, cols and ld must be multiples of 4, ld >= cols)", fn, batch, rows, cols, ld);
  const int tiles_r = (rows + kTT - 1) / kTT, tiles_c = (cols + kTT - 1) / kTT;
  hipLaunchKernelGGL(transpose_f32_kernel, dim3(tiles_r * tiles_c, batch), dim3(256), 0, static_cast<hipStream_t>(stream),
                     (const float*)src, (long)ld, (float*)dst, rows, cols, tiles_c);
  return check_launch(fn);
}
