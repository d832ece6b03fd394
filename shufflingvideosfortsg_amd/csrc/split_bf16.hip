// Operand preparation for the split-precision GEMM mode ("f32s"): an fp32 matrix is written as three bf16
// planes so that ONE bf16 MFMA GEMM with fp32 accumulation over the 3x longer contraction evaluates
//     A·B ≈ A_hi·B_hi + A_hi·B_lo + A_lo·B_hi         (hi = rne_bf16(x), lo = rne_bf16(x - hi)),
// which drops only the lo·lo term (2^-18 relative) — measured on MI355X at the LSTM GEMM shapes: 4.6e-6 max
// error relative to max|C|, the same as the fp32 rocBLAS GEMM (1.4e-6 .. 6.4e-6), at 2.5-3x its speed.
// The left operand uses the plane pattern (hi, hi, lo), the right operand (hi, lo, hi).
// HBM-bound elementwise pass: 4 B read + 6 B written per element.
#include "tsg_common.h"
#include <cstdlib>

namespace tsg {
namespace {

__device__ __forceinline__ unsigned bf16_rne(float x) {          // finite inputs; NaN stays NaN
  unsigned u = __float_as_uint(x);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}

// row r takes source row r - shift when that row exists and, with period > 0 (rows = consecutive sequences of `period`
// steps, the batch-major layout), lies in the same sequence
__device__ __forceinline__ bool shift_ok(long r, long shift, long rows, long period) {
  if (period > 0) { const long t = r % period - shift; return t >= 0 && t < period; }
  return r - shift >= 0 && r - shift < rows;
}

__global__ __launch_bounds__(256) void split_bf16_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                         long rows, int cols4, long ld_out, long plane, int right,
                                                         long ld_in, long row_shift, long period) {
  const long total = rows * cols4;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const long r = i / cols4;
    const int c = static_cast<int>(i - r * cols4) * 4;
    const long rs = r - row_shift;                                 // source row (zeros outside the matrix / the row's period)
    const float4 v = shift_ok(r, row_shift, rows, period) ? *reinterpret_cast<const float4*>(x + rs * ld_in + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float e[4] = {v.x, v.y, v.z, v.w};
    unsigned hi[4], lo[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      hi[k] = bf16_rne(e[k]);
      const float rem = e[k] - __uint_as_float(hi[k] << 16);
      lo[k] = (hi[k] & 0x7f80u) == 0x7f80u ? 0u : bf16_rne(rem);          // inf/NaN: keep it in the hi plane only
    }
    const uint2 H = make_uint2(hi[0] | (hi[1] << 16), hi[2] | (hi[3] << 16));
    const uint2 L = make_uint2(lo[0] | (lo[1] << 16), lo[2] | (lo[3] << 16));
    unsigned short* o = out + r * ld_out + c;
    *reinterpret_cast<uint2*>(o) = H;
    *reinterpret_cast<uint2*>(o + plane) = right ? L : H;
    *reinterpret_cast<uint2*>(o + 2 * plane) = right ? H : L;
  }
}


// Transposing variant: out[c][p*plane + r] (the contraction index r contiguous), for GEMM operands contracted over the
// ROWS of x -- the library's bf16 GEMM runs the weight-gradient shape [8h x 3TB] x [3TB x (I+2h)] 18 % faster when both
// operands are K-contiguous (tools/probe_dw_layout.py).  TR x TC tiles through LDS: float4 loads along the columns,
// (row, row+1) pairs packed per column on the way in, 32-byte segments along the rows on the way out.
template <int TR, int TC>
__global__ __launch_bounds__(256) void split_bf16_t_kernel(const float* __restrict__ x, unsigned short* __restrict__ out,
                                                           long rows, int cols, long ld_out, long plane, int right,
                                                           long ld_in, long row_shift, long period, int swap, long dup) {
  constexpr int LS = TR + 4;                 // LDS row stride in bf16 elements: (TR+4)/2 dwords = 2 mod 32 -> conflict-free b64
  constexpr int TPR = TC / 4;                // threads along a tile row (float4 each)
  constexpr int RPP = 4 * (256 / TPR);       // rows per load pass (4 consecutive rows per thread)
  static_assert(TR % RPP == 0 && (TC * TR / 16) % 256 == 0, "tile shape");
  __shared__ __align__(8) unsigned short Th[TC * LS], Tl[TC * LS];
  const int tid = threadIdx.x, tx = tid % TPR, ty = tid / TPR;
  // blockIdx.x runs along the COLUMNS: workgroups in flight together then read whole contiguous rows; with x along the rows
  // they all read the same 256-byte column window at a power-of-two row stride, i.e. the same memory channels
  // (measured: the read side alone runs at 5.2 TB/s, the write side alone -- 128-byte segments in 3 x 64 output rows per
  // workgroup -- at 3.7 TB/s, together 3.6; longer segments (128 x 64, 256 x 32 tiles) and grouping row tiles of one
  // column tile on neighbouring workgroups changed nothing, so 64 x 64 with the column-major grid stays)
  const long r0 = (long)TR * (swap ? blockIdx.x : blockIdx.y);
  const int c0 = TC * (swap ? blockIdx.y : blockIdx.x);
#pragma unroll
  for (int pass = 0; pass < TR / RPP; ++pass) {
    unsigned hi[4][4], lo[4][4];                                 // [row i of the thread's 4][column k of its 4]
    const int rl = pass * RPP + 4 * ty;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const long r = r0 + rl + i;
      const int c = c0 + 4 * tx;
      const float4 v = (r < rows && c < cols && shift_ok(r, row_shift, rows, period))
                           ? *reinterpret_cast<const float4*>(x + (r - row_shift) * ld_in + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[i][k] = bf16_rne(e[k]);
        const float rem = e[k] - __uint_as_float(hi[i][k] << 16);
        lo[i][k] = (hi[i][k] & 0x7f80u) == 0x7f80u ? 0u : bf16_rne(rem);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      *reinterpret_cast<uint2*>(Th + (4 * tx + k) * LS + rl) = make_uint2(hi[0][k] | (hi[1][k] << 16), hi[2][k] | (hi[3][k] << 16));
      *reinterpret_cast<uint2*>(Tl + (4 * tx + k) * LS + rl) = make_uint2(lo[0][k] | (lo[1][k] << 16), lo[2][k] | (lo[3][k] << 16));
    }
  }
  __syncthreads();
  constexpr int SPR = TR / 16;                                   // 16-element segments per output row
#pragma unroll
  for (int q = 0; q < TC * SPR / 256; ++q) {
    const int sidx = q * 256 + tid, cl = sidx / SPR, seg = sidx % SPR;     // output row (= input column) and its segment
    const int c = c0 + cl;
    const long r = r0 + 16 * seg;
    if (c < cols && r < rows) {                                  // rows % 16 == 0 (checked on the host): whole segments
      uint2 H[4], L[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        H[j] = *reinterpret_cast<const uint2*>(Th + cl * LS + 16 * seg + 4 * j);
        L[j] = *reinterpret_cast<const uint2*>(Tl + cl * LS + 16 * seg + 4 * j);
      }
      unsigned short* o = out + (long)c * ld_out + r;             // 16-byte aligned (host check)
      const uint4 H0 = make_uint4(H[0].x, H[0].y, H[1].x, H[1].y), H1 = make_uint4(H[2].x, H[2].y, H[3].x, H[3].y);
      const uint4 L0 = make_uint4(L[0].x, L[0].y, L[1].x, L[1].y), L1 = make_uint4(L[2].x, L[2].y, L[3].x, L[3].y);
      *reinterpret_cast<uint4*>(o) = H0;
      *reinterpret_cast<uint4*>(o + 8) = H1;
      *reinterpret_cast<uint4*>(o + plane) = right ? L0 : H0;
      *reinterpret_cast<uint4*>(o + plane + 8) = right ? L1 : H1;
      *reinterpret_cast<uint4*>(o + 2 * plane) = right ? H0 : L0;
      *reinterpret_cast<uint4*>(o + 2 * plane + 8) = right ? H1 : L1;
      if (dup) {                                                 // second copy of the same planes, `dup` elements further on
        unsigned short* o2 = o + dup;
        *reinterpret_cast<uint4*>(o2) = H0;
        *reinterpret_cast<uint4*>(o2 + 8) = H1;
        *reinterpret_cast<uint4*>(o2 + plane) = right ? L0 : H0;
        *reinterpret_cast<uint4*>(o2 + plane + 8) = right ? L1 : H1;
        *reinterpret_cast<uint4*>(o2 + 2 * plane) = right ? H0 : L0;
        *reinterpret_cast<uint4*>(o2 + 2 * plane + 8) = right ? H1 : L1;
      }
    }
  }
}

}  // namespace
}  // namespace tsg

extern "C" int tsg_split_bf16x3_shift(const void* x, long long ld_in, long long row_shift, long long period, void* out, long long rows,
                                      long long cols, long long ld_out, long long plane_stride, int right_operand, void* stream) {
  using namespace tsg;
  const char* fn = "tsg_split_bf16x3";
  if (!x || !out) return set_error(TSG_E_NULL, "%s: null pointer", fn);
  if (rows < 0 || cols < 0 || (cols & 3) || (ld_out & 3) || (plane_stride & 3) || (ld_in & 3) || ld_in < cols || period < 0)
    return set_error(TSG_E_SHAPE, "%s: rows=%lld cols=%lld ld_in=%lld ld_out=%lld plane=%lld (cols, ld_in, ld_out, plane must be multiples of 4, ld_in >= cols)",
                     fn, rows, cols, ld_in, ld_out, plane_stride);
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 7))
    return set_error(TSG_E_ALIGN, "%s: x must be 16-byte and out 8-byte aligned", fn);
  if (rows == 0 || cols == 0) return 0;
  const long total = rows * (cols / 4);
  const int grid = static_cast<int>(total / 256 + 1 < 256L * 16 ? total / 256 + 1 : 256L * 16);
  hipLaunchKernelGGL(split_bf16_kernel, dim3(grid), dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(x), static_cast<unsigned short*>(out), rows, static_cast<int>(cols / 4),
                     ld_out, plane_stride, right_operand, static_cast<long>(ld_in), static_cast<long>(row_shift), static_cast<long>(period));
  return check_launch(fn);
}

extern "C" int tsg_split_bf16x3(const void* x, void* out, long long rows, long long cols, long long ld_out, long long plane_stride,
                                int right_operand, void* stream) {
  return tsg_split_bf16x3_shift(x, cols, 0, 0, out, rows, cols, ld_out, plane_stride, right_operand, stream);
}

extern "C" int tsg_split_bf16x3_t(const void* x, long long ld_in, long long row_shift, long long period, void* out, long long rows,
                                  long long cols, long long ld_out, long long plane_stride, int right_operand, long long dup_offset,
                                  void* stream) {
  using namespace tsg;
  const char* fn = "tsg_split_bf16x3_t";
  if (!x || !out) return set_error(TSG_E_NULL, "%s: null pointer", fn);
  if (rows < 0 || cols < 0 || (cols & 3) || (rows & 15) || (ld_out & 7) || (plane_stride & 7) || (ld_in & 3) || ld_in < cols)
    return set_error(TSG_E_SHAPE, "%s: rows=%lld cols=%lld ld_in=%lld ld_out=%lld plane=%lld (rows %% 16, cols / ld_in %% 4, ld_out / plane %% 8, ld_in >= cols)",
                     fn, rows, cols, ld_in, ld_out, plane_stride);
  if ((reinterpret_cast<uintptr_t>(x) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) || (dup_offset & 7))
    return set_error(TSG_E_ALIGN, "%s: x and out must be 16-byte aligned, dup_offset a multiple of 8", fn);
  if (rows == 0 || cols == 0) return 0;
  static int tile = -1;                                            // TSG_SPLIT_T_TILE: 0 = 64x64, 1 = 64 rows x 128 cols, 2 = 128 x 64
  if (tile < 0) { const char* e = getenv("TSG_SPLIT_T_TILE"); tile = e ? atoi(e) : 0; }
  const int TR = tile == 3 ? 256 : tile == 2 ? 128 : 64, TC = tile == 3 ? 32 : tile == 1 ? 128 : 64;
  static int swap = -1;
  if (swap < 0) { const char* e = getenv("TSG_SPLIT_T_ROWMAJOR_GRID"); swap = e ? atoi(e) : 0; }
  const unsigned gr = static_cast<unsigned>((rows + TR - 1) / TR), gc = static_cast<unsigned>((cols + TC - 1) / TC);
  const dim3 grid(swap ? gr : gc, swap ? gc : gr);
  if (grid.y > 65535u) return set_error(TSG_E_SHAPE, "%s: rows=%lld cols=%lld too large", fn, rows, cols);
  auto kern = tile == 3 ? split_bf16_t_kernel<256, 32> : tile == 2 ? split_bf16_t_kernel<128, 64> : tile == 1 ? split_bf16_t_kernel<64, 128> : split_bf16_t_kernel<64, 64>;
  hipLaunchKernelGGL(kern, grid, dim3(256), 0, static_cast<hipStream_t>(stream),
                     static_cast<const float*>(x), static_cast<unsigned short*>(out), static_cast<long>(rows), static_cast<int>(cols),
                     static_cast<long>(ld_out), static_cast<long>(plane_stride), right_operand, static_cast<long>(ld_in),
                     static_cast<long>(row_shift), static_cast<long>(period), swap, static_cast<long>(dup_offset));
  return check_launch(fn);
}
