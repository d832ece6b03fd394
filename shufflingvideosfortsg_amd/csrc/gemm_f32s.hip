// Split-precision projection GEMM with the operands converted ON LOAD ("f32s" arithmetic, no operand planes in memory):
//     Y[M,N] = X[M,K] W[N,K]^T (+ bias[N])                       fp32 in, fp32 out, both operands contraction-contiguous
// i.e. torch.nn.Linear's layout -- the forward of the path's Linears and of nn.LSTM's input projection (reference:
// networks/attention.py:104-106,112-113; components/VideoEncoder.py:59; networks/RNN.py:31,42), and their input gradients
// dX = dY W with the weight passed transposed.  Every operand element x is split into hi = rne_bf16(x), lo = rne_bf16(x - hi) in
// registers and  hi*hi + hi*lo + lo*hi  is accumulated in fp32 on v_mfma_f32_32x32x16_bf16: the arithmetic of
// tsg_split_bf16x3 + a bf16 GEMM over the 3x longer contraction (functional._mm), without the split passes (fp32 read, three bf16
// planes written, read again) in front of it.
//
// Tiling.  Workgroup = 256 x 256 output tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 4 x 2 MFMA tiles (128 accumulator
// registers); persistent workgroups walk the tiles (grid = min(tiles, CUs)).  K advances in chunks of 32 fp32 columns.  A chunk's X
// and W row tiles (256 rows x 128 bytes each) are requested into registers (8 float4 per thread) one chunk ahead, split ONCE by
// the thread that loaded them, and written to LDS as four bf16 planes (X hi, X lo, W hi, W lo; 64 bytes per row; 64 KiB per chunk,
// double-buffered): a fragment is then one ds_read_b128 per plane and costs no conversion at the eight waves that share it.
// (First version: LDS-DMA of the raw fp32 tiles and a split per fragment read -- every X fragment was converted by four waves, every
// W fragment by two: 116 us of the 523 us launch at [16384 x 1024] x [4096 x 1024] was conversion VALU.)
// LDS image: the four 16-byte pieces of a plane row are XOR-swizzled by (row >> 2) & 3, on the write and on the read: a fragment
// read (32 rows x the same piece) touches 16 distinct bank quads per 16 lanes (SQ_LDS_BANK_CONFLICT = 0).
// One LDS-only barrier per chunk; behind it the registers (chunk ks + 1) are converted into the other buffer and chunk ks + 2 is
// requested; the steady-state loop body is one basic block (the tail chunks are instantiated separately), the next X fragment is
// read while the current one is multiplied.
// Measured (tools/gemm_f32s_time.py, one MI355X; "library" = tsg_split_bf16x3 of both operands + hipBLASLt's bf16 GEMM over 3K):
//   [16384 x 1024] x [1024 x 1024]   92-97 us vs 115-117      [16384 x 1024] x [2048 x 1024]  179-186 vs 198-205
//   [16384 x 2048] x [ 512 x 2048]  133-146 vs 171-174        [16384 x 4096] x [1024 x 4096]  380-420 vs 425-450
//   [16384 x 1024] x [4096 x 1024]  450-455 vs 436-454 (tie)  [ 2560 x 1024] x [1024 x 1024]   63-72 vs 41 (40 tiles: the library path stays)
// = 0.9-1.15 PFLOP/s of bf16 matrix work; PMC: the matrix pipe is busy 54 % of the launch (the clock under this load is 1.62 GHz, so
// the sustained peak is 1.7 PFLOP/s), the waves wait on their global requests 12-14 % of it (one chunk = 1.6 us of look-ahead; a
// second register set spills 35 registers, an L2 touch-prefetch by inline asm cost more than it hid), LDS waits 5 %.
// Requirements: M % 256 == 0, N % 256 == 0, K % 32 == 0, 16-byte aligned rows (the host code falls back to the planes + library
// path otherwise).
#include "tsg_common.h"
#include <type_traits>

namespace tsg {
namespace {

typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned g_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGT = 512;                       // threads
constexpr int kTM = 256, kTN = 256, kBK = 32;  // workgroup tile, fp32 columns per chunk
constexpr int kPlane = kTM * kBK / 2;           // dwords of one bf16 plane of an operand tile: 256 rows x 16 dwords (16 KiB)
constexpr size_t kGemmLds = sizeof(unsigned) * 2 * 4 * kPlane;  // [2 buffers][X hi | X lo | W hi | W lo] = 128 KiB

__device__ __forceinline__ unsigned g_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((g_f32x2){a, b}, g_bf16x2));
}
__device__ __forceinline__ void g_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = g_pk(a, b);
  lo = g_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ g_f32x16 g_mfma(g_u32x4 a, g_u32x4 b, g_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, a), __builtin_bit_cast(g_bf16x8, b), c, 0, 0, 0);
}

__global__ __launch_bounds__(kGT) void gemm_nt_f32s_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                           const float* __restrict__ bias, float* __restrict__ Y,
                                                           int M, int N, int K, int tiles_n) {
  extern __shared__ __align__(16) unsigned lds[];                   // [2 buffers][X hi | X lo | W hi | W lo], each [256 rows][16 dwords]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wv >> 2) * 128, wn = (wv & 3) * 64;
  const int jl = lane & 31, kg = lane >> 5;
  const int ntiles = (M / kTM) * tiles_n;
  // Persistent workgroups: workgroup w walks the tiles w, w + grid, ...: the fp32 stores of a finished tile drain under the next
  // tile's chunks (a 256 x 256 tile is 256 KiB of output: 15-20 us of store tail per tile when nothing runs beside it).
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  const int bid = xcd_remap(tile, ntiles, tiles_n);                 // the N tiles of one M tile share an XCD (X rows stay in its L2)
  const int m0 = (bid / tiles_n) * kTM, n0 = (bid % tiles_n) * kTN;

  // staging role: 8 threads per row (8 float4 = one 128-byte row segment), 64 rows per pass, 4 passes per operand
  const int sr = tid >> 3, sq = tid & 7;
  const float* xsrc = X + (size_t)(m0 + sr) * K + 4 * sq;
  const float* wsrc = W + (size_t)(n0 + sr) * K + 4 * sq;
  const size_t pass = (size_t)64 * K;
  float4 rx[4], rw[4];
  auto request = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      rx[p] = *reinterpret_cast<const float4*>(xsrc + p * pass + k0);
      rw[p] = *reinterpret_cast<const float4*>(wsrc + p * pass + k0);
    }
  };
  // split once, here, and write the bf16 planes: row r, 16-byte piece (sq >> 1) ^ ((r >> 2) & 3), 8-byte half sq & 1
  auto write_planes = [&](int buf) {
    unsigned* base = lds + buf * 4 * kPlane;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = sr + 64 * p;
      const int o = r * 16 + (((sq >> 1) ^ ((r >> 2) & 3)) << 2) + ((sq & 1) << 1);
      unsigned h0, l0, h1, l1;
      g_split_pair(rx[p].x, rx[p].y, h0, l0); g_split_pair(rx[p].z, rx[p].w, h1, l1);
      *reinterpret_cast<uint2*>(base + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(base + kPlane + o) = make_uint2(l0, l1);
      g_split_pair(rw[p].x, rw[p].y, h0, l0); g_split_pair(rw[p].z, rw[p].w, h1, l1);
      *reinterpret_cast<uint2*>(base + 2 * kPlane + o) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(base + 3 * kPlane + o) = make_uint2(l0, l1);
    }
  };
  // fragment of k step s: 8 consecutive k (one 16-byte piece 2 s + kg, swizzled) of row r, from the hi and the lo plane
  auto frag = [&](const unsigned* hi_plane, int r, int s, g_u32x4& hi, g_u32x4& lo) {
    const int o = r * 16 + (((2 * s + kg) ^ ((r >> 2) & 3)) << 2);
    hi = *reinterpret_cast<const g_u32x4*>(hi_plane + o);
    lo = *reinterpret_cast<const g_u32x4*>(hi_plane + kPlane + o);
  };

  g_f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / kBK;
  request(0);
  write_planes(0);                                                   // chunk 0 -> buffer 0
  if (nk > 1) request(kBK);
  // chunk ks: buffer ks & 1 is multiplied; the registers hold chunk ks + 1, which is converted into the other buffer right after the
  // barrier, and chunk ks + 2 is requested into them.  WRITE / REQ are compile-time so that the steady-state body is one basic block
  // (a branch inside it pins every fragment read in front of its own MFMAs).
  auto chunk = [&](int ks, auto w_tag, auto r_tag) {
    constexpr bool WRITE = decltype(w_tag)::value, REQ = decltype(r_tag)::value;
    lds_barrier();                                                   // chunk ks is in LDS; nobody reads the other buffer any more
    if (WRITE) write_planes((ks + 1) & 1);
    if (REQ) request((ks + 2) * kBK);
    const unsigned* xt = lds + (ks & 1) * 4 * kPlane;
    const unsigned* wt = xt + 2 * kPlane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      g_u32x4 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) frag(wt, wn + 32 * j + jl, s, bh[j], bl[j]);
      g_u32x4 ah, al, nh, nl;
      frag(xt, wm + jl, s, ah, al);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (i < 3) frag(xt, wm + 32 * (i + 1) + jl, s, nh, nl);     // the next X tile's fragment flies under this tile's MFMAs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = g_mfma(ah, bh[j], acc[i][j]);
          acc[i][j] = g_mfma(ah, bl[j], acc[i][j]);
          acc[i][j] = g_mfma(al, bh[j], acc[i][j]);
        }
        ah = nh; al = nl;
      }
    }
  };
  typedef std::true_type Y_; typedef std::false_type N_;
  int ks = 0;
#pragma unroll 1
  for (; ks + 2 < nk; ++ks) chunk(ks, Y_{}, Y_{});
  if (ks + 1 < nk) { chunk(ks, Y_{}, N_{}); ++ks; }
  chunk(ks, N_{}, N_{});
  // epilogue: accumulator register r of lane (jl, kg) = row (r & 3) + 8 (r >> 2) + 4 kg, column jl of the 32 x 32 tile
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = n0 + wn + 32 * j + jl;
    const float bv = bias ? bias[col] : 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* yp = Y + (size_t)(m0 + wm + 32 * i + 4 * kg) * N + col;
#pragma unroll
      for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * N] = acc[i][j][r] + bv;
    }
  }
  lds_barrier();                                                     // every wave is done with the last chunk's buffer before the next tile overwrites it
  }
}

}  // namespace
}  // namespace tsg

using namespace tsg;

// Y[M,N] = X[M,K] W[N,K]^T (+ bias, may be NULL) in the split-precision arithmetic, operands converted on load.
// M % 256 == 0, N % 256 == 0, K % 32 == 0 (TSG_E_SHAPE otherwise: the caller uses tsg_split_bf16x3 + a bf16 GEMM).
extern "C" int tsg_gemm_f32s(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, void* stream) {
  const char* fn = "tsg_gemm_f32s";
  for (const void* p : {x, w, (const void*)y}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (M <= 0 || N <= 0 || K <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension M=%d N=%d K=%d", fn, M, N, K);
  if (M % kTM || N % kTN || K % kBK)
    return set_error(TSG_E_SHAPE, "%s: M=%d, N=%d must be multiples of 256 and K=%d of 32", fn, M, N, K);
  auto kern = gemm_nt_f32s_kernel;
  hipError_t e = allow_lds(kern, kGemmLds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute(%zu): %s", fn, kGemmLds, hipGetErrorString(e));
  const int tiles_n = N / kTN;
  const int tiles = (M / kTM) * tiles_n, cus = device_cu_count();
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGT), kGemmLds, static_cast<hipStream_t>(stream), (const float*)x,
                     (const float*)w, (const float*)bias, (float*)y, M, N, K, tiles_n);
  return check_launch(fn);
}
