// Dense projection GEMM on the fp32 matrix cores (SURVEY §8b "tsg_gemm_*"): Y[M,N] = X[M,K] W[N,K]^T (+ bias[N]),
// i.e. torch.nn.Linear's layout (the d x d projections W_a, W_s, sent_linear, wq/wk/wv/wo and the first boundary
// Linear of the reference, e.g. attention.py:104-106, 63-66; SpanPredictor.py:62-67).
//
// Tiling for gfx950.  Workgroup = 128 x 128 output tile, 4 waves as 2 x 2, each wave 64 x 64 = 2 x 2 MFMA tiles of
// v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains; 64 accumulator registers).  K advances in 32-column chunks: the X
// and W row tiles (128 x 32 each, 128-byte coalesced row segments) go global -> registers -> LDS, double buffered, one
// workgroup barrier per chunk; the NEXT chunk's global loads are issued before the current chunk's MFMAs.  LDS rows
// have stride BK+2 floats (= 2 mod 32, 8-byte aligned): a lane's paired-k ds_read_b64 (columns 4s+2k', +1 of its row)
// is conflict-free per half-wave, and one b64 feeds two MFMAs (k order permuted identically for both operands).
// 2 x 34.8 KiB of LDS and ~130 VGPRs: two workgroups per CU, so one's staging overlaps the other's MFMAs.
// Requirements: K % 4 == 0, rows 16-byte aligned; M, N arbitrary (edge tiles are zero-filled / masked).
#include "tsg_common.h"

namespace tsg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int GT = 256;              // threads
constexpr int BM = 128, BN = 128, BK = 32;
constexpr int LS = BK + 4;           // LDS row stride (floats): = 4 mod 64 -> the 16-lane groups of a ds_read_b128 hit 64 distinct banks

__global__ __launch_bounds__(GT) void gemm_nt_f32_kernel(const float* __restrict__ X, const float* __restrict__ W,
                                                         const float* __restrict__ bias, float* __restrict__ Y,
                                                         int M, int N, int K, int tiles_n) {
  __shared__ __align__(16) float Al[2][BM * LS];       // double buffered: 2 x 2 x 18 KiB, two workgroups per CU
  __shared__ __align__(16) float Bl[2][BN * LS];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles_n);       // the N tiles of one M tile share an XCD (X rows stay in its L2)
  const int m0 = (bid / tiles_n) * BM, n0 = (bid % tiles_n) * BN;
  const int wm = (wv >> 1) * 64, wn = (wv & 1) * 64;
  const int jl = lane & 31, kk = lane >> 5;

  // staging role: 8 threads per row (8 float4 = 32 columns = 128 B), 32 rows per pass, 4 passes per operand
  const int sr = tid >> 3, sc = (tid & 7) * 4;
  constexpr int NP8 = BM * BK / 4 / GT;
  float4 ra[NP8], rb[NP8];
  auto gload = [&](int k0) {
#pragma unroll
    for (int p = 0; p < NP8; ++p) {
      const int r = sr + 32 * p;
      const bool kin = k0 + sc < K;
      ra[p] = (m0 + r < M && kin) ? *reinterpret_cast<const float4*>(X + (size_t)(m0 + r) * K + k0 + sc) : make_float4(0.f, 0.f, 0.f, 0.f);
      rb[p] = (n0 + r < N && kin) ? *reinterpret_cast<const float4*>(W + (size_t)(n0 + r) * K + k0 + sc) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto lstore = [&](int buf) {
#pragma unroll
    for (int p = 0; p < NP8; ++p) {
      float* a = &Al[buf][(sr + 32 * p) * LS + sc];
      float* b = &Bl[buf][(sr + 32 * p) * LS + sc];
      *reinterpret_cast<float4*>(a) = ra[p];
      *reinterpret_cast<float4*>(b) = rb[p];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = (K + BK - 1) / BK;
  gload(0);
  lstore(0);
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    const bool more = kc + 1 < nk;
    if (more) gload((kc + 1) * BK);                        // flies during the first half of this chunk's MFMAs
    const float* a0 = &Al[buf][(wm + jl) * LS + 4 * kk];
    const float* a1 = a0 + 32 * LS;
    const float* b0 = &Bl[buf][(wn + jl) * LS + 4 * kk];
    const float* b1 = b0 + 32 * LS;
    // One ds_read_b128 per row and 8 columns: lane (row, kk) holds columns 8u + 4kk .. +3, i.e. FOUR MFMAs' worth of k
    // (component m pairs k = 8u + m on the low half-wave with k = 8u + 4 + m on the high one, the same for both operands).
    // (Paired-k float2 reads get merged into ds_read2_b64 by hipcc: half the LDS rate and 2-way bank conflicts.)
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int u = 0; u < BK / 8; ++u) {
      const float4 x0 = *reinterpret_cast<const float4*>(a0 + 8 * u), x1 = *reinterpret_cast<const float4*>(a1 + 8 * u);
      const float4 w0 = *reinterpret_cast<const float4*>(b0 + 8 * u), w1 = *reinterpret_cast<const float4*>(b1 + 8 * u);
      const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
      const float wa[4] = {w0.x, w0.y, w0.z, w0.w}, wb[4] = {w1.x, w1.y, w1.z, w1.w};
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[m], wa[m], acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[m], wb[m], acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[m], wa[m], acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(xb[m], wb[m], acc[1][1], 0, 0, 0);
      }
      // the next chunk goes into the OTHER buffer in the middle of this chunk's MFMAs (its loads have had 32 MFMAs
      // = 2k cycles to land; the matrix pipe keeps draining the queued MFMAs while the stores issue)
      if (u == BK / 16 - 1 && more) lstore(buf ^ 1);
    }
    __builtin_amdgcn_s_setprio(0);
    __syncthreads();
  }

  // epilogue: accumulator register r of tile (i, j) = Y[m0 + wm + 32 i + rho(r)][n0 + wn + 32 j + jl], rho(r) = (r&3) + 8 (r>>2) + 4 kk
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int n = n0 + wn + 32 * j + jl;
    const float bv = (bias && n < N) ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (m < M && n < N) Y[(size_t)m * N + n] = acc[i][j][r] + bv;
      }
  }
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_linear_fwd(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, int dtype,
                              void* stream) {
  const char* fn = "tsg_linear_fwd";
  if (!x || !w || !y) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(x) || !aligned16(w)) return set_error(TSG_E_ALIGN, "%s: x / w not 16-byte aligned", fn);
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (M <= 0 || N <= 0 || K <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension M=%d N=%d K=%d", fn, M, N, K);
  if (K % 4) return set_error(TSG_E_ALIGN, "%s: K=%d must be a multiple of 4", fn, K);
  const int tiles_m = cdiv(M, BM), tiles_n = cdiv(N, BN);
  hipLaunchKernelGGL(gemm_nt_f32_kernel, dim3(tiles_m * tiles_n), dim3(GT), 0, static_cast<hipStream_t>(stream),
                     (const float*)x, (const float*)w, (const float*)bias, (float*)y, M, N, K, tiles_n);
  return check_launch(fn);
}
