// MomentPooling's masked means for gfx950 (reference grounding/model/components/TemporalOrderDiscriminator.py:29-46,
// `average_mask` applied to the target / fore / back clip ranges of the GMD's temporal-order discriminator):
//     pooled[b,k,:] = sum_t m_k[b,t] * feat[b,t,:] / (sum_t m_k[b,t] + 1e-6),      k = 0..2   (mask_logits with value 0 = a product)
// ONE pass over feat for the three ranges (the reference makes four [B,T,D] elementwise passes and a reduction per range; the torch
// formulation of rounds 1-3 was stack + bmm + divide), and one pass back:
//     dfeat[b,t,:] = sum_k m_k[b,t] / (cnt_k[b] + 1e-6) * dpooled[b,k,:].
// HBM-bound streaming kernels: forward reads feat once (T*D elements per item), backward writes dfeat once.
// Workgroup = (batch item, 256-column slice): 4 waves, wave w takes rows t = w, w + 4, ...; lanes own 4 consecutive columns
// (float4 / 8-byte bf16 pieces, coalesced rows); the masks of a row are wave-uniform (scalar loads); the four waves' partial sums
// meet in LDS in a fixed order (results are run-to-run identical: no atomics).
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kMpThreads = 256;
constexpr int kMpWaves = kMpThreads / kWave;
constexpr int kMpCols = 4 * kWave;               // columns per workgroup

template <typename ST>
__global__ __launch_bounds__(kMpThreads) void moment_pool_fwd_kernel(const ST* __restrict__ feat, const float* __restrict__ m0,
                                                                     const float* __restrict__ m1, const float* __restrict__ m2,
                                                                     float* __restrict__ pooled, int B, int T, int D, int slices) {
  __shared__ float4 red[kMpWaves][3][kWave];
  __shared__ float cnt[kMpWaves][3];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / slices, col = (blockIdx.x % slices) * kMpCols + lane * 4;
  const bool ok = col < D;                       // D % 4 == 0: a lane's four columns exist together
  const ST* base = feat + (size_t)b * T * D + (ok ? col : 0);
  const float* mk[3] = {m0 + (size_t)b * T, m1 + (size_t)b * T, m2 + (size_t)b * T};
  float4 acc[3];
  float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 3; ++k) acc[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
  for (int t = wv; t < T; t += kMpWaves) {
    const float4 v = ok ? ld4(base + (size_t)t * D) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float m = mk[k][t];                  // wave-uniform
      acc[k].x = fmaf(m, v.x, acc[k].x); acc[k].y = fmaf(m, v.y, acc[k].y);
      acc[k].z = fmaf(m, v.z, acc[k].z); acc[k].w = fmaf(m, v.w, acc[k].w);
      c[k] += m;
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    red[wv][k][lane] = acc[k];
    if (lane == 0) cnt[wv][k] = c[k];
  }
  __syncthreads();
  if (wv == 0 && ok) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      float4 s = red[0][k][lane];
      float n = cnt[0][k];
#pragma unroll
      for (int w = 1; w < kMpWaves; ++w) {
        const float4 r = red[w][k][lane];
        s.x += r.x; s.y += r.y; s.z += r.z; s.w += r.w;
        n += cnt[w][k];
      }
      const float inv = 1.f / (n + 1e-6f);       // TemporalOrderDiscriminator.py:31: sum(mask) + 1e-6
      *reinterpret_cast<float4*>(pooled + ((size_t)b * 3 + k) * D + col) = make_float4(s.x * inv, s.y * inv, s.z * inv, s.w * inv);
    }
  }
}

template <typename ST>
__global__ __launch_bounds__(kMpThreads) void moment_pool_bwd_kernel(const float* __restrict__ dpooled, const float* __restrict__ m0,
                                                                     const float* __restrict__ m1, const float* __restrict__ m2,
                                                                     ST* __restrict__ dfeat, int B, int T, int D, int slices) {
  __shared__ float cnt[kMpWaves][3];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / slices, col = (blockIdx.x % slices) * kMpCols + lane * 4;
  const bool ok = col < D;
  const float* mk[3] = {m0 + (size_t)b * T, m1 + (size_t)b * T, m2 + (size_t)b * T};
  // the three range sizes of this item (every workgroup of the item forms them: 3 T floats, L2-resident)
  float c[3] = {0.f, 0.f, 0.f};
  for (int t = tid; t < T; t += kMpThreads) {
#pragma unroll
    for (int k = 0; k < 3; ++k) c[k] += mk[k][t];
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    c[k] = wave_allsum(c[k]);
    if (lane == 0) cnt[wv][k] = c[k];
  }
  __syncthreads();
  float4 g[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    float n = 0.f;
#pragma unroll
    for (int w = 0; w < kMpWaves; ++w) n += cnt[w][k];
    const float inv = 1.f / (n + 1e-6f);
    const float4 d = ok ? *reinterpret_cast<const float4*>(dpooled + ((size_t)b * 3 + k) * D + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    g[k] = make_float4(d.x * inv, d.y * inv, d.z * inv, d.w * inv);
  }
  ST* base = dfeat + (size_t)b * T * D + (ok ? col : 0);
#pragma unroll 4
  for (int t = wv; t < T; t += kMpWaves) {
    const float a0 = mk[0][t], a1 = mk[1][t], a2 = mk[2][t];     // wave-uniform
    float4 o;
    o.x = fmaf(a0, g[0].x, fmaf(a1, g[1].x, a2 * g[2].x)); o.y = fmaf(a0, g[0].y, fmaf(a1, g[1].y, a2 * g[2].y));
    o.z = fmaf(a0, g[0].z, fmaf(a1, g[1].z, a2 * g[2].z)); o.w = fmaf(a0, g[0].w, fmaf(a1, g[1].w, a2 * g[2].w));
    if (ok) st4(base + (size_t)t * D, o);
  }
}

int mp_check(const char* fn, std::initializer_list<const void*> ptrs, int B, int T, int D, int dtype) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  }
  if (dtype != TSG_F32 && dtype != TSG_BF16) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, or TSG_BF16 = feat / dfeat stored as bf16)", fn, dtype);
  if (B <= 0 || T <= 0 || D <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d D=%d", fn, B, T, D);
  if (D % 4) return set_error(TSG_E_ALIGN, "%s: D=%d must be a multiple of 4", fn, D);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_moment_pool_fwd(const void* feat, const void* m_target, const void* m_fore, const void* m_back, void* pooled,
                                   int B, int T, int D, int dtype, void* stream) {
  const char* fn = "tsg_moment_pool_fwd";
  int rc = mp_check(fn, {feat, m_target, m_fore, m_back, (const void*)pooled}, B, T, D, dtype);
  if (rc) return rc;
  if (!aligned16(feat) || !aligned16(pooled)) return set_error(TSG_E_ALIGN, "%s: feat / pooled not 16-byte aligned", fn);
  const int slices = cdiv(D, kMpCols);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL(moment_pool_fwd_kernel<bf16_t>, dim3(B * slices), dim3(kMpThreads), 0, st, (const bf16_t*)feat, (const float*)m_target,
                       (const float*)m_fore, (const float*)m_back, (float*)pooled, B, T, D, slices);
  else
    hipLaunchKernelGGL(moment_pool_fwd_kernel<float>, dim3(B * slices), dim3(kMpThreads), 0, st, (const float*)feat, (const float*)m_target,
                       (const float*)m_fore, (const float*)m_back, (float*)pooled, B, T, D, slices);
  return check_launch(fn);
}

extern "C" int tsg_moment_pool_bwd(const void* dpooled, const void* m_target, const void* m_fore, const void* m_back, void* dfeat,
                                   int B, int T, int D, int dtype, void* stream) {
  const char* fn = "tsg_moment_pool_bwd";
  int rc = mp_check(fn, {dpooled, m_target, m_fore, m_back, (const void*)dfeat}, B, T, D, dtype);
  if (rc) return rc;
  if (!aligned16(dfeat) || !aligned16(dpooled)) return set_error(TSG_E_ALIGN, "%s: dfeat / dpooled not 16-byte aligned", fn);
  const int slices = cdiv(D, kMpCols);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL(moment_pool_bwd_kernel<bf16_t>, dim3(B * slices), dim3(kMpThreads), 0, st, (const float*)dpooled, (const float*)m_target,
                       (const float*)m_fore, (const float*)m_back, (bf16_t*)dfeat, B, T, D, slices);
  else
    hipLaunchKernelGGL(moment_pool_bwd_kernel<float>, dim3(B * slices), dim3(kMpThreads), 0, st, (const float*)dpooled, (const float*)m_target,
                       (const float*)m_fore, (const float*)m_back, (float*)dfeat, B, T, D, slices);
  return check_launch(fn);
}
