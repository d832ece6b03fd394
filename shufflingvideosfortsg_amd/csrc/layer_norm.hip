// LayerNorm over the channel axis for gfx950 -- the final `nn.LayerNorm(d)` (eps 1e-5) of QueryAwareEncoder.forward (reference
// grounding/model/components/VideoEncoder.py:96,112) on the [2B*T, d] encoder output, forward and backward.
//     y = (x - mean) * rstd * gamma + beta,   mean / var over the d channels of a row (biased variance, as torch.nn.LayerNorm)
//     dx = rstd * (g - mean_c(g) - xhat * mean_c(g * xhat)),  g = dy * gamma;   dgamma = sum_rows dy * xhat;   dbeta = sum_rows dy
// HBM-bound streaming kernels.  One wave per row: lanes own float4 pieces (d <= 2048: up to 8 pieces per lane), the row statistics
// are wave reductions, the row is read ONCE (kept in registers between the statistics and the output).  The backward reads x and dy
// once, writes dx once, and folds the per-column sums for dgamma / dbeta over the workgroup's rows in registers and LDS into one
// partial row per workgroup; a second small kernel adds the partial rows in workgroup order (run-to-run identical, no atomics).
// (torch's backward at [16384, 1024]: 68.6 + 36.5 us in two kernels that each read x and dy; this one: one pass.)
// ST: storage type of x / y / dy / dx (float, or bf16_t with dtype TSG_BF16); gamma, beta, the statistics and all sums fp32.
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kLnThreads = 256;
constexpr int kLnWaves = kLnThreads / kWave;
constexpr int kLnMaxP = 8;                        // float4 pieces per lane: d <= 64 * 4 * 8 = 2048 (the backward's LDS fold: 128 KB at 8)

template <typename ST, int NPc>
__global__ __launch_bounds__(kLnThreads) void layer_norm_fwd_kernel(const ST* __restrict__ x, const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta, ST* __restrict__ y,
                                                                    float* __restrict__ mean, float* __restrict__ rstd, long rows, int d, float eps) {
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float inv_d = 1.f / (float)d;
  for (long r = (long)blockIdx.x * kLnWaves + wv; r < rows; r += (long)gridDim.x * kLnWaves) {
    const ST* xr = x + r * d;
    float4 v[NPc];
    float s = 0.f;
#pragma unroll
    for (int p = 0; p < NPc; ++p) {
      const int c = (p * 64 + lane) * 4;
      v[p] = c < d ? ld4(xr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[p].x + v[p].y) + (v[p].z + v[p].w);
    }
    const float m = wave_allsum(s) * inv_d;
    float q = 0.f;
#pragma unroll
    for (int p = 0; p < NPc; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c < d) {
        const float a0 = v[p].x - m, a1 = v[p].y - m, a2 = v[p].z - m, a3 = v[p].w - m;
        q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
      }
    }
    const float rs = rsqrtf(wave_allsum(q) * inv_d + eps);
    if (lane == 0) { mean[r] = m; rstd[r] = rs; }
    ST* yr = y + r * d;
#pragma unroll
    for (int p = 0; p < NPc; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c < d) {
        const float4 g = *reinterpret_cast<const float4*>(gamma + c), b = *reinterpret_cast<const float4*>(beta + c);
        st4(yr + c, make_float4((v[p].x - m) * rs * g.x + b.x, (v[p].y - m) * rs * g.y + b.y, (v[p].z - m) * rs * g.z + b.z,
                                (v[p].w - m) * rs * g.w + b.w));
      }
    }
  }
}

// backward: one wave per row; per-column sums of the wave's rows in registers, the workgroup's waves folded in LDS -> partial row
constexpr int kLnBwdThreads = 512;                  // backward: 8 waves per workgroup, at most one workgroup per CU: <= 256 partial rows
constexpr int kLnBwdWaves = kLnBwdThreads / kWave;
template <typename ST, int NPc>
__global__ __launch_bounds__(kLnBwdThreads) void layer_norm_bwd_kernel(const ST* __restrict__ x, const ST* __restrict__ dy,
                                                                    const float* __restrict__ gamma, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, ST* __restrict__ dx,
                                                                    float* __restrict__ part, long rows, int d) {
  extern __shared__ float4 red[];                 // [kLnBwdWaves][2][NPc * 64]
  const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const float inv_d = 1.f / (float)d;
  float4 ag[NPc], ab[NPc], gm[NPc];
#pragma unroll
  for (int p = 0; p < NPc; ++p) {
    const int c = (p * 64 + lane) * 4;
    ag[p] = make_float4(0.f, 0.f, 0.f, 0.f); ab[p] = ag[p];
    gm[p] = c < d ? *reinterpret_cast<const float4*>(gamma + c) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long r = (long)blockIdx.x * kLnBwdWaves + wv; r < rows; r += (long)gridDim.x * kLnBwdWaves) {
    const ST* xr = x + r * d;
    const ST* gr = dy + r * d;
    const float m = mean[r], rs = rstd[r];
    float4 xh[NPc], g[NPc];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int p = 0; p < NPc; ++p) {
      const int c = (p * 64 + lane) * 4;
      const bool ok = c < d;
      const float4 xv = ok ? ld4(xr + c) : make_float4(m, m, m, m);
      const float4 dv = ok ? ld4(gr + c) : make_float4(0.f, 0.f, 0.f, 0.f);
      xh[p] = make_float4((xv.x - m) * rs, (xv.y - m) * rs, (xv.z - m) * rs, (xv.w - m) * rs);
      g[p] = make_float4(dv.x * gm[p].x, dv.y * gm[p].y, dv.z * gm[p].z, dv.w * gm[p].w);
      s1 += (g[p].x + g[p].y) + (g[p].z + g[p].w);
      s2 += (g[p].x * xh[p].x + g[p].y * xh[p].y) + (g[p].z * xh[p].z + g[p].w * xh[p].w);
      ag[p].x += dv.x * xh[p].x; ag[p].y += dv.y * xh[p].y; ag[p].z += dv.z * xh[p].z; ag[p].w += dv.w * xh[p].w;
      ab[p].x += dv.x; ab[p].y += dv.y; ab[p].z += dv.z; ab[p].w += dv.w;
    }
    const float c1 = wave_allsum(s1) * inv_d, c2 = wave_allsum(s2) * inv_d;
    ST* dr = dx + r * d;
#pragma unroll
    for (int p = 0; p < NPc; ++p) {
      const int c = (p * 64 + lane) * 4;
      if (c < d)
        st4(dr + c, make_float4(rs * (g[p].x - c1 - xh[p].x * c2), rs * (g[p].y - c1 - xh[p].y * c2), rs * (g[p].z - c1 - xh[p].z * c2),
                                rs * (g[p].w - c1 - xh[p].w * c2)));
    }
  }
  // fold the waves' column sums (wave order), one partial row [2][d] per workgroup
#pragma unroll
  for (int p = 0; p < NPc; ++p) {
    red[(wv * 2 + 0) * (NPc * 64) + p * 64 + lane] = ag[p];
    red[(wv * 2 + 1) * (NPc * 64) + p * 64 + lane] = ab[p];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 2 * NPc * 64; i += kLnBwdThreads) {
    const int which = i / (NPc * 64), j = i % (NPc * 64);
    float4 s = red[(0 * 2 + which) * (NPc * 64) + j];
#pragma unroll
    for (int w = 1; w < kLnBwdWaves; ++w) {
      const float4 t = red[(w * 2 + which) * (NPc * 64) + j];
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    const int c = j * 4;
    if (c < d) *reinterpret_cast<float4*>(part + ((size_t)blockIdx.x * 2 + which) * d + c) = s;
  }
}

// dgamma / dbeta = sum over the workgroups' partial rows: 64 float4 columns x 4 part groups per workgroup (part p goes to group p % 4,
// summed in p order; the four groups are folded in group order: a fixed order, run-to-run identical)
__global__ __launch_bounds__(256) void layer_norm_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int nparts, int d) {
  __shared__ float4 fold[4][64];
  const int cj = threadIdx.x & 63, pg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + cj;                    // over 2 * d / 4 float4s
  const bool ok = i < 2 * (d / 4);
  const int which = ok ? i / (d / 4) : 0, c = ok ? (i % (d / 4)) * 4 : 0;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (ok) {
#pragma unroll 4
    for (int p = pg; p < nparts; p += 4) {
      const float4 t = *reinterpret_cast<const float4*>(part + ((size_t)p * 2 + which) * d + c);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  fold[pg][cj] = s;
  __syncthreads();
  if (pg == 0 && ok) {
#pragma unroll
    for (int q = 1; q < 4; ++q) { const float4 t = fold[q][cj]; s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w; }
    *reinterpret_cast<float4*>((which ? dbeta : dgamma) + c) = s;
  }
}

int ln_check(const char* fn, std::initializer_list<const void*> ptrs, long long rows, int d, int dtype) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (dtype != TSG_F32 && dtype != TSG_BF16) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, or TSG_BF16 = x / y / dy / dx stored as bf16)", fn, dtype);
  if (rows <= 0 || d <= 0 || d % 4 || d > 64 * 4 * kLnMaxP || rows >= (1LL << 31))
    return set_error(TSG_E_SHAPE, "%s: rows=%lld d=%d (d must be a multiple of 4, <= %d)", fn, rows, d, 64 * 4 * kLnMaxP);
  return 0;
}

inline int ln_grid(long long rows) {
  const long long wgs = (rows + kLnWaves - 1) / kLnWaves;
  const int cap = 8 * device_cu_count();                 // persistent-style: every workgroup's waves stride over the rows
  return (int)(wgs < cap ? wgs : cap);
}

template <typename ST>
int ln_fwd(const char* fn, const void* x, const void* gamma, const void* beta, void* y, void* mean, void* rstd, long long rows, int d,
           float eps, hipStream_t st) {
  const int np = (d + 255) / 256, grid = ln_grid(rows);
#define TSG_LN_FWD(NPV) hipLaunchKernelGGL((layer_norm_fwd_kernel<ST, NPV>), dim3(grid), dim3(kLnThreads), 0, st, (const ST*)x, (const float*)gamma, \
                                           (const float*)beta, (ST*)y, (float*)mean, (float*)rstd, (long)rows, d, eps)
  if (np <= 1) TSG_LN_FWD(1); else if (np <= 2) TSG_LN_FWD(2); else if (np <= 4) TSG_LN_FWD(4); else TSG_LN_FWD(8);
#undef TSG_LN_FWD
  return check_launch(fn);
}

template <typename ST>
int ln_bwd(const char* fn, const void* x, const void* dy, const void* gamma, const void* mean, const void* rstd, void* dx, void* part,
           int grid, long long rows, int d, hipStream_t st) {
  const int np = (d + 255) / 256;
#define TSG_LN_BWD(NPV) { auto kern = layer_norm_bwd_kernel<ST, NPV>; const size_t lds = sizeof(float4) * kLnBwdWaves * 2 * NPV * 64;                 \
    hipError_t e = allow_lds(kern, lds); if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e)); \
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kLnBwdThreads), lds, st, (const ST*)x, (const ST*)dy, (const float*)gamma, (const float*)mean,      \
                       (const float*)rstd, (ST*)dx, (float*)part, (long)rows, d); }
  if (np <= 1) TSG_LN_BWD(1) else if (np <= 2) TSG_LN_BWD(2) else if (np <= 4) TSG_LN_BWD(4) else TSG_LN_BWD(8)
#undef TSG_LN_BWD
  return check_launch(fn);
}

// the backward's grid is fixed by the workspace it was sized for: one partial row pair per workgroup
inline int ln_bwd_grid(long long rows) {
  const long long wgs = (rows + kLnBwdWaves - 1) / kLnBwdWaves;
  const int cap = device_cu_count();
  return (int)(wgs < cap ? wgs : cap);
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_layer_norm_fwd(const void* x, const void* gamma, const void* beta, void* y, void* mean, void* rstd,
                                  long long rows, int d, float eps, int dtype, void* stream) {
  const char* fn = "tsg_layer_norm_fwd";
  int rc = ln_check(fn, {x, gamma, beta, (const void*)y, (const void*)mean, (const void*)rstd}, rows, d, dtype);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  return dtype == TSG_BF16 ? ln_fwd<bf16_t>(fn, x, gamma, beta, y, mean, rstd, rows, d, eps, st)
                           : ln_fwd<float>(fn, x, gamma, beta, y, mean, rstd, rows, d, eps, st);
}

extern "C" long long tsg_layer_norm_bwd_ws_bytes(long long rows, int d) {
  if (rows <= 0 || d <= 0 || d % 4) return -1;
  return (long long)sizeof(float) * ln_bwd_grid(rows) * 2 * d;
}

extern "C" int tsg_layer_norm_bwd(const void* x, const void* dy, const void* gamma, const void* mean, const void* rstd, void* dx,
                                  void* dgamma, void* dbeta, void* ws, long long ws_bytes, long long rows, int d, int dtype, void* stream) {
  const char* fn = "tsg_layer_norm_bwd";
  int rc = ln_check(fn, {x, dy, gamma, mean, rstd, (const void*)dx, (const void*)dgamma, (const void*)dbeta, (const void*)ws}, rows, d, dtype);
  if (rc) return rc;
  const int grid = ln_bwd_grid(rows);
  if (ws_bytes < (long long)sizeof(float) * grid * 2 * d)
    return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_layer_norm_bwd_ws_bytes)", fn, ws_bytes, (long long)sizeof(float) * grid * 2 * d);
  auto st = static_cast<hipStream_t>(stream);
  rc = dtype == TSG_BF16 ? ln_bwd<bf16_t>(fn, x, dy, gamma, mean, rstd, dx, ws, grid, rows, d, st)
                         : ln_bwd<float>(fn, x, dy, gamma, mean, rstd, dx, ws, grid, rows, d, st);
  if (rc) return rc;
  hipLaunchKernelGGL(layer_norm_reduce_kernel, dim3((2 * (d / 4) + 63) / 64), dim3(256), 0, st, (const float*)ws, (float*)dgamma, (float*)dbeta, grid, d);
  return check_launch(fn);
}
