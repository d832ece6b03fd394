// K5 -- matching head for gfx950: the part of VideoTextSemanticMatch (reference grounding/model/components/
// DistributionAlign.py:51-98: VideoTextConcat -> TwoLayerdMLP) that follows the video-half GEMM.  As in K3 the first
// Linear is split W1 = [W1v | W1s]: the sentence half is a per-pair row cs[b] = W1s q[b] + b1, so
//     z[b,t,:] = y[b,t,:] + cs[b,:],   y = W1v video;      l[b,t] = w2 . act(z[b,t,:]) + b2,   act in {relu, tanh, sigmoid}
// and the [B,T,Dv+Dq] concat tensor never exists.  As torch ops the tail (add, activation, the 1-output Linear) is three
// passes over [B,T,H] forward and two degenerate GEMMs (K = 1 and M = 1) plus two passes backward.
// forward : one wave per clip row (float4 per lane, coalesced), wave reduction of the dot product.  Reads y once.
// backward: dy[b,t,:] = dl[b,t] act'(z) w2;  dcs[b,:] = sum_t dy[b,t,:];  dw2 = sum_{b,t} dl act(z);  db2 = sum dl.
//           Workgroup = (pair, 32 clips): lanes own hidden columns, so dcs / dw2 are per-lane running sums over the
//           workgroup's rows, folded across its waves in LDS and added to the outputs with atomics (zeroed by the call).
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kMhThreads = 512;
constexpr int kMhWaves = kMhThreads / kWave;
constexpr int kMhRows = 32;              // clip rows per workgroup
constexpr int kMhMaxH4 = 4;              // H <= 64 lanes * 4 * kMhMaxH4 = 1024 hidden columns

__device__ __forceinline__ int mh_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

template <int ACT> __device__ __forceinline__ float act_f(float z) {          // 0 relu, 1 tanh, 2 sigmoid
  if (ACT == 0) return fmaxf(z, 0.f);
  if (ACT == 1) { const float e = fast_exp2(clampf(z, -44.f, 44.f) * k2Log2e); return 1.f - 2.f * fast_rcp(e + 1.f); }
  return fast_rcp(1.f + fast_exp2(-z * kLog2e));
}
template <int ACT> __device__ __forceinline__ float act_d(float z, float a) {  // derivative given z and a = act(z)
  if (ACT == 0) return z > 0.f ? 1.f : 0.f;
  if (ACT == 1) return 1.f - a * a;
  return a * (1.f - a);
}

// ST: storage type of y (and dy): float or bf16_t (dtype TSG_BF16); cs, w2, b2, the logits and all sums stay fp32.
template <int ACT, typename ST>
__global__ __launch_bounds__(kMhThreads) void match_head_fwd_kernel(const ST* __restrict__ y, const float* __restrict__ cs,
                                                                    const float* __restrict__ w2, const float* __restrict__ b2,
                                                                    float* __restrict__ logit, int B, int T, int H) {
  const int lane = threadIdx.x & 63, wv = mh_wave_id();
  const int tiles = (T + kMhRows - 1) / kMhRows;
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * kMhRows;
  const int H4 = H / 4;
  float4 c[kMhMaxH4], w[kMhMaxH4];
#pragma unroll
  for (int q = 0; q < kMhMaxH4; ++q) {
    const int j = lane + 64 * q;
    c[q] = j < H4 ? reinterpret_cast<const float4*>(cs + (size_t)b * H)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    w[q] = j < H4 ? reinterpret_cast<const float4*>(w2)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const float bias = b2[0];
  constexpr int NR = kMhRows / kMhWaves;                            // rows per wave: all requested up front (one round trip)
  float4 v[NR][kMhMaxH4];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int t = t0 + wv + kMhWaves * i;
    const ST* row = y + ((size_t)b * T + (t < T ? t : 0)) * H;
#pragma unroll
    for (int q = 0; q < kMhMaxH4; ++q) {
      const int j = lane + 64 * q;
      v[i][q] = (t < T && j < H4) ? ld4(row + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const int t = t0 + wv + kMhWaves * i;
    if (t >= T) break;                                             // wave-uniform
    float acc = 0.f;
#pragma unroll
    for (int q = 0; q < kMhMaxH4; ++q) {
      if (lane + 64 * q < H4)
        acc += w[q].x * act_f<ACT>(v[i][q].x + c[q].x) + w[q].y * act_f<ACT>(v[i][q].y + c[q].y) +
               w[q].z * act_f<ACT>(v[i][q].z + c[q].z) + w[q].w * act_f<ACT>(v[i][q].w + c[q].w);
    }
    acc = wave_allsum(acc);
    if (lane == 0) logit[(size_t)b * T + t] = acc + bias;
  }
}

template <int ACT, typename ST>
__global__ __launch_bounds__(kMhThreads) void match_head_bwd_kernel(const ST* __restrict__ y, const float* __restrict__ cs,
                                                                    const float* __restrict__ w2, const float* __restrict__ dl,
                                                                    ST* __restrict__ dy, float* __restrict__ dcs,
                                                                    float* __restrict__ dw2, float* __restrict__ db2,
                                                                    int B, int T, int H) {
  __shared__ float fold[kMhWaves][256 * kMhMaxH4];                 // per wave: H partial sums (<= 1024 floats)
  const int lane = threadIdx.x & 63, wv = mh_wave_id();
  const int tiles = (T + kMhRows - 1) / kMhRows;
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * kMhRows;
  const int H4 = H / 4;
  float4 c[kMhMaxH4], w[kMhMaxH4], sc[kMhMaxH4], sw[kMhMaxH4];
#pragma unroll
  for (int q = 0; q < kMhMaxH4; ++q) {
    const int j = lane + 64 * q;
    c[q] = j < H4 ? reinterpret_cast<const float4*>(cs + (size_t)b * H)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    w[q] = j < H4 ? reinterpret_cast<const float4*>(w2)[j] : make_float4(0.f, 0.f, 0.f, 0.f);
    sc[q] = make_float4(0.f, 0.f, 0.f, 0.f); sw[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float sb = 0.f;
  // rows of this wave, all requested up front (one round trip).  (Batches of two -- 114 instead of ~150 VGPRs, two
  // workgroups per CU -- measured 69.8 vs 63.7 us: the kernel is bound by its fold + atomics tail, not by occupancy.)
  constexpr int NR = kMhRows / kMhWaves, NB = NR;
#pragma unroll 1
  for (int i0 = 0; i0 < NR; i0 += NB) {
    float4 v[NB][kMhMaxH4];
    float g[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int t = t0 + wv + kMhWaves * (i0 + i);
      const bool ok = t < T;                                       // wave-uniform
      const size_t r = (size_t)b * T + (ok ? t : 0);
      g[i] = ok ? dl[r] : 0.f;
      const ST* row = y + r * H;
#pragma unroll
      for (int q = 0; q < kMhMaxH4; ++q) {
        const int j = lane + 64 * q;
        v[i][q] = (ok && j < H4) ? ld4(row + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int t = t0 + wv + kMhWaves * (i0 + i);
      if (t >= T) break;
      const size_t r = (size_t)b * T + t;
      sb += g[i];
      ST* drow = dy + r * H;
#pragma unroll
      for (int q = 0; q < kMhMaxH4; ++q) {
        const int j = lane + 64 * q;
        if (j < H4) {
          const float z[4] = {v[i][q].x + c[q].x, v[i][q].y + c[q].y, v[i][q].z + c[q].z, v[i][q].w + c[q].w};
          const float ww[4] = {w[q].x, w[q].y, w[q].z, w[q].w};
          float d[4], a[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { a[k] = act_f<ACT>(z[k]); d[k] = g[i] * ww[k] * act_d<ACT>(z[k], a[k]); }
          st4(drow + 4 * j, make_float4(d[0], d[1], d[2], d[3]));
          sc[q].x += d[0]; sc[q].y += d[1]; sc[q].z += d[2]; sc[q].w += d[3];
          sw[q].x += g[i] * a[0]; sw[q].y += g[i] * a[1]; sw[q].z += g[i] * a[2]; sw[q].w += g[i] * a[3];
        }
      }
    }
  }
  // fold the waves' column sums, then one atomic per column and workgroup
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kMhMaxH4; ++q) {
      const float4 v = pass == 0 ? sc[q] : sw[q];
      reinterpret_cast<float4*>(&fold[wv][0])[lane + 64 * q] = v;
    }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += kMhThreads) {
      float s = 0.f;
#pragma unroll
      for (int u = 0; u < kMhWaves; ++u) s += fold[u][col];
      if (pass == 0) atomicAdd(dcs + (size_t)b * H + col, s); else atomicAdd(dw2 + col, s);
    }
  }
  sb = (lane == 0) ? sb : 0.f;                                      // every lane of a wave carries the same row gradients
  sb = wave_allsum(sb);
  if (lane == 0) atomicAdd(db2, sb);
}

// backward, round 5: COLUMN-SLICED workgroups.  The kernel above ends every workgroup (pair, 32 clips) with 2 H float atomics -- 512 workgroups x 2048
// at [128, 128, 1024]: a million atomics on 132 K addresses, 512 deep on each dw2 column; it measured 75 us for a 128 MB pass (1.7 TB/s), its own
// batching experiment already said "bound by its fold + atomics tail".  Here a workgroup owns 128 hidden columns of a GROUP OF WHOLE PAIRS: a half
// wave handles one clip row (32 lanes x 4 columns = 512 contiguous bytes, fp32), dcs[b, cols] is complete inside the workgroup (plain stores, no
// atomics), dw2 costs 128 atomics per workgroup (32-64 deep per column instead of 512), db2 one per column-0 workgroup.
constexpr int kMcCols = 128;             // hidden columns per workgroup
template <int ACT, typename ST>
__global__ __launch_bounds__(kMhThreads) void match_head_bwd_cols_kernel(const ST* __restrict__ y, const float* __restrict__ cs,
                                                                         const float* __restrict__ w2, const float* __restrict__ dl,
                                                                         ST* __restrict__ dy, float* __restrict__ dcs,
                                                                         float* __restrict__ dw2, float* __restrict__ db2,
                                                                         int B, int T, int H, int ipg) {
  __shared__ float4 fold[2 * kMhWaves][32];                        // [half wave][4 columns of a lane]
  __shared__ float gsum[2 * kMhWaves];
  const int lane = threadIdx.x & 63, wv = mh_wave_id();
  const int hw = 2 * wv + (lane >> 5), l32 = lane & 31;            // half wave 0..15 = row slot; lane of the half wave = 4 columns
  const int nslices = H / kMcCols;
  const int slice = blockIdx.x % nslices, grp = blockIdx.x / nslices;
  const int col = slice * kMcCols + 4 * l32;
  const float4 w = *reinterpret_cast<const float4*>(w2 + col);
  const float ww[4] = {w.x, w.y, w.z, w.w};
  float4 sw = make_float4(0.f, 0.f, 0.f, 0.f);
  float sb = 0.f;
  constexpr int RS = 2 * kMhWaves;                                 // 16 rows per pass
  constexpr int NB = 4;                                            // rows per half wave requested together
  for (int b = grp * ipg; b < B && b < (grp + 1) * ipg; ++b) {
    const float4 c = *reinterpret_cast<const float4*>(cs + (size_t)b * H + col);
    const float cc[4] = {c.x, c.y, c.z, c.w};
    float4 sc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int t0 = hw; t0 < T; t0 += RS * NB) {
      float4 v[NB];
      float g[NB];
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int t = t0 + RS * i;
        const bool ok = t < T;
        const size_t r = (size_t)b * T + (ok ? t : 0);
        g[i] = ok ? dl[r] : 0.f;
        v[i] = ld4(y + r * H + col);
      }
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int t = t0 + RS * i;
        if (t < T) {
          const size_t r = (size_t)b * T + t;
          const float z[4] = {v[i].x + cc[0], v[i].y + cc[1], v[i].z + cc[2], v[i].w + cc[3]};
          float d[4], a[4];
#pragma unroll
          for (int k = 0; k < 4; ++k) { a[k] = act_f<ACT>(z[k]); d[k] = g[i] * ww[k] * act_d<ACT>(z[k], a[k]); }
          st4(dy + r * H + col, make_float4(d[0], d[1], d[2], d[3]));
          sc.x += d[0]; sc.y += d[1]; sc.z += d[2]; sc.w += d[3];
          sw.x += g[i] * a[0]; sw.y += g[i] * a[1]; sw.z += g[i] * a[2]; sw.w += g[i] * a[3];
          sb += g[i];
        }
      }
    }
    // dcs[b, cols]: the 16 row slots meet in LDS, in slot order (run-to-run identical); complete here -- a plain store
    __syncthreads();
    fold[hw][l32] = sc;
    __syncthreads();
    if (threadIdx.x < 32) {
      float4 s4 = fold[0][threadIdx.x];
#pragma unroll
      for (int u = 1; u < RS; ++u) { const float4 q = fold[u][threadIdx.x]; s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w; }
      *reinterpret_cast<float4*>(dcs + (size_t)b * H + slice * kMcCols + 4 * threadIdx.x) = s4;
    }
  }
  __syncthreads();
  fold[hw][l32] = sw;
  if (l32 == 0) gsum[hw] = sb;                                      // every lane of a half wave carries the same row gradients
  __syncthreads();
  if (threadIdx.x < 32) {
    float4 s4 = fold[0][threadIdx.x];
#pragma unroll
    for (int u = 1; u < RS; ++u) { const float4 q = fold[u][threadIdx.x]; s4.x += q.x; s4.y += q.y; s4.z += q.z; s4.w += q.w; }
    float* o = dw2 + slice * kMcCols + 4 * threadIdx.x;
    atomicAdd(o, s4.x); atomicAdd(o + 1, s4.y); atomicAdd(o + 2, s4.z); atomicAdd(o + 3, s4.w);
  }
  if (slice == 0 && threadIdx.x == 0) {
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < RS; ++u) s += gsum[u];
    atomicAdd(db2, s);
  }
}

int mh_check(const char* fn, int B, int T, int H, int act, int dtype) {
  if (dtype != TSG_F32 && dtype != TSG_BF16)
    return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, or TSG_BF16 = y / dy stored as bf16)", fn, dtype);
  if (B <= 0 || T <= 0 || H <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d H=%d", fn, B, T, H);
  if (H % 4 || H > 256 * kMhMaxH4) return set_error(TSG_E_SHAPE, "%s: H=%d must be a multiple of 4 and <= %d", fn, H, 256 * kMhMaxH4);
  if (act < 0 || act > 2) return set_error(TSG_E_SHAPE, "%s: activation %d (0 relu, 1 tanh, 2 sigmoid)", fn, act);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_match_head_fwd(const void* y, const void* cs, const void* w2, const void* b2, void* logits,
                                  int B, int T, int H, int activation, int dtype, void* stream) {
  const char* fn = "tsg_match_head_fwd";
  for (const void* p : {y, cs, w2, b2, (const void*)logits}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (p != b2 && !aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = mh_check(fn, B, T, H, activation, dtype);
  if (rc) return rc;
  const int grid = B * cdiv(T, kMhRows);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    auto k = activation == 0 ? match_head_fwd_kernel<0, S> : activation == 1 ? match_head_fwd_kernel<1, S> : match_head_fwd_kernel<2, S>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kMhThreads), 0, st, (const S*)y, (const float*)cs, (const float*)w2, (const float*)b2,
                       (float*)logits, B, T, H);
  } else {
    auto k = activation == 0 ? match_head_fwd_kernel<0, float> : activation == 1 ? match_head_fwd_kernel<1, float> : match_head_fwd_kernel<2, float>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kMhThreads), 0, st, (const float*)y, (const float*)cs, (const float*)w2, (const float*)b2,
                       (float*)logits, B, T, H);
  }
  return check_launch(fn);
}

extern "C" int tsg_match_head_bwd(const void* y, const void* cs, const void* w2, const void* dlogits, void* dy, void* dcs,
                                  void* dw2, void* db2, int B, int T, int H, int activation, int dtype, void* stream) {
  const char* fn = "tsg_match_head_bwd";
  for (const void* p : {y, cs, w2, dlogits, (const void*)dy, (const void*)dcs, (const void*)dw2, (const void*)db2}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (p != db2 && !aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = mh_check(fn, B, T, H, activation, dtype);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  static const bool cols_off = getenv("TSG_MH_BWD_COLS") && atoi(getenv("TSG_MH_BWD_COLS")) == 0;     // A/B: the (pair, 32 clips) kernel
  const bool cols = !cols_off && H % kMcCols == 0 && (long long)B * T >= 2048;
  // one zero-fill node, not three (the column-sliced kernel writes dcs whole: only the two atomic targets are zeroed)
  hipError_t e = cols ? zero3_async(dw2, sizeof(float) * H, db2, sizeof(float), nullptr, 0, st)
                      : zero3_async(dcs, sizeof(float) * (size_t)B * H, dw2, sizeof(float) * H, db2, sizeof(float), st);
  if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
  if (cols) {
    const int nslices = H / kMcCols;
    int groups = 512 / nslices;                                      // ~512 workgroups
    if (groups > B) groups = B;
    if (groups < 1) groups = 1;
    const int ipg = cdiv(B, groups);
    const int grid2 = nslices * cdiv(B, ipg);
    if (dtype == TSG_BF16) {
      using S = bf16_t;
      auto k = activation == 0 ? match_head_bwd_cols_kernel<0, S> : activation == 1 ? match_head_bwd_cols_kernel<1, S> : match_head_bwd_cols_kernel<2, S>;
      hipLaunchKernelGGL(k, dim3(grid2), dim3(kMhThreads), 0, st, (const S*)y, (const float*)cs, (const float*)w2,
                         (const float*)dlogits, (S*)dy, (float*)dcs, (float*)dw2, (float*)db2, B, T, H, ipg);
    } else {
      auto k = activation == 0 ? match_head_bwd_cols_kernel<0, float> : activation == 1 ? match_head_bwd_cols_kernel<1, float> : match_head_bwd_cols_kernel<2, float>;
      hipLaunchKernelGGL(k, dim3(grid2), dim3(kMhThreads), 0, st, (const float*)y, (const float*)cs, (const float*)w2,
                         (const float*)dlogits, (float*)dy, (float*)dcs, (float*)dw2, (float*)db2, B, T, H, ipg);
    }
    return check_launch(fn);
  }
  const int grid = B * cdiv(T, kMhRows);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    auto k = activation == 0 ? match_head_bwd_kernel<0, S> : activation == 1 ? match_head_bwd_kernel<1, S> : match_head_bwd_kernel<2, S>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kMhThreads), 0, st, (const S*)y, (const float*)cs, (const float*)w2,
                       (const float*)dlogits, (S*)dy, (float*)dcs, (float*)dw2, (float*)db2, B, T, H);
  } else {
    auto k = activation == 0 ? match_head_bwd_kernel<0, float> : activation == 1 ? match_head_bwd_kernel<1, float> : match_head_bwd_kernel<2, float>;
    hipLaunchKernelGGL(k, dim3(grid), dim3(kMhThreads), 0, st, (const float*)y, (const float*)cs, (const float*)w2,
                       (const float*)dlogits, (float*)dy, (float*)dcs, (float*)dw2, (float*)db2, B, T, H);
  }
  return check_launch(fn);
}
