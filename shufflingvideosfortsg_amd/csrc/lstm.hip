// Bidirectional LSTM recurrence for gfx950 -- the glue adjacent to the hot path (reference
// grounding/model/networks/RNN.py:26-48, an nn.LSTM; SURVEY.md 8f "next #1": it is ~85 % of the GPU
// time of a training step when left to MIOpen, which launches one small GEMM + one pointwise kernel
// per time step and direction).
//
// Split of work.  The input projections Gx = X W_ih^T + b_ih + b_hh for all T steps and both
// directions are ONE large GEMM done by the caller (no recurrence in it).  This file does the
// sequential part: per time step ONE launch covers both directions:
//     G_t = Gx[:, t] + h_{t-1} W_hh^T ;  i,f,o = sigmoid, g = tanh ;  c_t = f c_{t-1} + i g ;  h_t = o tanh(c_t)
// Workgroup = (direction, 4 hidden units) -> the 16 gate rows (4 units x i,f,g,o) of W_hh form the
// A operand of v_mfma_f32_16x16x4_f32, h_{t-1}^T (16 batch rows per wave) the B operand, so lane
// (b = lane&15, u = lane>>4) ends up with the four gate pre-activations of ONE (batch, unit) pair in
// its four accumulator registers and the cell update is lane-local.  W_hh rows sit in LDS for the
// whole launch, h_{t-1} streams through LDS in 64-column chunks (row strides = 4 mod 64 floats:
// the paired-k ds_read_b64 operand reads are conflict-free).  fp32 MFMA = exact fp32 FMA chains.
//
// Backward mirrors it: workgroup = (direction, 16 hidden units, 32 batch rows), W_hh passed transposed;
//     dh_t = dOut_t + dG_{t+1} W_hh   (MFMA, K = 4h), then the lane-local cell backward writes
// dG_t [B,T,2,4h] (consumed by the next step and, afterwards, by the caller's weight-gradient GEMMs).
//
// Saved for backward (caller-owned): R [T][2][B][h][4] activated gates, Cs [T][2][B][h] cell states.
#include "tsg_common.h"

namespace tsg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;       // 8 waves x 16 batch rows = 128 rows per pass
constexpr int KC = 64;                          // h_{t-1} columns per LDS chunk
constexpr int HS = KC + 4;                      // chunk row stride (floats), = 4 mod 64

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.f + fast_exp2(-x * kLog2e)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = fast_exp2(clampf(x, -44.f, 44.f) * k2Log2e);
  return 1.f - 2.f * fast_rcp(e + 1.f);
}

// ---------------------------------------------------------------------------------------------
// forward step.  grid = 2 * ceil(h/4).  tt = time index of this step for direction d.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void lstm_fwd_step_kernel(
    const float* __restrict__ Gx, const float* __restrict__ Whh, float* __restrict__ out,
    float* __restrict__ R, float* __restrict__ Cs, int B, int T, int h, int step, int WS) {
  extern __shared__ __align__(16) float lds[];
  float* Wl = lds;                              // [16][WS]   rows (u,g) -> W_hh[d][g*h + u0+u][:]
  float* Hl = lds + 16 * WS;                    // [2][128][HS]
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int uslices = (h + 3) / 4;
  const int d = blockIdx.x / uslices, u0 = (blockIdx.x % uslices) * 4;
  const int tt = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? tt - 1 : tt + 1;      // time index of h_{t-1}
  const bool first = step == 0;
  const int jb = lane & 15, ku = lane >> 4;     // batch row within the tile / unit (and k phase)

  if (!first) {
    const float* Wd = Whh + (size_t)d * 4 * h * h;
    const int wc4 = (WS - 4) / 4;                // padded row length (multiple of 64 columns) in float4
    for (int idx = tid; idx < 16 * wc4; idx += kThreads) {
      const int row = idx / wc4, k = (idx % wc4) * 4;
      const int u = row >> 2, g = row & 3;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero beyond h: the MFMA loop runs over whole chunks
      if (u0 + u < h && k < h) v = *reinterpret_cast<const float4*>(Wd + (size_t)(g * h + u0 + u) * h + k);
      *reinterpret_cast<float4*>(Wl + row * WS + k) = v;
    }
  }

  for (int b0 = 0; b0 < B; b0 += 128) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!first) {
      const int nchunks = (h + KC - 1) / KC;
      // chunk loader: 128 rows x 64 cols = 2048 float4, 4 per thread
      float4 stage[4];
      auto gload = [&](int c) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = tid + q * kThreads;
          const int r = idx >> 4, k = c * KC + (idx & 15) * 4;
          stage[q] = (b0 + r < B && k < h) ? *reinterpret_cast<const float4*>(out + ((size_t)(b0 + r) * T + tp) * 2 * h + d * h + k)
                                           : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto lstore = [&](int buf) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int idx = tid + q * kThreads;
          *reinterpret_cast<float4*>(Hl + (buf * 128 + (idx >> 4)) * HS + (idx & 15) * 4) = stage[q];
        }
      };
      gload(0);
      lstore(0);
      __syncthreads();
      for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) gload(c + 1);
        const float* hrow = Hl + ((c & 1) * 128 + wv * 16 + jb) * HS + 2 * ku;
        const float* wrow = Wl + jb * WS + c * KC + 2 * ku;
#pragma unroll
        for (int j = 0; j < KC / 8; ++j) {
          const float2 a = *reinterpret_cast<const float2*>(wrow + 8 * j);
          const float2 b = *reinterpret_cast<const float2*>(hrow + 8 * j);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
        }
        if (c + 1 < nchunks) lstore((c + 1) & 1);
        __syncthreads();
      }
    }
    // lane-local cell update for (batch b, unit u); accumulator register g = gate (i,f,g,o)
    const int b = b0 + wv * 16 + jb, u = u0 + ku;
    if (b < B && u < h) {
      const float* gx = Gx + (((size_t)b * T + tt) * 2 + d) * 4 * h + u;
      const float pi = acc[0] + gx[0], pf = acc[1] + gx[h], pg = acc[2] + gx[2 * h], po = acc[3] + gx[3 * h];
      const float cprev = first ? 0.f : Cs[(((size_t)tp * 2 + d) * B + b) * h + u];
      const float gi = sigmoid_f(pi), gf = sigmoid_f(pf), gg = tanh_f(pg), go = sigmoid_f(po);
      const float c = fmaf(gf, cprev, gi * gg);
      const float hv = go * tanh_f(c);
      const size_t s = (((size_t)tt * 2 + d) * B + b) * h + u;
      Cs[s] = c;
      *reinterpret_cast<float4*>(R + s * 4) = make_float4(gi, gf, gg, go);
      out[((size_t)b * T + tt) * 2 * h + d * h + u] = hv;
    }
    if (b0 + 128 < B) __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// backward step.  grid = 2 * ceil(h/16) * ceil(B/32); 128 threads (2 waves x 16 batch rows).
// dG layout [B,T,2,4h].  dCn [2][B][h] carries dL/dc across steps (in/out).
// ---------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 256;                // 4 waves = 2 batch tiles x 2 halves of every K chunk
constexpr int KCB = 64;
constexpr int HSB = KCB + 4;

__global__ __launch_bounds__(kBwdThreads) void lstm_bwd_step_kernel(
    const float* __restrict__ WhhT, const float* __restrict__ R, const float* __restrict__ Cs,
    const float* __restrict__ dOut, const float* __restrict__ dHn, float* __restrict__ dG, float* __restrict__ dCn,
    int B, int T, int h, int step, int WS) {
  extern __shared__ __align__(16) float lds[];
  float* Wl = lds;                              // [16 units][WS]: Wl[u][col] = W_hh[d][col][u0+u] = WhhT[d][u0+u][col]
  float* Gl = lds + 16 * WS;                    // [2][32][HSB] chunk of dG_{t+1}
  float* Xl = Gl + 2 * 32 * HSB;                // [2 tiles][64 lanes][4] partial sums of the second K half
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int tile = wv & 1, khalf = wv >> 1;
  const int uslices = (h + 15) / 16, bslices = (B + 31) / 32;
  const int d = blockIdx.x / (uslices * bslices);
  const int rem = blockIdx.x % (uslices * bslices);
  const int u0 = (rem / bslices) * 16, b0 = (rem % bslices) * 32;
  // processing order: the LAST forward step first.  step s handles forward step fs = T-1-s.
  const int fs = T - 1 - step;
  const int tt = d == 0 ? fs : T - 1 - fs;      // time index handled now
  const int tn = d == 0 ? tt + 1 : tt - 1;      // time index of the step processed just before (forward-later)
  const int tp = d == 0 ? tt - 1 : tt + 1;      // forward-earlier neighbour (c_{t-1})
  const bool last = step == 0;                  // no recurrent gradient yet
  const int jb = lane & 15, ku = lane >> 4;
  const int K = 4 * h;

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (!last) {
    const float* Wd = WhhT + (size_t)d * h * K;
    const int wc4 = (WS - 4) / 4;
    for (int idx = tid; idx < 16 * wc4; idx += kBwdThreads) {
      const int u = idx / wc4, k = (idx % wc4) * 4;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (u0 + u < h && k < K) v = *reinterpret_cast<const float4*>(Wd + (size_t)(u0 + u) * K + k);
      *reinterpret_cast<float4*>(Wl + u * WS + k) = v;
    }
    const int nchunks = (K + KCB - 1) / KCB;
    float4 stage[2];
    auto gload = [&](int c) {                   // 32 rows x 64 cols = 512 float4, 2 per thread
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * kBwdThreads;
        const int r = idx >> 4, k = c * KCB + (idx & 15) * 4;
        stage[q] = (b0 + r < B && k < K) ? *reinterpret_cast<const float4*>(dG + (((size_t)(b0 + r) * T + tn) * 2 + d) * K + k)
                                         : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto lstore = [&](int buf) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int idx = tid + q * kBwdThreads;
        *reinterpret_cast<float4*>(Gl + (buf * 32 + (idx >> 4)) * HSB + (idx & 15) * 4) = stage[q];
      }
    };
    gload(0);
    lstore(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
      if (c + 1 < nchunks) gload(c + 1);
      const float* grow = Gl + ((c & 1) * 32 + tile * 16 + jb) * HSB + 2 * ku + khalf * (KCB / 2);
      const float* wrow = Wl + jb * WS + c * KCB + 2 * ku + khalf * (KCB / 2);
#pragma unroll
      for (int j = 0; j < KCB / 16; ++j) {
        const float2 a = *reinterpret_cast<const float2*>(wrow + 8 * j);
        const float2 b = *reinterpret_cast<const float2*>(grow + 8 * j);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc, 0, 0, 0);
      }
      if (c + 1 < nchunks) lstore((c + 1) & 1);
      __syncthreads();
    }
    if (khalf == 1) *reinterpret_cast<float4*>(Xl + (tile * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (khalf == 0) {
      const float4 o = *reinterpret_cast<const float4*>(Xl + (tile * 64 + lane) * 4);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
  }
  // lane: batch b = b0 + tile*16 + jb, units u0 + 4*ku + r (r = accumulator register)
  const int b = b0 + tile * 16 + jb;
  if (khalf == 0 && b < B) {
    float dgate[4][4];                           // [unit r][gate]
    bool ok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int u = u0 + 4 * ku + r;
      ok[r] = u < h;
#pragma unroll
      for (int g = 0; g < 4; ++g) dgate[r][g] = 0.f;
      if (!ok[r]) continue;
      const size_t s = (((size_t)tt * 2 + d) * B + b) * h + u;
      const float4 g4 = *reinterpret_cast<const float4*>(R + s * 4);
      const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
      const float c = Cs[s];
      const bool has_prev = (d == 0) ? (tt > 0) : (tt < T - 1);
      const float cprev = has_prev ? Cs[(((size_t)tp * 2 + d) * B + b) * h + u] : 0.f;
      const float tc = tanh_f(c);
      float dh = dOut[((size_t)b * T + tt) * 2 * h + d * h + u] + acc[r];
      const size_t cs = ((size_t)d * B + b) * h + u;
      float dc = 0.f;
      if (last) { if (dHn) dh += dHn[cs]; } else dc = dCn[cs];
      dc = fmaf(dh * go, 1.f - tc * tc, dc);
      dgate[r][0] = dc * gg * gi * (1.f - gi);
      dgate[r][1] = dc * cprev * gf * (1.f - gf);
      dgate[r][2] = dc * gi * (1.f - gg * gg);
      dgate[r][3] = dh * tc * go * (1.f - go);
      dCn[cs] = dc * gf;
    }
    float* dst = dG + (((size_t)b * T + tt) * 2 + d) * K + u0 + 4 * ku;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (ok[3]) {
        *reinterpret_cast<float4*>(dst + g * h) = make_float4(dgate[0][g], dgate[1][g], dgate[2][g], dgate[3][g]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (ok[r]) dst[g * h + r] = dgate[r][g];
      }
    }
  }
}

}  // namespace
}  // namespace tsg

using namespace tsg;

static int lstm_check(const char* fn, int B, int T, int h, int dtype) {
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (B <= 0 || T <= 0 || h <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d h=%d", fn, B, T, h);
  if (h % 4) return set_error(TSG_E_ALIGN, "%s: hidden size %d must be a multiple of 4", fn, h);
  return 0;
}

extern "C" int tsg_lstm_fwd(const void* Gx, const void* Whh, void* out, void* R, void* Cs,
                            int B, int T, int h, int dtype, void* stream) {
  const char* fn = "tsg_lstm_fwd";
  for (const void* p : {Gx, Whh, (const void*)out, (const void*)R, (const void*)Cs}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = lstm_check(fn, B, T, h, dtype);
  if (rc) return rc;
  const int WS = roundup(h, 64) + 4;                     // = 4 mod 64
  const size_t lds = sizeof(float) * ((size_t)16 * WS + 2 * 128 * HS);
  if (lds > (size_t)kLdsBytes) return set_error(TSG_E_LDS, "%s: h=%d needs %zu B of LDS", fn, h, lds);
  auto kern = lstm_fwd_step_kernel;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  auto st = static_cast<hipStream_t>(stream);
  const int grid = 2 * cdiv(h, 4);
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, (const float*)Gx, (const float*)Whh, (float*)out,
                       (float*)R, (float*)Cs, B, T, h, step, WS);
  return check_launch(fn);
}

extern "C" int tsg_lstm_bwd(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                            void* dG, void* dC_ws, int B, int T, int h, int dtype, void* stream) {
  const char* fn = "tsg_lstm_bwd";
  for (const void* p : {WhhT, R, Cs, dOut, (const void*)dG, (const void*)dC_ws}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = lstm_check(fn, B, T, h, dtype);
  if (rc) return rc;
  const int WS = roundup(4 * h, 64) + 4;
  const size_t lds = sizeof(float) * ((size_t)16 * WS + 2 * 32 * HSB + 2 * 64 * 4);
  if (lds > (size_t)kLdsBytes) return set_error(TSG_E_LDS, "%s: h=%d needs %zu B of LDS", fn, h, lds);
  auto kern = lstm_bwd_step_kernel;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  auto st = static_cast<hipStream_t>(stream);
  const int grid = 2 * cdiv(h, 16) * cdiv(B, 32);
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), lds, st, (const float*)WhhT, (const float*)R, (const float*)Cs,
                       (const float*)dOut, (const float*)dHn, (float*)dG, (float*)dC_ws, B, T, h, step, WS);
  return check_launch(fn);
}
