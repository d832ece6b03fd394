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
constexpr int KC = 32;                          // h_{t-1} columns per LDS chunk
constexpr int HS = KC + 4;                      // chunk row stride (floats), = 4 mod 32: conflict-free b64 reads
constexpr int NBUF = 6;                         // LDS ring depth
constexpr int PF = 4;                           // chunks in flight (global -> registers) ahead of the MFMAs

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.f + fast_exp2(-x * kLog2e)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = fast_exp2(clampf(x, -44.f, 44.f) * k2Log2e);
  return 1.f - 2.f * fast_rcp(e + 1.f);
}

// ---------------------------------------------------------------------------------------------
// forward step.  grid = 2 * ceil(h/4).  tt = time index of this step for direction d.
// Latency, not bandwidth, bounds a step: everything the step needs is requested up front (W_hh
// rows, the first PF chunks of h_{t-1}, the Gx / c_{t-1} operands of the cell update) and the K
// loop keeps PF chunks in flight through a 6-deep LDS ring.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void lstm_fwd_step_kernel(
    const float* __restrict__ Gx, const float* __restrict__ Whh, float* __restrict__ out,
    float* __restrict__ R, float* __restrict__ Cs, int B, int T, int h, int step, int WS) {
  extern __shared__ __align__(16) float lds[];
  float* Wl = lds;                              // [16][WS]   rows (u,g) -> W_hh[d][g*h + u0+u][:]
  float* Hl = lds + 16 * WS;                    // [NBUF][128][HS]
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int uslices = (h + 3) / 4;
  const int d = blockIdx.x / uslices, u0 = (blockIdx.x % uslices) * 4;
  const int tt = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? tt - 1 : tt + 1;      // time index of h_{t-1}
  const bool first = step == 0;
  const int jb = lane & 15, ku = lane >> 4;     // batch row within the tile / unit (and k phase)
  const int nchunks = (h + KC - 1) / KC;
  // chunk loader: 128 rows x 32 cols = 1024 float4, 2 per thread
  const int lr = tid >> 3, lc = (tid & 7) * 4;  // row / column of this thread's float4 (second one: row + 64)

  for (int b0 = 0; b0 < B; b0 += 128) {
    const int b = b0 + wv * 16 + jb, u = u0 + ku;
    const bool live = b < B && u < h;
    // cell-update operands: requested now, used after the MFMAs
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (live) {
      const float* g = Gx + (((size_t)b * T + tt) * 2 + d) * 4 * h + u;
      gx[0] = g[0]; gx[1] = g[h]; gx[2] = g[2 * h]; gx[3] = g[3 * h];
      if (!first) cprev = Cs[(((size_t)tp * 2 + d) * B + b) * h + u];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!first) {
      float4 stage[PF][2];
      auto gload = [&](int c, float4 (&st)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = lr + 64 * q, k = c * KC + lc;
          st[q] = (b0 + r < B && k < h) ? *reinterpret_cast<const float4*>(out + ((size_t)(b0 + r) * T + tp) * 2 * h + d * h + k)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto lstore = [&](int buf, const float4 (&st)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q)
          *reinterpret_cast<float4*>(Hl + (buf * 128 + lr + 64 * q) * HS + lc) = st[q];
      };
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < nchunks) gload(i, stage[i]);
      if (b0 == 0) {                              // W_hh rows of this workgroup's 4 units (once per launch)
        const float* Wd = Whh + (size_t)d * 4 * h * h;
        const int wc4 = (WS - 4) / 4;             // padded row length (multiple of 64 columns) in float4
        for (int idx = tid; idx < 16 * wc4; idx += kThreads) {
          const int row = idx / wc4, k = (idx % wc4) * 4;
          float4 v = make_float4(0.f, 0.f, 0.f, 0.f);   // zero beyond h: the MFMA loop runs over whole chunks
          if (u0 + (row >> 2) < h && k < h) v = *reinterpret_cast<const float4*>(Wd + (size_t)((row & 3) * h + u0 + (row >> 2)) * h + k);
          *reinterpret_cast<float4*>(Wl + row * WS + k) = v;
        }
      }
      lstore(0, stage[0]);
      __syncthreads();
      // steady state: chunk c is consumed from ring slot c % NBUF; chunk c+1 is written to its slot
      // (its data was requested PF chunks ago); chunk c+PF is requested.  Slot reuse distance NBUF >
      // PF + 1, so one barrier per chunk suffices.
#pragma unroll 1
      for (int c0 = 0; c0 < nchunks; c0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          const int c = c0 + i;
          if (c < nchunks) {
            const float* hrow = Hl + ((c % NBUF) * 128 + wv * 16 + jb) * HS + 2 * ku;
            const float* wrow = Wl + jb * WS + c * KC + 2 * ku;
#pragma unroll
            for (int j = 0; j < KC / 8; ++j) {
              const float2 a = *reinterpret_cast<const float2*>(wrow + 8 * j);
              const float2 bv = *reinterpret_cast<const float2*>(hrow + 8 * j);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv.x, acc, 0, 0, 0);
              acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv.y, acc, 0, 0, 0);
            }
            if (c + 1 < nchunks) lstore((c + 1) % NBUF, stage[(i + 1) % PF]);
            if (c + PF < nchunks) gload(c + PF, stage[i]);
            __syncthreads();
          }
        }
      }
    }
    // lane-local cell update for (batch b, unit u); accumulator register g = gate (i,f,g,o)
    if (live) {
      const float gi = sigmoid_f(acc[0] + gx[0]), gf = sigmoid_f(acc[1] + gx[1]);
      const float gg = tanh_f(acc[2] + gx[2]), go = sigmoid_f(acc[3] + gx[3]);
      const float c = fmaf(gf, cprev, gi * gg);
      const float hv = go * tanh_f(c);
      const size_t s = (((size_t)tt * 2 + d) * B + b) * h + u;
      Cs[s] = c;
      *reinterpret_cast<float4*>(R + s * 4) = make_float4(gi, gf, gg, go);
      out[((size_t)b * T + tt) * 2 * h + d * h + u] = hv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step.  grid = 2 * ceil(h/16) * ceil(B/32); 128 threads (2 waves x 16 batch rows).
// dG layout [B,T,2,4h].  dCn [2][B][h] carries dL/dc across steps (in/out).
// ---------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 256;                // 4 waves = 2 batch tiles x 2 halves of every K chunk
constexpr int KCB = 64;
constexpr int HSB = KCB + 4;
constexpr int NBUFB = 6;
constexpr int PFB = 4;
constexpr int kStageRows = 32 + 16;             // one ring slot: 32 rows of dG_{t+1} + 16 rows of W_hh^T

__global__ __launch_bounds__(kBwdThreads) void lstm_bwd_step_kernel(
    const float* __restrict__ WhhT, const float* __restrict__ R, const float* __restrict__ Cs,
    const float* __restrict__ dOut, const float* __restrict__ dHn, float* __restrict__ dG, float* __restrict__ dCn,
    int B, int T, int h, int step) {
  // ring slot = [48][HSB]: rows 0..31 = dG_{t+1}[b0+r][chunk], rows 32..47 = WhhT[d][u0+u][chunk].
  // Both operands stream (PFB chunks in flight); nothing is loaded wholesale up front.
  __shared__ __align__(16) float ring[NBUFB * kStageRows * HSB];
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
  const int b = b0 + tile * 16 + jb;            // lane: batch b, units u0 + 4*ku + r (r = accumulator register)
  const bool has_prev = (d == 0) ? (tt > 0) : (tt < T - 1);

  // cell-backward operands of this lane's 4 (b, unit) pairs: requested now, used after the MFMAs
  float4 g4[4]; float cc[4], cpv[4], dov[4], dcv[4]; bool ok[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int u = u0 + 4 * ku + r;
    ok[r] = khalf == 0 && b < B && u < h;
    g4[r] = make_float4(0.f, 0.f, 0.f, 0.f); cc[r] = cpv[r] = dov[r] = dcv[r] = 0.f;
    if (ok[r]) {
      const size_t s = (((size_t)tt * 2 + d) * B + b) * h + u;
      const size_t cs = ((size_t)d * B + b) * h + u;
      g4[r] = *reinterpret_cast<const float4*>(R + s * 4);
      cc[r] = Cs[s];
      if (has_prev) cpv[r] = Cs[(((size_t)tp * 2 + d) * B + b) * h + u];
      dov[r] = dOut[((size_t)b * T + tt) * 2 * h + d * h + u];
      if (last) { if (dHn) dov[r] += dHn[cs]; } else dcv[r] = dCn[cs];
    }
  }

  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  if (!last) {
    const float* Wd = WhhT + (size_t)d * h * K;
    const int nchunks = (K + KCB - 1) / KCB;
    // loader: 48 rows x 64 cols = 768 float4, 3 per thread: thread -> (row = tid/16 + 16q, col = (tid%16)*4)
    const int lr = tid >> 4, lc = (tid & 15) * 4;
    float4 stage[PFB][3];
    auto gload = [&](int c, float4 (&st)[3]) {
      const int k = c * KCB + lc;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = lr + 16 * q;
        st[q] = (b0 + r < B && k < K) ? *reinterpret_cast<const float4*>(dG + (((size_t)(b0 + r) * T + tn) * 2 + d) * K + k)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
      st[2] = (u0 + lr < h && k < K) ? *reinterpret_cast<const float4*>(Wd + (size_t)(u0 + lr) * K + k)
                                     : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto lstore = [&](int buf, const float4 (&st)[3]) {
      float* base = ring + buf * kStageRows * HSB;
#pragma unroll
      for (int q = 0; q < 3; ++q) *reinterpret_cast<float4*>(base + (lr + 16 * q) * HSB + lc) = st[q];
    };
#pragma unroll
    for (int i = 0; i < PFB; ++i)
      if (i < nchunks) gload(i, stage[i]);
    lstore(0, stage[0]);
    __syncthreads();
#pragma unroll 1
    for (int c0 = 0; c0 < nchunks; c0 += PFB) {
#pragma unroll
      for (int i = 0; i < PFB; ++i) {
        const int c = c0 + i;
        if (c < nchunks) {
          const float* base = ring + (c % NBUFB) * kStageRows * HSB;
          const float* grow = base + (tile * 16 + jb) * HSB + 2 * ku + khalf * (KCB / 2);
          const float* wrow = base + (32 + jb) * HSB + 2 * ku + khalf * (KCB / 2);
#pragma unroll
          for (int j = 0; j < KCB / 16; ++j) {
            const float2 a = *reinterpret_cast<const float2*>(wrow + 8 * j);
            const float2 bv = *reinterpret_cast<const float2*>(grow + 8 * j);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bv.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bv.y, acc, 0, 0, 0);
          }
          if (c + 1 < nchunks) lstore((c + 1) % NBUFB, stage[(i + 1) % PFB]);
          if (c + PFB < nchunks) gload(c + PFB, stage[i]);
          __syncthreads();
        }
      }
    }
    // fold the two K halves (the ring is free now)
    float* Xl = ring;
    if (khalf == 1) *reinterpret_cast<float4*>(Xl + (tile * 64 + lane) * 4) = make_float4(acc[0], acc[1], acc[2], acc[3]);
    __syncthreads();
    if (khalf == 0) {
      const float4 o = *reinterpret_cast<const float4*>(Xl + (tile * 64 + lane) * 4);
      acc[0] += o.x; acc[1] += o.y; acc[2] += o.z; acc[3] += o.w;
    }
  }
  if (khalf == 0 && b < B) {
    float dgate[4][4];                           // [unit r][gate]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int g = 0; g < 4; ++g) dgate[r][g] = 0.f;
      if (!ok[r]) continue;
      const float gi = g4[r].x, gf = g4[r].y, gg = g4[r].z, go = g4[r].w;
      const float tc = tanh_f(cc[r]);
      const float dh = dov[r] + acc[r];
      const float dc = fmaf(dh * go, 1.f - tc * tc, dcv[r]);
      dgate[r][0] = dc * gg * gi * (1.f - gi);
      dgate[r][1] = dc * cpv[r] * gf * (1.f - gf);
      dgate[r][2] = dc * gi * (1.f - gg * gg);
      dgate[r][3] = dh * tc * go * (1.f - go);
      dCn[((size_t)d * B + b) * h + u0 + 4 * ku + r] = dc * gf;
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
  const size_t lds = sizeof(float) * ((size_t)16 * WS + (size_t)NBUF * 128 * HS);
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
  auto kern = lstm_bwd_step_kernel;
  auto st = static_cast<hipStream_t>(stream);
  const int grid = 2 * cdiv(h, 16) * cdiv(B, 32);
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kBwdThreads), 0, st, (const float*)WhhT, (const float*)R, (const float*)Cs,
                       (const float*)dOut, (const float*)dHn, (float*)dG, (float*)dC_ws, B, T, h, step);
  return check_launch(fn);
}
