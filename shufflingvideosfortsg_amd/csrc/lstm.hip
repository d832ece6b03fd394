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
// dG_t [T,B,2,4h] (consumed by the next step and, afterwards, by the caller's weight-gradient GEMMs).
//
// All sequence tensors are TIME-MAJOR (Gx [T,B,2,4h], out [T,B,2h], dOut, dG): one step touches one
// contiguous slab.  (Batch-first layouts put the rows of a step 0.5-2 MiB apart -- 128 pages per
// workgroup and step -- and the step time was dominated by address-translation misses.)
// Saved for backward (caller-owned): R [T][2][B][h][4] activated gates, Cs [T][2][B][h] cell states.
#include "tsg_common.h"
#include <map>
#include <mutex>
#include <utility>
#include <cstdlib>

namespace tsg {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// row of (time t, batch item b) in the caller's sequence tensors (Gx, out, dOut, dG): time-major [T,B,..] or, with bm != 0,
// batch-major [B,T,..] -- the layout of the model's activations, so that no transposed copy is needed around the recurrence
__device__ __forceinline__ size_t seq_row(int t, int b, int B, int T, int bm) { return bm ? (size_t)b * T + t : (size_t)t * B + b; }
constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;       // 8 waves x 16 batch rows = 128 rows per pass
constexpr int KC = 32;                          // operand columns per LDS chunk
constexpr int HS = KC + 4;                      // chunk row stride (floats), = 4 mod 32: conflict-free b64 reads
constexpr int NBUF = 3;                         // per-wave LDS ring depth (chunks wait in registers, not in LDS)
constexpr int PF = 4;                           // chunks in flight (global -> registers); 16 (= all of K at h=512) measured no faster

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ float sigmoid_f(float x) { return fast_rcp(1.f + fast_exp2(-x * kLog2e)); }
__device__ __forceinline__ float tanh_f(float x) {
  const float e = fast_exp2(clampf(x, -44.f, 44.f) * k2Log2e);
  return 1.f - 2.f * fast_rcp(e + 1.f);
}

// ---------------------------------------------------------------------------------------------
// forward step.  grid = 2 * ceil(h/4).  tt = time index of this step for direction d.
// A step is latency-bound (few KB per wave, ~100 MFMAs): everything is requested up front, the W_hh
// rows are staged with ONE batched round trip and one barrier, and after that the waves never meet
// again -- each wave streams the h_{t-1} rows of ITS 16 batch items through a private LDS ring
// (PF chunks in flight), so the K loop has no workgroup barrier.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void lstm_fwd_step_kernel(
    const float* __restrict__ Gx, const float* __restrict__ bias, const float* __restrict__ Whh, float* __restrict__ out,
    float* __restrict__ R, float* __restrict__ Cs, int B, int T, int h, int step, int WS, int bm) {
  extern __shared__ __align__(16) float lds[];
  float* Wl = lds;                              // [16][WS]   rows (u,g) -> W_hh[d][g*h + u0+u][:]
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  float* Hl = lds + 16 * WS + wv * (NBUF * 16 * HS);   // this wave's ring: [NBUF][16][HS]
  const int uslices = (h + 3) / 4;
  const int d = blockIdx.x / uslices, u0 = (blockIdx.x % uslices) * 4;
  const int tt = d == 0 ? step : T - 1 - step;
  const int tp = d == 0 ? tt - 1 : tt + 1;      // time index of h_{t-1}
  const bool first = step == 0;
  const int jb = lane & 15, ku = lane >> 4;     // batch row within the tile / unit (and k phase)
  const int nchunks = (h + KC - 1) / KC;
  const int lr = lane >> 3, lc = (lane & 7) * 4;  // chunk loader: 16 rows x 8 float4; lane -> rows lr, lr+8

  if (!first) {                                 // W_hh rows of this workgroup's 4 units: one batched round trip
    const float* Wd = Whh + (size_t)d * 4 * h * h;
    const int wc4 = (WS - 4) / 4;               // padded row length (multiple of 64 columns) in float4
    constexpr int WB = 5;                       // 16 rows * 129 float4 (h = 512) / 512 threads, rounded up
    for (int base = tid; base < 16 * wc4; base += WB * kThreads) {
      float4 v[WB];
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int idx = base + i * kThreads, row = idx / wc4, k = (idx % wc4) * 4;
        v[i] = (idx < 16 * wc4 && u0 + (row >> 2) < h && k < h)
                   ? *reinterpret_cast<const float4*>(Wd + (size_t)((row & 3) * h + u0 + (row >> 2)) * h + k)
                   : make_float4(0.f, 0.f, 0.f, 0.f);   // zero beyond h: the MFMA loop runs over whole chunks
      }
#pragma unroll
      for (int i = 0; i < WB; ++i) {
        const int idx = base + i * kThreads;
        if (idx < 16 * wc4) *reinterpret_cast<float4*>(Wl + (idx / wc4) * WS + (idx % wc4) * 4) = v[i];
      }
    }
  }

  for (int b0 = 0; b0 < B; b0 += 128) {
    const int bt = b0 + wv * 16;                 // first batch row of this wave's tile
    const int b = bt + jb, u = u0 + ku;
    const bool live = b < B && u < h;
    // cell-update operands: requested now, used after the MFMAs
    float gx[4] = {0.f, 0.f, 0.f, 0.f}, cprev = 0.f;
    if (live) {
      const float* g = Gx + (seq_row(tt, b, B, T, bm) * 2 + d) * 4 * h + u;
      gx[0] = g[0]; gx[1] = g[h]; gx[2] = g[2 * h]; gx[3] = g[3 * h];
      if (bias) {                                 // b_ih + b_hh not folded into Gx by the caller's GEMM
        const float* bb = bias + (size_t)d * 4 * h + u;
        gx[0] += bb[0]; gx[1] += bb[h]; gx[2] += bb[2 * h]; gx[3] += bb[3 * h];
      }
      if (!first) cprev = Cs[(((size_t)tp * 2 + d) * B + b) * h + u];
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (!first && bt < B) {                      // wave-uniform
      float4 stage[PF][2];
      auto gload = [&](int c, float4 (&st)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const int r = lr + 8 * q, k = c * KC + lc;
          st[q] = (bt + r < B && k < h) ? *reinterpret_cast<const float4*>(out + seq_row(tp, bt + r, B, T, bm) * 2 * h + d * h + k)
                                        : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      };
      auto lstore = [&](int buf, const float4 (&st)[2]) {
#pragma unroll
        for (int q = 0; q < 2; ++q) *reinterpret_cast<float4*>(Hl + (buf * 16 + lr + 8 * q) * HS + lc) = st[q];
      };
#pragma unroll
      for (int i = 0; i < PF; ++i)
        if (i < nchunks) gload(i, stage[i]);
      lstore(0, stage[0]);
      if (b0 == 0) __syncthreads();              // W_hh rows visible (the only workgroup barrier)
#pragma unroll 1
      for (int c0 = 0; c0 < nchunks; c0 += PF) {
#pragma unroll
        for (int i = 0; i < PF; ++i) {
          const int c = c0 + i;
          if (c < nchunks) {
            __builtin_amdgcn_wave_barrier();
            const float* hrow = Hl + ((c % NBUF) * 16 + jb) * HS + 2 * ku;
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
          }
        }
      }
    } else if (!first && b0 == 0) {
      __syncthreads();                           // idle waves still take part in the W barrier
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
      out[seq_row(tt, b, B, T, bm) * 2 * h + d * h + u] = hv;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward step.  grid = 2 * ceil(h/16) * ceil(B/32); 256 threads = 4 waves = the four quarters of
// the K = 4h contraction; each wave carries BOTH 16-row batch tiles of the workgroup (two independent
// MFMA chains) and streams its own operands -- 32 rows of dG_{t+1} and 16 rows of W_hh^T, its K
// quarter only -- through a private LDS ring, so the K loop has no workgroup barrier; the four
// partial sums meet once in LDS.   dG layout [T,B,2,4h].  dCn [2][B][h] carries dL/dc across steps.
// ---------------------------------------------------------------------------------------------
constexpr int kBwdThreads = 256;
constexpr int NBUFB = 5;
constexpr int PFB = 3;
constexpr int kStageRows = 32 + 16;             // one ring slot: 32 rows of dG_{t+1} + 16 rows of W_hh^T
constexpr int HSB = KC + 8;                     // slot row stride: = 8 mod 32 -> the 16-lane groups of a ds_read_b128 (rows x 4 k-quads) hit 64 distinct banks

__global__ __launch_bounds__(kBwdThreads) void lstm_bwd_step_kernel(
    const float* __restrict__ WhhT, const float* __restrict__ R, const float* __restrict__ Cs,
    const float* __restrict__ dOut, const float* __restrict__ dHn, float* __restrict__ dG, float* __restrict__ dCn,
    int B, int T, int h, int step, int bm) {
  __shared__ __align__(16) float ring_all[4 * NBUFB * kStageRows * HSB];
  const int tid = threadIdx.x, lane = tid & 63, kq = wave_id();
  float* ring = ring_all + kq * (NBUFB * kStageRows * HSB);
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
  const bool has_prev = (d == 0) ? (tt > 0) : (tt < T - 1);
  // after the fold, wave kq finishes rows: tile = kq & 1, units u0 + 4*ku + r for r in {2*(kq>>1), +1}
  const int tile = kq & 1, rsel = (kq >> 1) * 2;
  const int b = b0 + tile * 16 + jb;

  // cell-backward operands of this lane's 2 (b, unit) pairs: requested now, used after the MFMAs
  float4 g4[2]; float cc[2], cpv[2], dov[2], dcv[2]; bool ok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int u = u0 + 4 * ku + rsel + i;
    ok[i] = b < B && u < h;
    g4[i] = make_float4(0.f, 0.f, 0.f, 0.f); cc[i] = cpv[i] = dov[i] = dcv[i] = 0.f;
    if (ok[i]) {
      const size_t s = (((size_t)tt * 2 + d) * B + b) * h + u;
      const size_t cs = ((size_t)d * B + b) * h + u;
      g4[i] = *reinterpret_cast<const float4*>(R + s * 4);
      cc[i] = Cs[s];
      if (has_prev) cpv[i] = Cs[(((size_t)tp * 2 + d) * B + b) * h + u];
      dov[i] = dOut[seq_row(tt, b, B, T, bm) * 2 * h + d * h + u];
      if (last) { if (dHn) dov[i] += dHn[cs]; } else dcv[i] = dCn[cs];
    }
  }

  f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};      // batch tile 0 / 1
  if (!last) {
    const float* Wd = WhhT + (size_t)d * h * K;
    const int kchunks = (K + KC - 1) / KC;
    const int per = (kchunks + 3) / 4, cbeg = kq * per, cend = (cbeg + per < kchunks) ? cbeg + per : kchunks;
    const int lr = lane >> 3, lc = (lane & 7) * 4;           // 8 rows x 8 float4 per pass; 6 passes = 48 rows
    float4 stage[PFB][6];
    auto gload = [&](int c, float4 (&st)[6]) {
      const int k = c * KC + lc;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = lr + 8 * q;
        st[q] = (b0 + r < B && k < K) ? *reinterpret_cast<const float4*>(dG + (seq_row(tn, b0 + r, B, T, bm) * 2 + d) * K + k)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int r = lr + 8 * q;
        st[4 + q] = (u0 + r < h && k < K) ? *reinterpret_cast<const float4*>(Wd + (size_t)(u0 + r) * K + k)
                                          : make_float4(0.f, 0.f, 0.f, 0.f);
      }
    };
    auto lstore = [&](int buf, const float4 (&st)[6]) {
      float* base = ring + buf * kStageRows * HSB;
#pragma unroll
      for (int q = 0; q < 6; ++q) *reinterpret_cast<float4*>(base + (lr + 8 * q) * HSB + lc) = st[q];
    };
#pragma unroll
    for (int i = 0; i < PFB; ++i)
      if (cbeg + i < cend) gload(cbeg + i, stage[i]);
    if (cbeg < cend) lstore(0, stage[0]);
#pragma unroll 1
    for (int c0 = cbeg; c0 < cend; c0 += PFB) {
#pragma unroll
      for (int i = 0; i < PFB; ++i) {
        const int c = c0 + i;
        if (c < cend) {
          __builtin_amdgcn_wave_barrier();
          const float* base = ring + ((c - cbeg) % NBUFB) * kStageRows * HSB;
          // one ds_read_b128 per row and 16 columns: lane (row jb, quad ku) holds columns 16j + 4ku .. +3 = FOUR MFMAs' worth
          // of k (component m = k 16j + 4ku + m, the same for both operands).  Paired-k float2 reads are merged by hipcc into
          // ds_read2_b64, which banks mod 32 in 16-lane groups: 2-way conflicts at any stride = 4 mod 32.
          const float* g0 = base + jb * HSB + 4 * ku;
          const float* g1 = base + (16 + jb) * HSB + 4 * ku;
          const float* wrow = base + (32 + jb) * HSB + 4 * ku;
#pragma unroll
          for (int j = 0; j < KC / 16; ++j) {
            const float4 a4 = *reinterpret_cast<const float4*>(wrow + 16 * j);
            const float4 x4 = *reinterpret_cast<const float4*>(g0 + 16 * j);
            const float4 y4 = *reinterpret_cast<const float4*>(g1 + 16 * j);
            const float av[4] = {a4.x, a4.y, a4.z, a4.w}, xv[4] = {x4.x, x4.y, x4.z, x4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], xv[m], acc0, 0, 0, 0);
              acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], yv[m], acc1, 0, 0, 0);
            }
          }
          if (c + 1 < cend) lstore((c + 1 - cbeg) % NBUFB, stage[(i + 1) % PFB]);
          if (c + PFB < cend) gload(c + PFB, stage[i]);
        }
      }
    }
    // fold the four K quarters: Xl[kq][tile][lane][4]; wave kq then finishes (tile kq&1, registers rsel, rsel+1)
    __syncthreads();
    float* Xl = ring_all;
    *reinterpret_cast<float4*>(Xl + ((kq * 2 + 0) * 64 + lane) * 4) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
    *reinterpret_cast<float4*>(Xl + ((kq * 2 + 1) * 64 + lane) * 4) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
    __syncthreads();
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      s0 += Xl[((q * 2 + tile) * 64 + lane) * 4 + rsel];
      s1 += Xl[((q * 2 + tile) * 64 + lane) * 4 + rsel + 1];
    }
    acc0[0] = s0; acc0[1] = s1;
  }
  if (b < B) {
    const float rec[2] = {last ? 0.f : acc0[0], last ? 0.f : acc0[1]};
    float dgate[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int g = 0; g < 4; ++g) dgate[i][g] = 0.f;
      if (!ok[i]) continue;
      const float gi = g4[i].x, gf = g4[i].y, gg = g4[i].z, go = g4[i].w;
      const float tc = tanh_f(cc[i]);
      const float dh = dov[i] + rec[i];
      const float dc = fmaf(dh * go, 1.f - tc * tc, dcv[i]);
      dgate[i][0] = dc * gg * gi * (1.f - gi);
      dgate[i][1] = dc * cpv[i] * gf * (1.f - gf);
      dgate[i][2] = dc * gi * (1.f - gg * gg);
      dgate[i][3] = dh * tc * go * (1.f - go);
      dCn[((size_t)d * B + b) * h + u0 + 4 * ku + rsel + i] = dc * gf;
    }
    float* dst = dG + (seq_row(tt, b, B, T, bm) * 2 + d) * K + u0 + 4 * ku + rsel;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (ok[1]) *reinterpret_cast<float2*>(dst + g * h) = make_float2(dgate[0][g], dgate[1][g]);
      else if (ok[0]) dst[g * h] = dgate[0][g];
    }
  }
}

// ---------------------------------------------------------------------------------------------
// forward, PERSISTENT variant: one launch runs all T steps.  A per-step launch costs ~2.6 us of
// dispatch plus a memory round trip for W_hh before the first MFMA; here the recurrent weights never
// move.  Workgroup = (direction, 32 hidden units = 128 gate rows, 16 batch rows), 8 waves = 8 A-tiles
// (4 units each) x one batch tile.  Each wave keeps ITS A-tile of W_hh (16 gate rows x h) in registers as MFMA
// operand fragments (h/4 VGPRs, 128 at h = 512) and its lanes keep the cell state c of their (batch, unit)
// pair; per step only h_{t-1} of the 16 batch rows (32 KiB at h = 512) is fetched, staged in LDS (row stride
// = 8 mod 64) and read back as ds_read_b128 B operands several reads ahead of the MFMAs that consume them.
//
// Hand-off between workgroups: the data is the flag.  The 32-row slab of h_t a workgroup needs is produced by the
// h/16 workgroups with the same (direction, batch slice).  At launch every lane marks the elements of `out` it
// will produce with a sentinel (a signalling-NaN bit pattern h = o*tanh(c) can never take; `sc1` write-through
// stores) and the grid meets ONCE on an arrival counter; after that a consumer simply polls the slab itself with
// `sc1` loads (per-CU L1 bypassed) until none of its elements is the sentinel -- one memory round trip per step
// and no atomics, fences or store drains (the first version: store-drain -> atomic add -> counter poll -> slab
// load, measured 17 us/step like the launch-per-step kernels; this one 16 us at [128,128,512], 13.4 us with the
// waiting removed = its load + MFMA + gate floor).  Every element is checked, so partially landed lines are
// harmless.  Results do not depend on placement; all workgroups must be co-resident (checked on the host with
// the occupancy query, 1 workgroup per CU); every spin is bounded and raises the error word instead of hanging.
// The poll loads are issued unconditionally and consumed by the s_waitcnt asm directly: a conditional inline-asm
// load makes the compiler merge (copy) its destination registers before the data has landed.
// ---------------------------------------------------------------------------------------------
constexpr int kPersistMaxH = 512;
constexpr int kSpinLimit = 1 << 22;
constexpr int kSlabV = 16 * (kPersistMaxH / 4) / kThreads;   // float4 of the 16-row slab per thread (4 at h = 512)
constexpr unsigned kSentinel = 0x7fa5c3e1u;     // a signalling-NaN bit pattern: never the value of h = o * tanh(c)

__device__ __forceinline__ f32x4 load_sc1_x4(const float* p) {
  f32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
__device__ __forceinline__ void store_sc1(float* p, float v) {
  asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ u32x4 load_sc1_u4(const float* p) {           // integer typed: the sentinel is a NaN pattern
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
  return v;
}
// a wave-uniform pointer the compiler may hold in VGPRs (its divergence analysis is conservative): forced into an SGPR pair
__device__ __forceinline__ const unsigned* uniform_ptr(const unsigned* p) {
  const unsigned long long a = reinterpret_cast<unsigned long long>(p);
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)a), hi = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32));
  return reinterpret_cast<const unsigned*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ u32x4 load_sc1_u4_s(const unsigned* sbase, unsigned voff) {   // scalar base + 32-bit lane offset (bytes): one VGPR per address
  u32x4 v;
  asm volatile("global_load_dwordx4 %0, %1, %2 sc1" : "=v"(v) : "v"(voff), "s"(sbase) : "memory");
  return v;
}
__device__ __forceinline__ void store_sc1_u4(float* p, u32x4 v) {
  asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_sc1_u(float* p, unsigned v) {
  asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}

// Exchange stores.  Each XCD has its own L2, and an agent-scope (sc1) load is served by the L2 of the XCD it is issued on
// when the line was written there.  When every producer and consumer of an exchange group runs on ONE XCD the h / partial
// tiles can therefore stay in that L2 (plain stores: 4.6 vs 6.7 us per backward step, 4.5 vs 4.9 forward); a consumer on
// another XCD would never see them, so the fast path is taken only when the group has VERIFIED its placement: every
// workgroup ORs 1 << HW_REG_XCC_ID into its group's mask word before the start barrier and reads the mask after it
// (one bit set = co-located).  Otherwise the stores are write-through as before.  The workgroup -> role mapping below puts
// the groups on one XCD each under the usual round-robin dispatch (workgroup i -> XCD i % 8) whenever their size divides
// the per-XCD share; correctness never depends on that.
__device__ __forceinline__ void store_x(float* p, float v, bool local) {
  if (local) asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dword %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ void store_x_u4(float* p, u32x4 v, bool local) {
  if (local) asm volatile("global_store_dwordx4 %0, %1, off" : : "v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ int xcd_major_index() {          // position of this workgroup when the grid is ordered by (XCD, arrival)
  const int g = gridDim.x, x = blockIdx.x & 7, s = blockIdx.x >> 3;
  return x * (g >> 3) + min(x, g & 7) + s;
}
__device__ __forceinline__ unsigned xcc_id() {
  unsigned x;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
  return x & 15u;
}
constexpr int kSyncGroupWord = 16;               // sync[16 + group] = XCD mask of exchange group (direction, batch slice)
constexpr int kSyncBytes = TSG_LSTM_SYNC_BYTES;  // error word, arrival counter, debug / timing words, <= 256 group masks
static_assert(kSyncBytes >= 4 * (kSyncGroupWord + 256), "group masks fit the sync workspace");
// Workgroup barrier for the LDS hand-offs inside the step loops: release / acquire on the LDS address space only, so only
// this wave's LDS operations are waited for (s_waitcnt lgkmcnt(0)).  __syncthreads() also fences GLOBAL memory (vmcnt(0) in
// front of the barrier), which would put every load requested for the next step on the step's critical path.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// LDS-DMA of one 16-byte piece per lane (asm: hipcc neither counts it nor drains it in front of LDS reads).  lds_addr: wave-uniform LDS byte
// address of lane 0's piece; the hardware adds 16 bytes per lane.  M0 is the compiler's: saved and restored inside the statement.
template <bool NT = false> __device__ __forceinline__ void dma16(const void* src, unsigned lds_addr) {
  unsigned keep;
  if constexpr (NT)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(src), "s"(lds_addr) : "memory");
}
// The launch's error word, read inside the poll loops (every 32nd retry).  Load and wait in ONE asm statement: as a compiler-
// visible load its destination register stayed "pending" for the waitcnt pass across the loop, which then put a
// conservative s_waitcnt vmcnt(0) in front of the first MFMA that reused the register -- a wait for the loads requested
// for the next step, in the middle of the step.
__device__ __forceinline__ unsigned err_word(const unsigned* sync) {
  unsigned v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(sync) : "memory");
  return v;
}
// A bounded wait expired: raise the launch's error word (what the other workgroups look at) and, when the caller registered
// one (tsg_lstm_error_sink), the process-wide sink -- host-mapped memory the host can read without synchronising, so a
// failed launch is reported by the next call instead of silently leaving invalid results.
__device__ __forceinline__ void raise_error(unsigned* sync, ErrSink esink) {
  __hip_atomic_store(sync, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  report_expiry(esink);
}
// start barrier of the persistent kernels: publish this workgroup's XCD in its group's mask, meet the grid once, and
// report whether the whole group sits on one XCD.  false + error word set when the bounded wait expires.
__device__ __forceinline__ bool grid_start(unsigned* sync, int group, int l2x, ErrSink esink, unsigned nactive) {
  if (threadIdx.x == 0) {
    const unsigned old = __hip_atomic_fetch_or(sync + kSyncGroupWord + group, 1u << xcc_id(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" : : "v"(old) : "memory");       // the mask update has been performed before this workgroup counts as arrived
    __hip_atomic_fetch_add(sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    int spins = 0;
    while (__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nactive) {
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kSpinLimit || __hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
        raise_error(sync, esink);
        break;
      }
    }
  }
  __syncthreads();
  const unsigned mask = __hip_atomic_load(sync + kSyncGroupWord + group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const bool local = l2x != 0 && __builtin_popcount(__builtin_amdgcn_readfirstlane(mask)) == 1;
  if (local && threadIdx.x == 0) __hip_atomic_fetch_add(sync + 3, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sync[3] = workgroups on the L2-local path
  return local;
}

// Split-precision arithmetic of the recurrence (dtype TSG_F32S): both operands of the step's product are written as
// hi + lo with hi = rne_bf16(x), lo = rne_bf16(x - hi), and  W h ~= W_hi h_hi + W_hi h_lo + W_lo h_hi  runs on the bf16
// MFMA (v_mfma_f32_16x16x32_bf16, 16 cycles for 8x the k of the 32-cycle fp32 16x16x4) with fp32 accumulation: 48
// MFMAs of 16 cycles per wave and step at h = 512 instead of 128 of 32.  Only the lo*lo term (2^-18 relative) is
// dropped -- the same arithmetic as the split-precision GEMMs around the recurrence (split_bf16.hip).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // (rne(a), rne(b)) packed, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(a, b);
  lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split8(const f32x4 x0, const f32x4 x1, u32x4& hi, u32x4& lo) {
  unsigned h[4], l[4];
  split_pair(x0[0], x0[1], h[0], l[0]); split_pair(x0[2], x0[3], h[1], l[1]);
  split_pair(x1[0], x1[1], h[2], l[2]); split_pair(x1[2], x1[3], h[3], l[3]);
  hi = (u32x4){h[0], h[1], h[2], h[3]};
  lo = (u32x4){l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x4 mfma_bf16(u32x4 a, u32x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
constexpr int kHLB = kPersistMaxH / 2 + 8;       // bf16 slab plane row stride in dwords (= 8 mod 64)
constexpr int kSlabFloats = 2 * 16 * kHLB;       // LDS dwords of the slab region: two bf16 planes (>= the fp32 slab's 16 x 520)

// MODE 0: fp32 storage, fp32 MFMA.  MODE 1 (TSG_F32S): fp32 storage, split-precision bf16 MFMA arithmetic (needs HJ > 0, even).
// MODE 2 (TSG_BF16): bf16 STORAGE of Gx, out and R -- the sequence tensors; Cs, bias, W_hh and the cell state stay fp32 -- and ONE
// bf16 MFMA per k block (W_hh rounded to bf16 once in the prologue; h_t is exchanged, and fed back, as the bf16 value written to
// `out`).  The hand-off is the same data-is-the-flag protocol on 16-bit elements: the sentinel is a bf16 NaN pattern per
// element (0x7FA5: never the rounding of h = o tanh(c)), the slab a workgroup polls per step is 16 KiB instead of 32.
constexpr unsigned kSentinel16 = 0x7fa5u;
typedef bf16_t lstm_bf16;
// Streaming (use-once) operands of the persistent BACKWARD kernel -- R, Cs, dOut in, dG out -- carry the non-temporal hint: they pass through
// the L2s the partial-dh ring lives in (2.1 MB of ring + 0.65 MB of these per XCD and step against 4 MB of L2), and as ordinary lines they pushed
// dirty ring blocks out before the next generation could overwrite them in place: 1333 -> 728 MB written per launch at [128, 128, 512] (PMC),
// 3.91 -> 3.56 us per step (f32s), 3.33 -> 3.04 (bf16 storage); profiles/r4/lstm_bwd_nontemporal_ab_v1.txt.  Not in the forward kernel: there
// the hint on Gx / R / Cs measured 3.90 -> 4.10 us per step in f32s (its exchange is `out` itself, 0.26 MB per XCD and step -- nothing to protect).
// -DTSG_LSTM_NO_NT: plain accesses (A/B).
typedef unsigned u32x2nt __attribute__((ext_vector_type(2)));
#ifndef TSG_LSTM_NO_NT
__device__ __forceinline__ float4 ld4s(const float* p) { const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p)); return make_float4(v[0], v[1], v[2], v[3]); }
__device__ __forceinline__ float4 ld4s(const bf16_t* p) {
  const u32x2nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x2nt*>(p));
  return make_float4(bf16_lo(u[0]), bf16_hi(u[0]), bf16_lo(u[1]), bf16_hi(u[1]));
}
__device__ __forceinline__ float ld1s(const float* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ float ld1s(const bf16_t* p) { return __uint_as_float((unsigned)__builtin_nontemporal_load(&p->bits) << 16); }
__device__ __forceinline__ void st1s(float* p, float v) { __builtin_nontemporal_store(v, p); }
__device__ __forceinline__ void st1s(bf16_t* p, float v) { __builtin_nontemporal_store((unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu), &p->bits); }
#else
template <typename T> __device__ __forceinline__ float4 ld4s(const T* p) { return ld4(p); }
template <typename T> __device__ __forceinline__ float ld1s(const T* p) { return ld1(p); }
template <typename T> __device__ __forceinline__ void st1s(T* p, float v) { st1(p, v); }
#endif
template <int MODE> struct SeqT { typedef float type; };
template <> struct SeqT<2> { typedef lstm_bf16 type; };

// XR (round 5): the hand-off goes through a COMPACT exchange ring instead of `out` itself.  Polling `out` means polling lines that were marked
// once, up front, for all T steps (64 KB per XCD and step: far more than an L2 holds over a launch), so a poll that arrives before the producer's
// store misses the L2 and fetches the stale sentinel line from the memory side -- and `out` rows of one slab lie T * 2h elements apart in the
// model's batch-major layout.  The ring [4 slots][group][16 rows][h] (32 KiB per group and slot, 256 KiB per XCD at [128, ., 512]) never leaves
// the L2 it is exchanged through: every poll, early or late, is an L2 hit.  Slot = step % 4; the sentinel protocol is unchanged, a producer
// re-marks its part of slot (s + 2) % 4 right after its poll of step s (everybody has finished reading that slot's previous contents by then:
// see the loop), and `out` gets an ordinary store off the critical path.  In the split-precision mode the ring carries h already split into
// bf16 (hi, lo) -- one 128-byte line per (row, producer): hi halves then lo halves -- so the consumers copy planes instead of splitting the slab.
template <int HJ, int MODE, int NW, bool XR = false> // HJ = h / 16 when known at compile time (no branch between MFMAs), else 0;
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void lstm_fwd_persist_kernel(    // NW = waves = A-tiles per workgroup (8: 32 units, one workgroup
                                                                        // per CU; 4: 16 units, two independent chains per CU)
    const typename SeqT<MODE>::type* __restrict__ Gx, const float* __restrict__ bias, const float* __restrict__ Whh,
    typename SeqT<MODE>::type* __restrict__ out, typename SeqT<MODE>::type* __restrict__ R, float* __restrict__ Cs,
    unsigned* __restrict__ sync, int B, int Bs, int T, int h, int HLS, int flags, int bm, ErrSink esink, unsigned* __restrict__ xr = nullptr) {
  constexpr bool SPLIT = MODE >= 1, BF = MODE == 2;
  typedef typename SeqT<MODE>::type GT;
  constexpr int NT = 64 * NW, UW = 4 * NW, HTS = UW + 1;  // threads, units per workgroup, h-tile row stride
  constexpr int SV = (BF ? 8 : 16) * (kPersistMaxH / 4) / NT;   // 16-byte pieces of the 16-row slab per thread (fp32: 4 / 8 at h = 512; bf16: half)
  extern __shared__ __align__(16) float Hl[];            // [16][HLS]  h_{t-1} rows of this batch slice (SPLIT: two bf16 planes
  float* Ht = Hl + kSlabFloats;                          // [16][kHLB] dwords), then Ht [16][33] = this step's h tile (16 rows x
  unsigned* Hhi = reinterpret_cast<unsigned*>(Hl);       // 32 units), gathered for whole-line stores
  unsigned* Hlo = Hhi + 16 * kHLB;
  // The streamed tensors -- Gx in, R / Cs out -- go through LDS too (round 5), so that every global access is a 16-byte piece of a full line:
  // as per-lane accesses in the MFMA lane layout (four strided loads and two scattered stores per lane and step) they cost the step ~1 us
  // (profiles/r5/lstm_fwd_w64_streaming_ablation_v1.txt, lstm_fwd_staged_streams_ab_v1.txt).  Row strides = 4 / 8 / 4 mod 64 dwords: the gate phase
  // touches the tiles with 16 rows x 4 units per wave instruction, which on the natural strides is a 16-way bank conflict.
  constexpr int EW = BF ? 2 : 1;                           // elements per dword of the sequence tensors
  constexpr int GTS = 4 * UW / EW + 4, RTS = 4 * UW / EW + 8, CTS = UW + 4;
  // fp32 storage: the Gx tile of the NEXT step arrives by non-temporal LDS-DMA (no registers held over the step) into the other one of two
  // UNPADDED [16][4 UW] images; the DMA writes LDS lane-linearly, so the bank swizzle is applied to the piece each lane FETCHES: slot p of row r
  // holds piece p ^ (r & 15), and the gate phase reads dword 4 ((k UW/4 + at) ^ jb) + ku of row jb -- 16 rows x 4 units on 64 distinct banks.
  // (against the piece held in four registers over the step: 3.70 -> 3.63 us per step at [128, 128, 512] f32s, 2.91 -> 2.80 at 64 rows, the f32s kernel
  //  without scratch; bf16 storage keeps the register piece -- its Gx stream costs 0.06 us: profiles/r5/lstm_fwd_gx_dma_ab_v1.txt)
  constexpr bool GXD = !BF && UW >= 16;
  constexpr int GTW = GXD ? 2 * 16 * 4 * UW : 16 * GTS;      // dwords of the Gx tile region
  unsigned* Gt = reinterpret_cast<unsigned*>(Ht + 16 * 36 + (16 * HTS > 16 * 36 ? 16 * HTS - 16 * 36 : 0));   // [16][GTS] input gates of the step, as stored
  unsigned* Rt = Gt + GTW;                                 // [16][RTS] activated gates (i,f,g,o per unit), as stored
  float* Ct = reinterpret_cast<float*>(Rt + 16 * RTS);     // [16][CTS] cell states
  // raised by a wave whose bounded wait expired.  A static __shared__ variable: through a pointer derived from the dynamic
  // LDS block the compiler lost the address space and read the flag with a FLAT load, whose vmcnt(0) wait behind the barrier
  // also waited for the loads requested for the next step.
  __shared__ unsigned s_fail;
  if (threadIdx.x == 0) s_fail = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int uslices = h / UW, bslices = (B + 15) / 16;
  const int vidx = xcd_major_index(), group = vidx / uslices;     // exchange group = (direction, batch slice): its uslices
  const int us = vidx % uslices, d = group / bslices, bs = group % bslices;   // workgroups are neighbours in XCD-major order
  const int nactive = 2 * uslices * bslices;              // a PADDED grid (persist_grid) puts whole groups on one XCD each; the
  if (vidx >= nactive) return;                            // workgroups beyond the last group have no role (uniform exit, before any barrier)
  const int at = wv;                                      // 8 waves = 8 A-tiles (32 units) x ONE 16-row batch tile: the slab a
                                                          // workgroup fetches per step is 16 rows (32 KiB at h = 512), half of the
                                                          // 4 x 2 arrangement's, for the same MFMA work per wave
  const int jb = lane & 15, ku = lane >> 4;
  const int u0 = us * UW + at * 4;                        // first unit of this wave's A-tile
  const int b0 = bs * 16;
  // sync[0] = error word, sync[1] = arrival counter of the one start barrier

  // A fragments: row i = jb -> (unit u0 + (jb>>2), gate jb&3); lane quad ku holds columns 16j + 4ku .. +3 = the k values
  // of FOUR MFMAs (component m = k 16j + 4ku + m), matching the ds_read_b128 of the h slab below
  // SPLIT: lane quad ku holds k = 32j + 8ku .. +7 of the bf16 MFMA's k block j, as hi and lo planes (4 VGPRs each)
  constexpr int NA = SPLIT ? 1 : kPersistMaxH / 16, NJB = SPLIT ? HJ / 2 : 1;
  f32x4 areg[NA];
  u32x4 ahi[NJB], alo[NJB];
  if constexpr (SPLIT) {
    const float* wrow = Whh + (size_t)d * 4 * h * h + (size_t)((jb & 3) * h + u0 + (jb >> 2)) * h + 8 * ku;
#pragma unroll
    for (int j = 0; j < NJB; ++j)
      split8(*reinterpret_cast<const f32x4*>(wrow + 32 * j), *reinterpret_cast<const f32x4*>(wrow + 32 * j + 4), ahi[j], alo[j]);
  } else {
    const float* wrow = Whh + (size_t)d * 4 * h * h + (size_t)((jb & 3) * h + u0 + (jb >> 2)) * h + 4 * ku;
#pragma unroll
    for (int j = 0; j < kPersistMaxH / 16; ++j)
      areg[j] = (16 * j < h) ? *reinterpret_cast<const f32x4*>(wrow + 16 * j) : (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const int b = b0 + jb, u = u0 + ku;
  const bool live = b < B;
  float cprev = 0.f;
  const int nrow4 = h / 4;                                 // float4 per h row
  float bi[4] = {0.f, 0.f, 0.f, 0.f};                      // b_ih + b_hh of this lane's unit when the caller's GEMM left it out
  if (bias) {
#pragma unroll
    for (int k = 0; k < 4; ++k) bi[k] = bias[(size_t)d * 4 * h + k * h + u];
  }

  // exchange ring (XR): dword address of this workgroup's chunk of (slot, row): [slot][group][row][producer][CH dwords]
  constexpr int CH = BF ? UW / 2 : UW;                     // dwords per (row, producer) chunk: fp32 UW values | UW/2 hi + UW/2 lo pairs | UW/2 bf16 pairs
  const int rowdw = uslices * CH;                          // dwords per ring row (= h, or h / 2 with bf16 storage)
  const int ngroups = 2 * bslices;
  auto xr_row = [&](int slot, int row) { return xr + ((size_t)(slot * ngroups + group) * 16 + row) * rowdw; };
  auto xr_chunk = [&](int slot, int row) { return xr_row(slot, row) + us * CH; };
  {
    // Every lane marks the elements of `out` it will store (all T steps) with the sentinel -- the same lane, the same
    // address and the same write-through path as the later h store, so the two stay ordered -- and the grid meets once
    // (arrival counter sync[1]) so that no consumer can poll an element before its sentinel is in memory.
    if constexpr (XR) {                                      // the four ring slots instead of all T steps of `out`
      const int row = tid / UW, col = tid % UW;
      if (!BF || (col & 1) == 0)
        for (int slot = 0; slot < 4; ++slot)
          store_sc1_u(reinterpret_cast<float*>(xr_chunk(slot, row) + (BF ? col >> 1 : col)), MODE == 0 ? kSentinel : (kSentinel16 | (kSentinel16 << 16)));
    } else {
      const int row = tid / UW, col = tid % UW;              // the same whole-line pattern as the h stores of the step loop
      if (b0 + row < B && (!BF || (col & 1) == 0))           // bf16: the even columns mark (and later store) two elements per dword
        for (int t = 0; t < T; ++t)
          store_sc1_u(reinterpret_cast<float*>(out + seq_row(t, b0 + row, Bs, T, bm) * 2 * h + d * h + us * UW + col),
                      BF ? (kSentinel16 | (kSentinel16 << 16)) : kSentinel);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if ((flags & 2) && blockIdx.x == 0) return;                // TSG_LSTM_INJECT_TIMEOUT: workgroup 0 never arrives (test of the error path)
  const bool local = grid_start(sync, group, flags & 1, esink, (unsigned)nactive);
  if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;

#ifdef TSG_LSTM_TIMING
  unsigned long long tph[4] = {0, 0, 0, 0}, tm0 = 0, tm1 = 0;       // per-phase s_memtime sums of this wave (100 MHz ticks? no: shader clock)
#define TSG_TICK(i) { tm1 = __builtin_amdgcn_s_memtime(); tph[i] += tm1 - tm0; tm0 = tm1; }
#else
#define TSG_TICK(i) {}
#endif
  // thread -> 16-byte piece of the step's 16 x 4 x UW tile of Gx (row, gate, 16 / sizeof(GT) units), requested one step AHEAD
  constexpr int PPG = UW / (4 * EW);                       // pieces per (row, gate)
  const int grow = tid / (4 * PPG), ggate = (tid / PPG) % 4, gpart = tid % PPG;
  const bool gact = tid < 64 * PPG && b0 + grow < B;
  u32x4 gq = {0u, 0u, 0u, 0u};
#ifndef TSG_FWD_ABL
#define TSG_FWD_ABL 0            // timing-only ablations of the streams: 1 no R / Cs stores, 2 no Gx loads, 4 no `out` store
#endif
  auto load_gx = [&](int t) {
    if (gact && !(TSG_FWD_ABL & 2)) gq = *reinterpret_cast<const u32x4*>(Gx + (seq_row(t, b0 + grow, Bs, T, bm) * 2 + d) * 4 * h + ggate * h + us * UW + gpart * 4 * EW);
  };
  const unsigned gt_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Gt;
  auto request_gx = [&](int t, int buf) {                    // (GXD) thread = slot tid % UW of row tid / UW of the image
    if (TSG_FWD_ABL & 2) return;
    const int r = tid / UW, pi = (tid % UW) ^ (r & 15), br = b0 + r < B ? b0 + r : B - 1;      // rows beyond B re-read the last one (never used)
    dma16<true>(Gx + (seq_row(t, br, Bs, T, bm) * 2 + d) * 4 * h + (pi / (UW / 4)) * h + us * UW + 4 * (pi % (UW / 4)),
                __builtin_amdgcn_readfirstlane(gt_lds + (unsigned)(buf * 16 * 4 * UW * 4) + 1024u * (unsigned)wv));
  };
  if constexpr (GXD) request_gx(d == 0 ? 0 : T - 1, 0); else load_gx(d == 0 ? 0 : T - 1);
  for (int step = 0; step < T; ++step) {
#ifdef TSG_LSTM_TIMING
    tm0 = __builtin_amdgcn_s_memtime();
#endif
    const int tt = d == 0 ? step : T - 1 - step;
    const int tp = d == 0 ? tt - 1 : tt + 1;
    // The input gates of a step are requested one step AHEAD (right after the previous step's poll): the poll's
    // s_waitcnt vmcnt(0) waits for everything this wave has in flight, and a Gx load issued in front of it put an HBM
    // round trip (~1.5 us) into every hand-off that itself takes 0.24 us (tools/ubench/l2_pingpong.hip).
    // The piece is taken over (to LDS) right AFTER the poll (whose wait has covered it), never at the loop head, where the wait for
    // it would also wait for the stores of the step before.
    auto prefetch_gx = [&]() {
      if constexpr (GXD) {
        if (step + 1 < T) request_gx(d == 0 ? step + 1 : T - 2 - step, (step + 1) & 1);      // retired by the NEXT poll's vmcnt(0), read behind its barrier
      } else {
        if (tid < 64 * PPG) *reinterpret_cast<u32x4*>(Gt + grow * GTS + ggate * (UW / EW) + 4 * gpart) = gq;
        if (step + 1 < T) load_gx(d == 0 ? step + 1 : T - 2 - step);
      }
    };
    if (step == 0) {
      if constexpr (GXD) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the first tile has landed
      prefetch_gx();
      lds_barrier();
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (step > 0) {
      f32x4 v[SV];
      u32x4 q[SV];
      const int nrowp = BF ? h / 8 : nrow4;                  // 16-byte pieces per h row (bf16: 8 elements each)
      {
        // poll the slab until no element of this thread's 16-byte pieces is the sentinel (loads unconditional, see above)
        unsigned pending = 0u;
        const float* src[SV];
        unsigned xoff[SV];
#pragma unroll
        for (int i = 0; i < SV; ++i) {
          const int idx = tid + i * NT, r = idx / nrowp, c4 = idx % nrowp;
          const bool ok = r < 16 && b0 + r < B;
          if (ok) pending |= 1u << i;
          if constexpr (XR) { src[i] = nullptr; xoff[i] = ok ? 4u * (unsigned)(r * rowdw + c4 * 4) : 0u; }   // byte offset inside the group's slab: the same every step
          else src[i] = reinterpret_cast<const float*>(out + seq_row(tp, ok ? b0 + r : b0, Bs, T, bm) * 2 * h + d * h + (ok ? c4 * (BF ? 8 : 4) : 0));
        }
        const unsigned* xbase = XR ? uniform_ptr(xr_row((step - 1) & 3, 0)) : nullptr;     // (XR) wave-uniform slab base of the slot polled now
        // (measured and dropped: two or three staggered copies of the poll in flight -- the extra slab traffic costs more
        // than the shorter retry saves, 13.7 vs 11.5 us per step)
        static_assert(SV == 2 || SV == 4 || SV == 8, "the wait below lists 2, 4 or 8 loads");
        int spins = 0;
        while (true) {
#pragma unroll
          for (int i = 0; i < SV; ++i) {
            if constexpr (XR) q[i] = load_sc1_u4_s(xbase, xoff[i]); else q[i] = load_sc1_u4(src[i]);
          }
          if constexpr (SV == 2)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]) : : "memory");
          else if constexpr (SV == 4)
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2 % SV]), "+v"(q[3 % SV]) : : "memory");
          else
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2 % SV]), "+v"(q[3 % SV]), "+v"(q[4 % SV]), "+v"(q[5 % SV]),
                         "+v"(q[6 % SV]), "+v"(q[7 % SV]) : : "memory");
          unsigned raw = 0u;
#pragma unroll
          for (int i = 0; i < SV; ++i) {
            bool pend = false;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
              if constexpr (BF || (XR && SPLIT)) pend = pend || (q[i][m] & 0xffffu) == kSentinel16 || (q[i][m] >> 16) == kSentinel16;
              else pend = pend || q[i][m] == kSentinel;
            }
            if (pend) raw |= 1u << i;
          }
          raw &= pending;
          if (!raw) break;
          // the error word (a memory round trip) is looked at on every 32nd retry only; a wave that gives up also raises the
          // workgroup's LDS flag, which is what the other waves check after the barrier below
          if ((++spins & 31) == 0 && (spins > (kSpinLimit >> 6) || err_word(sync) != 0u)) {
            raise_error(sync, esink);
            __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            break;
          }
        }
#pragma unroll
        for (int i = 0; i < SV; ++i)
          if (!(pending & (1u << i))) q[i] = (u32x4){0u, 0u, 0u, 0u};
#pragma unroll
        for (int i = 0; i < SV; ++i)
          v[i] = (f32x4){__uint_as_float(q[i][0]), __uint_as_float(q[i][1]), __uint_as_float(q[i][2]), __uint_as_float(q[i][3])};
      }
      TSG_TICK(0)                                            // poll: slab complete in registers
      prefetch_gx();
      if constexpr (XR) {
        // re-mark this workgroup's part of slot (step + 2) % 4 (it held step - 2): this poll has seen step - 1 from every producer of the
        // group, each of which stored it only after ITS poll of step - 1 -- the last reader of step - 2 -- had returned.  The mark is
        // acknowledged before this workgroup's stores of step + 1 are issued (the next poll ends on vmcnt(0)), the data of step + 2 follows
        // it from the same lane to the same address
        const int row = tid / UW, col = tid % UW;
        if (step + 2 < T && (!BF || (col & 1) == 0))
          store_x(reinterpret_cast<float*>(xr_chunk((step + 2) & 3, row) + (BF ? col >> 1 : col)),
                  __uint_as_float(MODE == 0 ? kSentinel : (kSentinel16 | (kSentinel16 << 16))), local);
      }
#pragma unroll
      for (int i = 0; i < SV; ++i) {
        const int idx = tid + i * NT, r = idx / nrowp, c4 = idx % nrowp;
        if constexpr (XR && SPLIT && !BF) {
          // ring chunk = [hi pairs of the producer's UW units | lo pairs]: piece -> (producer, plane, 8 units) -> the LDS planes as they are
          constexpr int PPC = UW / 4, PPH = PPC / 2;          // 16-byte pieces per chunk / per plane half
          const int pp = c4 / PPC, qq = c4 % PPC;
          if (r < 16) *reinterpret_cast<u32x4*>((qq >= PPH ? Hlo : Hhi) + r * kHLB + pp * (UW / 2) + 4 * (qq % PPH)) = q[i];
        } else if constexpr (BF) {
          if (r < 16) *reinterpret_cast<u32x4*>(Hhi + r * kHLB + c4 * 4) = q[i];      // the bf16 row as it is: the one operand plane
        } else if constexpr (SPLIT) {
          uint2 hi2, lo2;
          split_pair(v[i][0], v[i][1], hi2.x, lo2.x);
          split_pair(v[i][2], v[i][3], hi2.y, lo2.y);
          if (r < 16) {
            *reinterpret_cast<uint2*>(Hhi + r * kHLB + c4 * 2) = hi2;
            *reinterpret_cast<uint2*>(Hlo + r * kHLB + c4 * 2) = lo2;
          }
        } else {
          if (r < 16) *reinterpret_cast<f32x4*>(Hl + r * HLS + c4 * 4) = v[i];
        }
      }
      lds_barrier();
      TSG_TICK(1)                                            // slab in LDS, workgroup met
      if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;     // a bounded wait expired in this workgroup
      if constexpr (SPLIT) {
        // B operand: k block j of row jb = 4 dwords at 16j + 4ku of each plane, one ds_read_b128 per plane, requested PFB
        // blocks ahead; three independent accumulator chains (hi*hi, hi*lo, lo*hi), the two small ones summed first
        const unsigned* hr = Hhi + jb * kHLB + 4 * ku;
        constexpr int PFB = NJB < 3 ? NJB : 3;
        u32x4 bh[PFB], bl[PFB];
        f32x4 acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < PFB; ++j) {
          bh[j] = *reinterpret_cast<const u32x4*>(hr + 16 * j);
          if constexpr (!BF) bl[j] = *reinterpret_cast<const u32x4*>(hr + 16 * kHLB + 16 * j); else bl[j] = bh[j];
        }
#pragma unroll
        for (int j = 0; j < NJB; ++j) {
          const u32x4 vh = bh[j % PFB], vl = bl[j % PFB];
          if (j + PFB < NJB) {
            bh[j % PFB] = *reinterpret_cast<const u32x4*>(hr + 16 * (j + PFB));
            if constexpr (!BF) bl[j % PFB] = *reinterpret_cast<const u32x4*>(hr + 16 * kHLB + 16 * (j + PFB));
          }
          if constexpr (BF) {                              // one product per k block, two accumulator chains
            if (j & 1) acc2 = mfma_bf16(ahi[j], vh, acc2); else acc = mfma_bf16(ahi[j], vh, acc);
          } else {
            acc = mfma_bf16(ahi[j], vh, acc);
            acc2 = mfma_bf16(ahi[j], vl, acc2);
            acc3 = mfma_bf16(alo[j], vh, acc3);
          }
        }
        acc += acc2 + acc3;
      } else {
      // B operand: one ds_read_b128 per 16 columns (row stride = 8 mod 64: conflict-free 16-lane groups), requested PFD
      // reads ahead of the MFMAs that consume it -- issued one at a time, each read's latency (~100+ cycles) sat in
      // front of its two MFMAs and the chain ran at 111 instead of 32-64 cycles per MFMA (s_memtime instrumentation)
      const float* hrow = Hl + jb * HLS + 4 * ku;
      constexpr int NJ = HJ > 0 ? HJ : kPersistMaxH / 16, PFD = NJ < 4 ? NJ : 4;
      f32x4 bq[PFD], acc2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < PFD; ++j) bq[j] = *reinterpret_cast<const f32x4*>(hrow + 16 * j);
#pragma unroll
      for (int j = 0; j < NJ; ++j) {
        const f32x4 bv = bq[j % PFD];
        if (j + PFD < NJ) bq[j % PFD] = *reinterpret_cast<const f32x4*>(hrow + 16 * (j + PFD));     // columns beyond h are never
        if (HJ > 0 || 16 * j < h) {                                                                  // consumed (wave-uniform)
#pragma unroll
          for (int m = 0; m < 4; m += 2) {                   // two accumulator chains: a dependent MFMA waits for its predecessor
            acc = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[j][m], bv[m], acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[j][m + 1], bv[m + 1], acc2, 0, 0, 0);
          }
        }
      }
      acc += acc2;
      }
    }
#ifdef TSG_LSTM_TIMING
    asm volatile("" : "+v"(acc));
    if (step > 0) TSG_TICK(2)                                // MFMA chain issued (and its result consumed below)
#endif
    // h_t goes out first and as WHOLE 128-byte lines: the workgroup's 16 x 32 tile is gathered in LDS and every wave
    // writes two complete rows per store instruction (write-through); the per-lane 4-byte stores of the first version
    // put eight partial writes from eight waves on every line.  R / Cs (not on the critical path) follow.
    if (live) {
      const int ul = at * 4 + ku;                          // this lane's unit inside the workgroup's UW
      float gx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        if constexpr (BF) { const unsigned w2 = Gt[jb * GTS + k * (UW / 2) + (ul >> 1)]; gx[k] = ((ul & 1) ? bf16_hi(w2) : bf16_lo(w2)) + bi[k]; }
        else if constexpr (GXD) gx[k] = __uint_as_float(Gt[(step & 1) * 16 * 4 * UW + jb * 4 * UW + 4 * ((k * (UW / 4) + at) ^ jb) + ku]) + bi[k];
        else gx[k] = __uint_as_float(Gt[jb * GTS + k * UW + ul]) + bi[k];
      }
      const float gi = sigmoid_f(acc[0] + gx[0]), gf = sigmoid_f(acc[1] + gx[1]);
      const float gg = tanh_f(acc[2] + gx[2]), go = sigmoid_f(acc[3] + gx[3]);
      const float c = fmaf(gf, cprev, gi * gg);
      cprev = c;
      Ht[jb * HTS + ul] = go * tanh_f(c);
      if constexpr (BF) *reinterpret_cast<uint2*>(Rt + jb * RTS + ul * 2) = make_uint2(pack_bf16x2(gi, gf), pack_bf16x2(gg, go));
      else *reinterpret_cast<f32x4*>(Rt + jb * RTS + ul * 4) = (f32x4){gi, gf, gg, go};
      Ct[jb * CTS + ul] = c;
    }
    lds_barrier();                                        // tile complete; the slab in LDS is free again
    if constexpr (XR) {
      // ring first (the hand-off); `out` -- which nobody polls any more -- is an ordinary store behind it.  Rows beyond B are neither stored
      // nor polled.  (Measured and dropped, profiles/r5/lstm_fwd_ring_deferred_stores_ab_v1.txt: R / Cs / out stores issued behind the NEXT
      // poll instead of in front of it: 3.85 -> 4.2 us per step -- stores in flight during the step delay the next h store behind them.)
      const int row = tid / UW, col = tid % UW;
      const bool rl = b0 + row < B;
      unsigned ov = 0u, rv = 0u;                             // dword for `out`, dword for the ring
      bool oact = rl, ract = rl && step + 1 < T;
      if constexpr (BF) {
        oact = oact && (col & 1) == 0; ract = ract && (col & 1) == 0;
        ov = rv = pack_bf16x2(Ht[row * HTS + (col & ~1)], Ht[row * HTS + (col | 1)]);
      } else if constexpr (SPLIT) {
        const int pc = col % (UW / 2);                       // dword `col` of the chunk: hi pair pc (col < UW / 2) or lo pair pc
        unsigned hi, lo;
        split_pair(Ht[row * HTS + 2 * pc], Ht[row * HTS + 2 * pc + 1], hi, lo);
        rv = col < UW / 2 ? hi : lo;
        ov = __float_as_uint(Ht[row * HTS + col]);
      } else {
        ov = rv = __float_as_uint(Ht[row * HTS + col]);
      }
      if (ract) store_x(reinterpret_cast<float*>(xr_chunk(step & 3, row) + (BF ? col >> 1 : col)), __uint_as_float(rv), local);
      unsigned* op = reinterpret_cast<unsigned*>(out + seq_row(tt, rl ? b0 + row : b0, Bs, T, bm) * 2 * h + d * h + us * UW + (BF ? (col & ~1) : col));
      if (oact && !(TSG_FWD_ABL & 4)) *op = ov;
    } else {
      const int row = tid / UW, col = tid % UW;
      if constexpr (BF) {
        if (b0 + row < B && (col & 1) == 0)
          store_x(reinterpret_cast<float*>(out + seq_row(tt, b0 + row, Bs, T, bm) * 2 * h + d * h + us * UW + col),
                  __uint_as_float(pack_bf16x2(Ht[row * HTS + col], Ht[row * HTS + col + 1])), local);
      } else {
        if (b0 + row < B) store_x(out + seq_row(tt, b0 + row, Bs, T, bm) * 2 * h + d * h + us * UW + col, Ht[row * HTS + col], local);
      }
    }
    {
      // R tile = 16 rows x (UW units x 4 gates) as stored, Cs tile = 16 rows x UW floats: 16-byte pieces, rows of whole lines
      // (R / Cs as non-temporal stores, Gx as a non-temporal load: nothing in the train step in rounds 4 and 5 -- profiles/r4/lstm_fwd_nontemporal_stores_ab_v1.txt,
      // profiles/r5/lstm_bwd_operand_dma_v1.txt (4) -- not used)
      constexpr int RPR = UW / EW, CPR = UW / 4;             // 16-byte pieces per row of the R tile / of the Cs tile
      const int rrow = tid / RPR, rpart = tid % RPR;
      if (rrow < 16 && b0 + rrow < B && !(TSG_FWD_ABL & 1)) {
        const size_t srow_ = (((size_t)tt * 2 + d) * Bs + b0 + rrow) * h + us * UW;
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned*>(R + srow_ * 4) + 4 * rpart) = *reinterpret_cast<const u32x4*>(Rt + rrow * RTS + 4 * rpart);
      }
      const int crow = tid / CPR, cpart = tid % CPR;
      if (crow < 16 && b0 + crow < B && !(TSG_FWD_ABL & 1))
        *reinterpret_cast<u32x4*>(Cs + (((size_t)tt * 2 + d) * Bs + b0 + crow) * h + us * UW + 4 * cpart) = *reinterpret_cast<const u32x4*>(Ct + crow * CTS + 4 * cpart);
    }
#ifdef TSG_LSTM_TIMING
    if (step > 0) TSG_TICK(3)                                // gates, stores issued, workgroup met
#endif
  }
#ifdef TSG_LSTM_TIMING
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 4; ++i) sync[8 + i] = (unsigned)(tph[i] / (unsigned long long)(T - 1));
#endif
}


// ---------------------------------------------------------------------------------------------
// forward, persistent, bf16 storage, 64-UNIT workgroups on HALF the CUs (round 5; h = 512, >= 96 rows).
// What paces the forward step at the full chip is not the work between its barriers but the all-gather of the 16-row h slab through
// one XCD's L2 -- every workgroup of a group polls the whole slab, two groups share an L2 -- and the clock of a fully busy chip
// (HISTORY.md, old section 5, "Round 5": only the poll phase differs between 64 and 128 rows; with bf16 storage the MFMA phase halves and the
// waiting grows by the same amount).  With bf16 storage W_hh is one plane: a wave can hold TWO A-tiles (8 units x 4 gates x h = 128 VGPRs), a
// workgroup 64 units, a group 8 workgroups -- half the slab readers per L2 and half the CUs idle -- for twice the MFMA and gate work per
// wave, which the waiting absorbs: a timing-only emulation (the 64-row grid with every wave's MFMAs and gates doubled:
// profiles/r5/lstm_fwd_half_chip_emulation_v1.txt) ran 3.18 us per step against 3.66 for the 32-unit kernel at 128 rows.
// Same protocol as lstm_fwd_persist_kernel<.., MODE = 2, NW = 8, XR = true>: data-as-flag through the 4-slot exchange ring (one 128-byte
// line per (row, producer)), sentinels, re-mark of slot (s + 2) % 4 behind the poll of step s, the group's XCD check, bounded waits.
// ---------------------------------------------------------------------------------------------
#ifndef TSG_W64_ABL
#define TSG_W64_ABL 0            // timing-only ablations: 1 no R / Cs stores, 2 no Gx loads, 4 no `out` store
#endif
template <int HJ>
__global__ __launch_bounds__(512) void lstm_fwd_persist_w64_kernel(
    const lstm_bf16* __restrict__ Gx, const float* __restrict__ bias, const float* __restrict__ Whh, lstm_bf16* __restrict__ out,
    lstm_bf16* __restrict__ R, float* __restrict__ Cs, unsigned* __restrict__ sync, int B, int Bs, int T, int h, int flags, int bm,
    ErrSink esink, unsigned* __restrict__ xr) {
  constexpr int NT = 512, AT = 2, UW = 64, HTS = UW + 1, NJB = HJ / 2, CH = UW / 2;      // CH: dwords of a (row, producer) ring chunk
  constexpr int SV = 2;                                    // 16-byte pieces of the 16-row bf16 slab per thread (16 x h x 2 bytes / 512 threads at h = 512)
  extern __shared__ __align__(16) float Hl[];
  unsigned* Hhi = reinterpret_cast<unsigned*>(Hl);         // [16][kHLB] dwords: the slab as it arrives (one bf16 plane)
  float* Ht = Hl + kSlabFloats;                            // [16][HTS] this step's h tile (16 rows x 64 units), gathered for whole-line stores
  // The streamed tensors go through LDS as well, so that every global access is a 16-byte piece of a full line.  (As per-lane accesses in the
  // MFMA lane layout -- 8 two-byte Gx loads and 4 scattered R / Cs stores per lane and step -- they cost 2.8 of the step's 4.9 us:
  // profiles/r5/lstm_fwd_w64_streaming_ablation_v1.txt.)
  // Row strides = 4 / 8 / 4 mod 64 dwords: the gate phase touches these tiles in the MFMA lane layout (16 rows x 4 units per wave) -- on the
  // natural strides (128 / 128 / 64 dwords) all 16 rows of a wave instruction fall on ONE bank (a 16-way conflict on 20 LDS accesses per lane
  // and step: measured +1.3 us per step)
  constexpr int GTS = 4 * 32 + 4, RTS = 64 * 2 + 8, CTS = 64 + 4;
  unsigned* Gt = reinterpret_cast<unsigned*>(Ht + 16 * HTS);       // [16 rows][GTS]: 4 gates x 32 dwords -- the step's input gates, bf16 pairs as stored
  unsigned* Rt = Gt + 16 * GTS;                                     // [16 rows][RTS]: 64 units x 2 dwords -- activated gates (i,f | g,o) as bf16 pairs
  float* Ct = reinterpret_cast<float*>(Rt + 16 * RTS);              // [16 rows][CTS]: 64 units -- cell states
  static_assert((kSlabFloats + 16 * HTS) % 4 == 0, "Gt is 16-byte aligned");
  __shared__ unsigned s_fail;
  if (threadIdx.x == 0) s_fail = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int uslices = h / UW, bslices = (B + 15) / 16;
  const int vidx = xcd_major_index(), group = vidx / uslices;
  const int us = vidx % uslices, d = group / bslices, bs = group % bslices;
  const int nactive = 2 * uslices * bslices;
  if (vidx >= nactive) return;
  const int jb = lane & 15, ku = lane >> 4, b0 = bs * 16;
  const int ngroups = 2 * bslices, rowdw = uslices * CH;   // (= h / 2 dwords per ring row)
  auto xr_row = [&](int slot, int row) { return xr + ((size_t)(slot * ngroups + group) * 16 + row) * rowdw; };
  auto xr_chunk = [&](int slot, int row) { return xr_row(slot, row) + us * CH; };
  // (32-bit element offsets from the scalar tensor bases: one VGPR per address; the host checks that the tensors stay below 2^31 elements)
  auto srow32 = [&](int t, int bb) { return bm ? (unsigned)bb * (unsigned)T + (unsigned)t : (unsigned)t * (unsigned)Bs + (unsigned)bb; };

  // A fragments of the wave's two tiles: tile ta = 2 wv + a holds units us * 64 + 4 ta .. + 3; row i = jb -> (unit + (jb >> 2), gate jb & 3);
  // lane quad ku holds k = 32 j + 8 ku .. + 7 of k block j, rounded to bf16 once (the storage mode's one plane)
  u32x4 ahi[AT][NJB];
  float bi[AT][4];
#pragma unroll
  for (int a = 0; a < AT; ++a) {
    const int u0 = us * UW + 4 * (AT * wv + a);
    const float* wrow = Whh + (size_t)d * 4 * h * h + (size_t)((jb & 3) * h + u0 + (jb >> 2)) * h + 8 * ku;
#pragma unroll
    for (int j = 0; j < NJB; ++j) {
      u32x4 lo;
      split8(*reinterpret_cast<const f32x4*>(wrow + 32 * j), *reinterpret_cast<const f32x4*>(wrow + 32 * j + 4), ahi[a][j], lo);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) bi[a][k] = bias ? bias[(size_t)d * 4 * h + k * h + u0 + ku] : 0.f;
  }
  const bool live = b0 + jb < B;
  float cprev[AT] = {0.f, 0.f};
  // thread -> (row, unit pair) of the 16 x 64 tile: marks, re-marks and h stores use the same lane for the same address
  const int srow = tid >> 5, spair = tid & 31;
  const bool rl = b0 + srow < B;
  const unsigned sent2 = kSentinel16 | (kSentinel16 << 16);
  for (int slot = 0; slot < 4; ++slot) store_sc1_u(reinterpret_cast<float*>(xr_chunk(slot, srow) + spair), sent2);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if ((flags & 2) && blockIdx.x == 0) return;              // TSG_LSTM_INJECT_TIMEOUT: workgroup 0 never arrives (test of the error path)
  const bool local = grid_start(sync, group, flags & 1, esink, (unsigned)nactive);
  if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;

  // poll pieces: the same every step (byte offsets inside the group's slab)
  const int nrowp = h / 8;                                 // 16-byte pieces per slab row
  unsigned xoff[SV], pending = 0u;
#pragma unroll
  for (int i = 0; i < SV; ++i) {
    const int idx = tid + i * NT, r = idx / nrowp, c4 = idx % nrowp;
    const bool ok = r < 16 && b0 + r < B;
    if (ok) pending |= 1u << i;
    xoff[i] = ok ? 4u * (unsigned)(r * rowdw + c4 * 4) : 0u;
  }
  // thread -> 16-byte piece of the step's 16 x 4 x 64 bf16 tile of Gx (row srow, gate, 8 units): 512 pieces = one per thread, requested a step ahead
  const int ggate = (tid >> 3) & 3, gpart = tid & 7;
  u32x4 gq = {0u, 0u, 0u, 0u};
  auto load_gx = [&](int t) {
    if (!rl || (TSG_W64_ABL & 2)) return;
    const unsigned g0 = (srow32(t, b0 + srow) * 2u + (unsigned)d) * 4u * (unsigned)h + (unsigned)(ggate * h + us * UW + 8 * gpart);
    gq = *reinterpret_cast<const u32x4*>(Gx + g0);
  };
  load_gx(d == 0 ? 0 : T - 1);
  for (int step = 0; step < T; ++step) {
    const int tt = d == 0 ? step : T - 1 - step;
    // the piece requested a step ago goes to LDS (its wait: the poll's vmcnt(0), or -- step 0 -- the compiler's), the next step's is requested behind it
    auto take_gx = [&]() {
      *reinterpret_cast<u32x4*>(Gt + srow * GTS + ggate * 32 + 4 * gpart) = gq;
      if (step + 1 < T) load_gx(d == 0 ? step + 1 : T - 2 - step);
    };
    f32x4 acc[AT][2];
#pragma unroll
    for (int a = 0; a < AT; ++a) { acc[a][0] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc[a][1] = acc[a][0]; }
    if (step == 0) { take_gx(); lds_barrier(); }
    if (step > 0) {
      u32x4 q[SV];
      const unsigned* xbase = uniform_ptr(xr_row((step - 1) & 3, 0));
      int spins = 0;
      while (true) {
#pragma unroll
        for (int i = 0; i < SV; ++i) q[i] = load_sc1_u4_s(xbase, xoff[i]);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]) : : "memory");
        unsigned raw = 0u;
#pragma unroll
        for (int i = 0; i < SV; ++i) {
          bool pend = false;
#pragma unroll
          for (int m = 0; m < 4; ++m) pend = pend || (q[i][m] & 0xffffu) == kSentinel16 || (q[i][m] >> 16) == kSentinel16;
          if (pend) raw |= 1u << i;
        }
        raw &= pending;
        if (!raw) break;
        if ((++spins & 31) == 0 && (spins > (kSpinLimit >> 6) || err_word(sync) != 0u)) {
          raise_error(sync, esink);
          __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          break;
        }
      }
      take_gx();
      // re-mark this workgroup's part of slot (step + 2) % 4 (it held step - 2: everybody has read it -- see the 32-unit kernel)
      if (step + 2 < T) store_x(reinterpret_cast<float*>(xr_chunk((step + 2) & 3, srow) + spair), __uint_as_float(sent2), local);
#pragma unroll
      for (int i = 0; i < SV; ++i) {
        const int idx = tid + i * NT, r = idx / nrowp, c4 = idx % nrowp;
        if (!(pending & (1u << i))) q[i] = (u32x4){0u, 0u, 0u, 0u};
        if (r < 16) *reinterpret_cast<u32x4*>(Hhi + r * kHLB + c4 * 4) = q[i];
      }
      lds_barrier();
      if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;
      // one B fragment (ds_read_b128, three blocks ahead) feeds both tiles' MFMAs; two accumulator chains per tile
      const unsigned* hr = Hhi + jb * kHLB + 4 * ku;
      constexpr int PFB = NJB < 3 ? NJB : 3;
      u32x4 bh[PFB];
#pragma unroll
      for (int j = 0; j < PFB; ++j) bh[j] = *reinterpret_cast<const u32x4*>(hr + 16 * j);
#pragma unroll
      for (int j = 0; j < NJB; ++j) {
        const u32x4 vh = bh[j % PFB];
        if (j + PFB < NJB) bh[j % PFB] = *reinterpret_cast<const u32x4*>(hr + 16 * (j + PFB));
#pragma unroll
        for (int a = 0; a < AT; ++a) acc[a][j & 1] = mfma_bf16(ahi[a][j], vh, acc[a][j & 1]);
      }
    }
#pragma unroll
    for (int a = 0; a < AT; ++a) {
      const int ul = 4 * (AT * wv + a) + ku;               // unit of this lane inside the workgroup's 64
      float gx[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned w2 = Gt[jb * GTS + k * 32 + (ul >> 1)];
        gx[k] = ((ul & 1) ? bf16_hi(w2) : bf16_lo(w2)) + bi[a][k];
      }
      const f32x4 z = acc[a][0] + acc[a][1];
      const float gi = sigmoid_f(z[0] + gx[0]), gf = sigmoid_f(z[1] + gx[1]), gg = tanh_f(z[2] + gx[2]), go = sigmoid_f(z[3] + gx[3]);
      const float c = fmaf(gf, cprev[a], gi * gg);
      cprev[a] = c;
      if (live) {
        Ht[jb * HTS + ul] = go * tanh_f(c);
        *reinterpret_cast<uint2*>(Rt + jb * RTS + ul * 2) = make_uint2(pack_bf16x2(gi, gf), pack_bf16x2(gg, go));
        Ct[jb * CTS + ul] = c;
      }
    }
    lds_barrier();                                         // tiles complete; the slab and Gt are free again
    if (rl) {
      // h first (the hand-off): ring, then `out`; then R (16 rows x 64 units x 8 bytes = 512 pieces of 16 bytes) and Cs (256 pieces)
      const unsigned pk = pack_bf16x2(Ht[srow * HTS + 2 * spair], Ht[srow * HTS + 2 * spair + 1]);
      if (step + 1 < T) store_x(reinterpret_cast<float*>(xr_chunk(step & 3, srow) + spair), __uint_as_float(pk), local);
      if (!(TSG_W64_ABL & 4)) *reinterpret_cast<unsigned*>(out + (srow32(tt, b0 + srow) * 2u * (unsigned)h + (unsigned)(d * h + us * UW + 2 * spair))) = pk;
      if (!(TSG_W64_ABL & 1)) {
        const unsigned sidx = (((unsigned)tt * 2u + (unsigned)d) * (unsigned)Bs + (unsigned)(b0 + srow)) * (unsigned)h + (unsigned)(us * UW);
        *reinterpret_cast<u32x4*>(reinterpret_cast<unsigned*>(R) + ((size_t)sidx * 2 + 4 * spair)) = *reinterpret_cast<const u32x4*>(Rt + srow * RTS + 4 * spair);
        if (spair < 16) *reinterpret_cast<f32x4*>(Cs + (sidx + 4u * spair)) = *reinterpret_cast<const f32x4*>(Ct + srow * CTS + 4 * spair);
      }
    }
  }
}


static int lstm_check(const char* fn, int B, int T, int h, int dtype) {
  if (dtype != TSG_F32 && dtype != TSG_F32S && dtype != TSG_BF16)
    return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, TSG_F32S or TSG_BF16)", fn, dtype);
  if (B <= 0 || T <= 0 || h <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d h=%d", fn, B, T, h);
  if (h % 4) return set_error(TSG_E_ALIGN, "%s: hidden size %d must be a multiple of 4", fn, h);
  return 0;
}

// ---------------------------------------------------------------------------------------------
// backward, PERSISTENT (the default when the caller provides the ring workspace): exchange dh, not dG.
// (A first version mirrored the forward literally -- consumers polling the sentinel-marked 4h-wide dG slab, 128 KiB per
// workgroup and step -- and measured 17.9 us per step against 14.9 for the launch-per-step kernels; see HISTORY.md.)
// Workgroup = (direction, 32 units, 16 batch rows) as in the forward.  Its OWN gate gradients of the step before
// (16 rows x 128 gate columns, in LDS) are the B operand, its 128 x h slice of W_hh (registers, 128 VGPRs at h = 512)
// the A operand of  partial_dh[16 rows][ALL h units] = dG_own W_hh[own gate rows, :]  -- no operand has to be fetched.
// What is exchanged is the reduction: every workgroup writes its partial tile as whole 128-byte write-through lines into
// a ring slot (one 2 KiB block per consumer), and polls the h/32 blocks addressed to it -- 32 KiB per workgroup and
// step, like the forward's h slab and a quarter of a dG slab.  Ring = 2 slots, slot = step % 2 (round 4; four slots until then: 33.5 MB
// at [128, ., 512], more than the chip's L2s hold, so the blocks were re-fetched from memory -- see launch_flags), and the
// "written" mark is a generation tag instead of a sentinel: every partial value carries (step/2) % 2 in its lowest
// mantissa bit (a perturbation of at most one ulp of a partial sum); a consumer accepts a float4 when all four tags
// match the generation it expects, so a slot needs no re-marking between uses (re-marking with sentinel lines doubled
// the write-through traffic: 14.1 us per step).  Consumers are never more than one step apart, so a slot's previous
// generation has been read by everyone before its next one is written.
// ---------------------------------------------------------------------------------------------
#ifndef TSG_BWD_ABL
#define TSG_BWD_ABL 0              // timing builds only: 1 no dG stores, 2 no operand streams
#endif
constexpr int kDLS = 128 + 8;                    // own-dG tile row stride (floats), = 8 mod 64
constexpr int kDLB = 64 + 8;                     // split-precision mode: row stride (dwords) of each bf16 plane of that tile
constexpr int kDlFloats = 2 * 16 * kDLB;         // LDS dwords of the dG tile region (>= 16 * kDLS)
constexpr int kPLS = kPersistMaxH + 8;           // partial-dh gather row stride
constexpr int kQLS = 36;                         // polled partial sums row stride
constexpr int kObBufs = 2;                       // LDS buffers of the backward's streamed operand tiles: this step's and the next one's (a second step of
                                                 // look-ahead measured no gain: profiles/r5/lstm_bwd_operand_dma_v1.txt (3))
constexpr int kObDw = 3072;                      // dwords of one buffer of the backward's streamed operand tiles (fp32: 8 + 2 + 2 KiB)

// MODE as in the forward kernel: 0 fp32, 1 split precision (TSG_F32S), 2 = bf16 storage of R, dOut and dG with one bf16 MFMA per
// k block (TSG_BF16; the partial-dh ring, Cs, dHn, dbias stay fp32).
template <int TW, int MODE>                       // TW = 16-unit tiles per wave = h / 128 (compile time: no branch between MFMAs)
__global__ __launch_bounds__(kThreads) void lstm_bwd_persist2_kernel(
    const float* __restrict__ WhhT, const typename SeqT<MODE>::type* __restrict__ R, const float* __restrict__ Cs,
    const typename SeqT<MODE>::type* __restrict__ dOut, const float* __restrict__ dHn, typename SeqT<MODE>::type* __restrict__ dG,
    float* __restrict__ ring, unsigned* __restrict__ sync, float* __restrict__ dbias, int B, int Bs, int T, int h, int flags, int bm, ErrSink esink) {
  constexpr bool SPLIT = MODE >= 1, BF = MODE == 2;
  typedef typename SeqT<MODE>::type GT;
  extern __shared__ __align__(16) float smem2[];
  float* Dl = smem2;                              // [16][kDLS]  this workgroup's dG tile, local column g*32 + ul
  unsigned* Dhi = reinterpret_cast<unsigned*>(Dl);          // (SPLIT: the same tile as two bf16 planes [16][kDLB] dwords)
  unsigned* Dlo = Dhi + 16 * kDLB;
  float* Pl = Dl + kDlFloats;                     // [16][kPLS]  partial dh of all h units, gathered for whole-line stores
  float* Ql = Pl + 16 * kPLS;                     // [4][16][kQLS] sums of the polled blocks per producer group
  unsigned* Ob = reinterpret_cast<unsigned*>(Ql + 4 * 16 * kQLS);    // [2][kObDw] the streamed operands of a step (R | c_(t-1) | dOut tiles), by LDS-DMA
  const unsigned ob_lds = (unsigned)(size_t)(__attribute__((address_space(3))) char*)Ob;
  __shared__ unsigned s_fail;                     // raised by a wave whose bounded wait expired (static: see the forward kernel)
  if (threadIdx.x == 0) s_fail = 0u;
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int nus = h / 32, bslices = (B + 15) / 16;         // TW = h / 128 = 16-unit tiles per wave = float4 per thread
  const int vidx = xcd_major_index(), group = vidx / nus;   // exchange group = (direction, batch slice), see the forward kernel
  const int us = vidx % nus, d = group / bslices, bs = group % bslices;
  const int nactive = 2 * nus * bslices;                    // padded grid (persist_grid): workgroups beyond the last group have no role
  if (vidx >= nactive) return;
  const int b0 = bs * 16, K = 4 * h;
  const int jb = lane & 15, ku = lane >> 4;
  // A fragments: tile t of this wave = output units 16*(wv*TW + t) + jb; local k' = 16s + 4ku + m -> gate s/2, unit
  // us*32 + 16*(s%2) + 4ku + m of the workgroup's own gate columns
  // SPLIT: k block kb of the bf16 MFMA = gate kb, units us*32 + 8ku .. +7 (8 consecutive floats of the W_hh^T row)
  constexpr int NA8 = SPLIT ? 1 : 8, NKB = SPLIT ? 4 : 1;
  f32x4 areg[TW][NA8];
  u32x4 ahi[TW][NKB], alo[TW][NKB];
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const float* wrow = WhhT + ((size_t)d * h + 16 * (wv * TW + t) + jb) * K + us * 32;
    if constexpr (SPLIT) {
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
        split8(*reinterpret_cast<const f32x4*>(wrow + kb * h + 8 * ku), *reinterpret_cast<const f32x4*>(wrow + kb * h + 8 * ku + 4),
               ahi[t][kb], alo[t][kb]);
    } else {
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8)
        areg[t][s8] = *reinterpret_cast<const f32x4*>(wrow + (s8 >> 1) * h + 16 * (s8 & 1) + 4 * ku);
    }
  }
  const int row = tid >> 5, ul = tid & 31;                  // epilogue role: (batch row, unit) of the 16 x 32 tile
  const int b = b0 + row, u = us * 32 + ul;
  const bool live = b < B;
  float dc_carry = 0.f;
  float dbsum[4] = {0.f, 0.f, 0.f, 0.f};                    // this thread's sum over the time steps of its four gate gradients
  const int hq = h / 4;                                     // float4 per partial row
  // ring depth: 2 slots (flags bit 3, the default: tag = (step / 2) % 2) or 4 (tag = (step / 4) % 2).  Two suffice: a workgroup stores step s + 2
  // only after it has polled step s + 1 from every producer of its group, and those stored step s + 1 only after their loads of step s had
  // returned (the poll loop ends on s_waitcnt vmcnt(0) with every tag checked) -- nobody can still be reading the slot that step s + 2 overwrites.
  const int rmask = (flags & 8) ? 1 : 3, rsh = (flags & 8) ? 1 : 2;
  auto slot_base = [&](int slot) { return ring + ((size_t)(slot * 2 + d) * bslices + bs) * nus * nus * 512; };
  // producer side: float4 i of this thread = (row pr, units 4*pc .. +3) of the partial tile -> block of consumer pc/8
  auto prod_ptr = [&](int slot, int i) {
    const int idx = tid + i * kThreads, pr = idx / hq, pc = idx % hq;
    return slot_base(slot) + ((size_t)((pc >> 3) * nus + us) * 16 + pr) * 32 + 4 * (pc & 7);
  };

  {  // all four slots start with the "odd generation" tag (low mantissa bit 1) on every element, then the grid meets once
    const u32x4 sent = {kSentinel | 1u, kSentinel | 1u, kSentinel | 1u, kSentinel | 1u};
    for (int slot = 0; slot <= rmask; ++slot)
      for (int i = 0; i < TW; ++i) store_sc1_u4(prod_ptr(slot, i), sent);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  if ((flags & 2) && blockIdx.x == 0) return;                // TSG_LSTM_INJECT_TIMEOUT: workgroup 0 never arrives (test of the error path)
  const bool local = grid_start(sync, group, flags & 1, esink, (unsigned)nactive);
  if (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;

#ifdef TSG_LSTM_TIMING
  unsigned long long tph[6] = {0, 0, 0, 0, 0, 0}, tm0 = 0, tm1 = 0;
#endif
  // The streamed operands of a step -- R (four gates), c_(t-1), dOut of the workgroup's 16 x 32 tile -- are moved by LDS-DMA, 16-byte pieces of
  // whole lines, ONE STEP AHEAD: requested behind the poll of the step before into the other half of `Ob`, landed by the time the next poll's
  // s_waitcnt vmcnt(0) returns (the operations of a wave retire in order), read from LDS by the (row, unit) threads behind that poll's barrier.
  // Until round 5 they were per-thread loads in front of the poll, which then waited for them: 0.6-0.76 us of a 3.1-3.8 us step
  // (profiles/r5/lstm_bwd_streams_ablation_v1.txt); as register loads a step ahead hipcc put the wait for them at the loop's end (a loop-carried
  // register copy, a bf16 conversion): in front of the next poll again.  c_t itself is the c_(t-1) of the step before.
  // non-temporal: train step 12.36 -> 12.27 ms (f32s), 7.34 -> 7.20 (bf16 storage); without the hint 12.5 / 7.36 -- the streamed lines then stay in the
  // L2s the exchange rings live in (profiles/r5/lstm_bwd_operand_dma_v1.txt (5))
  constexpr bool ONT = true;
  constexpr int kObR = BF ? 4096 : 8192, kObC = kObR;                 // byte offsets inside a buffer: R tile | c_(t-1) tile (2 KiB) | dOut tile
  auto request = [&](int s1) {
    if (TSG_BWD_ABL & 2) return;
    const int fs1 = T - 1 - s1, tt1 = d == 0 ? fs1 : T - 1 - fs1;
    const bool hp1 = (d == 0) ? (tt1 > 0) : (tt1 < T - 1);
    const int tp1 = hp1 ? (d == 0 ? tt1 - 1 : tt1 + 1) : tt1;
    const unsigned base = ob_lds + (unsigned)((s1 % kObBufs) * kObDw * 4);
    auto rowc = [&](int r) { const int bb = b0 + r; return bb < B ? bb : B - 1; };       // rows beyond B re-read the last one (never used)
    if constexpr (BF) {
      if (wv < 4) {                                           // R: 256 pieces = (row, 2 units)
        const int p = tid;
        dma16<ONT>(R + ((((size_t)tt1 * 2 + d) * Bs + rowc(p >> 4)) * h + us * 32 + 2 * (p & 15)) * 4, __builtin_amdgcn_readfirstlane(base + 1024u * wv));
      } else if (wv < 6) {                                    // c_(t-1): 128 pieces = (row, 4 units)
        const int p = tid - 256;
        dma16<ONT>(Cs + (((size_t)tp1 * 2 + d) * Bs + rowc(p >> 3)) * h + us * 32 + 4 * (p & 7), __builtin_amdgcn_readfirstlane(base + kObC + 1024u * (wv - 4)));
      } else if (wv == 6) {                                   // dOut: 64 pieces = (row, 8 units)
        const int p = tid - 384;
        dma16<ONT>(dOut + seq_row(tt1, rowc(p >> 2), Bs, T, bm) * 2 * h + d * h + us * 32 + 8 * (p & 3), __builtin_amdgcn_readfirstlane(base + kObC + 2048u));
      }
    } else {
      dma16<ONT>(R + ((((size_t)tt1 * 2 + d) * Bs + rowc(tid >> 5)) * h + us * 32 + (tid & 31)) * 4, __builtin_amdgcn_readfirstlane(base + 1024u * wv));   // R: piece = (row, unit)
      if (wv < 2) {
        const int p = tid;
        dma16<ONT>(Cs + (((size_t)tp1 * 2 + d) * Bs + rowc(p >> 3)) * h + us * 32 + 4 * (p & 7), __builtin_amdgcn_readfirstlane(base + kObC + 1024u * wv));
      } else if (wv < 4) {
        const int p = tid - 128;
        dma16<ONT>(dOut + seq_row(tt1, rowc(p >> 3), Bs, T, bm) * 2 * h + d * h + us * 32 + 4 * (p & 7), __builtin_amdgcn_readfirstlane(base + kObC + 2048u + 1024u * (wv - 2)));
      }
    }
  };
  float cc, dhn0 = 0.f;
  {
    const int tt0 = d == 0 ? T - 1 : 0;
    cc = ld1s(Cs + (((size_t)tt0 * 2 + d) * Bs + (live ? b : B - 1)) * h + u);
    if (dHn && live) dhn0 = dHn[((size_t)d * Bs + b) * h + u];
    request(0);
  }
  for (int step = 0; step < T; ++step) {
#ifdef TSG_LSTM_TIMING
    tm0 = __builtin_amdgcn_s_memtime();
#endif
    const int fs = T - 1 - step;
    const int tt = d == 0 ? fs : T - 1 - fs;                // time index handled now
    const int tp = d == 0 ? tt - 1 : tt + 1;                // forward-earlier neighbour (c_{t-1})
    const bool has_prev = (d == 0) ? (tt > 0) : (tt < T - 1);
    if (step == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); lds_barrier(); }     // the first step's operand tiles have landed (later steps: the poll's wait)
    float rec = 0.f;
    if (step > 0) {
      // poll the nus blocks addressed to this workgroup in slot (step-1)%4: thread = (producer group pg, row pr, 4 units pc);
      // its loads i = producers pg + 4i (always four loads: the ones beyond TW repeat the first and are ignored)
      const int pg = tid >> 7, pr = (tid & 127) >> 3, pc = tid & 7;
      const float* cbase = slot_base((step - 1) & rmask) + (size_t)us * nus * 512;
      const float* src[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) src[i] = cbase + ((size_t)(pg + 4 * (i < TW ? i : 0)) * 16 + pr) * 32 + 4 * pc;
      const unsigned pending = (1u << TW) - 1u;
      const unsigned gen = ((unsigned)(step - 1) >> rsh) & 1u;         // generation of slot (step-1)%4: every element carries it in its low bit
      u32x4 q[4];
      int spins = 0;
      while (true) {
#pragma unroll
        for (int i = 0; i < 4; ++i) q[i] = load_sc1_u4(src[i]);
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(q[0]), "+v"(q[1]), "+v"(q[2]), "+v"(q[3]) : : "memory");
        unsigned raw = 0u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (((q[i][0] & q[i][1] & q[i][2] & q[i][3]) & 1u) != gen || ((q[i][0] | q[i][1] | q[i][2] | q[i][3]) & 1u) != gen) raw |= 1u << i;
        raw &= pending;
        if (!raw) break;
        if ((++spins & 31) == 0 && (spins > (kSpinLimit >> 6) || err_word(sync) != 0u)) {
          raise_error(sync, esink);
          __hip_atomic_store(&s_fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // (error word read on every 32nd retry only)
          break;
        }
      }
      TSG_TICK(0)                                            // poll: all partial blocks landed
      f32x4 sum = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (i < TW) sum += (f32x4){__uint_as_float(q[i][0]), __uint_as_float(q[i][1]), __uint_as_float(q[i][2]), __uint_as_float(q[i][3])};
      *reinterpret_cast<f32x4*>(Ql + (pg * 16 + pr) * kQLS + 4 * pc) = sum;
      lds_barrier();
      if (__hip_atomic_load(&s_fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) return;     // a bounded wait expired in this workgroup
#pragma unroll
      for (int g = 0; g < 4; ++g) rec += Ql[(g * 16 + row) * kQLS + ul];
    }
    float4 g4;
    float cpv, dov;
    {
      const unsigned* ob = Ob + (step % kObBufs) * kObDw;
      if constexpr (BF) {
        const uint2 r2 = *reinterpret_cast<const uint2*>(ob + (row * 32 + ul) * 2);
        g4 = make_float4(bf16_lo(r2.x), bf16_hi(r2.x), bf16_lo(r2.y), bf16_hi(r2.y));
        const unsigned w2 = ob[kObC / 4 + 512 + (row * 32 + ul) / 2];
        dov = (ul & 1) ? bf16_hi(w2) : bf16_lo(w2);
      } else {
        const f32x4 r4 = *reinterpret_cast<const f32x4*>(ob + tid * 4);
        g4 = make_float4(r4[0], r4[1], r4[2], r4[3]);
        dov = __uint_as_float(ob[kObC / 4 + 512 + row * 32 + ul]);
      }
      cpv = has_prev ? __uint_as_float(ob[kObC / 4 + row * 32 + ul]) : 0.f;
      if (step == 0) dov += dhn0;                            // (loaded in front of the loop: a load inside it, even on a path taken once, makes hipcc drain vmcnt -- the DMAs -- at the join)
    }
    {
      const float gi = g4.x, gf = g4.y, gg = g4.z, go = g4.w;
      const float tc = tanh_f(cc);
      const float dh = dov + rec;
      const float dc = fmaf(dh * go, 1.f - tc * tc, dc_carry);
      dc_carry = dc * gf;
      cc = cpv;                                              // c_t of the next step is this step's c_(t-1): one stream less
      const float dg[4] = {dc * gg * gi * (1.f - gi), dc * cpv * gf * (1.f - gf), dc * gi * (1.f - gg * gg), dh * tc * go * (1.f - go)};
      if (!BF && live && !(TSG_BWD_ABL & 1)) {               // (bf16 storage: the tile goes out as 16-byte pieces behind the barrier below)
        GT* g = dG + (seq_row(tt, b, Bs, T, bm) * 2 + d) * K + u;
#pragma unroll
        for (int gate = 0; gate < 4; ++gate) st1s(g + gate * h, dg[gate]);
      }
#pragma unroll
      for (int gate = 0; gate < 4; ++gate) {
        const float dgv = live ? dg[gate] : 0.f;
        if constexpr (SPLIT) {
          unsigned hi, lo;
          split_pair(dgv, 0.f, hi, lo);
          reinterpret_cast<unsigned short*>(Dhi)[row * 2 * kDLB + gate * 32 + ul] = (unsigned short)hi;   // = the bf16 written to dG (BF)
          if constexpr (!BF) reinterpret_cast<unsigned short*>(Dlo)[row * 2 * kDLB + gate * 32 + ul] = (unsigned short)lo;
        } else {
          Dl[row * kDLS + gate * 32 + ul] = dgv;
        }
        if (live) dbsum[gate] += dg[gate];
      }
    }
    if (step > 0) TSG_TICK(1)                                // reduce + cell backward + dG / Dl stores issued
    lds_barrier();                                         // the dG tile is complete (and Ql is free again)
    if constexpr (BF) {
      // bf16 storage: the hi plane IS the dG tile as stored -- 16 rows x 4 gates x 64 bytes = 256 pieces of 16 bytes, one store for half the threads
      // instead of four 2-byte stores from each (0.12-0.19 us per step, profiles/r5/lstm_bwd_streams_ablation_v1.txt)
      const int prow = tid >> 4, pgate = (tid >> 2) & 3, ppart = tid & 3;
      if (tid < 256 && b0 + prow < B && !(TSG_BWD_ABL & 1))
        __builtin_nontemporal_store(*reinterpret_cast<const u32x4*>(Dhi + prow * kDLB + pgate * 16 + ppart * 4),
                                    reinterpret_cast<u32x4*>(dG + (seq_row(tt, b0 + prow, Bs, T, bm) * 2 + d) * K + pgate * h + us * 32 + ppart * 8));
    }
    if (step + 1 < T) {
      if (step + 1 < T) request(step + 1);                 // the next step's operand tiles (behind the poll's barrier measured the same or 0.1 us slower)
      if (step > 0) TSG_TICK(4)                              // (timing builds: the wait at this barrier)
      f32x4 acc[TW];
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT) {
        // per k block: hi*hi, hi*lo and lo*hi; consecutive MFMAs always sit on different accumulators.  With >= 3 tiles per
        // wave the three products of a tile share ONE accumulator (the tiles' round robin keeps dependent MFMAs >= 3 apart),
        // which frees the 16 VGPRs the operands requested for the next step need (with a second accumulator set the kernel
        // spilled a W_hh fragment to scratch and re-read it in front of an MFMA: 5.0 instead of 4.6 us per step); fewer tiles:
        // a second accumulator per tile for the two small products
        constexpr bool ONE = TW >= 3;
        f32x4 accc[ONE ? 1 : TW];
        if constexpr (!ONE) {
#pragma unroll
          for (int t = 0; t < TW; ++t) accc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
        }
        const unsigned* dr = Dhi + jb * kDLB + 4 * ku;
        u32x4 nh = *reinterpret_cast<const u32x4*>(dr), nl = BF ? nh : *reinterpret_cast<const u32x4*>(dr + 16 * kDLB);
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const u32x4 vh = nh, vl = nl;
          if (kb + 1 < 4) {
            nh = *reinterpret_cast<const u32x4*>(dr + 16 * (kb + 1));
            if constexpr (!BF) nl = *reinterpret_cast<const u32x4*>(dr + 16 * kDLB + 16 * (kb + 1));
          }
#pragma unroll
          for (int t = 0; t < TW; ++t) acc[t] = mfma_bf16(ahi[t][kb], vh, acc[t]);
          if constexpr (BF) {
            // bf16 storage: the one product per k block is the whole arithmetic
          } else if constexpr (ONE) {
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[t] = mfma_bf16(ahi[t][kb], vl, acc[t]);
#pragma unroll
            for (int t = 0; t < TW; ++t) acc[t] = mfma_bf16(alo[t][kb], vh, acc[t]);
          } else {
#pragma unroll
            for (int t = 0; t < TW; ++t) accc[t] = mfma_bf16(ahi[t][kb], vl, accc[t]);
#pragma unroll
            for (int t = 0; t < TW; ++t) accc[t] = mfma_bf16(alo[t][kb], vh, accc[t]);
          }
        }
        if constexpr (!ONE) {
#pragma unroll
          for (int t = 0; t < TW; ++t) acc[t] += accc[t];
        }
      } else {
      const float* drow = Dl + jb * kDLS + 4 * ku;
      f32x4 bnext = *reinterpret_cast<const f32x4*>(drow);   // B fragment of step s8+1 requested before the MFMAs of step s8
#pragma unroll                                               // (all eight up front pushed the kernel into scratch: 256 VGPRs + spills)
      for (int s8 = 0; s8 < 8; ++s8) {
        const f32x4 bv = bnext;
        if (s8 + 1 < 8) bnext = *reinterpret_cast<const f32x4*>(drow + 16 * (s8 + 1));
#pragma unroll
        for (int m = 0; m < 4; ++m)                          // round-robin over the tiles: consecutive MFMAs on different accumulators
#pragma unroll
          for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(areg[t][s8][m], bv[m], acc[t], 0, 0, 0);
      }
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) *reinterpret_cast<f32x4*>(Pl + jb * kPLS + 16 * (wv * TW + t) + 4 * ku) = acc[t];
      if (step > 0) TSG_TICK(5)                              // (timing builds: MFMAs issued and their results written to LDS)
      lds_barrier();
      if (step > 0) TSG_TICK(2)                              // MFMA + gather
      const unsigned gtag = ((unsigned)step >> rsh) & 1u;      // generation of slot step%4, carried in the low mantissa bit
      for (int i = 0; i < TW; ++i) {
        const int idx = tid + i * kThreads, pr = idx / hq, pc = idx % hq;
        const f32x4 v = *reinterpret_cast<const f32x4*>(Pl + pr * kPLS + 4 * pc);
        store_x_u4(prod_ptr(step & rmask, i), (u32x4){(__float_as_uint(v[0]) & ~1u) | gtag, (__float_as_uint(v[1]) & ~1u) | gtag,
                                                 (__float_as_uint(v[2]) & ~1u) | gtag, (__float_as_uint(v[3]) & ~1u) | gtag}, local);
      }
      if (step > 0) TSG_TICK(3)                              // partial stores issued
    }
  }
#ifdef TSG_LSTM_TIMING
  if (blockIdx.x == 0 && tid == 0)
    for (int i = 0; i < 6; ++i) sync[8 + i] = (unsigned)(tph[i] / (unsigned long long)(T - 2));
#endif
  if (dbias) {                                              // d(b_ih + b_hh)[d][gate*h + u] += sum over this workgroup's rows and all steps
    __syncthreads();
#pragma unroll
    for (int gate = 0; gate < 4; ++gate) Dl[row * kDLS + gate * 32 + ul] = dbsum[gate];
    __syncthreads();
    if (tid < 128) {
      float sgu = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) sgu += Dl[r * kDLS + tid];
      atomicAdd(dbias + (size_t)d * K + (tid >> 5) * h + us * 32 + (tid & 31), sgu);
    }
  }
}

}  // namespace
}  // namespace tsg

#include <atomic>
#include <mutex>
using namespace tsg;

// TSG_LSTM_PERSIST: unset = auto (persistent launch for sequences of >= 8 steps: 8.2 vs 17.3 us/step at
// [B=128,T=128,h=512], 7.4 vs 11.3 us/step at [64,20,512]), 0 = never, 1 = whenever the grid fits.
static std::atomic<int> g_persist{-2};   // -2: not decided yet (TSG_LSTM_PERSIST); tsg_lstm_set_persist overrides
static int persist_mode() {
  int v = g_persist.load(std::memory_order_relaxed);
  if (v == -2) { const char* e = getenv("TSG_LSTM_PERSIST"); v = e ? atoi(e) : -1; g_persist.store(v, std::memory_order_relaxed); }
  return v;
}
extern "C" int tsg_lstm_set_persist(int mode) { g_persist.store(mode < 0 ? -1 : (mode != 0), std::memory_order_relaxed); return 0; }
static std::atomic<int> g_l2x{-1};   // -1: not decided yet (TSG_LSTM_L2X, default on); tsg_lstm_set_l2_exchange overrides
static int l2_exchange() {        // TSG_LSTM_L2X=0: always write-through exchange stores (A/B measurements)
  int v = g_l2x.load(std::memory_order_relaxed);
  if (v < 0) { const char* e = getenv("TSG_LSTM_L2X"); v = e ? (atoi(e) != 0) : 1; g_l2x.store(v, std::memory_order_relaxed); }
  return v;
}
extern "C" int tsg_lstm_set_l2_exchange(int on) { g_l2x.store(on != 0, std::memory_order_relaxed); return 0; }

// Workgroups of a persistent kernel that can be co-resident: ONE per CU is counted (margin against over-reporting), cached
// per (device, kernel INSTANTIATION) -- the instantiations differ in LDS and register footprint, so an occupancy answer is never shared
// between them (a slot number used to stand for several kernels).  0 = the kernel cannot run persistently here.
// (`slot` is the callers' old cache index: unused.)
template <class K>
static int persist_capacity(int /*slot*/, K kern, int threads, size_t lds, int per_cu) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int> cache;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  // the LDS attribute is per (device, instantiation): requested on every call (a table hit after the first)
  hipError_t e1 = allow_lds(kern, lds);
  if (e1 != hipSuccess) return 0;
  const auto key = std::make_pair(dev, reinterpret_cast<const void*>(kern));
  {
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(key);
    if (it != cache.end()) return it->second;
  }
  int cus = 0, per = 0;
  e1 = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  if (e1 == hipSuccess) e1 = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per, kern, threads, lds);
  const int cap = (e1 == hipSuccess && per >= per_cu) ? cus * per_cu : 0;
  std::lock_guard<std::mutex> lock(mu);
  cache[key] = cap;
  return cap;
}
// Rows per launch when the B rows of a layer need more workgroups than fit: balanced chunks of whole 16-row slices, each
// with 2 * wgs_per_slice * slices <= cap workgroups.  0 = not even one slice fits.
static int persist_chunk_rows(int B, int wgs_per_slice, int cap) {
  const int max_slices = cap / (2 * wgs_per_slice);
  if (max_slices <= 0) return 0;
  const int slices = cdiv(B, 16);
  const int chunks = cdiv(slices, max_slices);
  return cdiv(slices, chunks) * 16;
}
// Grid of a persistent launch with `slices` 16-row batch slices: 2 * slices exchange groups of `wgs` workgroups each.  Workgroup i
// runs on XCD i % 8 and the kernels order their roles XCD-major (xcd_major_index), so a group sits on ONE XCD -- and takes the
// L2-local exchange -- when the group size divides the per-XCD share of the grid.  With fewer than 8 groups (at h = 512: up to 48
// rows -- the 16-pair per-GPU shards of BASELINE configs 3 / 4) the natural grid's share is smaller than a group and EVERY group
// straddles two XCDs; the grid is then PADDED to 8 * wgs: XCD x hosts group x, and the workgroups beyond the last group return at
// once (they never reach the start barrier, which counts the active ones only).  Backward step 3.50 -> 3.13 us (f32s) / 2.97 -> 2.69
// (bf16 storage) at [32, 512, 512], 3.31 -> 2.87 / 3.00 -> 2.48 at [40, 128, 512]; the forward does not change (profiles/r4/
// lstm_padded_grid_ab_v1.txt).  Not for 8 .. 15 groups that do not divide evenly (96 rows: 12 groups, 8 of them local on the natural
// grid): packing them two per XCD on six XCDs measured 4.68 -> 4.15 us in f32s but 3.53 -> 3.75 in bf16.
// TSG_LSTM_PAD=0: natural grids (A/B).  Correctness never depends on the placement: the groups verify it (grid_start).
static int persist_grid(int slices, int wgs, int cap) {
  static const bool pad = !(getenv("TSG_LSTM_PAD") && atoi(getenv("TSG_LSTM_PAD")) == 0);
  const int natural = 2 * slices * wgs, padded = 8 * wgs;
  return (pad && 2 * slices < 8 && padded <= cap) ? padded : natural;
}
static int launch_flags() {       // bit 0: L2-local exchange allowed; bit 1: inject a start-barrier timeout (tests)
  const char* e = getenv("TSG_LSTM_INJECT_TIMEOUT");
  // bit 3: two-slot partial-dh ring in the backward (default; TSG_LSTM_RING=4 = the four slots of rounds 1-3, A/B).  Four slots of 8.4 MB are
  // 33.5 MB -- more than the eight 4 MB L2s together -- so the ring fell out of the L2s it is exchanged through: 1.0 GB of re-fetches per
  // launch at [128, 128, 512] (PMC: 1502 -> 454 MB read), 4.49 -> 3.88 us per step (f32s), 4.05 -> 3.27 (bf16 storage)
  static const int ring2 = [] { const char* r = getenv("TSG_LSTM_RING"); return (r && atoi(r) == 4) ? 0 : 8; }();
  return (l2_exchange() ? 1 : 0) | ((e && atoi(e) != 0) ? 2 : 0) | ring2;
}
static bool persist_wanted(int T) { const int m = persist_mode(); return m > 0 || (m < 0 && T >= 8); }

static int lstm_fwd_impl(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* sync_ws, long long ws_bytes,
                         int B, int T, int h, int dtype, int batch_major, void* stream);

extern "C" int tsg_lstm_fwd(const void* Gx, const void* Whh, void* out, void* R, void* Cs, void* sync_ws,
                            int B, int T, int h, int dtype, void* stream) {
  return lstm_fwd_impl(Gx, nullptr, Whh, out, R, Cs, sync_ws, sync_ws ? kSyncBytes : 0, B, T, h, dtype, 0, stream);
}

extern "C" int tsg_lstm_fwd_bias(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* sync_ws,
                                 int B, int T, int h, int dtype, int batch_major, void* stream) {
  return lstm_fwd_impl(Gx, bias, Whh, out, R, Cs, sync_ws, sync_ws ? kSyncBytes : 0, B, T, h, dtype, batch_major, stream);
}

// workspace of tsg_lstm_fwd_ws: the sync words + the exchange ring [4 slots][2 * ceil(B/16) groups][16 rows][h] dwords (0: no ring for this h)
extern "C" long long tsg_lstm_fwd_ws_bytes(int B, int T, int h) {
  (void)T;
  if (B <= 0 || h <= 0 || h % 128 || h > kPersistMaxH) return 0;
  return kSyncBytes + 4LL * 2 * cdiv(B, 16) * 16 * h * (long long)sizeof(float);
}

extern "C" int tsg_lstm_fwd_ws(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* ws, long long ws_bytes,
                               int B, int T, int h, int dtype, int batch_major, void* stream) {
  if (ws && ws_bytes < kSyncBytes) return set_error(TSG_E_SHAPE, "tsg_lstm_fwd_ws: workspace of %lld bytes is smaller than TSG_LSTM_SYNC_BYTES", ws_bytes);
  return lstm_fwd_impl(Gx, bias, Whh, out, R, Cs, ws, ws ? ws_bytes : 0, B, T, h, dtype, batch_major, stream);
}

// Where the ring pays (profiles/r5/lstm_fwd_ring_*_v1.txt, one MI355X, us per step out-polling -> ring): at the full-chip grids -- [128, 128, 512]
// f32s 3.85-3.91 -> 3.81-3.83 batch-major, 3.76 -> 3.60 time-major; bf16 storage 3.75-3.77 -> 3.69-3.70 -- and, with bf16 storage, at the
// 16-pair shards of BASELINE configs 3 / 4 (<= 32 rows), where it makes the 4-wave workgroups possible (one A-tile per wave needs no second
// register set: 146 VGPRs): 3.26 -> 2.87 at [32, 512, 512], 3.16 -> 2.67 at [16, 128, 512].  It LOSES at half-chip grids ([64, ., 512]: 3.32 ->
// 3.49 f32s: one group per XCD, nothing contends for the L2 and the extra re-mark / `out` stores are pure cost) and in f32s at <= 32 rows (the
// 4-wave f32s ring kernel spills: 2.91 -> 3.32).  TSG_LSTM_XR=1 / 0: always / never (A/B).
static std::atomic<int> g_xr{-2};     // -2: not decided yet (TSG_LSTM_XR); -1 auto, 0 never, 1 always; tsg_lstm_set_ring overrides
static bool xr_wanted(int B, bool bf) {
  int v = g_xr.load(std::memory_order_relaxed);
  if (v == -2) { const char* e = getenv("TSG_LSTM_XR"); v = e ? (atoi(e) != 0) : -1; g_xr.store(v, std::memory_order_relaxed); }
  if (v >= 0) return v != 0;
  return B >= 96 || (bf && B <= 32);
}
extern "C" int tsg_lstm_set_ring(int mode) { g_xr.store(mode < 0 ? -1 : (mode != 0), std::memory_order_relaxed); return 0; }
// 64-unit workgroups (lstm_fwd_persist_w64_kernel: bf16 storage, h = 512, ring workspace): -1 automatic (more than 128 rows: one launch where the
// 32-unit kernel needs two), 0 never, 1 whenever possible
static std::atomic<int> g_w64{-2};
static bool w64_wanted(int B) {
  int v = g_w64.load(std::memory_order_relaxed);
  if (v == -2) { const char* e = getenv("TSG_LSTM_W64"); v = e ? (atoi(e) != 0) : -1; g_w64.store(v, std::memory_order_relaxed); }
  return v >= 0 ? v != 0 : B > 128;      // (up to 128 rows the 32-unit kernel with staged streams is faster: 2.19 vs 2.6 us per step)
}
extern "C" int tsg_lstm_set_wide(int mode) { g_w64.store(mode < 0 ? -1 : (mode != 0), std::memory_order_relaxed); return 0; }

template <int HJ, int MODE, int NW, bool XR>
static void launch_fwd_persist(int grid, size_t lds, hipStream_t st, const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs,
                               void* ws, int Bc, int B, int T, int h, int HLS, int bm, size_t seq, int c0) {
  typedef typename SeqT<MODE>::type GT;
  const size_t K8 = (size_t)8 * h, H2 = (size_t)2 * h;
  hipLaunchKernelGGL((lstm_fwd_persist_kernel<HJ, MODE, NW, XR>), dim3(grid), dim3(64 * NW), lds, st, (const GT*)Gx + seq * K8, (const float*)bias,
                     (const float*)Whh, (GT*)out + seq * H2, (GT*)R + (size_t)c0 * h * 4, (float*)Cs + (size_t)c0 * h, (unsigned*)ws, Bc, B, T, h,
                     HLS, launch_flags(), bm, error_sink(), XR ? (unsigned*)((char*)ws + kSyncBytes) : nullptr);
}

static int lstm_fwd_impl(const void* Gx, const void* bias, const void* Whh, void* out, void* R, void* Cs, void* sync_ws, long long ws_bytes,
                         int B, int T, int h, int dtype, int batch_major, void* stream) {
  const int bm = batch_major != 0;
  const char* fn = "tsg_lstm_fwd";
  if (bias && !aligned16(bias)) return set_error(TSG_E_ALIGN, "%s: bias %p is not 16-byte aligned", fn, bias);
  for (const void* p : {Gx, Whh, (const void*)out, (const void*)R, (const void*)Cs}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (sync_ws && !aligned16(sync_ws)) return set_error(TSG_E_ALIGN, "%s: workspace %p is not 16-byte aligned", fn, sync_ws);
  int rc = lstm_check(fn, B, T, h, dtype);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  // persistent path: weights stationary, one launch for all T steps -- needs every workgroup resident
  const bool bf = dtype == TSG_BF16;
  // bf16 storage exists in the persistent kernels only (hidden sizes 128 / 256 / 384 / 512): other shapes are reported as
  // unsupported and the caller runs them through the fp32-storage entry points
  if (bf && !(sync_ws && persist_mode() != 0 && h % 128 == 0 && h <= kPersistMaxH && T > 1))
    return set_error(TSG_E_SHAPE, "%s: dtype TSG_BF16 needs the persistent kernel (sync_ws given, TSG_LSTM_PERSIST != 0, T > 1, h in "
                     "{128,256,384,512}); got T=%d h=%d", fn, T, h);
  if (sync_ws && (bf || persist_wanted(T)) && h % 32 == 0 && h <= kPersistMaxH && T > 1) {
    // 16-unit workgroups of 4 waves (half the MFMA and gate work per workgroup and step, twice the workgroups): slower when they
    // have to share CUs (two per CU at 128 rows: 5.15 vs 4.59 us per step), faster when the chip is mostly idle -- up to 32 rows
    // (the 16-pair per-GPU shards of BASELINE configs 3 / 4): 3.01 -> 2.82 us per step at [32, 512, 512], 2.85 -> 2.68 at 16 rows,
    // no difference at 40 / 48 rows (profiles/r4/lstm_fwd_nw4_small_batch_ab_v1.txt).  TSG_LSTM_NW=4 / 8 forces one of them.
    static int nw_env = -1;
    if (nw_env < 0) { const char* e = getenv("TSG_LSTM_NW"); nw_env = e ? (atoi(e) == 4 ? 4 : 8) : 0; }
    const bool split = dtype == TSG_F32S;                  // other hidden sizes: the fp32 arithmetic (more accurate, slower)
    // the exchange ring (XR kernels): the caller's workspace holds it, hidden sizes 128 / 256 / 384 / 512
    const long long need = tsg_lstm_fwd_ws_bytes(B, T, h);
    if (bf && h == 512 && need > 0 && ws_bytes >= need && w64_wanted(B) && (long long)T * B * 8 * h < (1LL << 31)) {
      // bf16 storage at the full-chip sizes: 64-unit workgroups, 8 per exchange group, on half the CUs (see the kernel)
      auto kw = lstm_fwd_persist_w64_kernel<32>;
      const size_t wlds = sizeof(float) * ((size_t)kSlabFloats + 16 * 65 + 16 * (4 * 32 + 4) + 16 * (64 * 2 + 8) + 16 * (64 + 4) + 4);
      const int capw = persist_capacity(3, kw, 512, wlds, 1);
      const int rowsw = persist_chunk_rows(B, h / 64, capw);
      if (rowsw > 0) {
        const size_t K8 = (size_t)8 * h, H2 = (size_t)2 * h;
        for (int c0 = 0; c0 < B; c0 += rowsw) {
          const int Bc = B - c0 < rowsw ? B - c0 : rowsw;
          const size_t seq = bm ? (size_t)c0 * T : (size_t)c0;
          hipError_t e = zero_async(sync_ws, kSyncBytes, st);
          if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
          hipLaunchKernelGGL(kw, dim3(persist_grid(cdiv(Bc, 16), h / 64, capw)), dim3(512), wlds, st, (const lstm_bf16*)Gx + seq * K8, (const float*)bias,
                             (const float*)Whh, (lstm_bf16*)out + seq * H2, (lstm_bf16*)R + (size_t)c0 * h * 4, (float*)Cs + (size_t)c0 * h,
                             (unsigned*)sync_ws, Bc, B, T, h, launch_flags(), bm, error_sink(), (unsigned*)((char*)sync_ws + kSyncBytes));
        }
        return check_launch(fn);
      }
    }
    const bool xr = need > 0 && ws_bytes >= need && xr_wanted(B, bf);
    const int NW = ((nw_env == 4 || (nw_env == 0 && B <= 32)) && h == 512 && (split || (bf && xr))) ? 4 : 8;
    const int HLS = kPersistMaxH + 8;                      // fixed: the prefetch above may read (never use) columns up to kPersistMaxH
    const size_t plds = sizeof(float) * ((size_t)kSlabFloats + 16 * 36 + 2 * 16 * 128 + 16 * (136 + 36) + 4);     // slab, h tile, Gx / R / Cs tiles (fp32, 32 units: the largest)
    static_assert(kSlabFloats >= 16 * (kPersistMaxH + 8), "slab region holds the fp32 slab too");
    typedef void (*Launch)(int, size_t, hipStream_t, const void*, const void*, const void*, void*, void*, void*, void*, int, int, int, int, int, int, size_t, int);
    Launch go = nullptr;
    int cap = 0;
#define TSG_FWD_PICK(HJ_, MODE_, NW_, XR_, SLOT_)                                                                          \
    { go = launch_fwd_persist<HJ_, MODE_, NW_, XR_>;                                                                        \
      cap = persist_capacity(SLOT_, lstm_fwd_persist_kernel<HJ_, MODE_, NW_, XR_>, 64 * NW_, plds, NW_ == 4 ? 2 : 1); }
#define TSG_FWD_PICK_H(MODE_, XR_, SLOT_)                                                                                  \
    { if (h == 512) TSG_FWD_PICK(32, MODE_, 8, XR_, SLOT_) else if (h == 384) TSG_FWD_PICK(24, MODE_, 8, XR_, SLOT_)        \
      else if (h == 256) TSG_FWD_PICK(16, MODE_, 8, XR_, SLOT_) else if (h == 128) TSG_FWD_PICK(8, MODE_, 8, XR_, SLOT_) }
    if (NW == 4) {
      if (bf) TSG_FWD_PICK(32, 2, 4, true, 1)
      else if (xr) TSG_FWD_PICK(32, 1, 4, true, 1)
      else TSG_FWD_PICK(32, 1, 4, false, 1)
    } else if (bf) {
      if (xr) TSG_FWD_PICK_H(2, true, 3) else TSG_FWD_PICK_H(2, false, 3)
    } else if (split) {
      if (xr) TSG_FWD_PICK_H(1, true, 0) else TSG_FWD_PICK_H(1, false, 0)
    } else {
      if (xr) TSG_FWD_PICK_H(0, true, 0) else TSG_FWD_PICK_H(0, false, 0)
    }
    if (!go) TSG_FWD_PICK(0, 0, 8, false, 0)               // other hidden sizes (multiples of 32): the generic fp32 kernel, polling `out`
#undef TSG_FWD_PICK_H
#undef TSG_FWD_PICK
    // more rows than co-resident workgroups allow (B = 256 at h = 512: grid 512 on 256 CUs): the rows are independent, so the
    // layer runs as consecutive launches over balanced row chunks (pointer offsets; Bs = B keeps the tensors' strides)
    const int rows = persist_chunk_rows(B, h / (4 * NW), cap);
    if (bf && rows <= 0) return set_error(TSG_E_SHAPE, "%s: the persistent kernel cannot be co-resident on this device (TSG_BF16)", fn);
    if (rows > 0) {
      for (int c0 = 0; c0 < B; c0 += rows) {
        const int Bc = B - c0 < rows ? B - c0 : rows;
        const size_t seq = bm ? (size_t)c0 * T : (size_t)c0;              // first sequence row of the chunk
        hipError_t e = zero_async(sync_ws, kSyncBytes, st);
        if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
        go(persist_grid(cdiv(Bc, 16), h / (4 * NW), cap), plds, st, Gx, bias, Whh, out, R, Cs, sync_ws, Bc, B, T, h, HLS, bm, seq, c0);
      }
      return check_launch(fn);
    }
  }
  const int WS = roundup(h, 64) + 4;                     // = 4 mod 64
  const size_t lds = sizeof(float) * ((size_t)16 * WS + (size_t)kWaves * NBUF * 16 * HS);
  if (lds > (size_t)kLdsBytes) return set_error(TSG_E_LDS, "%s: h=%d needs %zu B of LDS", fn, h, lds);
  auto kern = lstm_fwd_step_kernel;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  const int grid = 2 * cdiv(h, 4);
  for (int step = 0; step < T; ++step)
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kThreads), lds, st, (const float*)Gx, (const float*)bias, (const float*)Whh, (float*)out,
                       (float*)R, (float*)Cs, B, T, h, step, WS, bm);
  return check_launch(fn);
}

static int lstm_bwd_steps(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                         void* dG, void* dC_ws, int B, int T, int h, int dtype, int bm, void* stream);

extern "C" int tsg_lstm_bwd(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                            void* dG, void* dC_ws, int B, int T, int h, int dtype, void* stream) {
  return lstm_bwd_steps(WhhT, R, Cs, dOut, dHn, dG, dC_ws, B, T, h, dtype, 0, stream);
}

static int lstm_bwd_steps(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                          void* dG, void* dC_ws, int B, int T, int h, int dtype, int bm, void* stream) {
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
                       (const float*)dOut, (const float*)dHn, (float*)dG, (float*)dC_ws, B, T, h, step, bm);
  return check_launch(fn);
}

extern "C" long long tsg_lstm_bwd_ws_bytes(int B, int T, int h) {
  (void)T;
  if (B <= 0 || h <= 0 || h % 128 || h > kPersistMaxH) return 0;
  const long long nus = h / 32, bslices = cdiv(B, 16);
  return kSyncBytes + 4LL * 2 * bslices * nus * nus * 512 * (long long)sizeof(float);
}

static constexpr size_t kBwd2Lds = sizeof(float) * ((size_t)kDlFloats + 16 * kPLS + 4 * 16 * kQLS + kObBufs * kObDw + 4);
static_assert(kDlFloats >= 16 * kDLS, "dG tile region holds the fp32 tile too");
static int bwd_persist_capacity() { return persist_capacity(2, lstm_bwd_persist2_kernel<4, 0>, kThreads, kBwd2Lds, 1); }

extern "C" int tsg_lstm_bwd_ws_persistent(int B, int T, int h, long long ws_bytes) {
  const long long need = tsg_lstm_bwd_ws_bytes(B, T, h);
  return need > 0 && ws_bytes >= need && T > 1 && persist_wanted(T) && persist_chunk_rows(B, h / 32, bwd_persist_capacity()) > 0;
}

extern "C" int tsg_lstm_bwd_ws(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                               void* dG, void* dC_ws, void* ws, long long ws_bytes, void* dbias, int B, int T, int h,
                               int dtype, void* stream) {
  return tsg_lstm_bwd_ws_layout(WhhT, R, Cs, dOut, dHn, dG, dC_ws, ws, ws_bytes, dbias, B, T, h, dtype, 0, stream);
}

extern "C" int tsg_lstm_bwd_ws_layout(const void* WhhT, const void* R, const void* Cs, const void* dOut, const void* dHn,
                                      void* dG, void* dC_ws, void* ws, long long ws_bytes, void* dbias, int B, int T, int h,
                                      int dtype, int batch_major, void* stream) {
  const char* fn = "tsg_lstm_bwd_ws";
  const int bm = batch_major != 0;
  const long long need = tsg_lstm_bwd_ws_bytes(B, T, h);
  const bool bf = dtype == TSG_BF16;
  const bool bf_ok = bf && ws && aligned16(ws) && need > 0 && ws_bytes >= need && T > 1 && persist_mode() != 0 &&
                     persist_chunk_rows(B, h / 32, bwd_persist_capacity()) > 0;
  if (bf && !bf_ok)
    return set_error(TSG_E_SHAPE, "%s: dtype TSG_BF16 needs the persistent kernel (ring workspace of tsg_lstm_bwd_ws_bytes, T > 1, h in "
                     "{128,256,384,512}, TSG_LSTM_PERSIST != 0); got T=%d h=%d ws=%lld", fn, T, h, ws_bytes);
  if (bf_ok || (ws && aligned16(ws) && tsg_lstm_bwd_ws_persistent(B, T, h, ws_bytes))) {
    for (const void* p : {WhhT, R, Cs, dOut, (const void*)dG}) {
      if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
      if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
    }
    int rc = lstm_check(fn, B, T, h, dtype);
    if (rc) return rc;
    auto st = static_cast<hipStream_t>(stream);
    hipError_t e = hipSuccess;                               // (dbias is zeroed together with the first chunk's sync words: one node)
    const bool split = dtype == TSG_F32S;
    const int rows = persist_chunk_rows(B, h / 32, bwd_persist_capacity());      // > 0 (tsg_lstm_bwd_ws_persistent)
    const size_t K8 = (size_t)8 * h, H2 = (size_t)2 * h;
    if (bf) {
      auto pb = h == 512 ? lstm_bwd_persist2_kernel<4, 2> : h == 384 ? lstm_bwd_persist2_kernel<3, 2>
              : h == 256 ? lstm_bwd_persist2_kernel<2, 2> : lstm_bwd_persist2_kernel<1, 2>;
      e = allow_lds(pb, kBwd2Lds);
      if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
      for (int c0 = 0; c0 < B; c0 += rows) {
        const int Bc = B - c0 < rows ? B - c0 : rows;
        const size_t seq = bm ? (size_t)c0 * T : (size_t)c0;
        e = zero3_async(ws, kSyncBytes, c0 == 0 ? dbias : nullptr, sizeof(float) * 8 * h, nullptr, 0, st);
        if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
        hipLaunchKernelGGL(pb, dim3(persist_grid(cdiv(Bc, 16), h / 32, bwd_persist_capacity())), dim3(kThreads), kBwd2Lds, st, (const float*)WhhT,
                           (const lstm_bf16*)R + (size_t)c0 * h * 4, (const float*)Cs + (size_t)c0 * h, (const lstm_bf16*)dOut + seq * H2,
                           dHn ? (const float*)dHn + (size_t)c0 * h : nullptr, (lstm_bf16*)dG + seq * K8, (float*)((char*)ws + kSyncBytes),
                           (unsigned*)ws, (float*)dbias, Bc, B, T, h, launch_flags(), bm, error_sink());
      }
      return check_launch(fn);
    }
    auto pk = h == 512 ? (split ? lstm_bwd_persist2_kernel<4, 1> : lstm_bwd_persist2_kernel<4, 0>)
            : h == 384 ? (split ? lstm_bwd_persist2_kernel<3, 1> : lstm_bwd_persist2_kernel<3, 0>)
            : h == 256 ? (split ? lstm_bwd_persist2_kernel<2, 1> : lstm_bwd_persist2_kernel<2, 0>)
            : (split ? lstm_bwd_persist2_kernel<1, 1> : lstm_bwd_persist2_kernel<1, 0>);
    e = allow_lds(pk, kBwd2Lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    for (int c0 = 0; c0 < B; c0 += rows) {                   // row chunks as in the forward; the ring is reused, dbias accumulates
      const int Bc = B - c0 < rows ? B - c0 : rows;
      const size_t seq = bm ? (size_t)c0 * T : (size_t)c0;
      e = zero3_async(ws, kSyncBytes, c0 == 0 ? dbias : nullptr, sizeof(float) * 8 * h, nullptr, 0, st);
      if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
      hipLaunchKernelGGL(pk, dim3(persist_grid(cdiv(Bc, 16), h / 32, bwd_persist_capacity())), dim3(kThreads), kBwd2Lds, st, (const float*)WhhT,
                         (const float*)R + (size_t)c0 * h * 4, (const float*)Cs + (size_t)c0 * h, (const float*)dOut + seq * H2,
                         dHn ? (const float*)dHn + (size_t)c0 * h : nullptr, (float*)dG + seq * K8, (float*)((char*)ws + kSyncBytes),
                         (unsigned*)ws, (float*)dbias, Bc, B, T, h, launch_flags(), bm, error_sink());
    }
    return check_launch(fn);
  }
  (void)need;
  return lstm_bwd_steps(WhhT, R, Cs, dOut, dHn, dG, dC_ws, B, T, h, dtype, bm, stream);
}
