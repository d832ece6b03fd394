// K3 -- boundary-score head for gfx950: the part of VideoSentenceConcat + MLP_predictor
// (reference grounding/model/components/CrossModalInteraction.py:44-47, SpanPredictor.py:71-85,
// gate from SpanGroundMatchDisc.py:86) that follows the video-half GEMM.
//
// The reference concatenates [video_t | sent] (a [B,T,Dv+Ds] tensor) and runs two Linear(Dv+Ds,Hm).
// Split W1 = [W1v | W1s]: the sentence half is a per-sample row cs[b] = W1s sent[b] (T-invariant),
// so   z[b,t,:] = g[b,t] * (y[b,t,:] + cs[b,:]) + b1,   y = W1v video   (g = 1 without the GMD gate)
//      l[b,t]   = w2 . tanh(z) + b2 ;  mask_logits ;  p = softmax_t(l)
// with the start and end branches stacked along the hidden axis (J = 2*Hm columns: [start | end]).
// The concat tensor never exists.
//
// Workgroup = (batch item, 32 clips) so that [64,128] still covers the chip; a wave owns 4 clip rows
// (all requested up front), lanes own hidden columns (coalesced float4 rows of y), the two dot
// products are wave reductions; a second tiny kernel does the T-softmax in LDS.
// HBM-bound: reads y once (T*J*4 B per pair), writes 2T probabilities.
// (Round 3 measured a ONE-launch forward -- a 16-wave workgroup per item keeping the logits in LDS for the softmax -- at 18 us per
// launch against 9.6 + 3 us for the two kernels at [64,128,256]: 64 workgroups cannot keep enough of the 17 MB in flight.  Dropped.)
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kThreads = 512;
constexpr int kWaves = kThreads / kWave;
constexpr int kMaxJ4 = 4;            // J <= 64 lanes * 4 * kMaxJ4 = 1024 hidden columns (2*Hm)

__device__ __forceinline__ int wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }

__device__ __forceinline__ float tanh_fast(float x) {        // 1 - 2/(exp(2x)+1), exact to ~2 ulp
  const float e = fast_exp2(clampf(x, -44.f, 44.f) * k2Log2e);
  return 1.f - 2.f * fast_rcp(e + 1.f);
}

// mask_logits (attention.py:129-133): l*m + (-1e30)*(1-m), m cast to float
__device__ __forceinline__ float mask_logit(float l, float m) { return l * m + (-1e30f) * (1.f - m); }

// block-wide softmax over T values held in LDS (in place); returns nothing, all threads take part
__device__ void block_softmax(float* v, int T, float* scratch) {
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  float m = -INFINITY;
  for (int t = tid; t < T; t += kThreads) m = fmaxf(m, v[t]);
  m = wave_allmax(m);
  if (lane == 0) scratch[wv] = m;
  __syncthreads();
  m = scratch[0];
#pragma unroll
  for (int u = 1; u < kWaves; ++u) m = fmaxf(m, scratch[u]);
  __syncthreads();
  float s = 0.f;
  for (int t = tid; t < T; t += kThreads) {
    const float e = __expf(v[t] - m);
    v[t] = e;
    s += e;
  }
  s = wave_allsum(s);
  if (lane == 0) scratch[wv] = s;
  __syncthreads();
  s = 0.f;
#pragma unroll
  for (int u = 0; u < kWaves; ++u) s += scratch[u];
  const float inv = 1.f / s;
  for (int t = tid; t < T; t += kThreads) v[t] *= inv;
  __syncthreads();
}

// ---- forward, kernel 1: raw (masked) logits.  grid = B * ceil(T/TR): 32 clip rows per workgroup so
// that a [64,128] problem still covers all 256 CUs; each wave owns 4 rows and requests all of them
// before the first tanh (the kernel is latency-bound otherwise: one 2 KiB row per iteration).
constexpr int TR = 32;                 // rows per workgroup
constexpr int RW = TR / kWaves;        // rows per wave

// ST: storage type of y (and dy in the backward): float or bf16_t (dtype TSG_BF16).  cs, the parameters, gate, the
// probabilities and every accumulator stay fp32 (they are [B,T] / [B,J]-sized).
template <typename ST>
__global__ __launch_bounds__(kThreads) void boundary_logits_kernel(
    const ST* __restrict__ y, const float* __restrict__ cs, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ b2, const float* __restrict__ gate,
    const int* __restrict__ mask, float* __restrict__ ls, float* __restrict__ le, int B, int T, int Hm, int tiles) {
  const int lane = threadIdx.x & 63, wv = wave_id();
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * TR + wv * RW;
  const int J = 2 * Hm;
  float4 yv[RW][kMaxJ4];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < kMaxJ4; ++i) {
      const int j = i * 256 + lane * 4;
      yv[r][i] = (t0 + r < T && j < J) ? ld4(y + ((size_t)b * T + t0 + r) * J + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float c[kMaxJ4][4], bb[kMaxJ4][4], ww[kMaxJ4][4];
#pragma unroll
  for (int i = 0; i < kMaxJ4; ++i) {
    const int j = i * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = j + q < J;
      c[i][q] = ok ? cs[(size_t)b * J + j + q] : 0.f;
      bb[i][q] = ok ? b1[j + q] : 0.f;
      ww[i][q] = ok ? w2[j + q] : 0.f;
    }
  }
  const float b2s = b2[0], b2e = b2[1];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int t = t0 + r;
    if (t >= T) break;
    const float g = gate ? gate[(size_t)b * T + t] : 1.f;
    float as = 0.f, ae = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxJ4; ++i) {
      const int j = i * 256 + lane * 4;
      const float v4[4] = {yv[r][i].x, yv[r][i].y, yv[r][i].z, yv[r][i].w};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float u = tanh_fast(fmaf(g, v4[q] + c[i][q], bb[i][q]));     // ww = 0 beyond J
        if (j + q < Hm) as = fmaf(ww[i][q], u, as); else ae = fmaf(ww[i][q], u, ae);
      }
    }
    as = wave_allsum(as); ae = wave_allsum(ae);
    if (lane == 0) {
      float l0 = as + b2s, l1 = ae + b2e;
      if (mask) { const float m = (float)mask[(size_t)b * T + t]; l0 = mask_logit(l0, m); l1 = mask_logit(l1, m); }
      ls[(size_t)b * T + t] = l0; le[(size_t)b * T + t] = l1;
    }
  }
}

// ---- forward, kernel 2: softmax over T, in place (grid = B) ----
__global__ __launch_bounds__(kThreads) void boundary_softmax_kernel(float* __restrict__ ps, float* __restrict__ pe, int T) {
  extern __shared__ float lds[];
  float* ls = lds; float* le = lds + T; float* scratch = lds + 2 * T;
  const int b = blockIdx.x;
  for (int t = threadIdx.x; t < T; t += kThreads) { ls[t] = ps[(size_t)b * T + t]; le[t] = pe[(size_t)b * T + t]; }
  __syncthreads();
  block_softmax(ls, T, scratch);
  block_softmax(le, T, scratch);
  for (int t = threadIdx.x; t < T; t += kThreads) { ps[(size_t)b * T + t] = ls[t]; pe[(size_t)b * T + t] = le[t]; }
}

// ---- backward, kernel 1 (grid = B): dl = p*(dp - <p,dp>)*m into the workspace, db2 partials ----
__global__ __launch_bounds__(kThreads) void boundary_dl_kernel(
    const int* __restrict__ mask, const float* __restrict__ ps, const float* __restrict__ pe,
    const float* __restrict__ dps, const float* __restrict__ dpe, float* __restrict__ dl, float* __restrict__ db2p,
    float* __restrict__ z0, float* __restrict__ z1, float* __restrict__ z2, int J, int B, int T) {
  __shared__ float scratch[2 * kWaves];
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id(), b = blockIdx.x;
  // this batch item's rows of the three atomic accumulators of kernel 2 (instead of three memset launches)
  for (int j = tid; j < J; j += kThreads) { z0[(size_t)b * J + j] = 0.f; z1[(size_t)b * J + j] = 0.f; z2[(size_t)b * J + j] = 0.f; }
  float d0 = 0.f, d1 = 0.f;
  for (int t = tid; t < T; t += kThreads) {
    d0 = fmaf(ps[(size_t)b * T + t], dps[(size_t)b * T + t], d0);
    d1 = fmaf(pe[(size_t)b * T + t], dpe[(size_t)b * T + t], d1);
  }
  d0 = wave_allsum(d0); d1 = wave_allsum(d1);
  if (lane == 0) { scratch[wv] = d0; scratch[kWaves + wv] = d1; }
  __syncthreads();
  d0 = 0.f; d1 = 0.f;
#pragma unroll
  for (int u = 0; u < kWaves; ++u) { d0 += scratch[u]; d1 += scratch[kWaves + u]; }
  __syncthreads();
  float sb0 = 0.f, sb1 = 0.f;
  for (int t = tid; t < T; t += kThreads) {
    const float m = mask ? (float)mask[(size_t)b * T + t] : 1.f;
    const float a0 = ps[(size_t)b * T + t] * (dps[(size_t)b * T + t] - d0) * m;
    const float a1 = pe[(size_t)b * T + t] * (dpe[(size_t)b * T + t] - d1) * m;
    dl[((size_t)b * T + t) * 2] = a0; dl[((size_t)b * T + t) * 2 + 1] = a1;
    sb0 += a0; sb1 += a1;
  }
  sb0 = wave_allsum(sb0); sb1 = wave_allsum(sb1);
  if (lane == 0) { scratch[wv] = sb0; scratch[kWaves + wv] = sb1; }
  __syncthreads();
  if (tid == 0) {
    float s0 = 0.f, s1 = 0.f;
    for (int u = 0; u < kWaves; ++u) { s0 += scratch[u]; s1 += scratch[kWaves + u]; }
    db2p[(size_t)b * 2] = s0; db2p[(size_t)b * 2 + 1] = s1;
  }
}

// ---- backward, kernel 2 (grid = B * ceil(T/TR)):  dz = dl*w2*(1-u^2);  dy = g*dz;
// dcs[b,:] += sum_t g*dz;  db1p[b,:] += sum_t dz;  dw2p[b,:] += sum_t dl*u  (atomics: ceil(T/32) adders
// per address, buffers zeroed by kernel 1);  dgate[b,t] = sum_j dz*(y+cs). ----
template <typename ST>
__global__ __launch_bounds__(kThreads) void boundary_bwd_kernel(
    const ST* __restrict__ y, const float* __restrict__ cs, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ gate, const float* __restrict__ dl,
    ST* __restrict__ dy, float* __restrict__ dcs, float* __restrict__ db1p,
    float* __restrict__ dw2p, float* __restrict__ dgate, int B, int T, int Hm, int tiles) {
  extern __shared__ float red[];                 // [kWaves][3][J]
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * TR + wv * RW;
  const int J = 2 * Hm;
  float4 yv[RW][kMaxJ4];
#pragma unroll
  for (int r = 0; r < RW; ++r)
#pragma unroll
    for (int i = 0; i < kMaxJ4; ++i) {
      const int j = i * 256 + lane * 4;
      yv[r][i] = (t0 + r < T && j < J) ? ld4(y + ((size_t)b * T + t0 + r) * J + j) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  float c[kMaxJ4][4], bb[kMaxJ4][4], ww[kMaxJ4][4];
  float acs[kMaxJ4][4], ab1[kMaxJ4][4], aw2[kMaxJ4][4];
#pragma unroll
  for (int i = 0; i < kMaxJ4; ++i) {
    const int j = i * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool ok = j + q < J;
      c[i][q] = ok ? cs[(size_t)b * J + j + q] : 0.f;
      bb[i][q] = ok ? b1[j + q] : 0.f;
      ww[i][q] = ok ? w2[j + q] : 0.f;
      acs[i][q] = 0.f; ab1[i][q] = 0.f; aw2[i][q] = 0.f;
    }
  }
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    const int t = t0 + r;
    if (t >= T) break;
    const float g = gate ? gate[(size_t)b * T + t] : 1.f;
    const float dl0 = dl[((size_t)b * T + t) * 2], dl1 = dl[((size_t)b * T + t) * 2 + 1];
    ST* dyr = dy + ((size_t)b * T + t) * J;
    float dg = 0.f;
#pragma unroll
    for (int i = 0; i < kMaxJ4; ++i) {
      const int j = i * 256 + lane * 4;
      if (j < J) {
        const float v4[4] = {yv[r][i].x, yv[r][i].y, yv[r][i].z, yv[r][i].w};
        float o[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float pre = v4[q] + c[i][q];
          const float u = tanh_fast(fmaf(g, pre, bb[i][q]));
          const float dlv = (j + q < Hm) ? dl0 : dl1;
          const float dz = dlv * ww[i][q] * (1.f - u * u);
          o[q] = g * dz;
          acs[i][q] += o[q];
          ab1[i][q] += dz;
          aw2[i][q] = fmaf(dlv, u, aw2[i][q]);
          dg = fmaf(dz, pre, dg);
        }
        st4(dyr + j, make_float4(o[0], o[1], o[2], o[3]));
      }
    }
    if (dgate) {
      dg = wave_allsum(dg);
      if (lane == 0) dgate[(size_t)b * T + t] = dg;
    }
  }
  // cross-wave sums of the three per-column accumulators, then one atomic per column and workgroup
#pragma unroll
  for (int i = 0; i < kMaxJ4; ++i) {
    const int j = i * 256 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (j + q < J) {
        red[(wv * 3 + 0) * J + j + q] = acs[i][q];
        red[(wv * 3 + 1) * J + j + q] = ab1[i][q];
        red[(wv * 3 + 2) * J + j + q] = aw2[i][q];
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < 3 * J; idx += kThreads) {
    const int which = idx / J, j = idx % J;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < kWaves; ++u) s += red[(u * 3 + which) * J + j];
    float* dst = which == 0 ? dcs : (which == 1 ? db1p : dw2p);
    atomicAdd(dst + (size_t)b * J + j, s);
  }
}

// ---- backward, ONE launch (tsg_boundary_score_bwd_ws).  Workgroup = (batch item, 128 hidden columns), ALL T rows: the per-column
// sums over T (dcs, db1, dw2) are complete inside the workgroup -- direct stores, no float atomics, nothing to zero -- and so is db2
// (column group 0 writes it).  Every workgroup recomputes the two softmax dot products <p, dp> and dl for the T rows (no dl pass).
// What crosses workgroups is the per-ROW sum dgate[b,t] over all J columns (GMD gate only): each column group writes its partial
// row (agent-scope write-through stores) and takes a ticket from the item's counter; the group that draws the last ticket adds the
// partial rows in group order (deterministic) and puts the counter back to 0 -- the state the NEXT call on the stream expects: the
// counters are zeroed once, when the caller creates the workspace.  Nobody waits for anybody: no co-residency requirement.
// A wave covers two rows per instruction (32 lanes x 4 columns each); 8 row pairs per wave are in flight (at T = 128: the whole tile).
constexpr int kCG = 128;                         // columns per workgroup
constexpr int kBR = 8;                           // row pairs in flight per wave
__device__ __forceinline__ void store_agent(float* p, float v) {          // write-through: visible to every XCD once acknowledged
  asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float4 load_agent_x4(const float* p0, const float* p1, const float* p2, const float* p3) {
  float4 v;                                                               // four independent device-coherent loads, one wait
  asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\t"
               "global_load_dword %2, %6, off sc1\n\tglobal_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v.x), "=&v"(v.y), "=&v"(v.z), "=&v"(v.w) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  return v;
}

template <typename ST>
__global__ __launch_bounds__(kThreads) void boundary_bwd_one_kernel(
    const ST* __restrict__ y, const float* __restrict__ cs, const float* __restrict__ b1,
    const float* __restrict__ w2, const float* __restrict__ gate, const int* __restrict__ mask,
    const float* __restrict__ ps, const float* __restrict__ pe, const float* __restrict__ dps, const float* __restrict__ dpe,
    ST* __restrict__ dy, float* __restrict__ dcs, float* __restrict__ db1p, float* __restrict__ dw2p, float* __restrict__ db2p,
    float* __restrict__ dgate, float* part, unsigned* cnt, int B, int T, int Hm, int cgs) {
  extern __shared__ float lds[];
  float* dl = lds;                               // [T][2]
  float* red = lds + 2 * (size_t)T;              // [2 kWaves][3][kCG]
  __shared__ float sc[4 * kWaves];
  __shared__ unsigned last_one;
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int half = lane >> 5, l32 = lane & 31;
  const int b = blockIdx.x / cgs, cg = blockIdx.x % cgs;
  const int J = 2 * Hm;
  const int col = cg * kCG + l32 * 4;
  const bool cok = col < J;                      // J % 4 == 0: a lane's four columns exist together
  const int r0 = 2 * wv + half;                  // this half wave's rows: r0 + 16 k
  typedef typename Raw4T<ST>::type Raw4;
  const ST* yb = y + (size_t)b * T * J + (cok ? col : 0);
  Raw4 yv[kBR];
#pragma unroll
  for (int u = 0; u < kBR; ++u) yv[u] = ldraw4(yb + (size_t)min(r0 + 16 * u, T - 1) * J);      // clamped: never used beyond T

  // <p, dp> of both branches over the item's T clips, then dl of every row (LDS) and the db2 sums
  float d0 = 0.f, d1 = 0.f;
  for (int t = tid; t < T; t += kThreads) {
    d0 = fmaf(ps[(size_t)b * T + t], dps[(size_t)b * T + t], d0);
    d1 = fmaf(pe[(size_t)b * T + t], dpe[(size_t)b * T + t], d1);
  }
  d0 = wave_allsum(d0); d1 = wave_allsum(d1);
  if (lane == 0) { sc[wv] = d0; sc[kWaves + wv] = d1; }
  __syncthreads();
  d0 = 0.f; d1 = 0.f;
#pragma unroll
  for (int u = 0; u < kWaves; ++u) { d0 += sc[u]; d1 += sc[kWaves + u]; }
  float s0 = 0.f, s1 = 0.f;
  for (int t = tid; t < T; t += kThreads) {
    const float m = mask ? (float)mask[(size_t)b * T + t] : 1.f;
    const float a0 = ps[(size_t)b * T + t] * (dps[(size_t)b * T + t] - d0) * m;
    const float a1 = pe[(size_t)b * T + t] * (dpe[(size_t)b * T + t] - d1) * m;
    dl[2 * t] = a0; dl[2 * t + 1] = a1;
    s0 += a0; s1 += a1;
  }
  if (cg == 0) {                                 // workgroup-uniform
    s0 = wave_allsum(s0); s1 = wave_allsum(s1);
    if (lane == 0) { sc[2 * kWaves + wv] = s0; sc[3 * kWaves + wv] = s1; }
  }
  float c[4], bb[4], ww[4], acs[4], ab1[4], aw2[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    c[q] = cok ? cs[(size_t)b * J + col + q] : 0.f;
    bb[q] = cok ? b1[col + q] : 0.f;
    ww[q] = cok ? w2[col + q] : 0.f;
    acs[q] = 0.f; ab1[q] = 0.f; aw2[q] = 0.f;
  }
  __syncthreads();                               // dl, sc
  if (cg == 0 && tid == 0) {
    float t0 = 0.f, t1 = 0.f;
    for (int u = 0; u < kWaves; ++u) { t0 += sc[2 * kWaves + u]; t1 += sc[3 * kWaves + u]; }
    db2p[(size_t)b * 2] = t0; db2p[(size_t)b * 2 + 1] = t1;
  }
  const bool many = cgs > 1;
  float* mypart = part + ((size_t)b * cgs + cg) * T;
  for (int k0 = 0; k0 * 16 < T; k0 += kBR) {
#pragma unroll
    for (int u = 0; u < kBR; ++u) {
      const int t = r0 + 16 * (k0 + u);
      const float4 v = cvt4(yv[u]);
      yv[u] = ldraw4(yb + (size_t)min(t + 16 * kBR, T - 1) * J);          // refill the slot (unconditional, clamped)
      if (t < T) {                               // uniform per half wave
        const float g = gate ? gate[(size_t)b * T + t] : 1.f;
        const float dl0 = dl[2 * t], dl1 = dl[2 * t + 1];
        const float v4[4] = {v.x, v.y, v.z, v.w};
        float o[4], dg = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float pre = v4[q] + c[q];
          const float uu = tanh_fast(fmaf(g, pre, bb[q]));
          const float dlv = (col + q < Hm) ? dl0 : dl1;
          const float dz = dlv * ww[q] * (1.f - uu * uu);                  // ww = 0 beyond J
          o[q] = g * dz;
          acs[q] += o[q];
          ab1[q] += dz;
          aw2[q] = fmaf(dlv, uu, aw2[q]);
          dg = fmaf(dz, pre, dg);
        }
        if (cok) st4(dy + ((size_t)b * T + t) * J + col, make_float4(o[0], o[1], o[2], o[3]));
        if (dgate) {
          dg += __shfl_xor(dg, 16, 64); dg += __shfl_xor(dg, 8, 64); dg += __shfl_xor(dg, 4, 64);
          dg += __shfl_xor(dg, 2, 64); dg += __shfl_xor(dg, 1, 64);          // over the 32 lanes of the row
          if (l32 == 0) {
            if (many) store_agent(mypart + t, dg); else dgate[(size_t)b * T + t] = dg;
          }
        }
      }
    }
  }
  // per-column sums over the 16 row slots of the workgroup, then straight to the outputs
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    red[((2 * wv + half) * 3 + 0) * kCG + l32 * 4 + q] = acs[q];
    red[((2 * wv + half) * 3 + 1) * kCG + l32 * 4 + q] = ab1[q];
    red[((2 * wv + half) * 3 + 2) * kCG + l32 * 4 + q] = aw2[q];
  }
  __syncthreads();
  if (tid < 3 * kCG) {
    const int which = tid / kCG, j = tid % kCG;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 2 * kWaves; ++u) s += red[(u * 3 + which) * kCG + j];
    float* dst = which == 0 ? dcs : (which == 1 ? db1p : dw2p);
    if (cg * kCG + j < J) dst[(size_t)b * J + cg * kCG + j] = s;
  }
  if (!dgate || !many) return;
  // ticket: the partial rows above are acknowledged (write-through) before the counter moves
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) last_one = __hip_atomic_fetch_add(cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(cgs - 1);
  __syncthreads();
  if (!last_one) return;                         // workgroup-uniform
  const float* all = part + (size_t)b * cgs * T;
  for (int t = tid; t < T; t += kThreads) {
    float s = 0.f;
    for (int g0 = 0; g0 < cgs; g0 += 4) {        // column groups in order, four loads in flight
      const float4 v = load_agent_x4(all + (size_t)min(g0, cgs - 1) * T + t, all + (size_t)min(g0 + 1, cgs - 1) * T + t,
                                     all + (size_t)min(g0 + 2, cgs - 1) * T + t, all + (size_t)min(g0 + 3, cgs - 1) * T + t);
      s += v.x;
      if (g0 + 1 < cgs) s += v.y;
      if (g0 + 2 < cgs) s += v.z;
      if (g0 + 3 < cgs) s += v.w;
    }
    dgate[(size_t)b * T + t] = s;
  }
  if (tid == 0) __hip_atomic_store(cnt + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

inline long long bwd_ws_bytes(int B, int T, int Hm) {
  return (long long)sizeof(float) * (roundup(B, 4) + (long long)B * cdiv(2 * Hm, kCG) * T);
}

int check(const char* fn, int B, int T, int Hm, int dtype) {
  if (dtype != TSG_F32 && dtype != TSG_BF16)
    return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, or TSG_BF16 = y / dy stored as bf16)", fn, dtype);
  if (B <= 0 || T <= 0 || Hm <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d Hm=%d", fn, B, T, Hm);
  if ((2 * Hm) % 4 || 2 * Hm > 256 * kMaxJ4)
    return set_error(TSG_E_SHAPE, "%s: 2*Hm=%d must be a multiple of 4 and <= %d", fn, 2 * Hm, 256 * kMaxJ4);
  if (T > 8192) return set_error(TSG_E_SHAPE, "%s: T=%d > 8192 not supported", fn, T);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_boundary_score_fwd(const void* y, const void* cs, const void* b1, const void* w2, const void* b2,
                                      const void* gate, const int32_t* mask, void* p_start, void* p_end,
                                      int B, int T, int Hm, int dtype, void* stream) {
  const char* fn = "tsg_boundary_score_fwd";
  for (const void* p : {y, cs, b1, w2, b2, (const void*)p_start, (const void*)p_end})
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(y)) return set_error(TSG_E_ALIGN, "%s: y is not 16-byte aligned", fn);
  int rc = check(fn, B, T, Hm, dtype);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  const int tiles = cdiv(T, TR);
  if (dtype == TSG_BF16)
    hipLaunchKernelGGL(boundary_logits_kernel<bf16_t>, dim3(B * tiles), dim3(kThreads), 0, st,
                       (const bf16_t*)y, (const float*)cs, (const float*)b1, (const float*)w2, (const float*)b2,
                       (const float*)gate, mask, (float*)p_start, (float*)p_end, B, T, Hm, tiles);
  else
    hipLaunchKernelGGL(boundary_logits_kernel<float>, dim3(B * tiles), dim3(kThreads), 0, st,
                       (const float*)y, (const float*)cs, (const float*)b1, (const float*)w2, (const float*)b2,
                       (const float*)gate, mask, (float*)p_start, (float*)p_end, B, T, Hm, tiles);
  rc = check_launch(fn);
  if (rc) return rc;
  const size_t lds = sizeof(float) * (2 * (size_t)T + 2 * kWaves);
  hipLaunchKernelGGL(boundary_softmax_kernel, dim3(B), dim3(kThreads), lds, st, (float*)p_start, (float*)p_end, T);
  return check_launch(fn);
}

extern "C" int tsg_boundary_softmax(void* p_start, void* p_end, int B, int T, void* stream) {
  const char* fn = "tsg_boundary_softmax";
  if (!p_start || !p_end) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (B <= 0 || T <= 0 || T > 8192) return set_error(TSG_E_SHAPE, "%s: bad B=%d T=%d", fn, B, T);
  const size_t lds = sizeof(float) * (2 * (size_t)T + 2 * kWaves);
  hipLaunchKernelGGL(boundary_softmax_kernel, dim3(B), dim3(kThreads), lds, static_cast<hipStream_t>(stream), (float*)p_start, (float*)p_end, T);
  return check_launch(fn);
}

extern "C" int tsg_boundary_score_bwd(const void* y, const void* cs, const void* b1, const void* w2, const void* gate,
                                      const int32_t* mask, const void* p_start, const void* p_end,
                                      const void* dp_start, const void* dp_end, void* dy, void* dcs, void* db1_part,
                                      void* dw2_part, void* db2_part, void* dgate, void* dl_ws, int B, int T, int Hm,
                                      int dtype, void* stream) {
  const char* fn = "tsg_boundary_score_bwd";
  for (const void* p : {y, cs, b1, w2, p_start, p_end, dp_start, dp_end, (const void*)dy, (const void*)dcs,
                        (const void*)db1_part, (const void*)dw2_part, (const void*)db2_part, (const void*)dl_ws})
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(y) || !aligned16(dy)) return set_error(TSG_E_ALIGN, "%s: y/dy not 16-byte aligned", fn);
  int rc = check(fn, B, T, Hm, dtype);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  const size_t J = 2 * (size_t)Hm;
  hipLaunchKernelGGL(boundary_dl_kernel, dim3(B), dim3(kThreads), 0, st, mask, (const float*)p_start, (const float*)p_end,
                     (const float*)dp_start, (const float*)dp_end, (float*)dl_ws, (float*)db2_part, (float*)dcs,
                     (float*)db1_part, (float*)dw2_part, (int)J, B, T);
  rc = check_launch(fn);
  if (rc) return rc;
  const size_t lds = sizeof(float) * (size_t)kWaves * 3 * J;
  if (lds > (size_t)kLdsBytes) return set_error(TSG_E_LDS, "%s: needs %zu B of LDS", fn, lds);
  const int tiles = cdiv(T, TR);
  if (dtype == TSG_BF16) {
    auto kern = boundary_bwd_kernel<bf16_t>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * tiles), dim3(kThreads), lds, st, (const bf16_t*)y, (const float*)cs, (const float*)b1,
                       (const float*)w2, (const float*)gate, (const float*)dl_ws, (bf16_t*)dy, (float*)dcs, (float*)db1_part,
                       (float*)dw2_part, (float*)dgate, B, T, Hm, tiles);
  } else {
    auto kern = boundary_bwd_kernel<float>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * tiles), dim3(kThreads), lds, st, (const float*)y, (const float*)cs, (const float*)b1,
                       (const float*)w2, (const float*)gate, (const float*)dl_ws, (float*)dy, (float*)dcs, (float*)db1_part,
                       (float*)dw2_part, (float*)dgate, B, T, Hm, tiles);
  }
  return check_launch(fn);
}

extern "C" long long tsg_boundary_score_bwd_ws_bytes(int B, int T, int Hm) {
  if (B <= 0 || T <= 0 || Hm <= 0) return -1;
  return bwd_ws_bytes(B, T, Hm);
}

extern "C" int tsg_boundary_score_bwd_ws(const void* y, const void* cs, const void* b1, const void* w2, const void* gate,
                                         const int32_t* mask, const void* p_start, const void* p_end,
                                         const void* dp_start, const void* dp_end, void* dy, void* dcs, void* db1_part,
                                         void* dw2_part, void* db2_part, void* dgate, void* ws, long long ws_bytes,
                                         int B, int T, int Hm, int dtype, void* stream) {
  const char* fn = "tsg_boundary_score_bwd_ws";
  for (const void* p : {y, cs, b1, w2, p_start, p_end, dp_start, dp_end, (const void*)dy, (const void*)dcs,
                        (const void*)db1_part, (const void*)dw2_part, (const void*)db2_part, (const void*)ws})
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(y) || !aligned16(dy) || !aligned16(ws)) return set_error(TSG_E_ALIGN, "%s: y / dy / ws not 16-byte aligned", fn);
  int rc = check(fn, B, T, Hm, dtype);
  if (rc) return rc;
  if (ws_bytes < bwd_ws_bytes(B, T, Hm))
    return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_boundary_score_bwd_ws_bytes)", fn, ws_bytes, bwd_ws_bytes(B, T, Hm));
  auto st = static_cast<hipStream_t>(stream);
  const int cgs = cdiv(2 * Hm, kCG);
  const size_t lds = sizeof(float) * (2 * (size_t)T + (size_t)2 * kWaves * 3 * kCG);
  if (lds > (size_t)kLdsBytes - 1024) return set_error(TSG_E_LDS, "%s: needs %zu B of LDS", fn, lds);
  unsigned* cnt = static_cast<unsigned*>(ws);
  float* part = static_cast<float*>(ws) + roundup(B, 4);
  if (dtype == TSG_BF16) {
    auto kern = boundary_bwd_one_kernel<bf16_t>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * cgs), dim3(kThreads), lds, st, (const bf16_t*)y, (const float*)cs, (const float*)b1,
                       (const float*)w2, (const float*)gate, mask, (const float*)p_start, (const float*)p_end,
                       (const float*)dp_start, (const float*)dp_end, (bf16_t*)dy, (float*)dcs, (float*)db1_part,
                       (float*)dw2_part, (float*)db2_part, (float*)dgate, part, cnt, B, T, Hm, cgs);
  } else {
    auto kern = boundary_bwd_one_kernel<float>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * cgs), dim3(kThreads), lds, st, (const float*)y, (const float*)cs, (const float*)b1,
                       (const float*)w2, (const float*)gate, mask, (const float*)p_start, (const float*)p_end,
                       (const float*)dp_start, (const float*)dp_end, (float*)dy, (float*)dcs, (float*)db1_part,
                       (float*)dw2_part, (float*)db2_part, (float*)dgate, part, cnt, B, T, Hm, cgs);
  }
  return check_launch(fn);
}
