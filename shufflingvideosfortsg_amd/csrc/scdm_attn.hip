// K1 -- fused SCDM additive cross-attention for gfx950 (MI355X).
//
// Replaces the arithmetic of SCDM_Attention.forward (reference grounding/model/networks/
// attention.py:109-121) after its two projections:
//     e[b,t,n] = sum_k w[k] * tanh(a[b,t,k] + s[b,n,k]);  P = softmax_n(e);  C = P @ sent
// The reference materialises the [B,T,N,H] tanh tensor (N separate [B,T,H] tensors kept for
// autograd); here it never exists: forward and backward both recompute it in registers.
//
// Arithmetic.  tanh(a+s) = 1 - 2/(Ea*Es + 1) with Ea = exp(2a) (T*H per pair) and Es = exp(2s)
// (N*H per pair), so the T*N*H inner volume costs ONE transcendental (v_rcp_f32) and two FMAs per
// element instead of exp+rcp.  The constant sum_k w[k] is the same for every word n and cancels in
// the softmax, so e'[t,n] = -2 sum_k w[k]*r (r = 1/(Ea*Es+1)) is what is accumulated.  a and s are
// clamped to +-40 before the exponential: Ea, Es stay finite and non-zero, Ea*Es may overflow to
// +inf (r = 0, tanh = 1) or underflow (r = 1, tanh = -1) but can never be NaN.  (tanh saturates to
// +-1 in fp32 beyond |x| = 9.1, so the clamp only matters if |a| or |s| alone exceeds 40.)
//
// Layout / tiling (forward).  One 512-thread workgroup = one batch item b x TT consecutive clips.
// LDS holds Es[b] as [NP][HP] fp32 (NP = N rounded to 4, HP = H rounded to 256; 80 KiB at
// N=20,H=1024) plus the tile's P rows.  Each wave owns TT/8 clip rows and sweeps k in 256-column
// chunks with R rows blocked in registers: lane l holds k = k0 + 4*l .. +3 of each row (coalesced 16-B loads, 1 KiB
// per wave-instruction), reads the matching Es float4 from LDS (conflict-free ds_read_b128) and
// reuses it for the R rows.  The k-reduction is a wave all-reduce (DPP inside 16-lane rows, LDS
// crossbar across rows), the N-softmax runs redundantly in every lane, and C = P @ sent streams
// sent[b] from L2 with the wave's rows blocked in registers.  Blocks are remapped so that the
// tiles of one batch item share an XCD (one L2 holds that item's s / sent rows).
//
// Backward = two kernels (no atomics on the big outputs, every output written exactly once):
//   bwd_rows : per clip row  dP = dC . sent^T,  de = P*(dP - <P,dP>)           -> de workspace
//   bwd_cols : per (b, 256-wide column slice), sweeping all T rows:
//                da[t,k]  = 4 w[k] sum_n de[t,n] q        q = r(1-r)  (= (1-tanh^2)/4)
//                ds[n,k]  = 4 w[k] sum_t de[t,n] q
//                dw[k]   += -2 sum_{t,n} de[t,n] r        (sum_n de = 0, so the "+1" of tanh drops)
//                dsent[n,j] = sum_t P[t,n] dC[t,j]
//              Es for the slice lives in registers (NP*4 per lane); de / P rows are wave-uniform.
#include "tsg_common.h"
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <type_traits>

namespace tsg {
namespace {

constexpr int kThreads = 256;                 // backward kernels
constexpr int kWaves = kThreads / kWave;
constexpr int kFwdThreads = 512;              // forward: 8 waves share one Es tile (1 workgroup / CU)
constexpr int kFwdWaves = kFwdThreads / kWave;
constexpr float kClamp = 40.0f;

// Timing-only ablation (developer builds with -DTSG_ABLATE: phases are skipped per bit of the env
// var TSG_ABLATE_MASK, outputs are then wrong).  In the product build TSG_SKIP() folds to false.
#ifdef TSG_ABLATE
#define TSG_SKIP(bit) ((dbg & (bit)) != 0)
static int ablate_mask() { const char* e = getenv("TSG_ABLATE_MASK"); return e ? atoi(e) : 0; }
#else
#define TSG_SKIP(bit) false
static int ablate_mask() { return 0; }
#endif

typedef float v2f __attribute__((ext_vector_type(2)));

__host__ __device__ __forceinline__ int roundup256(int x) { return (x + 255) & ~255; }

__device__ __forceinline__ int wave_id() {    // wave-uniform (SGPR) wave index inside the workgroup
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
// One wave-level k-chunk step for R clip rows: acc[r][n] += sum_c w2[c] / (Ea[r][c]*Es[n][c] + 1).
// The words are walked in groups of 2 with the next group's Es float4s already in flight; the
// sched_barrier keeps hipcc from hoisting every x of the chunk ahead of the first rcp (which costs
// 160+ VGPRs and spills).  (Calibration, tools/ubench: v_rcp_f32 ~10 cyc, v_fma_f32 ~2.8 cyc per
// wave-instruction; v_pk_fma_f32 costs two v_fma_f32, and sharing one rcp between two elements
// (1/x0 = x1/(x0 x1)) buys nothing, so the plain fma-rcp-fma triple is the floor: ~15.6 cyc.)
template <int NP, int R, int G = (R == 1) ? 4 : 2, int NA = NP>     // G = words per scheduling group (~16 triples of VALU work at G R = 4); NA >= NP: slots of `acc`
__device__ __forceinline__ void scdm_chunk_step(const float (&Ea)[R][4], const float* __restrict__ esp, int HP,
                                                const float (&w2)[4], float (&acc)[R][NA]) {
  static_assert(NP % G == 0, "word groups");
  float4 cur[G], nxt[G];
#pragma unroll
  for (int u = 0; u < G; ++u) cur[u] = *reinterpret_cast<const float4*>(esp + u * HP);
#pragma unroll
  for (int n0 = 0; n0 < NP; n0 += G) {
    if (n0 + G < NP) {
#pragma unroll
      for (int u = 0; u < G; ++u) nxt[u] = *reinterpret_cast<const float4*>(esp + (n0 + G + u) * HP);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float e4[4] = {cur[u].x, cur[u].y, cur[u].z, cur[u].w};
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[r][n0 + u] = fmaf(w2[c], fast_rcp(fmaf(Ea[r][c], e4[c], 1.f)), acc[r][n0 + u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u) cur[u] = nxt[u];
  }
}

// The same chunk step with ONE reciprocal per lane and word instead of four (round 6): the lane's four k elements share a denominator,
//     sum_c w_c / d_c = [ (w0 d1 + w1 d0) d2 d3 + (w2 d3 + w3 d2) d0 d1 ] / (d0 d1 d2 d3)
// -- 14 full-rate VALU instructions + 1 quarter-rate v_rcp_f32 per 4 elements instead of 8 + 4 (tools/ubench: v_rcp_f32 ~9.4 cycles, an FMA ~2.55).
// Range: the caller passes Ea' = exp(2a) 2^-13 and Es' = exp(2s) 2^-13, so d'_c = clamp01(Ea' Es' + u) = u (1 + x) saturated at 1 (u = kQU = 2^-26; the
// clamp is the FMA's output modifier and costs nothing): u <= d' <= 1, the product of four stays above 2^-104, an overflowing Ea' Es' lands on 1 and NaN is
// impossible.  Saturation sets tanh = 1 - 2^-25 where x >= 2^26 (fp32 rounds the true value to 1 there: half an ulp).  w2 carries -2 w u.
constexpr int kK1PRows = 64;             // scdm_fwd_ws_kernel: P rows kept in LDS (the largest tile)
constexpr int kK1TrHalf = 4 * 32 + 32;   // ... its consumers' transpose block: [2 halves][4 rows][32 columns], the halves 32 banks apart
constexpr int kK1TrWave = 2 * kK1TrHalf; // floats per consumer wave
constexpr float kQU = 0x1p-26f;          // u
constexpr float kQHalfExp = 13.f;        // Ea', Es' = exp2(.. - kQHalfExp) each
__device__ __forceinline__ float fma_sat(float a, float b, float c) { return __builtin_amdgcn_fmed3f(fmaf(a, b, c), 0.f, 1.f); }
template <int NP, int G, int NA>
__device__ __forceinline__ void scdm_chunk_step_q(const float (&Ea)[4], const float* __restrict__ esp, int HP,
                                                  const float (&w2)[4], float (&acc)[NA]) {
  static_assert(NP % G == 0, "word groups");
  float4 cur[G], nxt[G];
#pragma unroll
  for (int u = 0; u < G; ++u) cur[u] = *reinterpret_cast<const float4*>(esp + u * HP);
#pragma unroll
  for (int n0 = 0; n0 < NP; n0 += G) {
    if (n0 + G < NP) {
#pragma unroll
      for (int u = 0; u < G; ++u) nxt[u] = *reinterpret_cast<const float4*>(esp + (n0 + G + u) * HP);
    }
#pragma unroll
    for (int u = 0; u < G; ++u) {
      const float d0 = fma_sat(Ea[0], cur[u].x, kQU), d1 = fma_sat(Ea[1], cur[u].y, kQU);
      const float d2 = fma_sat(Ea[2], cur[u].z, kQU), d3 = fma_sat(Ea[3], cur[u].w, kQU);
      const float n01 = fmaf(w2[1], d0, w2[0] * d1), p01 = d0 * d1;
      const float n23 = fmaf(w2[3], d2, w2[2] * d3), p23 = d2 * d3;
      const float num = fmaf(n23, p01, n01 * p23);
      acc[n0 + u] = fmaf(num, fast_rcp(p01 * p23), acc[n0 + u]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < G; ++u) cur[u] = nxt[u];
  }
}

// Sum NP per-lane partials over the 64 lanes with the gfx950 swap instructions: v_permlane32_swap
// folds the two wave halves of TWO values at once, v_permlane16_swap the two 16-lane rows of each
// half, then 4 DPP steps finish inside a row.  NP values -> NP/4 registers; afterwards register j,
// 16-lane row q holds the total for word n = 4j + {0,2,1,3}[q].   (NP is a multiple of 4.)
template <int NP>
__device__ __forceinline__ void wave_transpose_sum(const float (&v)[NP], float (&z)[NP / 4]) {
  float h[NP / 2];
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[2 * i]), __float_as_uint(v[2 * i + 1]), false, false);
    h[i] = __uint_as_float(r[0]) + __uint_as_float(r[1]);
  }
#pragma unroll
  for (int j = 0; j < NP / 4; ++j) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(h[2 * j]), __float_as_uint(h[2 * j + 1]), false, false);
    float t = __uint_as_float(r[0]) + __uint_as_float(r[1]);
    t += dpp_mov<0xB1>(t);    // quad_perm [1,0,3,2]
    t += dpp_mov<0x4E>(t);    // quad_perm [2,3,0,1]
    t += dpp_mov<0x141>(t);   // row_half_mirror
    t += dpp_mov<0x140>(t);   // row_mirror
    z[j] = t;
  }
}

__device__ __forceinline__ float xrow_max(float v) {     // max over the four 16-lane rows
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));
  return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xrow_sum(float v) {
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));
  return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ float4 exp2x4(float4 v) {     // exp(2 v), v clamped to +-kClamp
  return make_float4(fast_exp2(clampf(v.x, -kClamp, kClamp) * k2Log2e), fast_exp2(clampf(v.y, -kClamp, kClamp) * k2Log2e),
                     fast_exp2(clampf(v.z, -kClamp, kClamp) * k2Log2e), fast_exp2(clampf(v.w, -kClamp, kClamp) * k2Log2e));
}

// exp(2 v) 2^-13 for scdm_chunk_step_q (lower clamp -39: the scaled value stays a normal number)
__device__ __forceinline__ float exp2q(float v) { return fast_exp2(fmaf(clampf(v, -kClamp + 1.f, kClamp), k2Log2e, -kQHalfExp)); }
__device__ __forceinline__ float4 exp2x4q(float4 v) { return make_float4(exp2q(v.x), exp2q(v.y), exp2q(v.z), exp2q(v.w)); }

// Forward kernel.  Workgroup = (batch item b, TT consecutive clips), 8 waves.  The TT rows are
// processed as TT/SUB sub-tiles of SUB = 8R rows (R rows per wave): scores+softmax of sub-tile i+1
// run while the C rows of sub-tile i are still draining to HBM (a CU retires stores at only ~7
// B/clk, so an un-overlapped 128 KiB C tile costs ~8 us of store tail).  In phase 2 wave w owns a
// fixed (column group, row slot); its sent[b,:,cols] slice stays in registers for the whole kernel.
// GATE = true is the fused tail of rnn_recalibration_layer (components/VideoEncoder.py:65-72):
//   out = r * sigmoid(sent_linear(C)) with sent_linear(P sent) = P (sent W^T) + bias, so V is the
//   pre-multiplied VW = sent W^T [B,N,Ds], and the epilogue applies bias, sigmoid and the gate:
//   C itself, the [B*T,d]x[d,d] GEMM on it and three elementwise passes never happen.
// ST = storage type of the activations a, s, V (sent / VW), gr (r) and C (out): float, or bf16_t for dtype TSG_BF16 (half the
// HBM bytes; the arithmetic, P, w and gbias stay fp32).
template <int NP, int R, bool GATE, typename ST>
__global__ __launch_bounds__(kFwdThreads) void scdm_fwd_kernel(
    const ST* __restrict__ a, const ST* __restrict__ s, const float* __restrict__ w,
    const ST* __restrict__ V, ST* __restrict__ C, float* __restrict__ P,
    const ST* __restrict__ gr, const float* __restrict__ gbias,
    int B, int T, int N, int H, int Ds, int TT, int tiles, int dbg) {
  constexpr int SUB = kFwdWaves * R;
  constexpr int CW = NP <= 20 ? 4 : 2;               // sentence columns per lane in phase 2 (VGPR budget)
  const int HP = roundup256(H);
  extern __shared__ __align__(16) float lds[];
  float* Es = lds;                       // [NP][HP]
  float* Wl = lds + NP * HP;             // [HP]   -2*w (0 beyond H: padded columns add nothing)
  float* Pl = Wl + HP;                   // [2][SUB][NP]

  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles);
  const int b = bid / tiles, tile = bid % tiles;
  const int t_tile = tile * TT;
  const ST* ab = a + (size_t)b * T * H;

  // A wave keeps its current clip row(s) in registers (H <= 1024: 4 float4 per lane and row) and
  // loads the NEXT sub-tile's rows before the current score loop starts.  They are waited for right
  // after that loop -- before this sub-tile's P / C stores are issued -- so that no wait in the
  // steady state sits behind a store (vmcnt counts loads and stores together, in order; a wait
  // that follows the C stores would drain them and serialise the store tail with the next loop).
  // (raw pieces: with bf16 storage the conversion happens where the row is consumed, not where it is requested)
  typedef typename Raw4T<ST>::type Raw4;
  Raw4 q[R][4], qn[R][4];
  auto load_rows = [&](Raw4 (&dst)[R][4], int t_first) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = t_first + wv * R + r;
      const ST* row = ab + (size_t)(t < T ? t : T - 1) * H + lane * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        dst[r][i] = (i * 256 + lane * 4 < H) ? ldraw4(row + i * 256) : zero_raw4(row);
    }
  };
  load_rows(q, t_tile);

  // ---- prologue: Es = exp(2 s[b]) and -2w into LDS (zero padding of Es: r = 1, masked below).
  // Up to 12 float4 per thread per round, every load issued before the first exp (the prologue is
  // latency-, not bandwidth-bound: s[b] is 80 KiB).
  const ST* sb = s + (size_t)b * N * H;
  const int hp4 = HP / 4, total4 = TSG_SKIP(8) ? 0 : NP * hp4;
  constexpr int PU = 12;
  for (int base = tid; base < total4; base += PU * kFwdThreads) {
    float4 v[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * kFwdThreads;
      const int n = idx / hp4, k = (idx % hp4) * 4;
      v[u] = (idx < total4 && n < N && k < H) ? ld4(sb + (size_t)n * H + k) : make_float4(-1e30f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * kFwdThreads;
      if (idx < total4) {
        const int n = idx / hp4, k = (idx % hp4) * 4;
        float4 e = exp2x4(v[u]);
        if (v[u].x == -1e30f) e = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Es + n * HP + k) = e;
      }
    }
  }
  for (int k = tid * 4; k < HP; k += 4 * kFwdThreads) {
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < H) wq = *reinterpret_cast<const float4*>(w + k);
    *reinterpret_cast<float4*>(Wl + k) = make_float4(-2.f * wq.x, -2.f * wq.y, -2.f * wq.z, -2.f * wq.w);
  }

  // phase-2 role of this wave: column group cg (64*CW columns), row slot rs of every `rslots`
  const int cgroups = (Ds + 64 * CW - 1) / (64 * CW);
  int cgp = 1;
  while (cgp < cgroups && cgp < kFwdWaves) cgp <<= 1;
  const int cg = wv % cgp, rs = wv / cgp, rslots = kFwdWaves / cgp;   // host guarantees cgroups <= 8
  const ST* Vb = V + (size_t)b * N * Ds;
  v2f vreg[NP][CW / 2];
  auto load_v = [&](int j) {
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      const ST* src = Vb + (size_t)(n < N ? n : 0) * Ds + (j < Ds ? j : 0);
      if (CW == 4) {
        const float4 q = ld4(src);
        vreg[n][0] = (v2f){q.x, q.y}; vreg[n][CW / 2 - 1] = (v2f){q.z, q.w};
      } else {
        const float2 q = ld2(src);
        vreg[n][0] = (v2f){q.x, q.y};
      }
      if (n >= N) {                                    // padded words: P is 0 there, keep V finite
#pragma unroll
        for (int h = 0; h < CW / 2; ++h) vreg[n][h] = (v2f){0.f, 0.f};
      }
    }
  };
  const int jcol = cg * 64 * CW + lane * CW;
  if (!TSG_SKIP(4)) load_v(jcol);
  // GATE: bias of this lane's columns, and the r rows of the current sub-tile's phase-2 rows (requested
  // before the score loop, landed with the next a rows -- i.e. ahead of this sub-tile's stores)
  typedef typename std::conditional<CW == 4, typename Raw4T<ST>::type, typename Raw2T<ST>::type>::type RawR;
  float gb[CW];
  RawR rgv[SUB];                                          // raw r pieces of the phase-2 rows (converted in the epilogue)
#pragma unroll
  for (int c = 0; c < CW; ++c) gb[c] = (GATE && jcol + c < Ds) ? gbias[jcol + c] : 0.f;
  auto load_r = [&](int t_first) {
#pragma unroll
    for (int i = 0; i < SUB; ++i) {
      const int tl = rs + i * rslots, t = t_first + tl;
      if (tl < SUB) {
        const ST* src = gr + ((size_t)b * T + (t < T ? t : T - 1)) * Ds + (jcol < Ds ? jcol : 0);
        if constexpr (CW == 4) rgv[i] = ldraw4(src); else rgv[i] = ldraw2(src);
      }
    }
  };
  lds_barrier();

  const int nsub = TT / SUB;
  for (int st = 0; st < nsub; ++st) {
    const int t0 = t_tile + st * SUB;                  // first clip of the sub-tile
    float* Pcur = Pl + (st & 1) * SUB * NP;

    // ---- phase 1: scores + softmax for this wave's R rows; k in 256-column chunks --------------
    float acc[R][NP];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NP; ++n) acc[r][n] = 0.f;
    if (st + 1 < nsub) load_rows(qn, t0 + SUB);          // flies during the score loop below
    if (GATE) load_r(t0);

#pragma unroll 1
    for (int k0 = 0; k0 < (TSG_SKIP(1) ? 0 : HP); k0 += 256) {
      const int k = k0 + lane * 4;
      const float4 wq = *reinterpret_cast<const float4*>(Wl + k);
      const float w2[4] = {wq.x, wq.y, wq.z, wq.w};
      float Ea[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 e = exp2x4(cvt4(q[r][0]));
        Ea[r][0] = e.x; Ea[r][1] = e.y; Ea[r][2] = e.z; Ea[r][3] = e.w;
        q[r][0] = q[r][1]; q[r][1] = q[r][2]; q[r][2] = q[r][3];   // H <= 1024: the whole row is here
      }
      scdm_chunk_step<NP, R>(Ea, Es + k, HP, w2, acc);
    }
    // land the next rows now (long since arrived), ahead of this sub-tile's stores
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (storage_is_bf16<ST>::value) asm volatile("" : "+v"(qn[r][i].x), "+v"(qn[r][i].y));
        else asm volatile("" : "+v"(qn[r][i].x), "+v"(qn[r][i].y), "+v"(qn[r][i].z), "+v"(qn[r][i].w));
        q[r][i] = qn[r][i];
      }
    if (GATE) {
#pragma unroll
      for (int i = 0; i < SUB; ++i) {
        if constexpr (storage_is_bf16<ST>::value) {
          if constexpr (CW == 4) asm volatile("" : "+v"(rgv[i].x), "+v"(rgv[i].y)); else asm volatile("" : "+v"(rgv[i]));
        } else {
          if constexpr (CW == 4) asm volatile("" : "+v"(rgv[i].x), "+v"(rgv[i].y), "+v"(rgv[i].z), "+v"(rgv[i].w));
          else asm volatile("" : "+v"(rgv[i].x), "+v"(rgv[i].y));
        }
      }
    }

    // k-reduction + softmax over the N words, one clip row at a time
#pragma unroll
    for (int r = 0; r < R; ++r) {
      float z[NP / 4];
      if (TSG_SKIP(2)) {
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) z[j] = acc[r][j];
      } else {
        wave_transpose_sum<NP>(acc[r], z);
      }
      const int q = lane >> 4;
      const int nq = ((q & 1) << 1) | (q >> 1);         // row q holds word 4j + {0,2,1,3}[q]
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        if (4 * j + nq >= N) z[j] = -INFINITY;
        m = fmaxf(m, z[j]);
      }
      m = xrow_max(m);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        z[j] = fast_exp2((z[j] - m) * kLog2e);            // exp(-inf) = 0 for padded words
        sum += z[j];
      }
      const float inv = 1.f / xrow_sum(sum);
      const int tl = wv * R + r;                        // row inside the sub-tile
      const int t = t0 + tl;
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
          const int n = 4 * j + nq;
          const float pv = z[j] * inv;
          Pcur[tl * NP + n] = pv;
          if (n < N && t < T) P[((size_t)b * T + t) * N + n] = pv;
        }
      }
    }
    lds_barrier();                                       // the sub-tile's P rows are in LDS (stores keep flying)

    // ---- phase 2: C[t, cols] = sum_n P[t,n] * sent[b,n,cols] for rows rs, rs+rslots, ... ---------
    if (!TSG_SKIP(4)) {
      const bool jok = jcol < Ds;
      auto row = [&](int tl, const RawR& rraw) {
        const int t = t0 + tl;
        float rr[CW];
        if constexpr (CW == 4) { const float4 f = cvt4(rraw); rr[0] = f.x; rr[1] = f.y; rr[2] = f.z; rr[3] = f.w; }
        else { const float2 f = cvt2(rraw); rr[0] = f.x; rr[1] = f.y; }
        v2f c[CW / 2];
#pragma unroll
        for (int h = 0; h < CW / 2; ++h) c[h] = (v2f){0.f, 0.f};
#pragma unroll
        for (int n4 = 0; n4 < NP; n4 += 4) {
          const float4 p = *reinterpret_cast<const float4*>(Pcur + tl * NP + n4);
          const float pp[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int h = 0; h < CW / 2; ++h)
              c[h] = __builtin_elementwise_fma((v2f){pp[u], pp[u]}, vreg[n4 + u][h], c[h]);
        }
        float o[CW];
#pragma unroll
        for (int h = 0; h < CW / 2; ++h) { o[2 * h] = c[h].x; o[2 * h + 1] = c[h].y; }
        if (GATE) {
#pragma unroll
          for (int cc = 0; cc < CW; ++cc)
            o[cc] = rr[cc] * fast_rcp(1.f + fast_exp2(-(o[cc] + gb[cc]) * kLog2e));      // r * sigmoid(G)
        }
        if (t < T && jok) {
          ST* dst = C + ((size_t)b * T + t) * Ds + jcol;
          if (CW == 4) st4(dst, make_float4(o[0], o[1], o[CW - 2], o[CW - 1]));
          else st2(dst, make_float2(o[0], o[1]));
        }
      };
      if (GATE) {
#pragma unroll
        for (int i = 0; i < SUB; ++i)
          if (rs + i * rslots < SUB) row(rs + i * rslots, rgv[i]);
      } else {
        const RawR none = {};
#pragma unroll 2
        for (int tl = rs; tl < SUB; tl += rslots) row(tl, none);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, split-precision variant (dtype TSG_F32S): phase 2 on the bf16 matrix pipe.
//
// The forward above is bound by VALU issue (PMC: the SIMDs' vector pipes are busy ~70 % of the launch, HBM traffic is 1.00x the
// algorithmic bytes): per 64-row workgroup tile a wave spends ~46 k cycles of issue in the score loop (fma, rcp, fma per (t,n,k))
// and ~16 k in phase 2 (C = P VW: T N Ds multiply-adds as v_pk_fma_f32, which costs two v_fma_f32 here) + the gate epilogue.
// Phase 2 is a [32 rows x N words] x [N words x Ds] product per 32-row group: on v_mfma_f32_32x32x16_bf16 it is 6 MFMAs per
// 32 x 32 output tile (2 k steps of 16 words x the three split-precision products hi*hi + hi*lo + lo*hi; P in [0,1] and VW are
// written as hi + lo with hi = rne_bf16(x), lo = rne_bf16(x - hi): error 2^-16 relative, the arithmetic of the "f32s" mode's
// GEMMs) = 24 MFMAs per wave and group against ~8 k cycles of VALU issue, and the matrix pipe runs beside the other wave's
// score loop.  What it costs is the accumulator layout: a lane holds ONE output column and 16 rows, so the r loads and the out
// stores are one dword per lane (two 128-byte row segments per wave instruction) instead of 16 bytes per lane: 4x the vector
// memory instructions for the same bytes.  (Round 2 tried this with the exact fp32 MFMA -- 10 x 64 cycles per tile -- and an LDS
// transpose to keep 16-byte rows, restructured around 32-row sub-tiles: slower.  Here the 8-row score pipeline is untouched,
// the P rows of four sub-tiles collect in LDS, and the MFMA phase runs once per 32 rows with the VW fragments resident.)
// Used for fp32 storage when H = Ds = 256 * CT, CT in {1, 2, 4} (SCDM_Attention's default hidden_dim = video_dim at d = 256, 512,
// 1024); everything else runs the kernel above.
// ------------------------------------------------------------------------------------------
typedef __bf16 k1_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned k1_u32x4 __attribute__((ext_vector_type(4)));
typedef float k1_f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ void k1_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack_bf16x2(a, b);
  lo = pack_bf16x2(a - bf16_lo(hi), b - bf16_hi(hi));
}
__device__ __forceinline__ void k1_split8(const float (&v)[8], k1_u32x4& hi, k1_u32x4& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) k1_split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
  hi = (k1_u32x4){h[0], h[1], h[2], h[3]};
  lo = (k1_u32x4){l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ k1_f32x16 k1_mfma(k1_u32x4 a, k1_u32x4 b, k1_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k1_bf16x8, a), __builtin_bit_cast(k1_bf16x8, b), c, 0, 0, 0);
}

template <int NP, bool GATE, int CT>
__global__ __launch_bounds__(kFwdThreads) void scdm_fwd_mm_kernel(
    const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ w,
    const float* __restrict__ V, float* __restrict__ C, float* __restrict__ P,
    const float* __restrict__ gr, const float* __restrict__ gbias,
    int B, int T, int N, int H, int Ds, int TT, int tiles, int dbg) {
  constexpr int SUB = kFwdWaves;                     // 8 rows per sub-tile, one per wave
  constexpr int PP = 36;                             // P tile row pitch (32 word slots + 4: 16-byte aligned rows)
  constexpr int KS = NP > 16 ? 2 : 1;                // MFMA k steps of 16 words
  const int HP = roundup256(H);
  extern __shared__ __align__(16) float lds[];
  float* Es = lds;                       // [NP][HP]
  float* Wl = lds + NP * HP;             // [HP]   -2*w
  float* Pl = Wl + HP;                   // [2][SUB][PP]  softmax rows of a sub-tile (double-buffered), zero beyond NP

  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int jl = lane & 31, hh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles);
  const int b = bid / tiles, tile = bid % tiles;
  const int t_tile = tile * TT;
  const float* ab = a + (size_t)b * T * H;

  // The wave's current clip row lives in 4 float4 (H <= 1024); chunk c of the NEXT sub-tile's row is requested into slot c as soon
  // as the score loop has consumed it (three chunks = ~4 k cycles of cover): 16 registers instead of a second row buffer.
  float4 q[4];
  auto row_ptr = [&](int t_first) {
    const int t = t_first + wv;
    return ab + (size_t)(t < T ? t : T - 1) * H + lane * 4;
  };
  {
    const float* row = row_ptr(t_tile);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      q[i] = (i * 256 + lane * 4 < H) ? *reinterpret_cast<const float4*>(row + i * 256) : make_float4(0.f, 0.f, 0.f, 0.f);
  }

  // ---- prologue: Es = exp(2 s[b]), -2w, zeroed P tiles (as the kernel above)
  const float* sb = s + (size_t)b * N * H;
  const int hp4 = HP / 4, total4 = NP * hp4;
  constexpr int PU = 12;
  for (int base = tid; base < total4; base += PU * kFwdThreads) {
    float4 v[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * kFwdThreads;
      const int n = idx / hp4, k = (idx % hp4) * 4;
      v[u] = (idx < total4 && n < N && k < H) ? *reinterpret_cast<const float4*>(sb + (size_t)n * H + k) : make_float4(-1e30f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * kFwdThreads;
      if (idx < total4) {
        const int n = idx / hp4, k = (idx % hp4) * 4;
        float4 e = exp2x4(v[u]);
        if (v[u].x == -1e30f) e = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Es + n * HP + k) = e;
      }
    }
  }
  for (int k = tid * 4; k < HP; k += 4 * kFwdThreads) {
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < H) wq = *reinterpret_cast<const float4*>(w + k);
    *reinterpret_cast<float4*>(Wl + k) = make_float4(-2.f * wq.x, -2.f * wq.y, -2.f * wq.z, -2.f * wq.w);
  }
  for (int i = tid; i < 2 * SUB * PP; i += kFwdThreads) Pl[i] = 0.f;

  // ---- this wave's strip of VW (32*CT columns) as resident B-operand fragments of v_mfma_f32_32x32x16_bf16: lane (jl, hh) holds
  // words 16 ks + 8 hh .. +7 of column col0 + 32 ct + jl, as hi and lo bf16 planes (hi = rne_bf16(x), lo = rne_bf16(x - hi))
  const int col0 = wv * 32 * CT;
  const float* Vb = V + (size_t)b * N * Ds + col0 + jl;      // Ds = 256 CT: every column exists
  k1_u32x4 vh[CT][KS], vl[CT][KS];
  float gb[CT];
#pragma unroll
  for (int ct = 0; ct < CT; ++ct) {
    gb[ct] = GATE ? -kLog2e * gbias[col0 + 32 * ct + jl] : 0.f;      // folded into the sigmoid's exponent
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      float e[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int n = 16 * ks + 8 * hh + j;
        e[j] = Vb[(size_t)(n < N ? n : 0) * Ds + 32 * ct];
        if (n >= N) e[j] = 0.f;
      }
      k1_split8(e, vh[ct][ks], vl[ct][ks]);
    }
  }
  // The MFMA tile is 32 rows; a sub-tile has 8: rows 8 .. 31 of the A operand are zero, and of the 16 accumulator registers of a
  // lane only i = 0 .. 3 (rows i + 4 hh) are real -- every lane of the wave has live outputs, 4 per column tile.  (The matrix pipe is
  // otherwise idle: 6 CT MFMAs per wave and sub-tile run beside the partner wave's score loop.)
  // r / out addresses: wave-uniform row base (scalar registers) + ONE 32-bit lane offset (4 hh rows + the lane's column).  Rows beyond
  // T (ragged last tile) read an in-bounds row instead (the uniform row clamped to T - 1; the upper half wave drops its 4-row offset
  // when that would leave the sequence -- a wave-uniform choice between two lane offsets) and are never stored.
  const unsigned lane_off = (unsigned)(4 * hh) * (unsigned)Ds + (unsigned)(col0 + jl);
  const unsigned lane_col = (unsigned)(col0 + jl);
  float rr[CT][4];
  auto load_rr = [&](int t0) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int u = min(t0 + i, T - 1);                    // wave-uniform
      const float* base = gr + ((size_t)b * T + u) * Ds;
      const unsigned off = u + 4 < T ? lane_off : lane_col;
#pragma unroll
      for (int ct = 0; ct < CT; ++ct) rr[ct][i] = (base + 32 * ct)[off];
    }
  };
  // nothing may be pending at the loop head (see the landing point inside the loop): the first row is consumed right away anyway
#pragma unroll
  for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));
  lds_barrier();

  const int nsub = TT / SUB;
#pragma unroll 1
  for (int st = 0; st < nsub; ++st) {
    const int t0 = t_tile + st * SUB;
    float* Pcur = Pl + (st & 1) * SUB * PP;

    // ---- phase 1: scores + softmax of this wave's row (as above) ---------------------------------
    float acc[1][NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) acc[0][n] = 0.f;
    if (GATE) load_rr(t0);                                 // lands under the score loop below: phase 2 never waits for HBM
    // next sub-tile's row (beyond the tile / the sequence: a clamped, valid row that is never used).  The loads below are
    // UNCONDITIONAL on clamped addresses: a load under a condition becomes a copy of its destination behind a vmcnt(0) wait right
    // after the chunk (the value must be merged with the old register contents) -- a stall on its own latency, four times per row.
    const float* nrow = row_ptr(t0 + SUB) - lane * 4;
#pragma unroll
    for (int c = 0; c < CT; ++c) {                         // H = Ds = 256 CT (host-checked): the chunk count is a compile-time constant,
      const int k0 = 256 * c;                              // so no chunk (and no load inside it) sits under a condition
      if (!TSG_SKIP(1)) {
        const int k = k0 + lane * 4;
        const float4 wq = *reinterpret_cast<const float4*>(Wl + k);
        const float w2[4] = {wq.x, wq.y, wq.z, wq.w};
        float Ea[1][4];
        const float4 e = exp2x4(q[c]);
        Ea[0][0] = e.x; Ea[0][1] = e.y; Ea[0][2] = e.z; Ea[0][3] = e.w;
        q[c] = *reinterpret_cast<const float4*>(nrow + k);
        scdm_chunk_step<NP, 1, 2>(Ea, Es + k, HP, w2, acc);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // Land the next row (and this sub-tile's r values) HERE, ahead of this sub-tile's P / out stores: vmcnt counts loads and stores
    // together, in order, and a wait that the compiler places at the loop head for a load still pending across the back edge is
    // `vmcnt(0)` -- it would drain the out stores of every sub-tile (an HBM write round trip, ~2 us, eight times per tile: what made
    // the first version of this kernel no faster than the VALU one).  After this point only stores are in flight.
#pragma unroll
    for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(q[i].x), "+v"(q[i].y), "+v"(q[i].z), "+v"(q[i].w));
    if (GATE) {
#pragma unroll
      for (int ct = 0; ct < CT; ++ct)
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(rr[ct][i]));
    }
    {
      float z[NP / 4];
      wave_transpose_sum<NP>(acc[0], z);
      const int qd = lane >> 4;
      const int nq = ((qd & 1) << 1) | (qd >> 1);
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        if (4 * j + nq >= N) z[j] = -INFINITY;
        m = fmaxf(m, z[j]);
      }
      m = xrow_max(m);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        z[j] = fast_exp2((z[j] - m) * kLog2e);
        sum += z[j];
      }
      const float inv = 1.f / xrow_sum(sum);
      const int t = t0 + wv;
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
          const int n = 4 * j + nq;
          const float pv = z[j] * inv;
          Pcur[wv * PP + n] = pv;
          if (n < N && t < T) P[((size_t)b * T + t) * N + n] = pv;
        }
      }
    }
    lds_barrier();                                        // the sub-tile's 8 P rows are in LDS (the other buffer is free again)
    if (TSG_SKIP(4)) continue;

    // ---- phase 2: out[8 rows, this wave's 32*CT columns] on the matrix pipe --------------------------
    k1_u32x4 ph[KS], pl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const float* pr = Pcur + (jl & 7) * PP + 16 * ks + 8 * hh;
      float4 x = *reinterpret_cast<const float4*>(pr), y = *reinterpret_cast<const float4*>(pr + 4);
      if (jl >= 8) { x = make_float4(0.f, 0.f, 0.f, 0.f); y = x; }         // rows 8 .. 31 of the MFMA tile do not exist
      const float e[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
      k1_split8(e, ph[ks], pl[ks]);
    }
#pragma unroll
    for (int ct = 0; ct < CT; ++ct) {
      k1_f32x16 o;
#pragma unroll
      for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        o = k1_mfma(ph[ks], vh[ct][ks], o);
        o = k1_mfma(ph[ks], vl[ct][ks], o);
        o = k1_mfma(pl[ks], vh[ct][ks], o);
      }
      float* dst = C + ((size_t)b * T + t0) * Ds + 32 * ct;
#pragma unroll
      for (int i = 0; i < 4; ++i) {                          // accumulator register i <-> row i + 4 hh
        float v = o[i];
        if (GATE) v = rr[ct][i] * fast_rcp(1.f + fast_exp2(fmaf(v, -kLog2e, gb[ct])));      // r * sigmoid(G + bias)
        if (t0 + i + 4 * hh < T) (dst + (size_t)i * Ds)[lane_off] = v;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, split-precision variant with ROLE-SPECIALISED waves (dtype TSG_F32S, the default for H = Ds = 256 CT).
//
// In the kernel above every wave alternates between the score loop (VALU issue) and phase 2 (r loads, MFMAs, out stores), and the
// barrier between them keeps the 8 waves of the CU in the same phase: the ablation (HISTORY.md, K1g round 3) shows the two phases
// adding up (34 us + 15 us) instead of overlapping.  Here the roles are split across waves instead of across time:
//   waves 0..PW-1     producers: scores + softmax of 8/PW rows each (one after the other) per 8-row sub-tile, nothing else;
//   waves PW..2PW-1   consumers: phase 2 of the PREVIOUS sub-tile -- all 8 rows x Ds/PW columns each on the matrix pipe, the r loads,
//                     the sigmoid gate and the out stores.
// PW = 8 (1024 threads: two producers and two consumers per SIMD, 128 VGPRs each) when the consumer's resident VW strip fits, i.e.
// N <= 24 (words 16..23 then ride a 32x32x8 k step with 2-register operands); PW = 4 (512 threads, one of each per SIMD) otherwise.
// One barrier per sub-tile hands the P rows over (double-buffered in LDS), so phase 2 of sub-tile i runs beside the score loop of
// sub-tile i + 1 by construction, on the SIMD's other wave.  A consumer's r rows are requested a whole score loop (~5 us) before they
// are used and its stores have as long to drain; a producer issues no stores but the 80-byte P rows.  Cost: the score loop runs on one
// wave per SIMD (tools/ubench: 18.2 instead of 16.1 cycles per fma-rcp-fma triple when nothing else fills the gaps).
// ------------------------------------------------------------------------------------------
#ifndef TSG_WS_PRIO_P
#define TSG_WS_PRIO_P 2                                   // wave priorities of the two roles (tuning: tools/build_variant.sh)
#endif
#ifndef TSG_WS_PRIO_C
#define TSG_WS_PRIO_C 0
#endif
#ifndef TSG_WS_PRIO_P1
#define TSG_WS_PRIO_P1 3                                  // the producer whose turn it is
#endif
typedef short k1_s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned k1_u32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ k1_f32x16 k1_mfma8(k1_u32x2 a, k1_u32x2 b, k1_f32x16 c) {      // 32x32x8: k = 4 (lane / 32) + j
  return __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(k1_s16x4, a), __builtin_bit_cast(k1_s16x4, b), c, 0, 0, 0);
}

// ST = bf16_t (dtype TSG_BF16): a, s, V, gr, C are bf16 in HBM; the rows are kept as raw 8-byte pieces until the score loop consumes them,
// VW is exact in bf16 so its lo plane (and the third MFMA of every k step) drops out, r / out are one 2-byte element per lane.
#ifdef TSG_K1_TICKS       // developer builds: per-wave shader-clock sums of the phases of scdm_fwd_ws_kernel (tools/k1_ticks.py reads them back)
__device__ unsigned long long g_k1_ticks[256 * 16 * 8];
#define K1_TICK(i) { const unsigned long long tnow_ = __builtin_amdgcn_s_memtime(); tph_[i] += tnow_ - tlast_; tlast_ = tnow_; }
#define K1_TICK_DECL unsigned long long tph_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast_ = __builtin_amdgcn_s_memtime();
#define K1_TICK_DUMP if (blockIdx.x < 256 && lane == 0) { tph_[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); /* HW_ID */ for (int i_ = 0; i_ < 8; ++i_) g_k1_ticks[(blockIdx.x * 16 + wv) * 8 + i_] = tph_[i_]; }
#else
#define K1_TICK(i)
#define K1_TICK_DECL
#define K1_TICK_DUMP
#endif
// s_setprio TSG_WS_PRIO_P + (hi ? 1 : 0) for a wave-uniform hi
#define TSG_STR2(x) #x
#define TSG_STR(x) TSG_STR2(x)
__device__ __forceinline__ void k1_prio_turn(int hi) {
  asm volatile("s_bitcmp1_b32 %0, 0\n\ts_cbranch_scc1 1f\n\ts_setprio " TSG_STR(TSG_WS_PRIO_P) "\n\ts_branch 2f\n1:\n\ts_setprio " TSG_STR(TSG_WS_PRIO_P1) "\n2:" :: "s"(hi) : "scc");
}
template <int NP, bool GATE, int CT, int PW, typename ST, int XW = 0>
__global__ __launch_bounds__(128 * PW) void scdm_fwd_ws_kernel(
    const ST* __restrict__ a, const ST* __restrict__ s, const float* __restrict__ w,
    const ST* __restrict__ V, ST* __restrict__ C, float* __restrict__ P,
    const ST* __restrict__ gr, const float* __restrict__ gbias,
    int B, int T, int N, int H, int Ds, int TT, int tiles, int dbg) {
  constexpr int NT = 128 * PW;                       // PW producer + PW consumer waves
  constexpr int SUB = 8;                             // rows per sub-tile
  constexpr int RPW = SUB / PW;                      // rows per producer wave and sub-tile (one after the other)
  constexpr int PP = 36;
  // XW > 0 (fp32 storage, N = 24 + XW <= 26 -- the ActivityNet configs' N = 25): words 0 .. 23 on the matrix pipe as for N <= 24, the XW words beyond
  // them as fp32 FMAs on the accumulator (P[row][24 + x] from LDS times one resident VW register per tile and word): the consumer's strip stays
  // within the 128 VGPRs of the 8 + 8 wave layout (a second 16-word k step does not: hi + lo of 32 words = 64 registers per wave)
  static_assert(XW == 0 || (NP == 28 && PW == 8 && XW <= 2), "XW: the 8 + 8 layout at N = 25, 26");
  constexpr bool K8 = (NP > 16 && NP <= 24) || XW > 0; // words 16 .. 23 as ONE 32x32x8 k step (2-register operands) instead of a 32x32x16
  constexpr int KS = NP > 16 && !K8 ? 2 : 1;         // full 16-word k steps
  constexpr int NS = XW > 0 ? 26 : NP;               // word slots the score loop walks (bf16 storage at N = 25: 26 instead of 28 slots measured no faster, 80 us)
  constexpr int CC = 8 * CT / PW;                    // 32-column tiles per consumer wave (Ds / PW columns)
#ifdef TSG_K1_TRIPLE
  constexpr bool QS = false;                         // (A/B builds: the round-3 fma-rcp-fma triple per element)
#else
  constexpr bool QS = NP <= 24;                      // score loop with one reciprocal per four k elements (scdm_chunk_step_q); 28 word slots: its six
                                                     // extra live registers spill (4-6 reloads per row: 310 vs 304 us at [128,512,25,1024]) -- the triple there
#endif
  const int HP = roundup256(H);
  extern __shared__ __align__(16) float lds[];
  float* Es = lds;                       // [NP][HP]
  float* Wl = lds + NP * HP;             // [HP]   -2*w
  float* Pl = Wl + HP;                   // [kK1PRows][PP]: the P rows of the whole tile (TT <= 64)
  unsigned* sync = reinterpret_cast<unsigned*>(Pl + kK1PRows * PP);     // [0 .. 7] rows scored per sub-tile, [8] the next row to hand out
  float* Tr = reinterpret_cast<float*>(sync + 16);                     // [PW][kK1TrWave]: the consumers' 8 x 32 transpose blocks

  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  K1_TICK_DECL
  const int jl = lane & 31, hh = lane >> 5;
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles);
  const int b = bid / tiles, tile = bid % tiles;
  const int t_tile = tile * TT;
  constexpr bool BF = storage_is_bf16<ST>::value;
  const ST* ab = a + (size_t)b * T * H;
  const int rows_valid = min(TT, T - t_tile);            // rows of this tile that exist

  // ---- prologue (all waves): Es = exp(2 s[b]), -2w, zeroed P tiles
  const ST* sb = s + (size_t)b * N * H;
  const int hp4 = HP / 4, total4 = TSG_SKIP(8) ? 0 : NP * hp4;
  constexpr int PU = PW == 4 ? 12 : 6;
  for (int base = tid; base < total4; base += PU * NT) {
    float4 v[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * NT;
      const int n = idx / hp4, k = (idx % hp4) * 4;
      v[u] = (idx < total4 && n < N && k < H) ? ld4(sb + (size_t)n * H + k) : make_float4(-1e30f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int idx = base + u * NT;
      if (idx < total4) {
        const int n = idx / hp4, k = (idx % hp4) * 4;
        float4 e = QS ? exp2x4q(v[u]) : exp2x4(v[u]);
        if (v[u].x == -1e30f) e = make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(Es + n * HP + k) = e;
      }
    }
  }
  for (int k = tid * 4; k < HP; k += 4 * NT) {
    float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k < H) wq = *reinterpret_cast<const float4*>(w + k);
    const float ws = QS ? -2.f * kQU : -2.f;
    *reinterpret_cast<float4*>(Wl + k) = make_float4(ws * wq.x, ws * wq.y, ws * wq.z, ws * wq.w);
  }
  for (int i = tid; i < kK1PRows * PP; i += NT) Pl[i] = 0.f;
  if (tid < 16) sync[tid] = tid == 8 ? PW : 0;             // (rows 0 .. PW - 1 are the producers' first rows)

  if (wv < PW) {
    // =================================== producer: one row at a time, handed out by an LDS counter =========================
    // (round 6: no workgroup barrier in the tile loop.  With one barrier per sub-tile every sub-tile cost the time of its SLOWEST wave: of the two
    // producers of a SIMD the older one wins the issue arbitration and scored a row in 7.7 k ticks, the other in 11 k, and the low-priority consumers
    // finished last of all -- each wave waited 15..35 % of its life at the barrier (tools/k1_ticks.py).  Now a producer takes the next unscored row
    // when it is done with one, the P rows of the whole tile stay in LDS, and the consumers wait for a sub-tile's row count.)
    __builtin_amdgcn_s_setprio(TSG_WS_PRIO_P);
    // The wave's current row lives in CT float4; chunk c of the row it scores NEXT is requested into slot c as soon as the score loop
    // has consumed it, unconditionally (clamped address): the next row is claimed before the current one is scored.
    typedef typename Raw4T<ST>::type Raw4;
    Raw4 q[CT];
    auto row_ptr = [&](int t) { return ab + (size_t)(t < T ? t : T - 1) * H + lane * 4; };
    auto land = [&]() {                                    // (an empty asm per piece: the wait for the requests lands HERE)
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        if constexpr (BF) asm volatile("" : "+v"(q[c].x), "+v"(q[c].y));
        else asm volatile("" : "+v"(q[c].x), "+v"(q[c].y), "+v"(q[c].z), "+v"(q[c].w));
      }
    };
    {
      const ST* row = row_ptr(t_tile + wv);
#pragma unroll
      for (int c = 0; c < CT; ++c) q[c] = ldraw4(row + c * 256);
    }
    land();
    lds_barrier();
    K1_TICK(0)                                             // 0: prologue
    const int turn = (wv >> 2) & 1;                        // which of the SIMD's two producers this wave is
    auto score_row = [&](float* Prow, int t, const ST* nrow) {
      float acc[1][NP];
#pragma unroll
      for (int n = 0; n < NP; ++n) acc[0][n] = 0.f;
#pragma unroll
      for (int c = 0; c < CT; ++c) {
        const int k = 256 * c + lane * 4;
#ifndef TSG_K1_NO_PRIO_SWAP
        // The SIMD's arbiter serves its OLDEST wave first: of the two producers of a SIMD the first-launched one scored a row in 7.7 k ticks, the other in
        // 11 k, and the sub-tile waited for the slower (tools/k1_ticks.py).  They take turns at the higher priority, chunk by chunk, and finish together.
        // (s_setprio takes an immediate.  The choice is a branch INSIDE one asm statement: a compiler-visible branch in the chunk loop cost 245 spilled
        // registers, the loop instantiated once per parity 7 -- their reloads wait behind the next row's requests)
        if constexpr (PW == 8 && CT >= 2) k1_prio_turn((turn ^ c) & 1);
#endif
        const float4 wq = *reinterpret_cast<const float4*>(Wl + k);
        const float w2[4] = {wq.x, wq.y, wq.z, wq.w};
        float Ea[1][4];
        const float4 e = QS ? exp2x4q(cvt4(q[c])) : exp2x4(cvt4(q[c]));
        Ea[0][0] = e.x; Ea[0][1] = e.y; Ea[0][2] = e.z; Ea[0][3] = e.w;
        q[c] = ldraw4(nrow + 256 * c);
        if (!TSG_SKIP(1)) {                                                            // (XW: 26 of the 28 slots are scored, the others stay 0 and are masked below)
          if constexpr (QS) scdm_chunk_step_q<NS, 2, NP>(Ea[0], Es + k, HP, w2, acc[0]);
          else scdm_chunk_step<NS, 1, 2, NP>(Ea, Es + k, HP, w2, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      K1_TICK(1)                                           // 1: score loop
      // land the next row ahead of the P stores (vmcnt counts loads and stores together, in order)
      land();
      K1_TICK(2)                                           // 2: wait for the next row
      float z[NP / 4];
      wave_transpose_sum<NP>(acc[0], z);
      const int qd = lane >> 4;
      const int nq = ((qd & 1) << 1) | (qd >> 1);
      float m = -INFINITY;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        if (4 * j + nq >= N) z[j] = -INFINITY;
        m = fmaxf(m, z[j]);
      }
      m = xrow_max(m);
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) {
        z[j] = fast_exp2((z[j] - m) * kLog2e);
        sum += z[j];
      }
      const float inv = 1.f / xrow_sum(sum);
      if ((lane & 15) == 0) {
#pragma unroll
        for (int j = 0; j < NP / 4; ++j) {
          const int n = 4 * j + nq;
          const float pv = z[j] * inv;
          Prow[n] = pv;
          if (n < N) P[((size_t)b * T + t) * N + n] = pv;
        }
      }
    };
    int row = wv;
#pragma unroll 1
    while (row < rows_valid) {
      unsigned claim = 0;
      if (lane == 0) claim = __hip_atomic_fetch_add(sync + 8, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      const int nxt = __builtin_amdgcn_readfirstlane((int)claim);
      K1_TICK(4)                                           // 4: claiming the next row
      score_row(Pl + row * PP, t_tile + row, row_ptr(t_tile + nxt));
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");     // the P row is in LDS before its sub-tile's count moves
      if (lane == 0) __hip_atomic_fetch_add(sync + (row >> 3), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      K1_TICK(3)                                           // 3: reduction, softmax, P stores, hand-over
      row = nxt;
    }
    K1_TICK_DUMP
  } else {
    // =================================== consumer: phase 2 of every sub-tile whose rows have all been scored ==============
    __builtin_amdgcn_s_setprio(TSG_WS_PRIO_C);
    const int col0 = (wv - PW) * 32 * CC;
    lds_barrier();                                         // (prologue barrier first: the producers do not wait for the VW loads below)
    const ST* Vb = V + (size_t)b * N * Ds + col0 + jl;
    k1_u32x4 vh[CC][KS], vl[CC][KS];
    k1_u32x2 vh8[CC], vl8[CC];
    float gb[CC];
#pragma unroll
    for (int ct = 0; ct < CC; ++ct) {
      gb[ct] = GATE ? -kLog2e * gbias[col0 + 32 * ct + jl] : 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        float e[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int n = 16 * ks + 8 * hh + j;
          e[j] = ld1(Vb + (size_t)(n < N ? n : 0) * Ds + 32 * ct);
          if (n >= N) e[j] = 0.f;
        }
        k1_split8(e, vh[ct][ks], vl[ct][ks]);
      }
      if (K8) {
        float e[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 16 + 4 * hh + j;
          e[j] = ld1(Vb + (size_t)(n < N ? n : 0) * Ds + 32 * ct);
          if (n >= N) e[j] = 0.f;
        }
        unsigned h0, l0, h1, l1;
        k1_split_pair(e[0], e[1], h0, l0); k1_split_pair(e[2], e[3], h1, l1);
        vh8[ct] = (k1_u32x2){h0, h1}; vl8[ct] = (k1_u32x2){l0, l1};
      }
    }
    float vx[CC][XW > 0 ? XW : 1];
    if constexpr (XW > 0) {
#pragma unroll
      for (int ct = 0; ct < CC; ++ct)
#pragma unroll
        for (int x = 0; x < XW; ++x) vx[ct][x] = ld1(Vb + (size_t)(24 + x) * Ds + 32 * ct);
    }
    // Epilogue layout (round 6): the accumulator gives a lane 4 consecutive ROWS of one column (dword loads / stores: 16 + 16 instructions per wave and
    // sub-tile, and the store issue was worth 6 us of the launch).  Each 8 x 32 tile goes through a per-wave LDS block instead and comes back as 4 consecutive
    // COLUMNS of one row per lane: one 16-byte r load and one 16-byte store per tile -- 8 rows x 128 bytes per instruction.
    float* trw = Tr + (wv - PW) * kK1TrWave;
    const int wr_off = hh * kK1TrHalf + jl;                // write: row i + 4 hh at [hh][i][jl] (the halves 32 banks apart: conflict-free)
    const int em = lane & 3, eq = (lane >> 2) & 7;         // read back: row em + 4 hh, columns 4 eq .. 4 eq + 3
    const int rd_off = hh * kK1TrHalf + em * 32 + 4 * eq;
    const int erow = em + 4 * hh;
    typedef typename Raw4T<ST>::type Raw4;
    Raw4 rr[CC];
    auto load_rr = [&](int t0) {
      const ST* base = gr + ((size_t)b * T + min(t0 + erow, T - 1)) * Ds + col0 + 4 * eq;
#pragma unroll
      for (int ct = 0; ct < CC; ++ct) rr[ct] = ldraw4(base + 32 * ct);
    };
    if (GATE) load_rr(t_tile);
    K1_TICK(0)                                             // 0: prologue + VW gathers
#pragma unroll 1
    for (int st = 0; st * SUB < rows_valid; ++st) {
      const int t0 = t_tile + st * SUB;
      const float* Pcur = Pl + st * SUB * PP;
      {                                                    // hand-over #st: all rows of the sub-tile have been scored
        const unsigned need = (unsigned)min(SUB, rows_valid - st * SUB);
        while ((unsigned)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(sync + st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < need)
          __builtin_amdgcn_s_sleep(2);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
      }
      K1_TICK(1)                                           // 1: waiting for the producers
      // land the r rows HERE (requested a whole score loop ago), before any of this sub-tile's stores is issued: a wait placed after
      // them would be vmcnt(0) (the stores sit under row predicates, the compiler cannot count them) and drain the store queue
      if (GATE) {
#pragma unroll
        for (int ct = 0; ct < CC; ++ct) {
          if constexpr (BF) asm volatile("" : "+v"(rr[ct].x), "+v"(rr[ct].y));
          else asm volatile("" : "+v"(rr[ct].x), "+v"(rr[ct].y), "+v"(rr[ct].z), "+v"(rr[ct].w));
        }
      }
      K1_TICK(2)                                           // 2: wait for the r rows
      if (TSG_SKIP(4)) continue;
      k1_u32x4 ph[KS], pl[KS];
      k1_u32x2 ph8, pl8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const float* pr = Pcur + (jl & 7) * PP + 16 * ks + 8 * hh;
        float4 x = *reinterpret_cast<const float4*>(pr), y = *reinterpret_cast<const float4*>(pr + 4);
        if (jl >= 8) { x = make_float4(0.f, 0.f, 0.f, 0.f); y = x; }
        const float e[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        k1_split8(e, ph[ks], pl[ks]);
      }
      float px[XW > 0 ? XW : 1][4];
      if constexpr (XW > 0) {
#pragma unroll
        for (int x = 0; x < XW; ++x)
#pragma unroll
          for (int i = 0; i < 4; ++i) px[x][i] = Pcur[(i + 4 * hh) * PP + 24 + x];
      }
      if (K8) {
        float4 x = *reinterpret_cast<const float4*>(Pcur + (jl & 7) * PP + 16 + 4 * hh);
        if (jl >= 8) x = make_float4(0.f, 0.f, 0.f, 0.f);
        unsigned h0, l0, h1, l1;
        k1_split_pair(x.x, x.y, h0, l0); k1_split_pair(x.z, x.w, h1, l1);
        ph8 = (k1_u32x2){h0, h1}; pl8 = (k1_u32x2){l0, l1};
      }
#pragma unroll
      for (int ct = 0; ct < CC; ++ct) {
        k1_f32x16 o;
#pragma unroll
        for (int i = 0; i < 16; ++i) o[i] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          o = k1_mfma(ph[ks], vh[ct][ks], o);
          if constexpr (!BF) o = k1_mfma(ph[ks], vl[ct][ks], o);            // (bf16 storage: VW has no lo part)
          o = k1_mfma(pl[ks], vh[ct][ks], o);
        }
        if (K8) {
          o = k1_mfma8(ph8, vh8[ct], o);
          if constexpr (!BF) o = k1_mfma8(ph8, vl8[ct], o);
          o = k1_mfma8(pl8, vh8[ct], o);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = o[i];
          if constexpr (XW > 0) {
#pragma unroll
            for (int x = 0; x < XW; ++x) v = fmaf(px[x][i], vx[ct][x], v);
          }
          if (GATE) v = fmaf(v, -kLog2e, gb[ct]);          // (the bias rides in the sigmoid's exponent, applied while the lane still owns ONE column)
          trw[wr_off + 32 * i] = v;
        }
        float4 e = *reinterpret_cast<const float4*>(trw + rd_off);
        if (GATE) {
          const float4 g = cvt4(rr[ct]);
          e = make_float4(g.x * fast_rcp(1.f + fast_exp2(e.x)), g.y * fast_rcp(1.f + fast_exp2(e.y)),
                          g.z * fast_rcp(1.f + fast_exp2(e.z)), g.w * fast_rcp(1.f + fast_exp2(e.w)));
        }
        if (t0 + erow < T) st4(C + ((size_t)b * T + t0 + erow) * Ds + col0 + 32 * ct + 4 * eq, e);
      }
      K1_TICK(3)                                           // 3: phase 2 + stores issued
      if (GATE) load_rr(t0 + SUB);                         // the next sub-tile's r rows: a whole score loop of cover
      K1_TICK(4)                                           // 4: r requests issued
    }
    K1_TICK_DUMP
  }
}

// ------------------------------------------------------------------------------------------
// backward, kernel 1: de[b,t,n] = P*(dP - <P,dP>),  dP[t,n] = <dC[t,:], sent[b,n,:]>
// Workgroup = (b, 32 clips), 8 waves = (column group cg, row slot rs) exactly as forward phase 2:
// the wave's sent[b,:,cols] slice stays in registers, its dC rows stream through (next row in
// flight), each row's NP partial dots are folded with the swap reduction, and the column groups
// meet in LDS.  sent[b] is read once per workgroup instead of once per clip row.
// ------------------------------------------------------------------------------------------
// GATE: dC is replaced by dout (gradient of out = r * sigmoid(G), G = P VW + bias); the kernel recomputes
// G from the VW slice in registers, writes dr = dout*sigmoid(G) and dG = dout*r*sigmoid'(G) (the latter is
// what the dot products below and the column kernel consume in place of dC).
template <int NP, bool GATE>
__global__ __launch_bounds__(kFwdThreads) void scdm_bwd_rows_kernel(
    const float* __restrict__ V, const float* __restrict__ P, const float* __restrict__ dC,
    float* __restrict__ de, const float* __restrict__ gr, const float* __restrict__ gbias,
    float* __restrict__ dGout, float* __restrict__ drout, int B, int T, int N, int Ds, int tiles) {
  constexpr int TT = 32;
  constexpr int CW = NP <= 20 ? 4 : 2;
  __shared__ float part[kFwdWaves][TT][NP + 1];          // [cg][row][n] (padded against bank conflicts)
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles);
  const int b = bid / tiles, t_tile = (bid % tiles) * TT;

  const int cgroups = (Ds + 64 * CW - 1) / (64 * CW);
  int cgp = 1;
  while (cgp < cgroups && cgp < kFwdWaves) cgp <<= 1;
  const int cg = wv % cgp, rs = wv / cgp, rslots = kFwdWaves / cgp;
  const int jcol = cg * 64 * CW + lane * CW;
  const bool jok = jcol < Ds;
  const float* Vb = V + (size_t)b * N * Ds;
  float vreg[NP][CW];
#pragma unroll
  for (int n = 0; n < NP; ++n) {
    const float* src = Vb + (size_t)(n < N ? n : 0) * Ds + (jok ? jcol : 0);
    if (CW == 4) {
      const float4 q = *reinterpret_cast<const float4*>(src);
      vreg[n][0] = q.x; vreg[n][1] = q.y; vreg[n][CW - 2] = q.z; vreg[n][CW - 1] = q.w;
    } else {
      const float2 q = *reinterpret_cast<const float2*>(src);
      vreg[n][0] = q.x; vreg[n][1] = q.y;
    }
    if (n >= N || !jok) {
#pragma unroll
      for (int c = 0; c < CW; ++c) vreg[n][c] = 0.f;
    }
  }
  auto load_g = [&](int tl, float (&g)[CW]) {
    const int t = t_tile + tl;
    const float* src = dC + ((size_t)b * T + (t < T ? t : T - 1)) * Ds + (jok ? jcol : 0);
    if (CW == 4) {
      const float4 q = *reinterpret_cast<const float4*>(src);
      g[0] = q.x; g[1] = q.y; g[CW - 2] = q.z; g[CW - 1] = q.w;
    } else {
      const float2 q = *reinterpret_cast<const float2*>(src);
      g[0] = q.x; g[1] = q.y;
    }
  };
  auto load_r = [&](int tl, float (&g)[CW]) {
    const int t = t_tile + tl;
    const float* src = gr + ((size_t)b * T + (t < T ? t : T - 1)) * Ds + (jok ? jcol : 0);
    if (CW == 4) {
      const float4 q = *reinterpret_cast<const float4*>(src);
      g[0] = q.x; g[1] = q.y; g[CW - 2] = q.z; g[CW - 1] = q.w;
    } else {
      const float2 q = *reinterpret_cast<const float2*>(src);
      g[0] = q.x; g[1] = q.y;
    }
  };
  float gb[CW];
#pragma unroll
  for (int c = 0; c < CW; ++c) gb[c] = (GATE && jcol + c < Ds) ? gbias[jcol + c] : 0.f;
  float g[CW], gn[CW], rr[CW] = {}, rn[CW] = {};
  load_g(rs, g);
  if (GATE) load_r(rs, rr);
  const int q4 = lane >> 4;
  const int nq = ((q4 & 1) << 1) | (q4 >> 1);
  for (int tl = rs; tl < TT; tl += rslots) {
    if (tl + rslots < TT) { load_g(tl + rslots, gn); if (GATE) load_r(tl + rslots, rn); }
    if (GATE) {
      const int t = t_tile + tl;
      const float* prow = P + ((size_t)b * T + (t < T ? t : T - 1)) * N;     // wave-uniform row
      float G[CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) G[c] = gb[c];
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        if (n < N) {
          const float pn = prow[n];
#pragma unroll
          for (int c = 0; c < CW; ++c) G[c] = fmaf(pn, vreg[n][c], G[c]);
        }
      }
      float dr[CW];
#pragma unroll
      for (int c = 0; c < CW; ++c) {
        const float sg = fast_rcp(1.f + fast_exp2(-G[c] * kLog2e));
        dr[c] = g[c] * sg;
        g[c] = g[c] * rr[c] * sg * (1.f - sg);           // dG: from here on "dC"
      }
      if (t < T && jok) {
        float* d0 = dGout + ((size_t)b * T + t) * Ds + jcol; float* d1 = drout + ((size_t)b * T + t) * Ds + jcol;
        if (CW == 4) {
          *reinterpret_cast<float4*>(d0) = make_float4(g[0], g[1], g[CW - 2], g[CW - 1]);
          *reinterpret_cast<float4*>(d1) = make_float4(dr[0], dr[1], dr[CW - 2], dr[CW - 1]);
        } else {
          *reinterpret_cast<float2*>(d0) = make_float2(g[0], g[1]);
          *reinterpret_cast<float2*>(d1) = make_float2(dr[0], dr[1]);
        }
      }
    }
    float dp[NP];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < CW; ++c) acc = fmaf(g[c], vreg[n][c], acc);
      dp[n] = acc;
    }
    float z[NP / 4];
    wave_transpose_sum<NP>(dp, z);
    if ((lane & 15) == 0) {
#pragma unroll
      for (int j = 0; j < NP / 4; ++j) part[cg][tl][4 * j + nq] = z[j];
    }
#pragma unroll
    for (int c = 0; c < CW; ++c) { g[c] = gn[c]; rr[c] = rn[c]; }
  }
  __syncthreads();
  // 32 rows x NP words: one thread per (row, word); the softmax-Jacobian row dot via LDS
  float* dots = &part[0][0][0];                          // reuse: after the sums are taken (2nd barrier)
  float dpv = 0.f, pv = 0.f;
  const int row = tid / NP, n = tid % NP;                // 512 threads cover 25 rows of NP=20 at a time
  for (int r0 = 0; r0 < TT; r0 += kFwdThreads / NP) {
    const int tl = r0 + row, t = t_tile + tl;
    const bool ok = row < kFwdThreads / NP && tl < TT && t < T && n < N;
    dpv = 0.f; pv = 0.f;
    if (ok) {
      for (int c = 0; c < cgp; ++c) dpv += part[c][tl][n];
      pv = P[((size_t)b * T + t) * N + n];
    }
    // <P,dP> over the row: NP consecutive threads hold one row
    __shared__ float prod[kFwdThreads];
    prod[tid] = pv * dpv;
    __syncthreads();
    if (ok) {
      float dot = 0.f;
      const int base = tid - n;
      for (int m = 0; m < N; ++m) dot += prod[base + m];
      de[((size_t)b * T + t) * N + n] = pv * (dpv - dot);
    }
    __syncthreads();
  }
  (void)dots;
}

// ------------------------------------------------------------------------------------------
// backward, kernel 2: one workgroup = (b, 256-column slice c), 8 waves striding over the T rows.
//   da[t,k] = 4 w[k] sum_n v      ds[n,k] = 4 w[k] sum_t v      dw[k] += -2 sum_{t,n} u
//   with u = de[t,n] r,  v = u (1 - r) = de * r(1-r),  r = 1/(Ea Es + 1)
//   dsent[n,j] = sum_t P[t,n] dC[t,j]
// Es for the slice lives in registers (NP*4 per lane); the de / P rows are wave-uniform (scalar
// loads); the next row's a / dC float4 and de / P row are in flight while the current one is used.
// Cross-wave sums go through LDS in word chunks (deterministic order, no atomics except dw).
// ------------------------------------------------------------------------------------------
constexpr int kColThreads = 256;                           // two workgroups per CU at ~234 VGPRs: their serial phases (Es prologue,
                                                          // row staging, cross-wave reductions) overlap the other one's row loop (512: 206 us, 256: 195 us gate bwd)
constexpr int kColWaves = kColThreads / kWave;
constexpr int kCplMax = 2;                                // columns per lane: 2 up to 20 words, 1 beyond (VGPR budget:
                                                          // 2*NP*CPL accumulators; NP = 28 at CPL = 2 spilled 57 VGPRs: 428 vs 207 us)
constexpr int kSliceMax = kWave * kCplMax;                // 128 columns per workgroup at most
constexpr int kRedChunk = 8;                              // words per LDS reduction round: 8*8*128*4 = 32 KiB
template <int NP> constexpr int cpl_of() { return NP <= 20 ? 2 : 1; }

template <int CPL> __device__ __forceinline__ void ld_cols(const float* __restrict__ p, float (&o)[CPL]) {
  if constexpr (CPL == 2) { const float2 t = *reinterpret_cast<const float2*>(p); o[0] = t.x; o[1] = t.y; }
  else o[0] = *p;
}
template <int CPL> __device__ __forceinline__ void st_cols(float* __restrict__ p, const float (&o)[CPL]) {
  if constexpr (CPL == 2) *reinterpret_cast<float2*>(p) = make_float2(o[0], o[1]);
  else *p = o[0];
}
template <int CPL> __device__ __forceinline__ void ld_cols(const bf16_t* __restrict__ p, float (&o)[CPL]) {   // bf16 storage
  if constexpr (CPL == 2) { const float2 t = ld2(p); o[0] = t.x; o[1] = t.y; }
  else o[0] = ld1(p);
}
template <int CPL> __device__ __forceinline__ void st_cols(bf16_t* __restrict__ p, const float (&o)[CPL]) {
  if constexpr (CPL == 2) st2(p, make_float2(o[0], o[1]));
  else st1(p, o[0]);
}
template <int CPL> __device__ __forceinline__ void exp2_cols(float (&v)[CPL]) {
#pragma unroll
  for (int q = 0; q < CPL; ++q) v[q] = fast_exp2(clampf(v[q], -kClamp, kClamp) * k2Log2e);
}

__device__ __forceinline__ float2 exp2x2(float2 v) {
  return make_float2(fast_exp2(clampf(v.x, -kClamp, kClamp) * k2Log2e), fast_exp2(clampf(v.y, -kClamp, kClamp) * k2Log2e));
}

template <int NP, int CPL>
__device__ __forceinline__ void cols_reduce_store(float (&acc)[NP][CPL], float* __restrict__ red, float* __restrict__ out,
                                                  const float* __restrict__ scale, int N, int width, int col0,
                                                  int tid, int lane, int wv) {
  // out[n*width + col0 + c] = scale[c] * sum_waves acc[n][c];  red = [kColWaves][kRedChunk][kSlice]
  constexpr int kSlice = kWave * CPL;
  for (int n0 = 0; n0 < NP; n0 += kRedChunk) {
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      if (n >= n0 && n < n0 + kRedChunk)
        st_cols<CPL>(red + ((wv * kRedChunk + (n - n0)) * kSlice) + lane * CPL, acc[n]);
    }
    __syncthreads();
    for (int idx = tid; idx < kRedChunk * kSlice; idx += kColThreads) {
      const int nn = n0 + idx / kSlice, c = idx % kSlice;
      if (nn < N && col0 + c < width) {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < kColWaves; ++u) sum += red[(u * kRedChunk + (nn - n0)) * kSlice + c];
        out[(size_t)nn * width + col0 + c] = (scale ? scale[c] : 1.f) * sum;
      }
    }
    __syncthreads();
  }
}

constexpr int kColTB = 256;                               // clip rows per LDS block of de / P
constexpr int kColPF = 4;                                 // rows in flight per wave

template <int NP>
__global__ __launch_bounds__(kColThreads) void scdm_bwd_cols_kernel(
    const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ w,
    const float* __restrict__ P, const float* __restrict__ dC, const float* __restrict__ de,
    float* __restrict__ da, float* __restrict__ ds, float* __restrict__ dw, float* __restrict__ dV,
    float* __restrict__ dbias, int B, int T, int N, int H, int Ds, int hslices, int slices) {
  constexpr int CPL = cpl_of<NP>(), kSlice = kWave * CPL;
  __shared__ __align__(16) float red[kColWaves * kRedChunk * kSliceMax];
  __shared__ float wscale[kSliceMax];
  extern __shared__ __align__(16) float rowsl[];          // [min(T,256)][NP]: de rows, later P rows (zero padded)
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, slices);
  const int b = bid / slices, c = bid % slices;
  const int k = c * kSlice + lane * CPL;
  // A wave owns rows wv, wv+8, ... of each 256-row block and keeps kColPF of its rows' operands in
  // flight: a row's compute (~0.5 us) is far shorter than a memory round trip, one-row-ahead starves it.
  auto stage_rows = [&](const float* __restrict__ src, int tb0, int tbn) {   // [tbn][N] -> LDS [tbn][NP]
    __syncthreads();
    for (int idx = tid; idx < tbn * NP; idx += kColThreads) {
      const int r = idx / NP, n = idx % NP;
      rowsl[idx] = (n < N) ? src[((size_t)b * T + tb0 + r) * N + n] : 0.f;
    }
    __syncthreads();
  };

  // ---------------- main: da, ds, dw for hidden columns k, k+1 ---------------------------------
  if (c < hslices) {
    const bool live = k < H;
    const float* arow = a + (size_t)b * T * H + (live ? k : 0);
    float es[NP][CPL];
    const float* sb = s + (size_t)b * N * H;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
#pragma unroll
      for (int q = 0; q < CPL; ++q) es[n][q] = 0.f;
      if (n < N && live) ld_cols<CPL>(sb + (size_t)n * H + k, es[n]);
      exp2_cols<CPL>(es[n]);
    }
    if (tid < kSlice) wscale[tid] = (c * kSlice + tid < H) ? 4.f * w[c * kSlice + tid] : 0.f;
    float dsacc[NP][CPL], dwacc[CPL], ws2[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) { dwacc[q] = 0.f; ws2[q] = 0.f; }
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int q = 0; q < CPL; ++q) dsacc[n][q] = 0.f;

    for (int tb0 = 0; tb0 < T; tb0 += kColTB) {
      const int tbn = (T - tb0 < kColTB) ? T - tb0 : kColTB;
      float ring[kColPF][CPL];
#pragma unroll
      for (int u = 0; u < kColPF; ++u) {
        const int tl = wv + kColWaves * u;
#pragma unroll
        for (int q = 0; q < CPL; ++q) ring[u][q] = 0.f;
        if (tl < tbn) ld_cols<CPL>(arow + (size_t)(tb0 + tl) * H, ring[u]);
      }
      stage_rows(de, tb0, tbn);
      ld_cols<CPL>(wscale + lane * CPL, ws2);
#pragma unroll 1
      for (int tl0 = wv; tl0 < tbn; tl0 += kColWaves * kColPF) {
#pragma unroll
        for (int u = 0; u < kColPF; ++u) {
          const int tl = tl0 + kColWaves * u;
          if (tl < tbn) {                                   // wave-uniform
            float ea[CPL], dasum[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) { ea[q] = ring[u][q]; dasum[q] = 0.f; }
            const int tnx = tl + kColWaves * kColPF;
            if (tnx < tbn) ld_cols<CPL>(arow + (size_t)(tb0 + tnx) * H, ring[u]);
            exp2_cols<CPL>(ea);
#pragma unroll
            for (int n4 = 0; n4 < NP; n4 += 4) {
              const float4 d4 = *reinterpret_cast<const float4*>(rowsl + tl * NP + n4);   // broadcast
              const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                  const float r = fast_rcp(fmaf(ea[q], es[n4 + j][q], 1.f));
                  const float u2 = dd[j] * r;
                  const float v = fmaf(-u2, r, u2);
                  dsacc[n4 + j][q] += v;
                  dasum[q] += v;
                  dwacc[q] += u2;
                }
            }
            if (live) {
#pragma unroll
              for (int q = 0; q < CPL; ++q) dasum[q] *= ws2[q];
              st_cols<CPL>(da + ((size_t)b * T + tb0 + tl) * H + k, dasum);
            }
          }
        }
      }
    }

    cols_reduce_store<NP, CPL>(dsacc, red, ds + (size_t)b * N * H, wscale, N, H, c * kSlice, tid, lane, wv);
    // dw: one more round through the same buffer, then one atomic per column and workgroup
    st_cols<CPL>(red + wv * kSlice + lane * CPL, dwacc);
    __syncthreads();
    if (tid < kSlice && c * kSlice + tid < H) {
      float sum = 0.f;
#pragma unroll
      for (int u = 0; u < kColWaves; ++u) sum += red[u * kSlice + tid];
      atomicAdd(dw + c * kSlice + tid, -2.f * sum);
    }
    __syncthreads();
  }

  // ---------------- dsent[b,n,j] = sum_t P[t,n] dC[t,j] for sentence columns of slice c --------
  if (c * kSlice < Ds) {
    const int j = c * kSlice + lane * CPL;
    const bool jok = j < Ds;
    const float* grow = dC + (size_t)b * T * Ds + (jok ? j : 0);
    float dv[NP][CPL], gsum[CPL];                           // gsum: column sums of dC (= d bias of the fused gate)
#pragma unroll
    for (int q = 0; q < CPL; ++q) gsum[q] = 0.f;
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int q = 0; q < CPL; ++q) dv[n][q] = 0.f;
    for (int tb0 = 0; tb0 < T; tb0 += kColTB) {
      const int tbn = (T - tb0 < kColTB) ? T - tb0 : kColTB;
      float ring[kColPF][CPL];
#pragma unroll
      for (int u = 0; u < kColPF; ++u) {
        const int tl = wv + kColWaves * u;
#pragma unroll
        for (int q = 0; q < CPL; ++q) ring[u][q] = 0.f;
        if (tl < tbn && jok) ld_cols<CPL>(grow + (size_t)(tb0 + tl) * Ds, ring[u]);
      }
      stage_rows(P, tb0, tbn);
#pragma unroll 1
      for (int tl0 = wv; tl0 < tbn; tl0 += kColWaves * kColPF) {
#pragma unroll
        for (int u = 0; u < kColPF; ++u) {
          const int tl = tl0 + kColWaves * u;
          if (tl < tbn) {
            float g[CPL];
#pragma unroll
            for (int q = 0; q < CPL; ++q) { g[q] = ring[u][q]; gsum[q] += g[q]; }
            const int tnx = tl + kColWaves * kColPF;
            if (tnx < tbn && jok) ld_cols<CPL>(grow + (size_t)(tb0 + tnx) * Ds, ring[u]);
#pragma unroll
            for (int n4 = 0; n4 < NP; n4 += 4) {
              const float4 p4 = *reinterpret_cast<const float4*>(rowsl + tl * NP + n4);
              const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
              for (int jn = 0; jn < 4; ++jn)
#pragma unroll
                for (int q = 0; q < CPL; ++q) dv[n4 + jn][q] = fmaf(pp[jn], g[q], dv[n4 + jn][q]);
            }
          }
        }
      }
    }
    cols_reduce_store<NP, CPL>(dv, red, dV + (size_t)b * N * Ds, nullptr, N, Ds, c * kSlice, tid, lane, wv);
    if (dbias) {
      st_cols<CPL>(red + wv * kSlice + lane * CPL, gsum);
      __syncthreads();
      if (tid < kSlice && c * kSlice + tid < Ds) {
        float sum = 0.f;
#pragma unroll
        for (int u = 0; u < kColWaves; ++u) sum += red[u * kSlice + tid];
        atomicAdd(dbias + c * kSlice + tid, sum);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, fused (default): ONE kernel; every input is read once and every output written once
// (the two-kernel path above moves dG and de through HBM: 1.38x the algorithmic traffic).
//
// Workgroup = (batch item b, column part pt), 8 waves.  The columns (of Ds in the row phase, of H in the column
// phase) are cut into slices of SW = 64*CPL; the part owns SP consecutive slices, wave (sw, rq) owns slice sw of
// them and the clip rows rq, rq+RS, rq+2RS, ... (RS = 8/SP row splits).  All T rows of the item stay with the
// workgroup, so the T-sums (ds, dVW) never leave it:
//   row phase   per row t, columns of the wave:  G = P[t,:] VW + bias, sg = sigmoid(G), dr = dout*sg (stored),
//               dG = dout*r*sg*(1-sg);  dVW += P[t,:]^T dG;  dbias += dG;  partial dP[t,n] = <dG, VW[n,:]> over the
//               wave's columns (swap reduction), folded over the part's slices in LDS  (GATE = false: dG = dC)
//   exchange    the dot products need ALL columns: the `parts` workgroups of an item publish their partial dP
//               ([T][NP] floats, agent-scope write-through stores, acknowledged before a RELAXED counter add: no release fence =
//               no L2 write-back of the dr rows just stored), meet on the counter and each sums the partials in part order
//               (deterministic).  de = P (dP - <P,dP>) -> LDS.  Co-dispatched neighbours (consecutive block ids),
//               bounded spin; 10 KiB per workgroup instead of the 1 MiB dG round trip.
//   column phase per row t, columns k of the wave: r = 1/(Ea Es[n]+1), q = r - r^2:
//               da[t,k] = 4w Σ_n de q (stored), ds[n,k] += de q, dw[k] += de r      (5 VALU + 1 rcp per element)
//   end         row splits folded through LDS in fixed order; ds, dVW stored; dw, dbias: one atomic per column.
// ------------------------------------------------------------------------------------------
constexpr int kFusedThreads = 512, kFusedWaves = kFusedThreads / kWave;
constexpr int kFusedSub = 32;       // rows per dP folding round
constexpr int kFusedPF = 4;         // rows in flight per wave
constexpr unsigned kXchSpinLimit = 1u << 22;   // bounded wait on the neighbours (~seconds); expiry sets the error sink
template <int NP> constexpr int fused_cpl() { return NP <= 20 ? 2 : 1; }

struct FusedPlan { int parts, SP, grid; size_t lds; long long ws_bytes; bool ok; bool mrow; size_t lds_mrow; };
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMrowWk = 32 * 33;               // floats of a wave's scratch block (dG transpose tile, then its dP tile)

// MROW = true: the variant with the row phase on the matrix pipe; its waves own 64-column slices (two 32-column MFMA
// tiles: 32 accumulator registers for dVW instead of 64 -- with 128-column slices the gate variant spilled 110 registers).
template <int NP, bool MROW>
FusedPlan fused_plan(int B, int T, int N, int H, int Ds) {
  constexpr int SW = MROW ? kWave : kWave * fused_cpl<NP>();
  FusedPlan p{};
  int S = cdiv(H, SW) > cdiv(Ds, SW) ? cdiv(H, SW) : cdiv(Ds, SW), S2 = 1;
  while (S2 < S) S2 <<= 1;
  int parts = 1;
  while (parts < S2 && ((long)B * parts < 256 || S2 / parts > kFusedWaves)) parts <<= 1;
  p.parts = parts; p.SP = S2 / parts; p.grid = B * parts;
  const int RS = kFusedWaves / p.SP;
  const size_t part_f = (size_t)p.SP * kFusedSub * (NP + 1), red_f = (size_t)(RS - 1) * p.SP * NP * SW;
  p.ws_bytes = (long long)sizeof(float) * ((long long)B * parts * T * NP + roundup(B, 4));
  if (!MROW) {
    p.lds = sizeof(float) * ((size_t)2 * T * NP + (part_f > red_f ? part_f : red_f));
    p.ok = p.SP <= kFusedWaves && p.lds <= (size_t)kLdsBytes - 1024;
    p.mrow = false; p.lds_mrow = 0;
    return p;
  }
  // P / de tiles padded to 32 rows, a scratch block and the VW slice [NP][SW+4] per wave (the row-split partials of the
  // column phase and of dVW overlay the VW slices at the end)
  const size_t t32 = (size_t)roundup(T, 32), vw_f = (size_t)kFusedWaves * NP * (SW + 4);
  const size_t red_m = (size_t)(RS - 1) * p.SP * (SW / 32) * 1024;                       // dVW accumulator tiles of the row splits
  const size_t red_c = (fused_cpl<NP>() == 2 && p.SP % 2 == 0) ? (size_t)(2 * RS - 1) * (p.SP / 2) * NP * 2 * kWave : 0;   // column phase regrouped
  size_t tail = vw_f > red_f ? (vw_f > red_m ? vw_f : red_m) : (red_f > red_m ? red_f : red_m);
  if (red_c > tail) tail = red_c;
  p.lds = p.lds_mrow = sizeof(float) * (2 * t32 * NP + (size_t)kFusedWaves * kMrowWk + tail);
  p.ok = p.mrow = p.SP <= kFusedWaves && p.lds <= (size_t)kLdsBytes - 1024;
  return p;
}

// MROW: the row phase on the matrix pipe (see the block comment inside).
// ST: storage type of the activations and their gradients (a, s, V, dC, gr in; da, ds, dV, dr out): float or bf16_t (TSG_BF16).
// P, w, gbias, dw, dbias, the exchange workspace and all arithmetic stay fp32.
template <int NP, bool GATE, bool MROW, typename ST>
__global__ __launch_bounds__(kFusedThreads) void scdm_bwd_fused_kernel(
    const ST* __restrict__ a, const ST* __restrict__ s, const float* __restrict__ w, const ST* __restrict__ V,
    const float* __restrict__ P, const ST* __restrict__ dC, const ST* __restrict__ gr, const float* __restrict__ gbias,
    ST* __restrict__ da, ST* __restrict__ ds, float* __restrict__ dw, ST* __restrict__ dV, float* __restrict__ dbias,
    ST* __restrict__ drout, float* xch, unsigned* __restrict__ cnt, ErrSink esink,
    int B, int T, int N, int H, int Ds, int parts, int SP, int dbg) {
  constexpr int CPL = MROW ? 1 : fused_cpl<NP>(), SW = kWave * CPL;
  extern __shared__ __align__(16) float lds[];
  __shared__ unsigned xch_failed;               // set by thread 0 when the bounded wait on the partner workgroups expired
  const int TL = MROW ? ((T + 31) & ~31) : T;   // rows of the P / de tiles in LDS (MROW: padded to whole 32-row MFMA tiles)
  float* Pl = lds;                              // [TL][NP]  P rows, zero beyond N (and beyond T)
  float* De = Pl + (size_t)TL * NP;             // [TL][NP]  partial dP -> de
  float* part = De + (size_t)TL * NP;           // [SP][kFusedSub][NP+1] per-slice partial dP of a folding round; later the
  float* red = MROW ? part + kFusedWaves * kMrowWk : part;   // [RS-1][SP][NP][SW] row-split partials of the T-sums
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, parts);
  const int b = bid / parts, pt = bid % parts;
  const int RS = kFusedWaves / SP, sw = wv % SP, rq = wv / SP;
  const int col = (pt * SP + sw) * SW + lane * CPL;
  const bool jok = col < Ds, kok = col < H;
  const size_t rowDs = (size_t)b * T * Ds + (jok ? col : 0), rowH = (size_t)b * T * H + (kok ? col : 0);
  const int nrows = rq < T ? (T - rq + RS - 1) / RS : 0;          // this wave's rows: t = rq + RS*i

  for (int idx = tid; idx < TL * NP; idx += kFusedThreads) {
    const int r = idx / NP, n = idx % NP;
    Pl[idx] = (n < N && r < T) ? P[((size_t)b * T + r) * N + n] : 0.f;
  }

  // Publish this part's partial dP rows (De complete, workgroup met) and count it in: called right after the row loop, BEFORE the
  // T-sum epilogue of the row phase (dVW stores, dbias atomics), so that the partners' round trip runs under that epilogue.
  // The rows are agent-scope (write-through, sc1) stores; EVERY thread waits for the acknowledgement of its own stores with an
  // explicit `s_waitcnt vmcnt(0)` before the workgroup barrier -- the barrier's workgroup-scope fence is `s_waitcnt lgkmcnt(0)` only
  // on gfx950 and does NOT wait for vector stores (round-3 review: the shipped ISA had store -> s_barrier -> atomic with no vmcnt
  // wait, so a partner could count this part in and read rows that had not reached L2).  With the wait the rows are visible
  // device-wide before the counter moves.  Still no release fence: a fence here is an L2 WRITE-BACK (buffer_wbl2), and at this
  // point the L2 holds this workgroup's 256 KiB of freshly stored dr rows -- every workgroup of the XCD would stall on flushing
  // them in the middle of the kernel (173 -> 147 us per launch at [128,128,20,1024], measured in round 3 with a fenced build).
  // tests/test_isa_cpu.py disassembles the built code object and asserts the vmcnt(0) wait between the last sc1 store and the
  // counter's atomic in every instantiation.
  // Two steps (round 4).  publish_dp: the stores, right after the row loop.  count_in: the wait + barrier + ticket, AFTER the T-sum
  // epilogue of the row phase -- vmcnt counts loads and stores together and in order, so the acknowledgement wait also covers the
  // wave's queue of dr row stores; behind the epilogue that queue has drained under the epilogue's LDS work.  A/B on one box
  // (profiles/r4/k1g_bwd_ack_wait_ab_v1.txt; the round-3 protocol without the wait): 165.4 / 166.4 us with the wait, 166.1 / 164.3
  // without, at [128,128,20,1024] -- the acknowledgement wait is free.
  auto publish_dp = [&]() {
    if (parts <= 1) return;
    float* mine = xch + ((size_t)b * parts + pt) * T * NP;
    for (int idx = tid; idx < T * NP; idx += kFusedThreads)
      __hip_atomic_store(mine + idx, De[idx], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };
  auto count_in = [&]() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // acknowledgement of THIS thread's partial rows (write-through to L2 / memory)
    __syncthreads();
    if (tid == 0) {
      xch_failed = 0u;
      __hip_atomic_fetch_add(cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  };

  const bool abl_nobar = TSG_SKIP(512), abl_swap = TSG_SKIP(1024) && wv >= kFusedWaves / 2;
  // ---------------- column phase -----------------------------------------------------------------
  // Generic in the columns per lane: the MFMA-row variant works on 64-column slices in its row phase (two MFMA tiles per wave), but
  // its column phase is the VALU loop of the other variant and amortises its per-row overhead (exponential of the a row, de broadcast,
  // da store, loop control) better over 2 columns per lane: the waves regroup as SP/2 slices of 128 columns x twice the row splits.
  auto column_phase = [&](auto cpl_tag, const int SP, const int RS) {
    constexpr int CPL = decltype(cpl_tag)::value, SW = kWave * CPL;
    const int sw = wv % SP, rq = wv / SP;
    const int col = (pt * SP + sw) * SW + lane * CPL;
    const bool kok = col < H;
    const size_t rowH = (size_t)b * T * H + (kok ? col : 0);
    const int nrows = rq < T ? (T - rq + RS - 1) / RS : 0;          // this wave's rows: t = rq + RS*i
    float es[NP][CPL], dsacc[NP][CPL], dwacc[CPL], w4[CPL];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) { es[n][c] = 0.f; dsacc[n][c] = 0.f; }
      if (n < N && kok) ld_cols<CPL>(s + ((size_t)b * N + n) * H + col, es[n]);
      exp2_cols<CPL>(es[n]);
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) { dwacc[c] = 0.f; w4[c] = (col + c < H) ? 4.f * w[col + c] : 0.f; }
    float aring[kFusedPF][CPL];
#pragma unroll
    for (int u = 0; u < kFusedPF; ++u) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) aring[u][c] = 0.f;
      if (u < nrows && kok) ld_cols<CPL>(a + rowH + (size_t)(rq + RS * u) * H, aring[u]);
    }
#pragma unroll 1
    for (int i0 = 0; i0 < nrows; i0 += kFusedPF) {
#pragma unroll
      for (int u = 0; u < kFusedPF; ++u) {
        const int i = i0 + u, t = rq + RS * i;
        if (i < nrows && !TSG_SKIP(32)) {                           // wave-uniform
          float ea[CPL], dasum[CPL];
#pragma unroll
          for (int c = 0; c < CPL; ++c) { ea[c] = aring[u][c]; dasum[c] = 0.f; }
          const int inx = i + kFusedPF;
          if (inx < nrows && kok) ld_cols<CPL>(a + rowH + (size_t)(rq + RS * inx) * H, aring[u]);
          exp2_cols<CPL>(ea);
          const float* drow = De + t * NP;
#pragma unroll
          for (int n4 = 0; n4 < NP; n4 += 4) {
            const float4 d4 = *reinterpret_cast<const float4*>(drow + n4);   // broadcast
            const float dd[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
              for (int c = 0; c < CPL; ++c) {
                const float r = fast_rcp(fmaf(ea[c], es[n4 + j][c], 1.f));
                const float q = fmaf(-r, r, r);
                dsacc[n4 + j][c] = fmaf(dd[j], q, dsacc[n4 + j][c]);
                dasum[c] = fmaf(dd[j], q, dasum[c]);
                dwacc[c] = fmaf(dd[j], r, dwacc[c]);
              }
          }
          if (kok) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) dasum[c] *= w4[c];
            st_cols<CPL>(da + rowH + (size_t)t * H, dasum);
          }
        }
      }
    }
    // T-sums of the column phase
    if (!abl_nobar) __syncthreads();                               // De no longer read; `red` overlays `part` only, but keep order simple
    for (int q = 1; q < RS; ++q) {
      if (rq == q) {
#pragma unroll
        for (int n = 0; n < NP; ++n) st_cols<CPL>(red + (((size_t)(q - 1) * SP + sw) * NP + n) * SW + lane * CPL, dsacc[n]);
      }
    }
    if (!abl_nobar) lds_barrier();
    if (rq == 0) {
      for (int q = 1; q < RS; ++q) {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
          float o[CPL];
          ld_cols<CPL>(red + (((size_t)(q - 1) * SP + sw) * NP + n) * SW + lane * CPL, o);
#pragma unroll
          for (int c = 0; c < CPL; ++c) dsacc[n][c] += o[c];
        }
      }
      if (kok) {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
          if (n < N) {
#pragma unroll
            for (int c = 0; c < CPL; ++c) dsacc[n][c] *= w4[c];
            st_cols<CPL>(ds + ((size_t)b * N + n) * H + col, dsacc[n]);
          }
        }
      }
    }
    if (!abl_nobar) lds_barrier();
    if (rq > 0) st_cols<CPL>(red + ((size_t)(rq - 1) * SP + sw) * SW + lane * CPL, dwacc);
    if (!abl_nobar) lds_barrier();
    if (rq == 0 && kok) {
      for (int q = 1; q < RS; ++q) {
        float o[CPL];
        ld_cols<CPL>(red + ((size_t)(q - 1) * SP + sw) * SW + lane * CPL, o);
#pragma unroll
        for (int c = 0; c < CPL; ++c) dwacc[c] += o[c];
      }
#pragma unroll
      for (int c = 0; c < CPL; ++c)
        if (col + c < H && !TSG_SKIP(256)) atomicAdd(dw + col + c, -2.f * dwacc[c]);
    }
  };
  auto run_column = [&]() {
    if constexpr (MROW && fused_cpl<NP>() == 2) {
      if (SP % 2 == 0) column_phase(std::integral_constant<int, 2>{}, SP / 2, 2 * RS);
      else column_phase(std::integral_constant<int, 1>{}, SP, RS);
    } else {
      column_phase(std::integral_constant<int, CPL>{}, SP, RS);
    }
  };
  // Timing-only experiment (-DTSG_ABLATE builds, results are garbage): bit 512 drops every workgroup barrier inside the two phases and the
  // partner exchange (the reference point), bit 1024 additionally lets waves 4..7 run their column phase BEFORE their row phase, so that
  // each SIMD holds one wave in the memory-bound row phase and one in the VALU-bound column phase: the upper bound of what a
  // tile-pipelined kernel (row phase of tile i + 1 under the column phase of tile i) could gain.
  // ---------------- row phase ------------------------------------------------------------------
  if constexpr (MROW) {
    // The row phase is three small GEMMs per 32-row tile and 32-column tile of the wave's slice -- G = P VW (K = N words),
    // dVW += P^T dG (K = rows) and dP += dG VW^T (K = columns) -- i.e. 3 T N Ds multiply-adds per pair, as much VALU work as
    // the column phase when done with FMAs plus a 64-lane reduction per row.  Here they run as v_mfma_f32_32x32x2_f32
    // (exact fp32), 42 per (row tile, column tile):
    //   G    : A = P[t = lane&31][n = 2s + kk] (LDS), B = VW[n][j = lane&31] (the wave's slice in LDS)  -> D: column j on the
    //          lane, 16 rows t = rho(v, kk) in registers.  dout / r are loaded and dr is stored in that layout (one dword per
    //          lane: 128 contiguous bytes per half wave); sigmoid, dr and dG are formed element-wise in it.
    //   dVW  : dG's ROW index is the contraction index, so the accumulator tile is the B operand as it stands (no lane
    //          movement); A = P^T[n = lane&31][t = rho(v, kk)] from LDS.  Accumulates over the wave's row tiles.
    //   dP   : contracts over dG's COLUMN (lane) index: the tile goes through a 32 x 33 LDS block once and comes back as the
    //          A operand [t = lane&31][j = 2s + kk]; B = VW^T.  The MFMA does the reduction the VALU version needs the swap /
    //          DPP ladder for; the wave's dP tile lands in the same block for the fold over the slices.
    constexpr int CTW = SW / 32, VS = SW + 4;
    float* Wk = part;                                               // [8][kMrowWk]
    float* myW = Wk + wv * kMrowWk;
    float* myV = Wk + kFusedWaves * kMrowWk + wv * NP * VS;          // [NP][VS]  VW[b, :, slice] (zero beyond N / Ds)
    const int jl = lane & 31, kk = lane >> 5;
    const int colw = (pt * SP + sw) * SW;
    for (int idx = lane; idx < NP * (SW / 4); idx += kWave) {
      const int n = idx / (SW / 4), c4 = (idx % (SW / 4)) * 4;
      float4 v4 = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && colw + c4 < Ds) v4 = ld4(V + ((size_t)b * N + n) * Ds + colw + c4);
      *reinterpret_cast<float4*>(myV + n * VS + c4) = v4;
    }
    f32x16 dvw[CTW];
    float gsum[CTW], gb[CTW];
#pragma unroll
    for (int ct = 0; ct < CTW; ++ct) {
#pragma unroll
      for (int v = 0; v < 16; ++v) dvw[ct][v] = 0.f;
      gsum[ct] = 0.f;
      gb[ct] = (GATE && colw + 32 * ct + jl < Ds) ? gbias[colw + 32 * ct + jl] : 0.f;
    }
    __syncthreads();                                               // Pl and the VW slices staged
    if (abl_swap) run_column();
    for (int r0 = 0; r0 < TL; r0 += 32 * RS) {
      const int t0 = r0 + 32 * rq;                                 // this wave's row tile of the round
      if (t0 < TL && !TSG_SKIP(16)) {                              // wave-uniform
        // every read below is UNCONDITIONAL on a clamped address and masked by a select afterwards: a conditional read
        // becomes an exec-masked branch per element and a wait in front of every MFMA
        const int jn = jl < NP ? jl : 0;
        const float nmask = jl < NP ? 1.f : 0.f;
        f32x16 dpacc;
#pragma unroll
        for (int v = 0; v < 16; ++v) dpacc[v] = 0.f;
        // RAG = false: all 32 rows of the tile exist (row v of this lane sits rho(v) * Ds elements behind its first: a constant
        // times a wave-uniform stride, no per-row registers); RAG = true: the last, partial tile (rows clamped and masked)
        auto run_tile = [&](auto rag_tag) {
          constexpr bool RAG = decltype(rag_tag)::value;
          float pa[NP / 2], ptt[16];
#pragma unroll
          for (int s2 = 0; s2 < NP / 2; ++s2) pa[s2] = Pl[(t0 + jl) * NP + 2 * s2 + kk];
#pragma unroll
          for (int v = 0; v < 16; ++v) ptt[v] = Pl[(t0 + (v & 3) + 8 * (v >> 2) + 4 * kk) * NP + jn] * nmask;
          const int tfirst = t0 + 4 * kk;
#pragma unroll
          for (int ct = 0; ct < CTW; ++ct) {
            const int col = colw + 32 * ct + jl;
            const bool cok = col < Ds;
            const float cmask = cok ? 1.f : 0.f;
            const size_t base = ((size_t)b * T + (RAG ? 0 : tfirst)) * Ds + (cok ? col : 0);
            const ST* gp = dC + base;
            const ST* rp = gr + base;
            ST* op = drout + base;
            auto ro = [&](int v) -> size_t {                        // element offset of row v from the base
              const int dt = (v & 3) + 8 * (v >> 2);
              if (!RAG) return (size_t)dt * Ds;
              const int t = tfirst + dt;
              return (size_t)(t < T ? t : T - 1) * Ds;
            };
            auto live = [&](int v) -> bool { return !RAG || tfirst + (v & 3) + 8 * (v >> 2) < T; };
            float g[16], rr[16];
#pragma unroll
            for (int v = 0; v < 16; ++v) g[v] = ld1(gp + ro(v));
            if (GATE) {
#pragma unroll
              for (int v = 0; v < 16; ++v) rr[v] = ld1(rp + ro(v));
            }
            float vb[NP / 2];
#pragma unroll
            for (int s2 = 0; s2 < NP / 2; ++s2) vb[s2] = myV[(2 * s2 + kk) * VS + 32 * ct + jl];
#pragma unroll
            for (int v = 0; v < 16; ++v) g[v] = live(v) ? g[v] * cmask : 0.f;
            if (GATE) {
              f32x16 G;
#pragma unroll
              for (int v = 0; v < 16; ++v) G[v] = 0.f;
#pragma unroll
              for (int s2 = 0; s2 < NP / 2; ++s2) G = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[s2], vb[s2], G, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int v = 0; v < 16; ++v) {
                const float sg = fast_rcp(1.f + fast_exp2(-(G[v] + gb[ct]) * kLog2e));
                const float drv = g[v] * sg;
                if (cok && live(v)) st1(op + ro(v), drv);
                g[v] = g[v] * rr[v] * sg * (1.f - sg);             // dG: from here on "dC" (g is 0 outside the matrix)
              }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) gsum[ct] += g[v];
#pragma unroll
            for (int v = 0; v < 16; ++v) myW[((v & 3) + 8 * (v >> 2) + 4 * kk) * 33 + jl] = g[v];
            __builtin_amdgcn_sched_barrier(0);                     // the operand reads below stay below (registers)
#pragma unroll
            for (int v = 0; v < 16; ++v) dvw[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(ptt[v], g[v], dvw[ct], 0, 0, 0);
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                       // four operand pairs at a time: reads fly under the MFMAs before them
              float av[4], vt[4];
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                av[u] = myW[jl * 33 + 2 * (4 * q4 + u) + kk];
                vt[u] = myV[jn * VS + 32 * ct + 2 * (4 * q4 + u) + kk] * nmask;
              }
#pragma unroll
              for (int u = 0; u < 4; ++u) dpacc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], vt[u], dpacc, 0, 0, 0);
              __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);                     // one column tile's operands at a time (registers)
          }
        };
        if (t0 + 32 <= T) run_tile(std::false_type{}); else run_tile(std::true_type{});
        if (jl < NP) {                                             // dP tile [t][n]: column n on the lane
#pragma unroll
          for (int v = 0; v < 16; ++v) myW[((v & 3) + 8 * (v >> 2) + 4 * kk) * (NP + 1) + jl] = dpacc[v];
        }
      }
      if (!abl_nobar) lds_barrier();
      for (int idx = tid; idx < 32 * RS * NP; idx += kFusedThreads) {
        const int row = idx / NP, n = idx % NP, t = r0 + row;
        if (t < TL) {
          float sum = 0.f;
          for (int c = 0; c < SP; ++c) sum += Wk[((row >> 5) * SP + c) * kMrowWk + (row & 31) * (NP + 1) + n];
          De[t * NP + n] = sum;
        }
      }
      if (!abl_nobar) lds_barrier();
    }
    publish_dp();
    // T-sums of the row phase.  The accumulator tiles hold dVW[n = rho(v, kk)][column jl]; row splits rq > 0 hand theirs to
    // rq = 0 through LDS (over the VW slices, which are dead now), in fixed order.
    float* redm = Wk + kFusedWaves * kMrowWk;                      // [RS-1][SP][CTW][16][64]
    for (int q = 1; q < RS; ++q) {
      if (rq == q) {
#pragma unroll
        for (int ct = 0; ct < CTW; ++ct)
#pragma unroll
          for (int v = 0; v < 16; ++v) redm[((((size_t)(q - 1) * SP + sw) * CTW + ct) * 16 + v) * 64 + lane] = dvw[ct][v];
      }
    }
    if (!abl_nobar) lds_barrier();
    if (rq == 0) {
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        const int col = colw + 32 * ct + jl;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
          float acc = dvw[ct][v];
          for (int q = 1; q < RS; ++q) acc += redm[((((size_t)(q - 1) * SP + sw) * CTW + ct) * 16 + v) * 64 + lane];
          const int n = (v & 3) + 8 * (v >> 2) + 4 * kk;
          if (n < N && col < Ds) st1(dV + ((size_t)b * N + n) * Ds + col, acc);
        }
      }
    }
    if (GATE) {                                                    // dbias: the two half waves hold different rows of a column
#pragma unroll
      for (int ct = 0; ct < CTW; ++ct) {
        const float gs = gsum[ct] + __shfl_xor(gsum[ct], 32, 64);
        const int col = colw + 32 * ct + jl;
        if (kk == 0 && col < Ds && !TSG_SKIP(256)) atomicAdd(dbias + col, gs);
      }
    }
    if (!abl_nobar) lds_barrier();                                                 // redm is read; the column phase may overlay it
  } else {
    float vreg[NP][CPL], dvacc[NP][CPL], gsum[CPL], gb[CPL];
#pragma unroll
    for (int n = 0; n < NP; ++n) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) { vreg[n][c] = 0.f; dvacc[n][c] = 0.f; }
      if (n < N && jok) ld_cols<CPL>(V + ((size_t)b * N + n) * Ds + col, vreg[n]);
    }
#pragma unroll
    for (int c = 0; c < CPL; ++c) { gsum[c] = 0.f; gb[c] = (GATE && col + c < Ds) ? gbias[col + c] : 0.f; }
    float gring[kFusedPF][CPL], rring[kFusedPF][CPL];
#pragma unroll
    for (int u = 0; u < kFusedPF; ++u) {
#pragma unroll
      for (int c = 0; c < CPL; ++c) { gring[u][c] = 0.f; rring[u][c] = 0.f; }
      if (u < nrows && jok) {
        ld_cols<CPL>(dC + rowDs + (size_t)(rq + RS * u) * Ds, gring[u]);
        if (GATE) ld_cols<CPL>(gr + rowDs + (size_t)(rq + RS * u) * Ds, rring[u]);
      }
    }
    const int q4 = lane >> 4, nq = ((q4 & 1) << 1) | (q4 >> 1);
    const int per_sub = kFusedSub / RS;                           // this wave's rows per folding round (RS divides 32)
    if (!abl_nobar) __syncthreads();                                              // Pl staged
    for (int sb0 = 0, i0 = 0; sb0 < T; sb0 += kFusedSub, i0 += per_sub) {
#pragma unroll 1
      for (int ib = 0; ib < per_sub; ib += kFusedPF) {
#pragma unroll
        for (int u = 0; u < kFusedPF; ++u) {
          const int i = i0 + ib + u, t = rq + RS * i;
          if (ib + u < per_sub && i < nrows && !TSG_SKIP(16)) {    // wave-uniform
            float g[CPL], rr[CPL];
#pragma unroll
            for (int c = 0; c < CPL; ++c) { g[c] = gring[u][c]; rr[c] = rring[u][c]; }
            const int inx = i + kFusedPF;                         // same ring slot, kFusedPF rows ahead (per_sub % PF == 0 or tail)
            if (inx < nrows && jok) {
              ld_cols<CPL>(dC + rowDs + (size_t)(rq + RS * inx) * Ds, gring[u]);
              if (GATE) ld_cols<CPL>(gr + rowDs + (size_t)(rq + RS * inx) * Ds, rring[u]);
            }
            const float* prow = Pl + t * NP;
            if (GATE) {
              float G[CPL];
#pragma unroll
              for (int c = 0; c < CPL; ++c) G[c] = gb[c];
#pragma unroll
              for (int n4 = 0; n4 < NP; n4 += 4) {
                const float4 p4 = *reinterpret_cast<const float4*>(prow + n4);
                const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                  for (int c = 0; c < CPL; ++c) G[c] = fmaf(pp[j], vreg[n4 + j][c], G[c]);
              }
              float drv[CPL];
#pragma unroll
              for (int c = 0; c < CPL; ++c) {
                const float sg = fast_rcp(1.f + fast_exp2(-G[c] * kLog2e));
                drv[c] = g[c] * sg;
                g[c] = g[c] * rr[c] * sg * (1.f - sg);             // dG: from here on "dC"
              }
              if (jok) st_cols<CPL>(drout + rowDs + (size_t)t * Ds, drv);
            }
            float dp[NP];
#pragma unroll
            for (int n4 = 0; n4 < NP; n4 += 4) {
              const float4 p4 = *reinterpret_cast<const float4*>(prow + n4);
              const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int c = 0; c < CPL; ++c) {
                  dvacc[n4 + j][c] = fmaf(pp[j], g[c], dvacc[n4 + j][c]);
                  acc = fmaf(g[c], vreg[n4 + j][c], acc);
                }
                dp[n4 + j] = acc;
              }
            }
#pragma unroll
            for (int c = 0; c < CPL; ++c) gsum[c] += g[c];
            float z[NP / 4];
            if (TSG_SKIP(128)) {
#pragma unroll
              for (int j = 0; j < NP / 4; ++j) z[j] = dp[j];
            } else {
              wave_transpose_sum<NP>(dp, z);
            }
            if ((lane & 15) == 0) {
#pragma unroll
              for (int j = 0; j < NP / 4; ++j) part[(sw * kFusedSub + (t - sb0)) * (NP + 1) + 4 * j + nq] = z[j];
            }
          }
        }
      }
      if (!abl_nobar) lds_barrier();
      const int sbn = T - sb0 < kFusedSub ? T - sb0 : kFusedSub;
      for (int idx = tid; idx < sbn * NP; idx += kFusedThreads) {
        const int r = idx / NP, n = idx % NP;
        float sum = 0.f;
        for (int c = 0; c < SP; ++c) sum += part[(c * kFusedSub + r) * (NP + 1) + n];
        De[(sb0 + r) * NP + n] = sum;
      }
      if (!abl_nobar) lds_barrier();
    }
    publish_dp();
    // T-sums of the row phase: dVW and the gate bias gradient.  Row splits rq > 0 hand theirs to rq = 0 through LDS.
    for (int q = 1; q < RS; ++q) {
      if (rq == q) {
#pragma unroll
        for (int n = 0; n < NP; ++n) st_cols<CPL>(red + (((size_t)(q - 1) * SP + sw) * NP + n) * SW + lane * CPL, dvacc[n]);
      }
    }
    if (!abl_nobar) lds_barrier();
    if (rq == 0) {
      for (int q = 1; q < RS; ++q) {
#pragma unroll
        for (int n = 0; n < NP; ++n) {
          float o[CPL];
          ld_cols<CPL>(red + (((size_t)(q - 1) * SP + sw) * NP + n) * SW + lane * CPL, o);
#pragma unroll
          for (int c = 0; c < CPL; ++c) dvacc[n][c] += o[c];
        }
      }
      if (jok) {
#pragma unroll
        for (int n = 0; n < NP; ++n)
          if (n < N) st_cols<CPL>(dV + ((size_t)b * N + n) * Ds + col, dvacc[n]);
      }
    }
    if (GATE) {                                                    // dbias: fold the row splits with a second round
      if (!abl_nobar) lds_barrier();
      if (rq > 0) st_cols<CPL>(red + ((size_t)(rq - 1) * SP + sw) * SW + lane * CPL, gsum);
      if (!abl_nobar) lds_barrier();
      if (rq == 0 && jok) {
        for (int q = 1; q < RS; ++q) {
          float o[CPL];
          ld_cols<CPL>(red + ((size_t)(q - 1) * SP + sw) * SW + lane * CPL, o);
#pragma unroll
          for (int c = 0; c < CPL; ++c) gsum[c] += o[c];
        }
#pragma unroll
        for (int c = 0; c < CPL; ++c)
          if (col + c < Ds && !TSG_SKIP(256)) atomicAdd(dbias + col + c, gsum[c]);
      }
    }
  }

  // ---------------- exchange: dP over ALL columns, then de -------------------------------------
  // (partial rows stored right after the row loop: `publish_dp`; counted in here, behind the T-sum epilogue: `count_in`)
  if (parts > 1 && !abl_nobar) {
    count_in();
    if (tid == 0) {
      unsigned spins = 0;
      while (!TSG_SKIP(64) && __hip_atomic_load(cnt + b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)parts) {
        if (++spins > kXchSpinLimit) {
          report_expiry(esink);                  // host sink + device word (the optimizer's guard reads the latter)
          xch_failed = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // invalidate only (buffer_inv): the partners' rows are read after this; no write-back
      // the invalidate completes asynchronously: this wait holds the workgroup barrier below until it has (MI355X_MICROARCH.md,
      // "Consumer, always: one relaxed poll -> one agent acquire -> s_waitcnt vmcnt(0) -> barrier -> plain loads")
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (!abl_nobar) __syncthreads();
    // every part's partial in part order (deterministic).  Plain loads: thread 0's agent-scope acquire followed by the
    // workgroup barrier orders them after the neighbours' published stores, and they can be issued back to back (atomic
    // loads are kept in program order by the compiler: one memory round trip each, ~40 in a row).
    // An expired wait poisons dP with NaN: da / ds / dw of this item then carry the failure instead of a finite sum of
    // incomplete partials (a finite-but-wrong gradient would pass every isfinite guard downstream).
    const float poison = xch_failed ? __int_as_float(0x7FC00000) : 0.f;
    for (int idx = tid; idx < T * NP; idx += kFusedThreads) {
      float v = poison;
      for (int p = 0; p < parts; ++p) v += xch[((size_t)b * parts + p) * T * NP + idx];
      De[idx] = v;
    }
    if (!abl_nobar) __syncthreads();
  } else {
    if (!abl_nobar) __syncthreads();
  }
  for (int r = tid; r < T; r += kFusedThreads) {
    float dp[NP], dot = 0.f;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      dp[n] = De[r * NP + n];
      dot = fmaf(Pl[r * NP + n], dp[n], dot);
    }
#pragma unroll
    for (int n = 0; n < NP; ++n) De[r * NP + n] = Pl[r * NP + n] * (dp[n] - dot);      // 0 for padded words (P = 0)
  }
  if (!abl_nobar) __syncthreads();

  if (!abl_swap) run_column();
}

// zero up to three small accumulator blocks with ONE launch (dw, dbias, the exchange counters)
__global__ void zero3_kernel(unsigned* p0, int n0, unsigned* p1, int n1, unsigned* p2, int n2) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += gridDim.x * blockDim.x) {
    if (i < n0) p0[i] = 0u;
    else if (i < n0 + n1) p1[i - n0] = 0u;
    else p2[i - n0 - n1] = 0u;
  }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------

template <int NP, bool GATE, typename ST>
int launch_fwd(const ST* a, const ST* s, const float* w, const ST* V, ST* C, float* P,
               const ST* gr, const float* gbias, int B, int T, int N, int H, int Ds, hipStream_t st) {
  constexpr int R = 1;                       // clip rows per wave and sub-tile (sub-tile = 8R rows)
  constexpr int SUB = kFwdWaves * R;
  // rows per workgroup: as many as still give every CU a workgroup (the Es prologue, the launch ramp and the store
  // tail are paid once per workgroup: 64 rows at 128 pairs x 128 clips measured 80 us vs 88 us for 32-row groups)
  int TT = 64;
  while (TT > SUB && (long)B * cdiv(T, TT) < 256) TT >>= 1;
  static const int tt_env = [] { const char* e = getenv("TSG_K1_TT"); return e ? atoi(e) : 0; }();   // tuning override
  if (tt_env >= SUB && tt_env % SUB == 0) TT = tt_env;
  const int tiles = cdiv(T, TT);
  const size_t lds = sizeof(float) * ((size_t)(NP + 1) * roundup256(H) + (size_t)2 * SUB * NP);
  if (H > 1024 || Ds > (NP <= 20 ? 2048 : 1024))
    return set_error(TSG_E_SHAPE, "scdm_attn_fwd: H=%d (max 1024) / Ds=%d (max %d at N=%d) not supported",
                     H, Ds, NP <= 20 ? 2048 : 1024, N);
  if (lds > (size_t)kLdsBytes)
    return set_error(TSG_E_LDS, "scdm_attn_fwd: N=%d H=%d needs %zu B of LDS (> %d)", N, H, lds, kLdsBytes);
  auto kern = scdm_fwd_kernel<NP, R, GATE, ST>;
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "scdm_attn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
  launch_timed(kern, dim3(B * tiles), dim3(kFwdThreads), lds, st, a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, ablate_mask());
  return check_launch("scdm_attn_fwd");
}

template <int NP, bool GATE, typename ST>
int launch_fwd_ws(const ST* a, const ST* s, const float* w, const ST* V, ST* C, float* P, const ST* gr, const float* gbias,
                  int B, int T, int N, int H, int Ds, int TT, int tiles, size_t lds, hipStream_t st);

// dtype TSG_F32S, fp32 storage: the forward with phase 2 on the bf16 matrix pipe where its tiling applies (Ds = 256, 512, 1024);
// TSG_K1_FWD=valu in the environment keeps the VALU kernel (A/B timing).  Returns -1000 when the shape is not covered.
template <int NP, bool GATE>
int launch_fwd_mm(const float* a, const float* s, const float* w, const float* V, float* C, float* P,
                  const float* gr, const float* gbias, int B, int T, int N, int H, int Ds, hipStream_t st) {
  static const bool valu_only = [] { const char* e = getenv("TSG_K1_FWD"); return e && e[0] == 'v'; }();
  if (valu_only || H != Ds || (Ds != 256 && Ds != 512 && Ds != 1024)) return -1000;
  int TT = 64;                               // rows per workgroup: as many as still give every CU a workgroup (as launch_fwd)
  while (TT > 8 && (long)B * cdiv(T, TT) < 256) TT >>= 1;
  static const int tt_env = [] { const char* e = getenv("TSG_K1_TT"); return e ? atoi(e) : 0; }();
  if (tt_env >= 8 && tt_env % 8 == 0 && tt_env <= kK1PRows) TT = tt_env;
  const int tiles = cdiv(T, TT);
  const size_t lds = sizeof(float) * ((size_t)(NP + 1) * roundup256(H) + (size_t)kK1PRows * 36 + 16 + 8 * kK1TrWave);
  if (lds > (size_t)kLdsBytes) return -1000;
  static const bool mm_only = [] { const char* e = getenv("TSG_K1_FWD"); return e && e[0] == 'm'; }();   // A/B: time-shared roles
  if (mm_only) {
    auto kern = Ds == 1024 ? scdm_fwd_mm_kernel<NP, GATE, 4> : Ds == 512 ? scdm_fwd_mm_kernel<NP, GATE, 2> : scdm_fwd_mm_kernel<NP, GATE, 1>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "scdm_attn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * tiles), dim3(kFwdThreads), lds, st, a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, ablate_mask());
    return check_launch("scdm_attn_fwd");
  }
  return launch_fwd_ws<NP, GATE, float>(a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, lds, st);
}

// role-specialised waves: 8 producers + 8 consumers (two score waves per SIMD) while the consumer's VW strip fits 128 VGPRs
// (8 < N <= 24), else 4 + 4.  TSG_K1_PW=4 forces the latter (A/B timing).  ST = float (TSG_F32S) or bf16_t (TSG_BF16).
template <int NP, bool GATE, typename ST>
int launch_fwd_ws(const ST* a, const ST* s, const float* w, const ST* V, ST* C, float* P, const ST* gr, const float* gbias,
                  int B, int T, int N, int H, int Ds, int TT, int tiles, size_t lds, hipStream_t st) {
  static const int pw_env = [] { const char* e = getenv("TSG_K1_PW"); return e ? atoi(e) : 0; }();
  // (NP = 8 at 1024 threads spills: 4 + 4 there; bf16 storage has no VW lo plane and 8-byte row pieces: 28 slots -- N = 25 -- still fit)
  constexpr bool kCanPw8 = NP > 8 && (NP <= 24 || (storage_is_bf16<ST>::value && NP <= 28));
  const bool pw8 = kCanPw8 && pw_env != 4;
  void (*kern)(const ST*, const ST*, const float*, const ST*, ST*, float*, const ST*, const float*, int, int, int, int, int, int, int, int);
  if constexpr (NP == 28 && !storage_is_bf16<ST>::value) {
    // N = 25, 26 in fp32 storage: 8 + 8 waves with the words beyond 24 on the VALU (round-4 review item 3a); N = 27, 28: 4 + 4
    if ((N == 25 || N == 26) && pw_env != 4) {
      if (N == 25) kern = Ds == 1024 ? scdm_fwd_ws_kernel<NP, GATE, 4, 8, ST, 1> : Ds == 512 ? scdm_fwd_ws_kernel<NP, GATE, 2, 8, ST, 1> : scdm_fwd_ws_kernel<NP, GATE, 1, 8, ST, 1>;
      else kern = Ds == 1024 ? scdm_fwd_ws_kernel<NP, GATE, 4, 8, ST, 2> : Ds == 512 ? scdm_fwd_ws_kernel<NP, GATE, 2, 8, ST, 2> : scdm_fwd_ws_kernel<NP, GATE, 1, 8, ST, 2>;
      hipError_t e = allow_lds(kern, lds);
      if (e != hipSuccess) return set_error((int)e, "scdm_attn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
      launch_timed(kern, dim3(B * tiles), dim3(1024), lds, st, a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, ablate_mask());
      return check_launch("scdm_attn_fwd");
    }
  }
  if constexpr (kCanPw8) {
    if (pw8) kern = Ds == 1024 ? scdm_fwd_ws_kernel<NP, GATE, 4, 8, ST> : Ds == 512 ? scdm_fwd_ws_kernel<NP, GATE, 2, 8, ST> : scdm_fwd_ws_kernel<NP, GATE, 1, 8, ST>;
    else kern = Ds == 1024 ? scdm_fwd_ws_kernel<NP, GATE, 4, 4, ST> : Ds == 512 ? scdm_fwd_ws_kernel<NP, GATE, 2, 4, ST> : scdm_fwd_ws_kernel<NP, GATE, 1, 4, ST>;
  } else {
    kern = Ds == 1024 ? scdm_fwd_ws_kernel<NP, GATE, 4, 4, ST> : Ds == 512 ? scdm_fwd_ws_kernel<NP, GATE, 2, 4, ST> : scdm_fwd_ws_kernel<NP, GATE, 1, 4, ST>;
  }
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "scdm_attn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
  launch_timed(kern, dim3(B * tiles), dim3(pw8 ? 1024 : 512), lds, st, a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, ablate_mask());
  return check_launch("scdm_attn_fwd");
}

// dtype TSG_BF16: the same kernel on bf16 storage where its tiling applies (H = Ds = 256, 512, 1024); -1000 otherwise (the VALU kernel runs).
// TSG_K1_FWD=valu keeps the VALU kernel (A/B timing).
template <int NP, bool GATE>
int launch_fwd_ws_bf16(const bf16_t* a, const bf16_t* s, const float* w, const bf16_t* V, bf16_t* C, float* P,
                       const bf16_t* gr, const float* gbias, int B, int T, int N, int H, int Ds, hipStream_t st) {
  static const bool valu_only = [] { const char* e = getenv("TSG_K1_FWD"); return e && e[0] == 'v'; }();
  if (valu_only || H != Ds || (Ds != 256 && Ds != 512 && Ds != 1024)) return -1000;
  int TT = 64;
  while (TT > 8 && (long)B * cdiv(T, TT) < 256) TT >>= 1;
  static const int tt_env = [] { const char* e = getenv("TSG_K1_TT"); return e ? atoi(e) : 0; }();
  if (tt_env >= 8 && tt_env % 8 == 0 && tt_env <= kK1PRows) TT = tt_env;
  const int tiles = cdiv(T, TT);
  const size_t lds = sizeof(float) * ((size_t)(NP + 1) * roundup256(H) + (size_t)kK1PRows * 36 + 16 + 8 * kK1TrWave);
  if (lds > (size_t)kLdsBytes) return -1000;
  return launch_fwd_ws<NP, GATE, bf16_t>(a, s, w, V, C, P, gr, gbias, B, T, N, H, Ds, TT, tiles, lds, st);
}

// Two-kernel path (kept for shapes whose P / de tiles do not fit the fused kernel's LDS, and for A/B timing with
// TSG_K1_BWD=split).  GATE: V = VW, dC = dout; extra outputs dbias [Ds], dr [B,T,Ds]; dG_ws [B,T,Ds] workspace.
template <int NP, bool GATE>
int launch_bwd_split(const float* a, const float* s, const float* w, const float* V, const float* P,
               const float* dC, float* da, float* ds, float* dw, float* dV, float* de,
               const float* gr, const float* gbias, float* dbias, float* dr, float* dG_ws,
               int B, int T, int N, int H, int Ds, hipStream_t st) {
  if (Ds > (NP <= 20 ? 2048 : 1024))
    return set_error(TSG_E_SHAPE, "scdm_attn_bwd: Ds=%d (max %d at N=%d) not supported", Ds, NP <= 20 ? 2048 : 1024, N);
  hipLaunchKernelGGL(zero3_kernel, dim3(4), dim3(256), 0, st, (unsigned*)dw, H, (unsigned*)dbias, GATE ? Ds : 0, (unsigned*)nullptr, 0);
  const int tiles = cdiv(T, 32);
  hipLaunchKernelGGL((scdm_bwd_rows_kernel<NP, GATE>), dim3(B * tiles), dim3(kFwdThreads), 0, st, V, P, dC, de, gr, gbias,
                     dG_ws, dr, B, T, N, Ds, tiles);
  int rc = check_launch("scdm_attn_bwd(rows)");
  if (rc) return rc;
  constexpr int kSlice = kWave * cpl_of<NP>();
  const int hslices = cdiv(H, kSlice), slices = hslices > cdiv(Ds, kSlice) ? hslices : cdiv(Ds, kSlice);
  const size_t rows_lds = sizeof(float) * (size_t)(T < kColTB ? T : kColTB) * NP;
  hipLaunchKernelGGL(scdm_bwd_cols_kernel<NP>, dim3(B * slices), dim3(kColThreads), rows_lds, st, a, s, w, P, GATE ? dG_ws : dC, de,
                     da, ds, dw, dV, GATE ? dbias : nullptr, B, T, N, H, Ds, hslices, slices);
  return check_launch("scdm_attn_bwd(cols)");
}

// Backward path selection: 0 = automatic (one fused launch where the plan fits, else the two kernels), 1 = always the two-kernel
// path ("split": no cross-workgroup exchange at all), 2 = fused with the row phase on the VALU.  Initialised from TSG_K1_BWD
// (s.. / v..), changed at run time by tsg_scdm_bwd_mode (the exchange stress test compares the fused result with the split one in
// one process).
static std::atomic<int> g_bwd_mode{-1};
static int bwd_mode() {
  int v = g_bwd_mode.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("TSG_K1_BWD");
    v = (e && e[0] == 's') ? 1 : ((e && e[0] == 'v') ? 2 : 0);
    g_bwd_mode.store(v, std::memory_order_relaxed);
  }
  return v;
}
static bool want_split() { return bwd_mode() == 1; }

// The plan launch_bwd runs for a shape: which fused variant, and whether the one-launch kernel takes it at all.
struct BwdChoice { FusedPlan plan; bool mrow; bool fused; };
template <int NP>
BwdChoice choose_bwd(int B, int T, int N, int H, int Ds) {
  const bool valu_rows = bwd_mode() == 2;                           // A/B: row phase on the VALU
  const FusedPlan pm = fused_plan<NP, true>(B, T, N, H, Ds), pv = fused_plan<NP, false>(B, T, N, H, Ds);
  // the MFMA row phase pays when its 64-column slices do not force more column parts per item than the 128-column
  // slices of the VALU variant would need (measured at [., 128, 20, 1024]: 128 pairs 183 vs 188 us gate-fused, 132 vs 158 us
  // plain; 256 pairs -- 2 parts instead of 1 -- 355 vs 326 us)
  const bool mrow = pm.ok && !valu_rows && (!pv.ok || pm.parts <= pv.parts);
  const FusedPlan& pl = mrow ? pm : pv;
  // The parts of an item wait for each other (bounded spin).  They are `parts` block ids inside a window of 8*(parts-1)+1
  // consecutive ids (xcd_remap keeps an item on one XCD); with in-order dispatch the oldest resident workgroup's partners
  // are all dispatched as long as that window fits the workgroups the chip holds at once (one per CU).  Half the CUs is the
  // margin kept for CUs a collective or another stream occupies; beyond it the two-kernel path (no waiting) runs.
  const bool window_ok = pl.parts == 1 || 8 * (pl.parts - 1) + 1 <= device_cu_count() / 2;
  const bool ds_ok = Ds <= (NP <= 20 ? 2048 : 1024);
  return BwdChoice{pl, mrow, ds_ok && pl.ok && window_ok && !want_split()};
}

inline long long split_ws_bytes(int B, int T, int N, int Ds, bool gate) {
  return (long long)sizeof(float) * ((long long)B * T * roundup(N, 4) + (gate ? (long long)B * T * Ds : 0));
}

template <int NP, bool GATE, typename ST>
int launch_bwd(const ST* a, const ST* s, const float* w, const ST* V, const float* P,
               const ST* dC, ST* da, ST* ds, float* dw, ST* dV, const ST* gr, const float* gbias,
               float* dbias, ST* dr, void* ws, long long ws_bytes, int B, int T, int N, int H, int Ds, hipStream_t st) {
  const char* fn = GATE ? "tsg_scdm_gate_bwd" : "tsg_scdm_attn_bwd";
  if (Ds > (NP <= 20 ? 2048 : 1024))
    return set_error(TSG_E_SHAPE, "%s: Ds=%d (max %d at N=%d) not supported", fn, Ds, NP <= 20 ? 2048 : 1024, N);
  const BwdChoice ch = choose_bwd<NP>(B, T, N, H, Ds);
  const FusedPlan& pl = ch.plan;
  const bool mrow = ch.mrow;
  if (ch.fused) {
    if (ws_bytes < pl.ws_bytes) return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_scdm_bwd_ws_bytes)", fn, ws_bytes, pl.ws_bytes);
    float* xch = static_cast<float*>(ws);
    unsigned* cnt = reinterpret_cast<unsigned*>(xch + (size_t)B * pl.parts * T * NP);
    hipLaunchKernelGGL(zero3_kernel, dim3(4), dim3(256), 0, st, (unsigned*)dw, H, (unsigned*)dbias, GATE ? Ds : 0, cnt, B);
    auto kern = mrow ? scdm_bwd_fused_kernel<NP, GATE, true, ST> : scdm_bwd_fused_kernel<NP, GATE, false, ST>;
    hipError_t e = allow_lds(kern, pl.lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute(%zu): %s", fn, pl.lds, hipGetErrorString(e));
    launch_timed(kern, dim3(pl.grid), dim3(kFusedThreads), pl.lds, st, a, s, w, V, P, dC, gr, gbias, da, ds, dw, dV, dbias, dr,
                       xch, cnt, error_sink(), B, T, N, H, Ds, pl.parts, pl.SP, ablate_mask());
    return check_launch(fn);
  }
  if constexpr (storage_is_bf16<ST>::value) {
    // the two-kernel path exists for fp32 storage only; the host code runs such a shape through it with fp32 copies
    return set_error(TSG_E_SHAPE, "%s: this shape (T=%d N=%d H=%d Ds=%d: P / de tiles beyond the fused kernel's LDS, or an exchange "
                     "window beyond the co-residency margin) is not supported with dtype TSG_BF16", fn, T, N, H, Ds);
  } else {
    if (ws_bytes < split_ws_bytes(B, T, N, Ds, GATE))
      return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_scdm_bwd_ws_bytes)", fn, ws_bytes, split_ws_bytes(B, T, N, Ds, GATE));
    float* de = static_cast<float*>(ws);
    float* dG = de + (size_t)B * T * roundup(N, 4);
    return launch_bwd_split<NP, GATE>(a, s, w, V, P, dC, da, ds, dw, dV, de, gr, gbias, dbias, dr, GATE ? dG : nullptr, B, T, N, H, Ds, st);
  }
}

int check_common(const char* fn, std::initializer_list<const void*> ptrs, int B, int T, int N, int H, int Ds, int dtype) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (dtype != TSG_F32 && dtype != TSG_BF16 && dtype != TSG_F32S)
    return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, TSG_F32S, or TSG_BF16 = bf16 storage of the activations)", fn, dtype);
  if (B <= 0 || T <= 0 || N <= 0 || H <= 0 || Ds <= 0)
    return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d N=%d H=%d Ds=%d", fn, B, T, N, H, Ds);
  if (N > 32) return set_error(TSG_E_SHAPE, "%s: N=%d > 32 words not supported", fn, N);
  // the kernels keep B * tiles in an int grid and B * T rows in 32-bit row arithmetic (found by the host sanitizer build: the workspace plan of
  // B = T = 2^30 overflowed a long long)
  if ((long long)B * T >= (1LL << 31)) return set_error(TSG_E_SHAPE, "%s: B * T = %lld rows (limit 2^31 - 1)", fn, (long long)B * T);
  if (H % 4 || Ds % 4) return set_error(TSG_E_ALIGN, "%s: H=%d and Ds=%d must be multiples of 4", fn, H, Ds);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

#define TSG_DISPATCH_NP(NPV, CALL)                                  \
  switch (NPV) {                                                    \
    case 4: { constexpr int NP = 4; return CALL; }                  \
    case 8: { constexpr int NP = 8; return CALL; }                  \
    case 12: { constexpr int NP = 12; return CALL; }                \
    case 16: { constexpr int NP = 16; return CALL; }                \
    case 20: { constexpr int NP = 20; return CALL; }                \
    case 24: { constexpr int NP = 24; return CALL; }                \
    case 28: { constexpr int NP = 28; return CALL; }                \
    default: { constexpr int NP = 32; return CALL; }                \
  }

extern "C" int tsg_scdm_attn_fwd(const void* a, const void* s, const void* w, const void* sent, void* C,
                                 void* P, int B, int T, int N, int H, int Ds, int dtype, void* stream) {
  int rc = check_common("tsg_scdm_attn_fwd", {a, s, w, sent, C, P}, B, T, N, H, Ds, dtype);
  if (rc) return rc;
  const int np = roundup(N, 4);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    auto ws = [&]() -> int {
      TSG_DISPATCH_NP(np, (launch_fwd_ws_bf16<NP, false>((const S*)a, (const S*)s, (const float*)w, (const S*)sent,
                                                         (S*)C, (float*)P, nullptr, nullptr, B, T, N, H, Ds, st)));
    };
    rc = ws();
    if (rc != -1000) return rc;
    TSG_DISPATCH_NP(np, (launch_fwd<NP, false, S>((const S*)a, (const S*)s, (const float*)w, (const S*)sent,
                                                  (S*)C, (float*)P, nullptr, nullptr, B, T, N, H, Ds, st)));
  }
  if (dtype == TSG_F32S) {
    auto mm = [&]() -> int {
      TSG_DISPATCH_NP(np, (launch_fwd_mm<NP, false>((const float*)a, (const float*)s, (const float*)w, (const float*)sent,
                                                    (float*)C, (float*)P, nullptr, nullptr, B, T, N, H, Ds, st)));
    };
    rc = mm();
    if (rc != -1000) return rc;
  }
  TSG_DISPATCH_NP(np, (launch_fwd<NP, false, float>((const float*)a, (const float*)s, (const float*)w, (const float*)sent,
                                                    (float*)C, (float*)P, nullptr, nullptr, B, T, N, H, Ds, st)));
}

extern "C" long long tsg_scdm_bwd_ws_bytes(int B, int T, int N, int H, int Ds, int gate) {
  if (B <= 0 || T <= 0 || N <= 0 || N > 32 || H <= 0 || Ds <= 0 || (long long)B * T >= (1LL << 31)) return 0;     // (shapes the entry points reject)
  const int np = roundup(N, 4);
  const long long split = split_ws_bytes(B, T, N, Ds, gate != 0);
  long long fused = 0;
  switch (np) {
    case 4: fused = std::max(fused_plan<4, true>(B, T, N, H, Ds).ws_bytes, fused_plan<4, false>(B, T, N, H, Ds).ws_bytes); break;
    case 8: fused = std::max(fused_plan<8, true>(B, T, N, H, Ds).ws_bytes, fused_plan<8, false>(B, T, N, H, Ds).ws_bytes); break;
    case 12: fused = std::max(fused_plan<12, true>(B, T, N, H, Ds).ws_bytes, fused_plan<12, false>(B, T, N, H, Ds).ws_bytes); break;
    case 16: fused = std::max(fused_plan<16, true>(B, T, N, H, Ds).ws_bytes, fused_plan<16, false>(B, T, N, H, Ds).ws_bytes); break;
    case 20: fused = std::max(fused_plan<20, true>(B, T, N, H, Ds).ws_bytes, fused_plan<20, false>(B, T, N, H, Ds).ws_bytes); break;
    case 24: fused = std::max(fused_plan<24, true>(B, T, N, H, Ds).ws_bytes, fused_plan<24, false>(B, T, N, H, Ds).ws_bytes); break;
    case 28: fused = std::max(fused_plan<28, true>(B, T, N, H, Ds).ws_bytes, fused_plan<28, false>(B, T, N, H, Ds).ws_bytes); break;
    default: fused = std::max(fused_plan<32, true>(B, T, N, H, Ds).ws_bytes, fused_plan<32, false>(B, T, N, H, Ds).ws_bytes); break;
  }
  return fused > split ? fused : split;       // either path can run in it
}

extern "C" int tsg_scdm_bwd_mode(int mode) {
  const int prev = bwd_mode();
  if (mode >= 0 && mode <= 2) g_bwd_mode.store(mode, std::memory_order_relaxed);
  return prev;
}

extern "C" int tsg_scdm_bwd_fused_ok(int B, int T, int N, int H, int Ds) {
  if (B <= 0 || T <= 0 || N <= 0 || N > 32 || H <= 0 || Ds <= 0 || H % 4 || Ds % 4) return 0;
  const int np = roundup(N, 4);
  TSG_DISPATCH_NP(np, (choose_bwd<NP>(B, T, N, H, Ds).fused ? 1 : 0));
}

extern "C" int tsg_scdm_attn_bwd(const void* a, const void* s, const void* w, const void* sent,
                                 const void* P, const void* dC, void* da, void* ds, void* dw, void* dsent,
                                 void* ws, long long ws_bytes, int B, int T, int N, int H, int Ds, int dtype, void* stream) {
  int rc = check_common("tsg_scdm_attn_bwd", {a, s, w, sent, P, dC, da, ds, dw, dsent, ws}, B, T, N, H, Ds, dtype);
  if (rc) return rc;
  const int np = roundup(N, 4);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    TSG_DISPATCH_NP(np, (launch_bwd<NP, false, S>((const S*)a, (const S*)s, (const float*)w, (const S*)sent,
                                                  (const float*)P, (const S*)dC, (S*)da, (S*)ds, (float*)dw,
                                                  (S*)dsent, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                                                  B, T, N, H, Ds, st)));
  }
  TSG_DISPATCH_NP(np, (launch_bwd<NP, false, float>((const float*)a, (const float*)s, (const float*)w, (const float*)sent,
                                                    (const float*)P, (const float*)dC, (float*)da, (float*)ds, (float*)dw,
                                                    (float*)dsent, nullptr, nullptr, nullptr, nullptr, ws, ws_bytes,
                                                    B, T, N, H, Ds, st)));
}

extern "C" int tsg_scdm_gate_fwd(const void* a, const void* s, const void* w, const void* VW, const void* gbias,
                                 const void* r, void* out, void* P, int B, int T, int N, int H, int Ds, int dtype,
                                 void* stream) {
  int rc = check_common("tsg_scdm_gate_fwd", {a, s, w, VW, gbias, r, out, P}, B, T, N, H, Ds, dtype);
  if (rc) return rc;
  const int np = roundup(N, 4);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    auto ws = [&]() -> int {
      TSG_DISPATCH_NP(np, (launch_fwd_ws_bf16<NP, true>((const S*)a, (const S*)s, (const float*)w, (const S*)VW,
                                                        (S*)out, (float*)P, (const S*)r, (const float*)gbias, B, T, N, H, Ds, st)));
    };
    rc = ws();
    if (rc != -1000) return rc;
    TSG_DISPATCH_NP(np, (launch_fwd<NP, true, S>((const S*)a, (const S*)s, (const float*)w, (const S*)VW,
                                                 (S*)out, (float*)P, (const S*)r, (const float*)gbias, B, T, N, H, Ds, st)));
  }
  if (dtype == TSG_F32S) {
    auto mm = [&]() -> int {
      TSG_DISPATCH_NP(np, (launch_fwd_mm<NP, true>((const float*)a, (const float*)s, (const float*)w, (const float*)VW,
                                                   (float*)out, (float*)P, (const float*)r, (const float*)gbias, B, T, N, H, Ds, st)));
    };
    rc = mm();
    if (rc != -1000) return rc;
  }
  TSG_DISPATCH_NP(np, (launch_fwd<NP, true, float>((const float*)a, (const float*)s, (const float*)w, (const float*)VW,
                                                   (float*)out, (float*)P, (const float*)r, (const float*)gbias, B, T, N, H, Ds, st)));
}

extern "C" int tsg_scdm_gate_bwd(const void* a, const void* s, const void* w, const void* VW, const void* gbias,
                                 const void* r, const void* P, const void* dout, void* da, void* ds, void* dw,
                                 void* dVW, void* dgbias, void* dr, void* ws, long long ws_bytes,
                                 int B, int T, int N, int H, int Ds, int dtype, void* stream) {
  int rc = check_common("tsg_scdm_gate_bwd", {a, s, w, VW, gbias, r, P, dout, da, ds, dw, dVW, dgbias, dr, ws},
                        B, T, N, H, Ds, dtype);
  if (rc) return rc;
  const int np = roundup(N, 4);
  auto st = static_cast<hipStream_t>(stream);
  if (dtype == TSG_BF16) {
    using S = bf16_t;
    TSG_DISPATCH_NP(np, (launch_bwd<NP, true, S>((const S*)a, (const S*)s, (const float*)w, (const S*)VW,
                                                 (const float*)P, (const S*)dout, (S*)da, (S*)ds, (float*)dw,
                                                 (S*)dVW, (const S*)r, (const float*)gbias,
                                                 (float*)dgbias, (S*)dr, ws, ws_bytes, B, T, N, H, Ds, st)));
  }
  TSG_DISPATCH_NP(np, (launch_bwd<NP, true, float>((const float*)a, (const float*)s, (const float*)w, (const float*)VW,
                                                   (const float*)P, (const float*)dout, (float*)da, (float*)ds, (float*)dw,
                                                   (float*)dVW, (const float*)r, (const float*)gbias,
                                                   (float*)dgbias, (float*)dr, ws, ws_bytes, B, T, N, H, Ds, st)));
}

#ifdef TSG_K1_TICKS
extern "C" int tsg_debug_k1_ticks(unsigned long long* host_dst) {
  return (int)hipMemcpyFromSymbol(host_dst, HIP_SYMBOL(tsg::g_k1_ticks), sizeof(unsigned long long) * 256 * 16 * 8);
}
#endif
