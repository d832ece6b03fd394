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

namespace tsg {
namespace {

constexpr int kThreads = 256;                 // backward kernels
constexpr int kWaves = kThreads / kWave;
constexpr int kFwdThreads = 512;              // forward: 8 waves share one Es tile (1 workgroup / CU)
constexpr int kFwdWaves = kFwdThreads / kWave;
constexpr float kClamp = 40.0f;

__host__ __device__ __forceinline__ int roundup256(int x) { return (x + 255) & ~255; }

__device__ __forceinline__ int wave_id() {    // wave-uniform (SGPR) wave index inside the workgroup
  return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

// ------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------
template <int NP, int R>
__global__ __launch_bounds__(kFwdThreads) void scdm_fwd_kernel(
    const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ w,
    const float* __restrict__ V, float* __restrict__ C, float* __restrict__ P,
    int B, int T, int N, int H, int Ds, int TT, int tiles) {
  const int HP = roundup256(H);
  extern __shared__ __align__(16) float lds[];
  float* Es = lds;                 // [NP][HP]
  float* Pl = lds + NP * HP;       // [TT][NP]

  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, tiles);
  const int b = bid / tiles, tile = bid % tiles;

  // ---- prologue: Es = exp(2 s[b]) into LDS (zero padding: r = 1, harmless, masked below) ----
  const float* sb = s + (size_t)b * N * H;
  const int hp4 = HP / 4;
  for (int idx = tid; idx < NP * hp4; idx += kFwdThreads) {
    const int n = idx / hp4, k = (idx % hp4) * 4;
    float4 e = make_float4(0.f, 0.f, 0.f, 0.f);
    if (n < N && k < H) {
      const float4 v = *reinterpret_cast<const float4*>(sb + (size_t)n * H + k);
      e.x = fast_exp2(clampf(v.x, -kClamp, kClamp) * k2Log2e);
      e.y = fast_exp2(clampf(v.y, -kClamp, kClamp) * k2Log2e);
      e.z = fast_exp2(clampf(v.z, -kClamp, kClamp) * k2Log2e);
      e.w = fast_exp2(clampf(v.w, -kClamp, kClamp) * k2Log2e);
    }
    *reinterpret_cast<float4*>(Es + n * HP + k) = e;
  }
  __syncthreads();

  const int rows_per_wave = TT / kFwdWaves;
  const int t_wave = tile * TT + wv * rows_per_wave;

  // ---- phase 1: scores + softmax, R rows at a time; k swept in 256-column chunks --------------
  for (int r0 = 0; r0 < rows_per_wave; r0 += R) {
    const float* arow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int t = t_wave + r0 + r;
      arow[r] = a + ((size_t)b * T + (t < T ? t : T - 1)) * H + lane * 4;
    }
    float acc[R][NP];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
      for (int n = 0; n < NP; ++n) acc[r][n] = 0.f;

    float4 av[R];
#pragma unroll
    for (int r = 0; r < R; ++r)
      av[r] = (lane * 4 < H) ? *reinterpret_cast<const float4*>(arow[r]) : make_float4(0.f, 0.f, 0.f, 0.f);

#pragma unroll 1
    for (int k0 = 0; k0 < HP; k0 += 256) {
      const int k = k0 + lane * 4;
      float4 nxt[R];                                    // prefetch the next chunk of the R rows
#pragma unroll
      for (int r = 0; r < R; ++r)
        nxt[r] = (k + 256 < H) ? *reinterpret_cast<const float4*>(arow[r] + k0 + 256) : make_float4(0.f, 0.f, 0.f, 0.f);
      float4 wq = make_float4(0.f, 0.f, 0.f, 0.f);     // 0 beyond H: padded columns add nothing
      if (k < H) wq = *reinterpret_cast<const float4*>(w + k);
      const float w2[4] = {-2.f * wq.x, -2.f * wq.y, -2.f * wq.z, -2.f * wq.w};
      float Ea[R][4];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Ea[r][0] = fast_exp2(clampf(av[r].x, -kClamp, kClamp) * k2Log2e);
        Ea[r][1] = fast_exp2(clampf(av[r].y, -kClamp, kClamp) * k2Log2e);
        Ea[r][2] = fast_exp2(clampf(av[r].z, -kClamp, kClamp) * k2Log2e);
        Ea[r][3] = fast_exp2(clampf(av[r].w, -kClamp, kClamp) * k2Log2e);
      }
      const float* esp = Es + k;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        const float4 es = *reinterpret_cast<const float4*>(esp + n * HP);
#pragma unroll
        for (int r = 0; r < R; ++r) {
          acc[r][n] = fmaf(w2[0], fast_rcp(fmaf(Ea[r][0], es.x, 1.f)), acc[r][n]);
          acc[r][n] = fmaf(w2[1], fast_rcp(fmaf(Ea[r][1], es.y, 1.f)), acc[r][n]);
          acc[r][n] = fmaf(w2[2], fast_rcp(fmaf(Ea[r][2], es.z, 1.f)), acc[r][n]);
          acc[r][n] = fmaf(w2[3], fast_rcp(fmaf(Ea[r][3], es.w, 1.f)), acc[r][n]);
        }
      }
#pragma unroll
      for (int r = 0; r < R; ++r) av[r] = nxt[r];
    }

#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int tl = wv * rows_per_wave + r0 + r;       // row inside the tile
      const int t = tile * TT + tl;
      float m = -INFINITY;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        acc[r][n] = wave_allsum(acc[r][n]);
        if (n < N) m = fmaxf(m, acc[r][n]);
      }
      float sum = 0.f;
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        acc[r][n] = (n < N) ? fast_exp2((acc[r][n] - m) * kLog2e) : 0.f;
        sum += acc[r][n];
      }
      const float inv = 1.f / sum;
      float mine = 0.f;
#pragma unroll
      for (int n = 0; n < NP; ++n) mine = (lane == n) ? acc[r][n] * inv : mine;
      if (r0 + r < rows_per_wave) {
        if (lane < NP) Pl[tl * NP + lane] = mine;
        if (lane < N && t < T) P[((size_t)b * T + t) * N + lane] = mine;
      }
    }
  }
  // Pl rows are written and read by the same wave only: no workgroup barrier needed, but the LDS
  // writes must have landed before the broadcast reads below.
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)

  // ---- phase 2: C[t,:] = sum_n P[t,n] * sent[b,n,:], rows blocked in registers ---------------
  const float* Vb = V + (size_t)b * N * Ds;
  constexpr int RC = 4;                                  // rows per register block
  for (int j0 = 0; j0 < Ds; j0 += 256) {
    const int j = j0 + lane * 4;
    for (int r0 = 0; r0 < rows_per_wave; r0 += RC) {
      float4 c[RC];
#pragma unroll
      for (int r = 0; r < RC; ++r) c[r] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (j < Ds) {
#pragma unroll
        for (int n4 = 0; n4 < NP; n4 += 4) {
          float4 v[4];
#pragma unroll
          for (int q = 0; q < 4; ++q)
            v[q] = (n4 + q < N) ? *reinterpret_cast<const float4*>(Vb + (size_t)(n4 + q) * Ds + j)
                                : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int r = 0; r < RC; ++r) {
            if (r0 + r < rows_per_wave) {
              const float4 p = *reinterpret_cast<const float4*>(Pl + (wv * rows_per_wave + r0 + r) * NP + n4);
              c[r].x = fmaf(p.x, v[0].x, c[r].x); c[r].y = fmaf(p.x, v[0].y, c[r].y);
              c[r].z = fmaf(p.x, v[0].z, c[r].z); c[r].w = fmaf(p.x, v[0].w, c[r].w);
              c[r].x = fmaf(p.y, v[1].x, c[r].x); c[r].y = fmaf(p.y, v[1].y, c[r].y);
              c[r].z = fmaf(p.y, v[1].z, c[r].z); c[r].w = fmaf(p.y, v[1].w, c[r].w);
              c[r].x = fmaf(p.z, v[2].x, c[r].x); c[r].y = fmaf(p.z, v[2].y, c[r].y);
              c[r].z = fmaf(p.z, v[2].z, c[r].z); c[r].w = fmaf(p.z, v[2].w, c[r].w);
              c[r].x = fmaf(p.w, v[3].x, c[r].x); c[r].y = fmaf(p.w, v[3].y, c[r].y);
              c[r].z = fmaf(p.w, v[3].z, c[r].z); c[r].w = fmaf(p.w, v[3].w, c[r].w);
            }
          }
        }
#pragma unroll
        for (int r = 0; r < RC; ++r) {
          const int t = t_wave + r0 + r;
          if (r0 + r < rows_per_wave && t < T)
            *reinterpret_cast<float4*>(C + ((size_t)b * T + t) * Ds + j) = c[r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, kernel 1: de[b,t,n] = P*(dP - <P,dP>),  dP[n] = <dC[t,:], sent[b,n,:]>
// one wave per clip row; sent rows come from L2 (80 KiB per batch item).
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kThreads) void scdm_bwd_rows_kernel(
    const float* __restrict__ V, const float* __restrict__ P, const float* __restrict__ dC,
    float* __restrict__ de, int B, int T, int N, int Ds) {
  const int lane = threadIdx.x & 63, wv = wave_id();
  const long row = (long)blockIdx.x * kWaves + wv;            // = b*T + t
  if (row >= (long)B * T) return;
  const int b = (int)(row / T);
  const float* Vb = V + (size_t)b * N * Ds;
  const float* g = dC + (size_t)row * Ds;
  float dp[NP];
#pragma unroll
  for (int n = 0; n < NP; ++n) dp[n] = 0.f;
  for (int j = lane * 4; j < Ds; j += 256) {
    const float4 gv = *reinterpret_cast<const float4*>(g + j);
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      if (n < N) {
        const float4 v = *reinterpret_cast<const float4*>(Vb + (size_t)n * Ds + j);
        dp[n] = fmaf(gv.x, v.x, fmaf(gv.y, v.y, fmaf(gv.z, v.z, fmaf(gv.w, v.w, dp[n]))));
      }
    }
  }
  const float p = (lane < N) ? P[(size_t)row * N + lane] : 0.f;
  float mine = 0.f;
#pragma unroll
  for (int n = 0; n < NP; ++n) {
    dp[n] = wave_allsum(dp[n]);
    mine = (lane == n) ? dp[n] : mine;
  }
  const float dot = wave_allsum(p * mine);                     // <P, dP>
  if (lane < N) de[(size_t)row * N + lane] = p * (mine - dot);
}

// ------------------------------------------------------------------------------------------
// backward, kernel 2: one workgroup = (b, 256-column slice c); waves stride over the T rows.
// ------------------------------------------------------------------------------------------
template <int NP>
__global__ __launch_bounds__(kThreads) void scdm_bwd_cols_kernel(
    const float* __restrict__ a, const float* __restrict__ s, const float* __restrict__ w,
    const float* __restrict__ P, const float* __restrict__ dC, const float* __restrict__ de,
    float* __restrict__ da, float* __restrict__ ds, float* __restrict__ dw, float* __restrict__ dV,
    int B, int T, int N, int H, int Ds, int hslices, int slices) {
  extern __shared__ __align__(16) float lds[];                 // [kWaves][NP][256] cross-wave reduce
  const int tid = threadIdx.x, lane = tid & 63, wv = wave_id();
  const int bid = xcd_remap(blockIdx.x, gridDim.x, slices);
  const int b = bid / slices, c = bid % slices;
  const int k = c * 256 + lane * 4;
  const float* deb = de + (size_t)b * T * N;
  const float* Pb = P + (size_t)b * T * N;

  // ---------------- main: da, ds, dw for hidden columns k..k+3 --------------------------------
  if (c < hslices) {
    const bool live = k < H;
    float es[NP][4];
    const float* sb = s + (size_t)b * N * H;
#pragma unroll
    for (int n = 0; n < NP; ++n) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (n < N && live) v = *reinterpret_cast<const float4*>(sb + (size_t)n * H + k);
      es[n][0] = fast_exp2(clampf(v.x, -kClamp, kClamp) * k2Log2e);
      es[n][1] = fast_exp2(clampf(v.y, -kClamp, kClamp) * k2Log2e);
      es[n][2] = fast_exp2(clampf(v.z, -kClamp, kClamp) * k2Log2e);
      es[n][3] = fast_exp2(clampf(v.w, -kClamp, kClamp) * k2Log2e);
    }
    float4 wv4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (live) wv4 = *reinterpret_cast<const float4*>(w + k);
    const float w4[4] = {4.f * wv4.x, 4.f * wv4.y, 4.f * wv4.z, 4.f * wv4.w};

    float dsacc[NP][4];
    float dwacc[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < NP; ++n)
#pragma unroll
      for (int q = 0; q < 4; ++q) dsacc[n][q] = 0.f;

    for (int t = wv; t < T; t += kWaves) {
      float4 av = make_float4(0.f, 0.f, 0.f, 0.f);
      if (live) av = *reinterpret_cast<const float4*>(a + ((size_t)b * T + t) * H + k);
      const float ea[4] = {fast_exp2(clampf(av.x, -kClamp, kClamp) * k2Log2e),
                           fast_exp2(clampf(av.y, -kClamp, kClamp) * k2Log2e),
                           fast_exp2(clampf(av.z, -kClamp, kClamp) * k2Log2e),
                           fast_exp2(clampf(av.w, -kClamp, kClamp) * k2Log2e)};
      float dasum[4] = {0.f, 0.f, 0.f, 0.f};
      const float* der = deb + (size_t)t * N;                  // wave-uniform row -> scalar loads
#pragma unroll
      for (int n = 0; n < NP; ++n) {
        if (n < N) {
          const float d = der[n];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float r = fast_rcp(fmaf(ea[q], es[n][q], 1.f));
            const float qq = fmaf(-r, r, r);                   // r(1-r)
            const float dq = d * qq;
            dsacc[n][q] += dq;
            dasum[q] += dq;
            dwacc[q] = fmaf(d, r, dwacc[q]);
          }
        }
      }
      if (live)
        *reinterpret_cast<float4*>(da + ((size_t)b * T + t) * H + k) =
            make_float4(w4[0] * dasum[0], w4[1] * dasum[1], w4[2] * dasum[2], w4[3] * dasum[3]);
    }

    // cross-wave reduction of dsacc / dwacc through LDS
    float* red = lds;                                          // [kWaves][NP+1][256]
#pragma unroll
    for (int n = 0; n < NP; ++n)
      *reinterpret_cast<float4*>(red + ((wv * (NP + 1) + n) * 256) + lane * 4) =
          make_float4(dsacc[n][0], dsacc[n][1], dsacc[n][2], dsacc[n][3]);
    *reinterpret_cast<float4*>(red + ((wv * (NP + 1) + NP) * 256) + lane * 4) =
        make_float4(dwacc[0], dwacc[1], dwacc[2], dwacc[3]);
    __syncthreads();
    // 256 threads: thread tid owns column (c*256 + tid) for every n
    {
      const int kk = c * 256 + tid;
      const float wk = (kk < H) ? w[kk] : 0.f;
      for (int n = 0; n <= NP; ++n) {
        float acc = 0.f;
#pragma unroll
        for (int u = 0; u < kWaves; ++u) acc += red[(u * (NP + 1) + n) * 256 + tid];
        if (kk < H) {
          if (n < N) ds[((size_t)b * N + n) * H + kk] = 4.f * wk * acc;
          else if (n == NP) atomicAdd(dw + kk, -2.f * acc);
        }
      }
    }
    __syncthreads();
  }

  // ---------------- dsent[b,n,j] = sum_t P[t,n] dC[t,j] for sentence columns of slice c --------
  {
    const int j = c * 256 + lane * 4;
    if (c * 256 < Ds) {
      float dv[NP][4];
#pragma unroll
      for (int n = 0; n < NP; ++n)
#pragma unroll
        for (int q = 0; q < 4; ++q) dv[n][q] = 0.f;
      for (int t = wv; t < T; t += kWaves) {
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < Ds) g = *reinterpret_cast<const float4*>(dC + ((size_t)b * T + t) * Ds + j);
        const float* pr = Pb + (size_t)t * N;
#pragma unroll
        for (int n = 0; n < NP; ++n) {
          if (n < N) {
            const float p = pr[n];
            dv[n][0] = fmaf(p, g.x, dv[n][0]); dv[n][1] = fmaf(p, g.y, dv[n][1]);
            dv[n][2] = fmaf(p, g.z, dv[n][2]); dv[n][3] = fmaf(p, g.w, dv[n][3]);
          }
        }
      }
      float* red = lds;
#pragma unroll
      for (int n = 0; n < NP; ++n)
        *reinterpret_cast<float4*>(red + ((wv * (NP + 1) + n) * 256) + lane * 4) =
            make_float4(dv[n][0], dv[n][1], dv[n][2], dv[n][3]);
      __syncthreads();
      const int jj = c * 256 + tid;
      if (jj < Ds) {
        for (int n = 0; n < N; ++n) {
          float acc = 0.f;
#pragma unroll
          for (int u = 0; u < kWaves; ++u) acc += red[(u * (NP + 1) + n) * 256 + tid];
          dV[((size_t)b * N + n) * Ds + jj] = acc;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------

template <int NP>
int launch_fwd(const float* a, const float* s, const float* w, const float* V, float* C, float* P,
               int B, int T, int N, int H, int Ds, hipStream_t st) {
  constexpr int R = 4;
  // rows per workgroup: 32 (4 per wave) when that still gives every CU a workgroup, else 16 / 8
  int TT = 32;
  while (TT > kFwdWaves && (long)B * cdiv(T, TT) < 256) TT >>= 1;
  const int tiles = cdiv(T, TT);
  const size_t lds = sizeof(float) * ((size_t)NP * roundup256(H) + (size_t)TT * NP);
  if (lds > (size_t)kLdsBytes)
    return set_error(TSG_E_LDS, "scdm_attn_fwd: N=%d H=%d needs %zu B of LDS (> %d)", N, H, lds, kLdsBytes);
  auto kern = scdm_fwd_kernel<NP, R>;
  static thread_local size_t allowed = 0;     // per NP instantiation
  if (lds > allowed) {
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "scdm_attn_fwd: hipFuncSetAttribute(%zu): %s", lds, hipGetErrorString(e));
    allowed = lds;
  }
  hipLaunchKernelGGL(kern, dim3(B * tiles), dim3(kFwdThreads), lds, st, a, s, w, V, C, P, B, T, N, H, Ds, TT, tiles);
  return check_launch("scdm_attn_fwd");
}

template <int NP>
int launch_bwd(const float* a, const float* s, const float* w, const float* V, const float* P,
               const float* dC, float* da, float* ds, float* dw, float* dV, float* de,
               int B, int T, int N, int H, int Ds, hipStream_t st) {
  hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * H, st);
  if (e != hipSuccess) return set_error((int)e, "scdm_attn_bwd: memset dw: %s", hipGetErrorString(e));
  const long rows = (long)B * T;
  hipLaunchKernelGGL(scdm_bwd_rows_kernel<NP>, dim3((unsigned)cdiv((int)rows, kWaves)), dim3(kThreads), 0, st,
                     V, P, dC, de, B, T, N, Ds);
  int rc = check_launch("scdm_attn_bwd(rows)");
  if (rc) return rc;
  const int hslices = cdiv(H, 256), slices = hslices > cdiv(Ds, 256) ? hslices : cdiv(Ds, 256);
  const size_t lds = sizeof(float) * kWaves * (NP + 1) * 256;
  auto kern = scdm_bwd_cols_kernel<NP>;
  static thread_local bool allowed = false;
  if (!allowed && lds > 64 * 1024) {
    e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "scdm_attn_bwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
    allowed = true;
  }
  hipLaunchKernelGGL(kern, dim3(B * slices), dim3(kThreads), lds, st, a, s, w, P, dC, de, da, ds, dw, dV,
                     B, T, N, H, Ds, hslices, slices);
  return check_launch("scdm_attn_bwd(cols)");
}

int check_common(const char* fn, std::initializer_list<const void*> ptrs, int B, int T, int N, int H, int Ds, int dtype) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (B <= 0 || T <= 0 || N <= 0 || H <= 0 || Ds <= 0)
    return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d N=%d H=%d Ds=%d", fn, B, T, N, H, Ds);
  if (N > 32) return set_error(TSG_E_SHAPE, "%s: N=%d > 32 words not supported", fn, N);
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
  TSG_DISPATCH_NP(np, (launch_fwd<NP>((const float*)a, (const float*)s, (const float*)w,
                                            (const float*)sent, (float*)C, (float*)P, B, T, N, H, Ds, st)));
}

extern "C" int tsg_scdm_attn_bwd(const void* a, const void* s, const void* w, const void* sent,
                                 const void* P, const void* dC, void* da, void* ds, void* dw, void* dsent,
                                 void* de_ws, int B, int T, int N, int H, int Ds, int dtype, void* stream) {
  int rc = check_common("tsg_scdm_attn_bwd", {a, s, w, sent, P, dC, da, ds, dw, dsent, de_ws}, B, T, N, H, Ds, dtype);
  if (rc) return rc;
  const int np = roundup(N, 4);
  auto st = static_cast<hipStream_t>(stream);
  TSG_DISPATCH_NP(np, (launch_bwd<NP>((const float*)a, (const float*)s, (const float*)w, (const float*)sent,
                                      (const float*)P, (const float*)dC, (float*)da, (float*)ds, (float*)dw,
                                      (float*)dsent, (float*)de_ws, B, T, N, H, Ds, st)));
}
