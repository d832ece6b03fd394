// K2 -- fused multi-head dot-product attention for gfx950, with the exact semantics of the
// reference's MultiHead / Attention (grounding/model/networks/attention.py:39-97) between its
// wq/wk/wv projections and wo:
//     Qh,Kh,Vh = chunk(n_heads);  A_h = (Qh Kh^T - 1e10*triu(1) [causal]) / scale;
//     S_h = softmax(A_h);  O = cat_h(S_h Vh);  side outputs  sum_h A_h  and  sum_h S_h (A_forward).
// `scale` is passed in by the caller: the reference divides by sqrt(d_key) of the FULL model width,
// not of the head -- nothing in here assumes sqrt(d_head).
//
// Forward: workgroup = (batch b, 32-query tile), 256 threads, heads processed one after another.
// Per head the Q tile, a 32-key K block and the matching V block are staged in LDS in 128-channel
// chunks (rows padded by 4 floats: the b128 reads below are conflict-free); thread (r = tid/8,
// s = tid%8) owns query row r: scores for keys {s, s+8, s+16, s+24} of the block, then output
// channels {4s + 32j}.  Softmax is online over key blocks (flash-style running max / sum in
// registers, 8-lane DPP reductions); the [Tq,Tk] score matrix only exists if the caller asks for
// the A_forward side outputs.  LSE per (b, head, query) is kept for the backward.
//
// Backward: workgroup = (b, head): every dK/dV/dQ element of that head is produced by exactly one
// workgroup -- no atomics, bitwise reproducible.  For each 32-key block the dK/dV accumulators live
// in registers while the query tiles stream through LDS; P is recomputed from LSE.
#include "tsg_common.h"
#include <cstdlib>

namespace tsg {
namespace {

constexpr int kThreads = 256;
constexpr int TQ = 32;          // queries per tile
constexpr int KB = 32;          // keys per block
constexpr int CC = 128;         // channels per LDS chunk
constexpr int LS = CC + 4;      // padded LDS row stride (floats)
constexpr int PS = KB + 4;      // padded stride of the P / dS tiles
constexpr int MAXVC = 4;        // head width of V <= MAXVC*CC = 512
constexpr float kNegBig = -3.0e38f;

// Attention dropout (reference: out = dropout(softmax(A)) V, attention.py:53-54; the returned A_softmax is the
// un-dropped softmax).  Counter-based: the keep decision of element (b, head, q, key) is a hash of its index and
// of (seed, offset), so forward and backward regenerate the same mask and nothing is stored.  thresh = p * 2^32
// (0 = no dropout), inv_keep = 1/(1-p).
struct DropCfg { unsigned thresh; float inv_keep; unsigned k0, k1; const unsigned long long* rng; };
// rng != NULL: (seed, offset) live in DEVICE memory ([0] = seed, [1] = offset) and the keys are derived in the kernel -- the
// launch can then sit in a captured HIP graph whose replays advance the offset with a captured add (tsg_mha_fwd_rng).
__device__ __forceinline__ void drop_resolve(DropCfg& dc) {
  if (dc.rng) {
    const unsigned long long seed = dc.rng[0], off = dc.rng[1];
    dc.k0 = (unsigned)seed ^ ((unsigned)off * 0x9E3779B1u);
    dc.k1 = (unsigned)(seed >> 32) ^ ((unsigned)(off >> 32) * 0x85EBCA77u + 0x165667B1u);
  }
}
__device__ __forceinline__ unsigned mix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float drop_scale(const DropCfg& dc, int b, int H, int hd, int Tq, int q, int Tk, int key) {
  const unsigned long long idx = ((unsigned long long)((size_t)b * H + hd) * Tq + q) * Tk + key;
  const unsigned h = mix32(mix32((unsigned)idx + dc.k0) ^ (unsigned)(idx >> 32) ^ dc.k1);
  return h >= dc.thresh ? dc.inv_keep : 0.f;
}

// reduce over the 8 consecutive lanes that share one query row
__device__ __forceinline__ float sum8(float v) {
  v += dpp_mov<0xB1>(v); v += dpp_mov<0x4E>(v); v += dpp_mov<0x141>(v);
  return v;
}
__device__ __forceinline__ float max8(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v)); v = fmaxf(v, dpp_mov<0x4E>(v)); v = fmaxf(v, dpp_mov<0x141>(v));
  return v;
}

// stage rows [row0, row0+32) x channels [c0, c0+CC) of a [rows_total, ld] matrix into LDS (zero fill)
__device__ __forceinline__ void stage_tile(float* __restrict__ dst, const float* __restrict__ src, int row0,
                                           int rows_total, int ld, int c0, int cend) {
  for (int idx = threadIdx.x; idx < 32 * (CC / 4); idx += kThreads) {
    const int r = idx / (CC / 4), c = (idx % (CC / 4)) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row0 + r < rows_total && c0 + c < cend) v = *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * ld + c0 + c);
    *reinterpret_cast<float4*>(dst + r * LS + c) = v;
  }
}

// s[j] += <X[r][:], Y[sub + 8j][:]> over one staged channel chunk
__device__ __forceinline__ void dot_rows(const float* __restrict__ Xs, const float* __restrict__ Ys, int r, int sub,
                                         float (&s)[4]) {
#pragma unroll 4
  for (int c = 0; c < CC; c += 4) {
    const float4 x = *reinterpret_cast<const float4*>(Xs + r * LS + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float4 y = *reinterpret_cast<const float4*>(Ys + (sub + 8 * j) * LS + c);
      s[j] = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, s[j]))));
    }
  }
}

// raw dot products of the tile -> scaled, causally shifted scores a[j] for keys k0 + sub + 8j
__device__ __forceinline__ void finish_scores(float (&a)[4], int q, int k0, int sub, int Tk, float inv_scale,
                                              int causal) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = k0 + sub + 8 * j;
    float v = a[j];
    if (causal && key > q) v -= 1e10f;          // attention.py:47-51: subtracted BEFORE the division
    v *= inv_scale;
    a[j] = (key < Tk) ? v : kNegBig;
  }
}

__global__ __launch_bounds__(kThreads) void mha_fwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ Asum, float* __restrict__ Ssum, float* __restrict__ LSE,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, int qtiles, DropCfg dc) {
  drop_resolve(dc);
  __shared__ __align__(16) float Qs[TQ * LS];
  __shared__ __align__(16) float Ks[KB * LS];      // K chunk, then reused for the V chunk
  __shared__ __align__(16) float Ps[TQ * PS];
  const int tid = threadIdx.x, r = tid >> 3, sub = tid & 7;
  const int b = blockIdx.x / qtiles, q0 = (blockIdx.x % qtiles) * TQ;
  const int q = q0 + r;
  const int dh = dk / H, dvh = dv / H;
  const float* Qb = Q + (size_t)b * Tq * dk;
  const float* Kb = K + (size_t)b * Tk * dk;
  const float* Vb = V + (size_t)b * Tk * dv;
  const int vchunks = (dvh + CC - 1) / CC;

  for (int hd = 0; hd < H; ++hd) {
    float m_run = kNegBig, l_run = 0.f;
    float4 o[MAXVC][4];
#pragma unroll
    for (int vc = 0; vc < MAXVC; ++vc)
#pragma unroll
      for (int j = 0; j < 4; ++j) o[vc][j] = make_float4(0.f, 0.f, 0.f, 0.f);

    for (int pass = 0; pass < (Ssum ? 2 : 1); ++pass) {
      // pass 0: online softmax + PV.  pass 1 (only for the A_forward side output): the final
      // probabilities exp(a - lse) need the complete row sum, so the scores are recomputed.
      const float lse = (pass == 1) ? m_run + __logf(l_run) : 0.f;
      for (int k0 = 0; k0 < Tk; k0 += KB) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c0 = 0; c0 < dh; c0 += CC) {
          __syncthreads();
          stage_tile(Qs, Qb + hd * dh, q0, Tq, dk, c0, dh);
          stage_tile(Ks, Kb + hd * dh, k0, Tk, dk, c0, dh);
          __syncthreads();
          dot_rows(Qs, Ks, r, sub, a);
        }
        finish_scores(a, q, k0, sub, Tk, inv_scale, causal);
        if (pass == 1) {
          if (q < Tq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int key = k0 + sub + 8 * j;
              if (key < Tk) Ssum[((size_t)b * Tq + q) * Tk + key] += __expf(a[j] - lse);
            }
          }
          continue;
        }
        if (Asum && q < Tq) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int key = k0 + sub + 8 * j;
            if (key < Tk) Asum[((size_t)b * Tq + q) * Tk + key] += a[j];
          }
        }
        const float m_new = fmaxf(m_run, max8(fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]))));
        const float alpha = __expf(m_run - m_new);
        float psum = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = (a[j] > kNegBig) ? __expf(a[j] - m_new) : 0.f;
          Ps[r * PS + sub + 8 * j] = dc.thresh ? p * drop_scale(dc, b, H, hd, Tq, q, Tk, k0 + sub + 8 * j) : p;
          psum += p;
        }
        l_run = l_run * alpha + sum8(psum);
        m_run = m_new;
#pragma unroll
        for (int vc = 0; vc < MAXVC; ++vc)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            o[vc][j].x *= alpha; o[vc][j].y *= alpha; o[vc][j].z *= alpha; o[vc][j].w *= alpha;
          }
#pragma unroll
        for (int vc = 0; vc < MAXVC; ++vc) {
          if (vc < vchunks) {
            __syncthreads();
            stage_tile(Ks, Vb + hd * dvh, k0, Tk, dv, vc * CC, dvh);
            __syncthreads();
#pragma unroll 2
            for (int n4 = 0; n4 < KB; n4 += 4) {
              const float4 p4 = *reinterpret_cast<const float4*>(Ps + r * PS + n4);
              const float pp[4] = {p4.x, p4.y, p4.z, p4.w};
#pragma unroll
              for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                  const float4 v = *reinterpret_cast<const float4*>(Ks + (n4 + u) * LS + sub * 4 + 32 * j);
                  o[vc][j].x = fmaf(pp[u], v.x, o[vc][j].x); o[vc][j].y = fmaf(pp[u], v.y, o[vc][j].y);
                  o[vc][j].z = fmaf(pp[u], v.z, o[vc][j].z); o[vc][j].w = fmaf(pp[u], v.w, o[vc][j].w);
                }
            }
          }
        }
      }
    }
    if (q < Tq) {
      const float inv = 1.f / l_run;
#pragma unroll
      for (int vc = 0; vc < MAXVC; ++vc)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = vc * CC + sub * 4 + 32 * j;
          if (vc < vchunks && c < dvh)
            *reinterpret_cast<float4*>(O + ((size_t)b * Tq + q) * dv + hd * dvh + c) =
                make_float4(o[vc][j].x * inv, o[vc][j].y * inv, o[vc][j].z * inv, o[vc][j].w * inv);
        }
      if (sub == 0) LSE[((size_t)b * H + hd) * Tq + q] = m_run + __logf(l_run);
    }
  }
}

// ------------------------------------------------------------------------------------------
// forward, MFMA path (head widths <= 128, no side outputs): fp32 v_mfma_f32_32x32x2_f32.
// Workgroup = (b, head, 128 queries), 4 waves x 32 queries.  Per 32-key block the K_h / V_h rows are
// staged once in LDS and shared by the four waves; each wave keeps its Q_h rows as MFMA B-operand
// fragments in registers.  The scores are computed TRANSPOSED, S^T = K_h Q_h^T, so that a lane owns
// one query (column) and 16 of the 32 keys (rows) in its accumulator registers: the softmax over keys
// is in-register plus one exchange with lane^32, the running rescale of O is lane-local, and the
// P^T accumulator registers are, unchanged, the B operand of the second product O^T = V_h^T P^T
// (register r pairs keys rho(r), rho(r)+4 -- the MFMA k-order is simply permuted to match).
// ------------------------------------------------------------------------------------------
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kMfmaThreads = 256;
constexpr int QB = 128;           // queries per workgroup
constexpr int DHMAX = 128;        // max head width on this path

__device__ __forceinline__ float xhalf_max(float v) {   // combine lane l with lane l^32
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xhalf_sum(float v) {
  const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Batched tile staging: every thread first REQUESTS its NV float4 (rows x cols4 float4 per row, zero fill
// outside the matrix), then writes them to LDS -- one round trip instead of NV dependent ones.
template <int NV, int NT = 256>
struct TileStage {
  float4 v[NV];
  __device__ __forceinline__ void load(const float* __restrict__ src, int ld, int row0, int rows_total, int rows,
                                       int cols, int cols4) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / cols4, c = (idx % cols4) * 4;
      v[i] = (r < rows && row0 + r < rows_total && c < cols) ? *reinterpret_cast<const float4*>(src + (size_t)(row0 + r) * ld + c)
                                                             : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  // 8-byte stores: row strides of the A-operand tiles are only 8-byte aligned (stride = 2 mod 64 floats)
  __device__ __forceinline__ void store(float* __restrict__ dst, int stride, int rows, int cols4) const {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const int idx = threadIdx.x + i * NT;
      const int r = idx / cols4, c = (idx % cols4) * 4;
      if (r < rows && c + 4 <= stride) {                 // the padded row (zeros beyond the head width)
        *reinterpret_cast<float2*>(dst + r * stride + c) = make_float2(v[i].x, v[i].y);
        *reinterpret_cast<float2*>(dst + r * stride + c + 2) = make_float2(v[i].z, v[i].w);
      }
    }
  }
};

// FULL: head widths are exactly DHMAX (the common d/8 = 128 case): the channel-count guards fold away, so no wave-uniform
// branch stands between the MFMAs (with the run-time guards every MFMA group sat behind one).
template <bool DROP, bool FULL>
__global__ __launch_bounds__(kMfmaThreads) __attribute__((amdgpu_waves_per_eu(2, 2))) void mha_fwd_mfma_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ LSE,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, int qblocks, int KS, int VS, DropCfg dc) {
  drop_resolve(dc);
  extern __shared__ __align__(16) float lds[];
  float* Kl = lds;                         // [32][KS]   KS = roundup(dh,64)+2  (conflict-free b64 A reads)
  float* Vl = lds + 32 * KS;               // [32][VS]   VS = roundup(dvh,32)+4, zero beyond dvh
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hd = blockIdx.x % H;
  const int qb = (blockIdx.x / H) % qblocks, b = blockIdx.x / (H * qblocks);
  const int dh = dk / H, dvh = dv / H;
  const int jq = lane & 31, kk = lane >> 5;
  const int q = qb * QB + wv * 32 + jq;    // this lane's query
  const float* Qb = Q + (size_t)b * Tq * dk + hd * dh;
  const float* Kb = K + (size_t)b * Tk * dk + hd * dh;
  const float* Vb = V + (size_t)b * Tk * dv + hd * dvh;
  const int nsteps = FULL ? DHMAX / 4 : (dh + 3) / 4;       // MFMA pairs over the head channels
  const int ctiles = FULL ? DHMAX / 32 : (dvh + 31) / 32;

  // K_h / V_h rows of a 32-key block: 4 float4 per thread each, requested first (they fly during the Q staging);
  // inside the loop the NEXT block is requested while the current one is in the MFMAs (zero fill beyond Tk / head width)
  TileStage<32 * (DHMAX / 4) / kMfmaThreads> ks, vs;
  ks.load(Kb, dk, 0, Tk, 32, dh, DHMAX / 4);
  vs.load(Vb, dv, 0, Tk, 32, dvh, DHMAX / 4);

  // Q fragments: B operand of S^T = K Q^T.  Staged through LDS (rows are read coalesced) once.
  float2 qf[DHMAX / 4];
  {
    float* Ql = lds;                       // [128][KS] -- fits: 128*130*4 = 66.6 KB <= allocated (see host)
    TileStage<QB * (DHMAX / 4) / kMfmaThreads> qs;       // 16 float4 per thread at dh = 128
    qs.load(Qb, dk, qb * QB, Tq, QB, dh, DHMAX / 4);
    qs.store(Ql, KS, QB, DHMAX / 4);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < DHMAX / 4; ++s)
      qf[s] = (FULL || (s < nsteps && 4 * s + 2 * kk < dh)) ? *reinterpret_cast<const float2*>(Ql + (wv * 32 + jq) * KS + 4 * s + 2 * kk)
                                                  : make_float2(0.f, 0.f);
    __syncthreads();
  }

  f32x16 o[DHMAX / 32];
#pragma unroll
  for (int ct = 0; ct < DHMAX / 32; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  for (int k0 = 0; k0 < Tk; k0 += 32) {
    ks.store(Kl, KS, 32, DHMAX / 4);
    vs.store(Vl, VS, 32, DHMAX / 4);
    __syncthreads();
    if (k0 + 32 < Tk) {
      ks.load(Kb, dk, k0 + 32, Tk, 32, dh, DHMAX / 4);
      vs.load(Vb, dv, k0 + 32, Tk, 32, dvh, DHMAX / 4);
    }

    // S^T tile: rows = keys (rho(r) + 4*kk), column = this lane's query
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const float* krow = Kl + jq * KS + 2 * kk;            // A operand: K[key = lane&31][channel pair]
#pragma unroll
    for (int s = 0; s < DHMAX / 4; ++s) {
      if (FULL || s < nsteps) {                           // wave-uniform
        const float2 a = *reinterpret_cast<const float2*>(krow + 4 * s);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[s].x, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[s].y, st, 0, 0, 0);
      }
    }
    float mb = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      float v = st[r];
      if (causal && key > q) v -= 1e10f;
      v *= inv_scale;
      v = (key < Tk) ? v : kNegBig;
      st[r] = v;
      mb = fmaxf(mb, v);
    }
    const float m_new = fmaxf(m_run, xhalf_max(mb));
    const float alpha = __expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (st[r] > kNegBig) ? __expf(st[r] - m_new) : 0.f;
      st[r] = DROP ? p * drop_scale(dc, b, H, hd, Tq, q, Tk, k0 + (r & 3) + 8 * (r >> 2) + 4 * kk) : p;
      ps += p;
    }
    l_run = l_run * alpha + xhalf_sum(ps);
    m_run = m_new;
    // O^T += V^T P^T : A operand V[key = rho(r) + 4*kk][channel = ct*32 + lane&31], B operand = st[r]
#pragma unroll
    for (int ct = 0; ct < DHMAX / 32; ++ct) {
      if (FULL || ct < ctiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = Vl[((r & 3) + 8 * (r >> 2) + 4 * kk) * VS + ct * 32 + jq];
          o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, st[r], o[ct], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: O[q][c] = o / l ; transpose through LDS for coalesced row stores
  float* Ol = lds;                                       // [128][VS]
  const float inv = 1.f / l_run;
#pragma unroll
  for (int ct = 0; ct < DHMAX / 32; ++ct)
    if (FULL || ct < ctiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (c < dvh) Ol[(wv * 32 + jq) * VS + c] = o[ct][r] * inv;
      }
    }
  if (kk == 0 && q < Tq) LSE[((size_t)b * H + hd) * Tq + q] = m_run + __logf(l_run);
  __syncthreads();
  for (int idx = tid; idx < QB * (dvh / 4); idx += kMfmaThreads) {
    const int r = idx / (dvh / 4), c = (idx % (dvh / 4)) * 4;
    const int qq = qb * QB + r;
    if (qq < Tq)
      *reinterpret_cast<float4*>(O + ((size_t)b * Tq + qq) * dv + hd * dvh + c) = *reinterpret_cast<const float4*>(Ol + r * VS + c);
  }
}

// ------------------------------------------------------------------------------------------
// forward, MFMA path for WIDE heads (128 < head width <= 256: Self_Attention_predictor at d = 1024 has 2d/8 = 256 channels
// per head, SpanPredictor.py:244-266).  Keeping a 256-wide Q fragment and a 256-wide O accumulator per lane as the kernel
// above does would take one wave per SIMD and still spill, so the head's channels are SPLIT OVER THE WAVES instead (as in
// the backward): workgroup = (b, head, 64 queries), 8 waves = 2 query tiles x 4 channel quarters.  Per 32-key block wave
// (qt, cw) forms the partial S^T = K Q^T over its 64 channels (32 MFMAs), the four quarters meet in LDS, every wave
// rebuilds the full tile and the (replicated) online softmax, and accumulates O^T = V^T P^T for ITS 64 output channels.
// Per wave: 32 registers of Q fragments, 32 of O accumulators -- two waves per SIMD with room to spare.
// ------------------------------------------------------------------------------------------
constexpr int kWideThreads = 512;
constexpr int QBW = 64;           // queries per workgroup
constexpr int DHW = 256;          // max head width

template <bool DROP>
__global__ __launch_bounds__(kWideThreads) void mha_fwd_mfma_wide_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    float* __restrict__ O, float* __restrict__ LSE,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, int qblocks, int KS, int VS, DropCfg dc) {
  drop_resolve(dc);
  extern __shared__ __align__(16) float lds[];
  float* Kl = lds;                         // [32][KS]   KS = 258 (= 2 mod 64: conflict-free b64 A reads), zero beyond dh
  float* Vl = lds + 32 * KS;               // [32][VS]   zero beyond dvh
  float* X = Vl + 32 * VS;                 // [2 query tiles][4 quarters][16][64] partial S^T
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qt = wv >> 2, cw = wv & 3;     // query tile, channel quarter
  const int hd = blockIdx.x % H;
  const int qb = (blockIdx.x / H) % qblocks, b = blockIdx.x / (H * qblocks);
  const int dh = dk / H, dvh = dv / H;
  const int jq = lane & 31, kk = lane >> 5;
  const int q = qb * QBW + qt * 32 + jq;   // this lane's query
  const float* Qb = Q + (size_t)b * Tq * dk + hd * dh;
  const float* Kb = K + (size_t)b * Tk * dk + hd * dh;
  const float* Vb = V + (size_t)b * Tk * dv + hd * dvh;
  const int c0 = cw * 64;                  // this wave's channels (of Q/K for the scores, of V/O for the output)
  const int ksteps = c0 < dh ? ((dh - c0 < 64 ? dh - c0 : 64) + 3) / 4 : 0;      // MFMA pairs over my score channels
  const int ctiles = c0 < dvh ? ((dvh - c0 < 64 ? dvh - c0 : 64) + 31) / 32 : 0; // my 32-channel output tiles (0..2)

  TileStage<32 * (DHW / 4) / kWideThreads, kWideThreads> ks, vs;                  // 4 float4 per thread each
  ks.load(Kb, dk, 0, Tk, 32, dh, DHW / 4);
  vs.load(Vb, dv, 0, Tk, 32, dvh, DHW / 4);

  float2 qf[16];                           // Q fragments of my 64 channels: B operand of S^T = K Q^T
  {
    float* Ql = lds;                       // [64][KS] = 66 KB, aliased with Kl / Vl / X (98 KB)
    TileStage<QBW * (DHW / 4) / kWideThreads, kWideThreads> qs;                   // 8 float4 per thread
    qs.load(Qb, dk, qb * QBW, Tq, QBW, dh, DHW / 4);
    qs.store(Ql, KS, QBW, DHW / 4);
    __syncthreads();
#pragma unroll
    for (int s = 0; s < 16; ++s)
      qf[s] = (s < ksteps) ? *reinterpret_cast<const float2*>(Ql + (qt * 32 + jq) * KS + c0 + 4 * s + 2 * kk) : make_float2(0.f, 0.f);
    __syncthreads();
  }

  f32x16 o[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[ct][r] = 0.f;
  float m_run = kNegBig, l_run = 0.f;

  for (int k0 = 0; k0 < Tk; k0 += 32) {
    ks.store(Kl, KS, 32, DHW / 4);
    vs.store(Vl, VS, 32, DHW / 4);
    __syncthreads();
    if (k0 + 32 < Tk) {
      ks.load(Kb, dk, k0 + 32, Tk, 32, dh, DHW / 4);
      vs.load(Vb, dv, k0 + 32, Tk, 32, dvh, DHW / 4);
    }
    // partial S^T over my channels: rows = keys rho(r) + 4 kk, column = this lane's query
    f32x16 st;
#pragma unroll
    for (int r = 0; r < 16; ++r) st[r] = 0.f;
    const float* krow = Kl + jq * KS + c0 + 2 * kk;      // A operand: K[key = lane&31][channel pair]
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      if (s < ksteps) {                                   // wave-uniform
        const float2 a = *reinterpret_cast<const float2*>(krow + 4 * s);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, qf[s].x, st, 0, 0, 0);
        st = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, qf[s].y, st, 0, 0, 0);
      }
    }
    float* Xq = X + qt * 4 * 16 * 64;
#pragma unroll
    for (int r = 0; r < 16; ++r) Xq[(cw * 16 + r) * 64 + lane] = st[r];
    __syncthreads();
    float mb = kNegBig;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = (Xq[(0 * 16 + r) * 64 + lane] + Xq[(1 * 16 + r) * 64 + lane]) + (Xq[(2 * 16 + r) * 64 + lane] + Xq[(3 * 16 + r) * 64 + lane]);
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * kk;
      if (causal && key > q) v -= 1e10f;
      v *= inv_scale;
      v = (key < Tk) ? v : kNegBig;
      st[r] = v;
      mb = fmaxf(mb, v);
    }
    const float m_new = fmaxf(m_run, xhalf_max(mb));
    const float alpha = __expf(m_run - m_new);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = (st[r] > kNegBig) ? __expf(st[r] - m_new) : 0.f;
      st[r] = DROP ? p * drop_scale(dc, b, H, hd, Tq, q, Tk, k0 + (r & 3) + 8 * (r >> 2) + 4 * kk) : p;
      ps += p;
    }
    l_run = l_run * alpha + xhalf_sum(ps);
    m_run = m_new;
    // O^T += V^T P^T for my output channels: A operand V[key = rho(r) + 4 kk][channel], B operand = st[r]
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      if (ct < ctiles) {
#pragma unroll
        for (int r = 0; r < 16; ++r) o[ct][r] *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float a = Vl[((r & 3) + 8 * (r >> 2) + 4 * kk) * VS + c0 + ct * 32 + jq];
          o[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, st[r], o[ct], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }

  // epilogue: O[q][c] = o / l ; transpose through LDS for coalesced row stores
  float* Ol = lds;                                       // [64][VS]
  const float inv = 1.f / l_run;
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
    if (ct < ctiles) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = c0 + ct * 32 + (r & 3) + 8 * (r >> 2) + 4 * kk;
        if (c < dvh) Ol[(qt * 32 + jq) * VS + c] = o[ct][r] * inv;
      }
    }
  if (cw == 0 && kk == 0 && q < Tq) LSE[((size_t)b * H + hd) * Tq + q] = m_run + __logf(l_run);
  __syncthreads();
  for (int idx = tid; idx < QBW * (dvh / 4); idx += kWideThreads) {
    const int r = idx / (dvh / 4), c = (idx % (dvh / 4)) * 4;
    const int qq = qb * QBW + r;
    if (qq < Tq)
      *reinterpret_cast<float4*>(O + ((size_t)b * Tq + qq) * dv + hd * dvh + c) = *reinterpret_cast<const float4*>(Ol + r * VS + c);
  }
}

// ------------------------------------------------------------------------------------------
// backward: workgroup = (b, head).
//   D[q]    = <dO[q,:], O[q,:]> (head channels)          P = exp(a - LSE)
//   dV[n,:] = sum_q P[q,n] dO[q,:]       dP[q,n] = <dO[q,:], V[n,:]>       dS = P (dP - D) / scale
//   dK[n,:] = sum_q dS[q,n] Q[q,:]       dQ[q,:] = sum_n dS[q,n] K[n,:]
// (the causal shift is a constant, so it only enters through P.)
// Key blocks outermost: dK/dV of the block accumulate in registers (thread (n = tid/8, s) owns
// channels {4s + 32j} of key n) across the query tiles; dQ is accumulated in global memory by the
// same thread across key blocks.  Head widths up to 128 per chunk pass; wider heads loop chunks.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kThreads) void mha_bwd_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ O, const float* __restrict__ dO, const float* __restrict__ LSE,
    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, DropCfg dc) {
  drop_resolve(dc);
  __shared__ __align__(16) float Xs[TQ * LS];      // Q or dO tile chunk
  __shared__ __align__(16) float Ys[KB * LS];      // K or V block chunk
  __shared__ __align__(16) float Ps[TQ * PS];      // P tile   [q][n]
  __shared__ __align__(16) float Ss[TQ * PS];      // dS tile  [q][n]
  __shared__ float Dq[TQ];
  const int tid = threadIdx.x, r = tid >> 3, sub = tid & 7;
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int dh = dk / H, dvh = dv / H;
  const float* Qb = Q + (size_t)b * Tq * dk + hd * dh;
  const float* Kb = K + (size_t)b * Tk * dk + hd * dh;
  const float* Vb = V + (size_t)b * Tk * dv + hd * dvh;
  const float* Ob = O + (size_t)b * Tq * dv + hd * dvh;
  const float* dOb = dO + (size_t)b * Tq * dv + hd * dvh;
  float* dQb = dQ + (size_t)b * Tq * dk + hd * dh;
  float* dKb = dK + (size_t)b * Tk * dk + hd * dh;
  float* dVb = dV + (size_t)b * Tk * dv + hd * dvh;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const int kchunks = (dh + CC - 1) / CC, vchunks = (dvh + CC - 1) / CC;

  for (int k0 = 0; k0 < Tk; k0 += KB) {
    // channel chunks of the dK / dV accumulators are handled one at a time (registers), which
    // re-runs the score computation per chunk; heads are <= 128 wide in every config of interest.
    const int nchunks = kchunks > vchunks ? kchunks : vchunks;
    for (int ch = 0; ch < nchunks; ++ch) {
      float4 dk_acc[4], dv_acc[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) { dk_acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); dv_acc[j] = make_float4(0.f, 0.f, 0.f, 0.f); }

      for (int q0 = 0; q0 < Tq; q0 += TQ) {
        const int q = q0 + r;
        // ---- scores a[q][n] and dP[q][n] for the tile (thread (r,sub): keys sub+8j) ----
        float a[4] = {0.f, 0.f, 0.f, 0.f}, dp[4] = {0.f, 0.f, 0.f, 0.f}, dpart = 0.f;
        for (int c0 = 0; c0 < dh; c0 += CC) {
          __syncthreads();
          stage_tile(Xs, Qb, q0, Tq, dk, c0, dh);
          stage_tile(Ys, Kb, k0, Tk, dk, c0, dh);
          __syncthreads();
          dot_rows(Xs, Ys, r, sub, a);
        }
        for (int c0 = 0; c0 < dvh; c0 += CC) {
          __syncthreads();
          stage_tile(Xs, dOb, q0, Tq, dv, c0, dvh);
          stage_tile(Ys, Vb, k0, Tk, dv, c0, dvh);
          __syncthreads();
          dot_rows(Xs, Ys, r, sub, dp);
          // D[q] partial: this thread's 1/8 of the chunk's channels
          if (q < Tq) {
            for (int c = sub * 4; c < CC && c0 + c < dvh; c += 32) {
              const float4 g = *reinterpret_cast<const float4*>(Xs + r * LS + c);
              const float4 ov = *reinterpret_cast<const float4*>(Ob + (size_t)q * dv + c0 + c);
              dpart = fmaf(g.x, ov.x, fmaf(g.y, ov.y, fmaf(g.z, ov.z, fmaf(g.w, ov.w, dpart))));
            }
          }
        }
        const float Drow = sum8(dpart);
        finish_scores(a, q, k0, sub, Tk, inv_scale, causal);
        const float l = (q < Tq) ? lse[q] : 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float p = (a[j] > kNegBig && q < Tq) ? __expf(a[j] - l) : 0.f;
          const float mk = dc.thresh ? drop_scale(dc, b, H, hd, Tq, q, Tk, k0 + sub + 8 * j) : 1.f;
          Ps[r * PS + sub + 8 * j] = p * mk;                               // dropped probabilities feed dV
          Ss[r * PS + sub + 8 * j] = p * (dp[j] * mk - Drow) * inv_scale;  // D = <dO, O> already contains the mask
        }
        __syncthreads();

        // ---- dQ[q, chunk ch] += sum_n dS[q][n] K[n][chunk]  (thread (r,sub): channels 4sub+32j) ----
        if (ch < kchunks) {
          __syncthreads();
          stage_tile(Ys, Kb, k0, Tk, dk, ch * CC, dh);
          __syncthreads();
          float4 dq[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) dq[j] = make_float4(0.f, 0.f, 0.f, 0.f);
          for (int n4 = 0; n4 < KB; n4 += 4) {
            const float4 s4 = *reinterpret_cast<const float4*>(Ss + r * PS + n4);
            const float ss[4] = {s4.x, s4.y, s4.z, s4.w};
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
              for (int j = 0; j < 4; ++j) {
                const float4 kv = *reinterpret_cast<const float4*>(Ys + (n4 + u) * LS + sub * 4 + 32 * j);
                dq[j].x = fmaf(ss[u], kv.x, dq[j].x); dq[j].y = fmaf(ss[u], kv.y, dq[j].y);
                dq[j].z = fmaf(ss[u], kv.z, dq[j].z); dq[j].w = fmaf(ss[u], kv.w, dq[j].w);
              }
          }
          if (q < Tq) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const int c = ch * CC + sub * 4 + 32 * j;
              if (c < dh) {
                float4* dst = reinterpret_cast<float4*>(dQb + (size_t)q * dk + c);
                float4 cur = make_float4(0.f, 0.f, 0.f, 0.f);
                if (k0 > 0) cur = *dst;                    // same thread wrote it in the previous key block
                *dst = make_float4(cur.x + dq[j].x, cur.y + dq[j].y, cur.z + dq[j].z, cur.w + dq[j].w);
              }
            }
          }
        }

        // ---- dK[n, chunk] += sum_q dS[q][n] Q[q][chunk];  dV[n, chunk] += sum_q P[q][n] dO[q][chunk]
        //      thread (n = r, sub): channels 4sub+32j of key k0+n.  The tiles are read column-wise:
        //      Ss[q*PS + n] -- 8 lanes share n, rows differ by PS=36 floats: conflict-free. ----
        if (ch < kchunks) {
          __syncthreads();
          stage_tile(Xs, Qb, q0, Tq, dk, ch * CC, dh);
          __syncthreads();
          for (int qq = 0; qq < TQ; ++qq) {
            const float sv = Ss[qq * PS + r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 x = *reinterpret_cast<const float4*>(Xs + qq * LS + sub * 4 + 32 * j);
              dk_acc[j].x = fmaf(sv, x.x, dk_acc[j].x); dk_acc[j].y = fmaf(sv, x.y, dk_acc[j].y);
              dk_acc[j].z = fmaf(sv, x.z, dk_acc[j].z); dk_acc[j].w = fmaf(sv, x.w, dk_acc[j].w);
            }
          }
        }
        if (ch < vchunks) {
          __syncthreads();
          stage_tile(Xs, dOb, q0, Tq, dv, ch * CC, dvh);
          __syncthreads();
          for (int qq = 0; qq < TQ; ++qq) {
            const float pv = Ps[qq * PS + r];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              const float4 x = *reinterpret_cast<const float4*>(Xs + qq * LS + sub * 4 + 32 * j);
              dv_acc[j].x = fmaf(pv, x.x, dv_acc[j].x); dv_acc[j].y = fmaf(pv, x.y, dv_acc[j].y);
              dv_acc[j].z = fmaf(pv, x.z, dv_acc[j].z); dv_acc[j].w = fmaf(pv, x.w, dv_acc[j].w);
            }
          }
        }
      }
      const int key = k0 + r;
      if (key < Tk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = ch * CC + sub * 4 + 32 * j;
          if (ch < kchunks && c < dh) *reinterpret_cast<float4*>(dKb + (size_t)key * dk + c) = dk_acc[j];
          if (ch < vchunks && c < dvh) *reinterpret_cast<float4*>(dVb + (size_t)key * dv + c) = dv_acc[j];
        }
      }
    }
  }
  (void)Dq;
}

// ------------------------------------------------------------------------------------------
// backward, MFMA path (head widths <= 128).  Workgroup = (b, head), 4 waves; wave w owns channel
// tile w (32 channels) of every product.  Orientation "key on the lane": S = Q K^T and dP = dO V^T
// are computed with the key as the MFMA column, so their accumulators (rows = queries) are directly
// the B operands of  dV^T = dO^T P  and  dK^T = Q^T dS  (sums over queries); only dS crosses LDS once,
// as the A operand of dQ = dS K (sum over keys).  The 128-channel contraction of S / dP is split over
// the four waves (32 channels each) and folded through LDS.  delta[q] = <dO[q], O[q]> (head channels)
// comes from a small pre-pass.  Key blocks outermost: dK^T / dV^T accumulate in registers across
// the query tiles; dQ is accumulated in global memory by its owner wave across key blocks.
// No atomics: every output element has exactly one writer.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mha_bwd_delta_kernel(const float* __restrict__ O, const float* __restrict__ dO,
                                                            float* __restrict__ delta, int B, int Tq, int dv, int H) {
  // one wave per (b, q): lanes stride the dv channels; heads are contiguous channel ranges
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wv;
  if (row >= (long)B * Tq) return;
  const int b = (int)(row / Tq), q = (int)(row % Tq), dvh = dv / H;
  const float* o = O + (size_t)row * dv; const float* g = dO + (size_t)row * dv;
  for (int hd = 0; hd < H; ++hd) {
    float acc = 0.f;
    for (int c = lane * 4; c < dvh; c += 256) {
      const float4 x = *reinterpret_cast<const float4*>(o + hd * dvh + c);
      const float4 y = *reinterpret_cast<const float4*>(g + hd * dvh + c);
      acc = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, acc))));
    }
    acc = wave_allsum(acc);
    if (lane == 0) delta[((size_t)b * H + hd) * Tq + q] = acc;
  }
}

__device__ __forceinline__ int rho(int r, int kk) { return (r & 3) + 8 * (r >> 2) + 4 * kk; }

template <bool DROP, bool FULL>
__global__ __launch_bounds__(kMfmaThreads) void mha_bwd_mfma_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ delta,
    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, int KS, int VS2, DropCfg dc) {
  drop_resolve(dc);
  // LDS: Kl [32][KS], Vl [32][VS2], Ql [32][KS], Gl (dO) [32][VS2]  (strides = 2 mod 64: b64 A/B reads),
  //      X [2][4 waves][16][64] partial S / dP, dSl [32][66], lsel [32], dl [32]
  extern __shared__ __align__(16) float lds[];
  float* Kl = lds; float* Vl = Kl + 32 * KS; float* Ql = Vl + 32 * VS2; float* Gl = Ql + 32 * KS;
  float* X = Gl + 32 * VS2; float* dSl = X + 2 * 4 * 16 * 64; float* lsel = dSl + 32 * 66; float* dl = lsel + 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int dh = dk / H, dvh = dv / H;
  const int jl = lane & 31, kk = lane >> 5;
  const float* Qb = Q + (size_t)b * Tq * dk + hd * dh;
  const float* Kb = K + (size_t)b * Tk * dk + hd * dh;
  const float* Vb = V + (size_t)b * Tk * dv + hd * dvh;
  const float* Gb = dO + (size_t)b * Tq * dv + hd * dvh;
  float* dQb = dQ + (size_t)b * Tq * dk + hd * dh;
  float* dKb = dK + (size_t)b * Tk * dk + hd * dh;
  float* dVb = dV + (size_t)b * Tk * dv + hd * dvh;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const float* dlt = delta + ((size_t)b * H + hd) * Tq;
  const int c0 = wv * 32;                                  // this wave's channel tile
  const bool has_k = FULL || c0 < dh, has_v = FULL || c0 < dvh;
  const int ks_steps = FULL ? 8 : has_k ? ((dh - c0 < 32 ? dh - c0 : 32) + 3) / 4 : 0;   // paired MFMA steps over my channels
  const int vs_steps = FULL ? 8 : has_v ? ((dvh - c0 < 32 ? dvh - c0 : 32) + 3) / 4 : 0;

  TileStage<32 * (DHMAX / 4) / kMfmaThreads> t0, t1;
  for (int k0 = 0; k0 < Tk; k0 += 32) {
    __syncthreads();
    t0.load(Kb, dk, k0, Tk, 32, dh, DHMAX / 4); t1.load(Vb, dv, k0, Tk, 32, dvh, DHMAX / 4);
    t0.store(Kl, KS, 32, DHMAX / 4); t1.store(Vl, VS2, 32, DHMAX / 4);
    f32x16 dkt, dvt;
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[r] = 0.f; dvt[r] = 0.f; }

    // Q / dO tiles: the next tile's rows are requested while the current tile is in the MFMAs
    t0.load(Qb, dk, 0, Tq, 32, dh, DHMAX / 4); t1.load(Gb, dv, 0, Tq, 32, dvh, DHMAX / 4);
    float lse_n = (tid < 32 && tid < Tq) ? lse[tid] : 0.f, dl_n = (tid < 32 && tid < Tq) ? dlt[tid] : 0.f;
    for (int q0 = 0; q0 < Tq; q0 += 32) {
      __syncthreads();                                     // previous tile's readers are done
      t0.store(Ql, KS, 32, DHMAX / 4); t1.store(Gl, VS2, 32, DHMAX / 4);
      if (tid < 32) { lsel[tid] = lse_n; dl[tid] = dl_n; }
      __syncthreads();
      if (q0 + 32 < Tq) {
        t0.load(Qb, dk, q0 + 32, Tq, 32, dh, DHMAX / 4); t1.load(Gb, dv, q0 + 32, Tq, 32, dvh, DHMAX / 4);
        if (tid < 32) { lse_n = (q0 + 32 + tid < Tq) ? lse[q0 + 32 + tid] : 0.f; dl_n = (q0 + 32 + tid < Tq) ? dlt[q0 + 32 + tid] : 0.f; }
      }

      // partial S = Q K^T and dP = dO V^T over this wave's 32 channels (A: rows = queries, B: cols = keys)
      f32x16 sp, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (FULL || s < ks_steps) {
          const float2 a = *reinterpret_cast<const float2*>(Ql + jl * KS + c0 + 4 * s + 2 * kk);
          const float2 bb = *reinterpret_cast<const float2*>(Kl + jl * KS + c0 + 4 * s + 2 * kk);
          sp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, sp, 0, 0, 0);
          sp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, sp, 0, 0, 0);
        }
        if (FULL || s < vs_steps) {
          const float2 a = *reinterpret_cast<const float2*>(Gl + jl * VS2 + c0 + 4 * s + 2 * kk);
          const float2 bb = *reinterpret_cast<const float2*>(Vl + jl * VS2 + c0 + 4 * s + 2 * kk);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, dp, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { X[((0 * 4 + wv) * 16 + r) * 64 + lane] = sp[r]; X[((1 * 4 + wv) * 16 + r) * 64 + lane] = dp[r]; }
      __syncthreads();
      // full S, dP (every wave), then P and dS in the accumulator layout: row q = rho(r,kk), col key = jl
      f32x16 pm, ds;
      const int key = k0 + jl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, dpv = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { sv += X[((0 * 4 + u) * 16 + r) * 64 + lane]; dpv += X[((1 * 4 + u) * 16 + r) * 64 + lane]; }
        const int ql = rho(r, kk), q = q0 + ql;
        if (causal && key > q) sv -= 1e10f;
        sv *= inv_scale;
        const float p = (key < Tk && q < Tq) ? __expf(sv - lsel[ql]) : 0.f;
        const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
        pm[r] = p * mk;
        ds[r] = p * (dpv * mk - dl[ql]) * inv_scale;
      }
      if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dSl[rho(r, kk) * 66 + jl] = ds[r];
      }
      __syncthreads();

      // dV^T[c][key] += dO^T[c][q] P[q][key];  dK^T[c][key] += Q^T[c][q] dS[q][key]   (sum over the 32 queries)
      if (has_v) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dvt = __builtin_amdgcn_mfma_f32_32x32x2f32(Gl[rho(r, kk) * VS2 + c0 + jl], pm[r], dvt, 0, 0, 0);
      }
      if (has_k) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          dkt = __builtin_amdgcn_mfma_f32_32x32x2f32(Ql[rho(r, kk) * KS + c0 + jl], ds[r], dkt, 0, 0, 0);
        // dQ[q][c] (+)= dS[q][key] K[key][c]: A = dS rows from LDS, B = K[key pair][my channel]
        f32x16 dq;
#pragma unroll
        for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const float2 a = *reinterpret_cast<const float2*>(dSl + jl * 66 + 4 * s + 2 * kk);
          dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, Kl[(4 * s + 2 * kk) * KS + c0 + jl], dq, 0, 0, 0);
          dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, Kl[(4 * s + 2 * kk + 1) * KS + c0 + jl], dq, 0, 0, 0);
        }
        if (c0 + jl < dh) {
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int q = q0 + rho(r, kk);
            if (q < Tq) {
              float* dst = dQb + (size_t)q * dk + c0 + jl;
              *dst = (k0 > 0 ? *dst : 0.f) + dq[r];       // same lane wrote it for the previous key block
            }
          }
        }
      }
    }
    // dK / dV rows of this key block: accumulators are [channel rows][key on lane] -> transpose through LDS
    __syncthreads();
    float* Tl = Ql;                                        // [32 keys][KS] and Gl as [32][VS2]
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (has_k) Ql[jl * KS + c0 + rho(r, kk)] = dkt[r];
      if (has_v) Gl[jl * VS2 + c0 + rho(r, kk)] = dvt[r];
    }
    __syncthreads();
    for (int idx = tid; idx < 32 * (dh / 4); idx += kMfmaThreads) {
      const int r = idx / (dh / 4), c = (idx % (dh / 4)) * 4;
      if (k0 + r < Tk)
        *reinterpret_cast<float4*>(dKb + (size_t)(k0 + r) * dk + c) = make_float4(Tl[r * KS + c], Tl[r * KS + c + 1], Tl[r * KS + c + 2], Tl[r * KS + c + 3]);
    }
    for (int idx = tid; idx < 32 * (dvh / 4); idx += kMfmaThreads) {
      const int r = idx / (dvh / 4), c = (idx % (dvh / 4)) * 4;
      if (k0 + r < Tk)
        *reinterpret_cast<float4*>(dVb + (size_t)(k0 + r) * dv + c) = make_float4(Gl[r * VS2 + c], Gl[r * VS2 + c + 1], Gl[r * VS2 + c + 2], Gl[r * VS2 + c + 3]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward, MFMA path for WIDE heads (128 < head width <= 256).  Same scheme as above with TWO 32-channel tiles per
// wave (wave w owns channels [64w, 64w+64) of every product).  Four [32][258] operand tiles do not fit the LDS next to
// the exchange buffer, so the V rows of the key block -- only ever the B operand of dP = dO V^T, 16 channel pairs per
// lane -- are held in registers for the length of the key block (32 VGPRs; one wave per SIMD has 512).
// ------------------------------------------------------------------------------------------
template <bool DROP>
__global__ __launch_bounds__(kMfmaThreads) void mha_bwd_mfma_wide_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
    const float* __restrict__ dO, const float* __restrict__ LSE, const float* __restrict__ delta,
    float* __restrict__ dQ, float* __restrict__ dK, float* __restrict__ dV,
    int B, int Tq, int Tk, int dk, int dv, int H, float inv_scale, int causal, int KS, int VS2, DropCfg dc) {
  drop_resolve(dc);
  // LDS: Kl [32][KS], Ql [32][KS], Gl (dO; V while its fragments are read) [32][VS2]  (strides = 2 mod 64: b64 A/B reads),
  //      X [2][4 waves][16][64] partial S / dP, dSl [32][66], lsel [32], dl [32]
  extern __shared__ __align__(16) float lds[];
  float* Kl = lds; float* Ql = Kl + 32 * KS; float* Gl = Ql + 32 * KS;
  float* X = Gl + 32 * VS2; float* dSl = X + 2 * 4 * 16 * 64; float* lsel = dSl + 32 * 66; float* dl = lsel + 32;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.x / H, hd = blockIdx.x % H;
  const int dh = dk / H, dvh = dv / H;
  const int jl = lane & 31, kk = lane >> 5;
  const float* Qb = Q + (size_t)b * Tq * dk + hd * dh;
  const float* Kb = K + (size_t)b * Tk * dk + hd * dh;
  const float* Vb = V + (size_t)b * Tk * dv + hd * dvh;
  const float* Gb = dO + (size_t)b * Tq * dv + hd * dvh;
  float* dQb = dQ + (size_t)b * Tq * dk + hd * dh;
  float* dKb = dK + (size_t)b * Tk * dk + hd * dh;
  float* dVb = dV + (size_t)b * Tk * dv + hd * dvh;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const float* dlt = delta + ((size_t)b * H + hd) * Tq;
  const int c0 = wv * 64;                                  // this wave's 64 channels = two 32-channel tiles
  const int kw = c0 < dh ? (dh - c0 < 64 ? dh - c0 : 64) : 0, vw = c0 < dvh ? (dvh - c0 < 64 ? dvh - c0 : 64) : 0;
  const int ks_steps = (kw + 3) / 4, vs_steps = (vw + 3) / 4;       // paired MFMA steps over my channels (<= 16)
  const int ktiles = (kw + 31) / 32, vtiles = (vw + 31) / 32;       // my 32-channel tiles (0..2)

  TileStage<32 * (DHW / 4) / kMfmaThreads> t0, t1;          // 8 float4 per thread each
  for (int k0 = 0; k0 < Tk; k0 += 32) {
    __syncthreads();
    t0.load(Kb, dk, k0, Tk, 32, dh, DHW / 4); t1.load(Vb, dv, k0, Tk, 32, dvh, DHW / 4);
    t0.store(Kl, KS, 32, DHW / 4); t1.store(Gl, VS2, 32, DHW / 4);
    // Q / dO tiles: the next tile's rows are requested while the current tile is in the MFMAs
    t0.load(Qb, dk, 0, Tq, 32, dh, DHW / 4); t1.load(Gb, dv, 0, Tq, 32, dvh, DHW / 4);
    __syncthreads();
    float2 vf[16];                                         // V[key = jl][my channel pairs]: B operand of dP
#pragma unroll
    for (int s = 0; s < 16; ++s)
      vf[s] = (s < vs_steps) ? *reinterpret_cast<const float2*>(Gl + jl * VS2 + c0 + 4 * s + 2 * kk) : make_float2(0.f, 0.f);
    f32x16 dkt[2], dvt[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) { dkt[ct][r] = 0.f; dvt[ct][r] = 0.f; }

    float lse_n = (tid < 32 && tid < Tq) ? lse[tid] : 0.f, dl_n = (tid < 32 && tid < Tq) ? dlt[tid] : 0.f;
    for (int q0 = 0; q0 < Tq; q0 += 32) {
      __syncthreads();                                     // previous tile's readers (and the V fragment reads) are done
      t0.store(Ql, KS, 32, DHW / 4); t1.store(Gl, VS2, 32, DHW / 4);
      if (tid < 32) { lsel[tid] = lse_n; dl[tid] = dl_n; }
      __syncthreads();
      if (q0 + 32 < Tq) {
        t0.load(Qb, dk, q0 + 32, Tq, 32, dh, DHW / 4); t1.load(Gb, dv, q0 + 32, Tq, 32, dvh, DHW / 4);
        if (tid < 32) { lse_n = (q0 + 32 + tid < Tq) ? lse[q0 + 32 + tid] : 0.f; dl_n = (q0 + 32 + tid < Tq) ? dlt[q0 + 32 + tid] : 0.f; }
      }

      // partial S = Q K^T and dP = dO V^T over this wave's 64 channels (A: rows = queries, B: cols = keys)
      f32x16 sp, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        if (s < ks_steps) {
          const float2 a = *reinterpret_cast<const float2*>(Ql + jl * KS + c0 + 4 * s + 2 * kk);
          const float2 bb = *reinterpret_cast<const float2*>(Kl + jl * KS + c0 + 4 * s + 2 * kk);
          sp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bb.x, sp, 0, 0, 0);
          sp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bb.y, sp, 0, 0, 0);
        }
        if (s < vs_steps) {
          const float2 a = *reinterpret_cast<const float2*>(Gl + jl * VS2 + c0 + 4 * s + 2 * kk);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, vf[s].x, dp, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, vf[s].y, dp, 0, 0, 0);
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) { X[((0 * 4 + wv) * 16 + r) * 64 + lane] = sp[r]; X[((1 * 4 + wv) * 16 + r) * 64 + lane] = dp[r]; }
      __syncthreads();
      __builtin_amdgcn_sched_barrier(0);
      // full S, dP (every wave), then P and dS in the accumulator layout: row q = rho(r,kk), col key = jl
      f32x16 pm, ds;
      const int key = k0 + jl;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sv = 0.f, dpv = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) { sv += X[((0 * 4 + u) * 16 + r) * 64 + lane]; dpv += X[((1 * 4 + u) * 16 + r) * 64 + lane]; }
        const int ql = rho(r, kk), q = q0 + ql;
        if (causal && key > q) sv -= 1e10f;
        sv *= inv_scale;
        const float p = (key < Tk && q < Tq) ? __expf(sv - lsel[ql]) : 0.f;
        const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
        pm[r] = p * mk;
        ds[r] = p * (dpv * mk - dl[ql]) * inv_scale;
      }
      if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dSl[rho(r, kk) * 66 + jl] = ds[r];
      }
      __syncthreads();

      // dV^T[c][key] += dO^T[c][q] P[q][key];  dK^T[c][key] += Q^T[c][q] dS[q][key]   (sum over the 32 queries)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        __builtin_amdgcn_sched_barrier(0);                 // one channel tile at a time (the A reads of the next stay behind)
        if (ct < vtiles) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dvt[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Gl[rho(r, kk) * VS2 + c0 + 32 * ct + jl], pm[r], dvt[ct], 0, 0, 0);
        }
        if (ct < ktiles) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dkt[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(Ql[rho(r, kk) * KS + c0 + 32 * ct + jl], ds[r], dkt[ct], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          // dQ[q][c] (+)= dS[q][key] K[key][c]: A = dS rows from LDS, B = K[key pair][my channel]
          f32x16 dq;
#pragma unroll
          for (int r = 0; r < 16; ++r) dq[r] = 0.f;
#pragma unroll
          for (int s = 0; s < 8; ++s) {
            const float2 a = *reinterpret_cast<const float2*>(dSl + jl * 66 + 4 * s + 2 * kk);
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, Kl[(4 * s + 2 * kk) * KS + c0 + 32 * ct + jl], dq, 0, 0, 0);
            dq = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, Kl[(4 * s + 2 * kk + 1) * KS + c0 + 32 * ct + jl], dq, 0, 0, 0);
          }
          if (c0 + 32 * ct + jl < dh) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int q = q0 + rho(r, kk);
              if (q < Tq) {
                float* dst = dQb + (size_t)q * dk + c0 + 32 * ct + jl;
                *dst = (k0 > 0 ? *dst : 0.f) + dq[r];     // same lane wrote it for the previous key block
              }
            }
          }
        }
      }
    }
    // dK / dV rows of this key block: accumulators are [channel rows][key on lane] -> transpose through LDS
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (ct < ktiles) Ql[jl * KS + c0 + 32 * ct + rho(r, kk)] = dkt[ct][r];
        if (ct < vtiles) Gl[jl * VS2 + c0 + 32 * ct + rho(r, kk)] = dvt[ct][r];
      }
    __syncthreads();
    for (int idx = tid; idx < 32 * (dh / 4); idx += kMfmaThreads) {
      const int r = idx / (dh / 4), c = (idx % (dh / 4)) * 4;
      if (k0 + r < Tk)
        *reinterpret_cast<float4*>(dKb + (size_t)(k0 + r) * dk + c) = make_float4(Ql[r * KS + c], Ql[r * KS + c + 1], Ql[r * KS + c + 2], Ql[r * KS + c + 3]);
    }
    for (int idx = tid; idx < 32 * (dvh / 4); idx += kMfmaThreads) {
      const int r = idx / (dvh / 4), c = (idx % (dvh / 4)) * 4;
      if (k0 + r < Tk)
        *reinterpret_cast<float4*>(dVb + (size_t)(k0 + r) * dv + c) = make_float4(Gl[r * VS2 + c], Gl[r * VS2 + c + 1], Gl[r * VS2 + c + 2], Gl[r * VS2 + c + 3]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// backward in SPLIT PRECISION (dtype TSG_F32S; head widths 32 / 64 / 96 / 128, d_key == d_value): the five products of the
// attention backward on the bf16 MFMA -- every fp32 operand x as hi = rne_bf16(x), lo = rne_bf16(x - hi), each product as
// hi*hi + hi*lo + lo*hi with fp32 accumulation (v_mfma_f32_32x32x16_bf16: 5.3x fewer matrix cycles than the exact-fp32
// 32x32x2 form; the arithmetic of the split-precision GEMMs and of the LSTM recurrence around it).  With the matrix time out
// of the way the kernel is organised around NOT moving data instead:
//   * TWO kernels, each the only writer of its outputs -- (A) dK, dV: workgroup = (b, head, 128 keys), wave = 32 keys;
//     (B) dQ: workgroup = (b, head, 128 queries), wave = 32 queries.  S and dP are recomputed in both (cheap now); there is
//     no global read-modify-write of dQ per (query tile, key block) and no cross-wave exchange of partial S / dP tiles.
//   * the wave's OWN rows (K, V in A; Q, dO in B) are B-operand fragments held in registers for the whole kernel; the rows it
//     streams past (Q, dO tiles in A; K, V tiles in B) are staged once per tile in LDS as bf16 planes in the two layouts the
//     products need: row-major (contraction over channels) and transposed (contraction over the tile's rows).
//   * P and dS never leave the registers: S is formed with the wave's rows on the LANES (A: S[q][key], B: S^T[key][q]), so
//     its accumulator registers -- 16 tile rows rho(r, hh) per lane -- are, packed to bf16 pairs, exactly the operand of the
//     next product whose contraction runs over the tile's rows in that same permuted order (the transposed planes are read
//     in the matching 4-row runs).
// LDS is double-buffered (the next tile's rows are requested before the MFMAs of the current one and split afterwards); one
// barrier per tile.  delta = <dO, O> comes from the same pre-pass as the fp32 kernels; dropout / causal masks as there.
// ------------------------------------------------------------------------------------------
typedef __bf16 bf16x8v __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2v __attribute__((ext_vector_type(2)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // (rne(a), rne(b)) packed, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2v){a, b}, bf16x2v));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(a, b);
  lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ void split8(const float (&v)[8], u32x4v& hi, u32x4v& lo) {
  unsigned h[4], l[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) split_pair(v[2 * i], v[2 * i + 1], h[i], l[i]);
  hi = (u32x4v){h[0], h[1], h[2], h[3]};
  lo = (u32x4v){l[0], l[1], l[2], l[3]};
}
__device__ __forceinline__ f32x16 mfma_bf(u32x4v a, u32x4v b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8v, a), __builtin_bit_cast(bf16x8v, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma3(u32x4v ah, u32x4v al, u32x4v bh, u32x4v bl, f32x16 c) {
  c = mfma_bf(ah, bh, c);
  c = mfma_bf(ah, bl, c);
  return mfma_bf(al, bh, c);
}

// One staged [32 rows x 32*DT channels] fp32 tile as bf16 planes in LDS.  Row-major planes R (pitch PR dwords per row: the
// b128 fragment reads of 16 lanes hit 16 distinct bank quads for DT = 1..4) and, when TR, transposed planes T (pitch 20 dwords
// per channel: 32 rows + 4).  256 threads: lane bits [1:0] = channel quad inside a 64-byte segment, [4:2] = group of 4 rows,
// the rest = further segments -- 64-byte global reads per lane quad and conflict-free transposing writes (csrc/wgrad_split.hip).
// ET = element type in HBM: float, or bf16_t (dtype TSG_BF16: the pieces stay raw in registers until they are staged, so the request
// still flies under the MFMAs; the lo planes are then all zero -- the arithmetic is exact for bf16-valued operands).
template <int DT, bool TR, bool ROW = true, typename ET = float>
struct SplitTile {
  static constexpr int PR = 16 * DT + 4, PT = 20;
  static constexpr int kRow = 32 * PR, kCol = 32 * DT * PT;                 // dwords per plane
  static constexpr int kT0 = ROW ? 2 * kRow : 0;                            // first dword of the transposed planes
  static constexpr int kDwords = kT0 + (TR ? 2 * kCol : 0);                 // [R hi][R lo][T hi][T lo]
  static constexpr int NR = DT > 4 ? 2 : 1;                                 // 128-channel column blocks per thread (head widths > 128)
  typename Raw4T<ET>::type v[NR][4];
  const ET* p0;                                                          // this thread's first row of tile 0
  int mg, c4, ld;
  bool on[NR];
  // src: the matrix (row stride ld_ elements) with the head's first channel already applied
  __device__ __forceinline__ SplitTile(const ET* __restrict__ src, int ld_) {
    const int tid = threadIdx.x;
    mg = (tid >> 2) & 7; c4 = (tid & 3) + 4 * (tid >> 5); ld = ld_;
#pragma unroll
    for (int k = 0; k < NR; ++k) on[k] = 4 * (c4 + 32 * k) < 32 * DT;
    p0 = src + (size_t)(4 * mg) * ld + 4 * c4;
  }
  // rows [row0, row0+32), zero fill beyond rows_total
  __device__ __forceinline__ void request(int row0, int rows_total) {
    const ET* p = p0 + (size_t)row0 * ld;
#pragma unroll
    for (int k = 0; k < NR; ++k)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        v[k][i] = (on[k] && row0 + 4 * mg + i < rows_total) ? ldraw4(p + i * ld + 128 * k) : zero_raw4(p);
  }
  __device__ __forceinline__ void stage(unsigned* __restrict__ base) const {
#pragma unroll
    for (int k = 0; k < NR; ++k) {
      if (!on[k]) continue;
      const int cq = c4 + 32 * k;
      const float4 f0 = cvt4(v[k][0]), f1 = cvt4(v[k][1]), f2 = cvt4(v[k][2]), f3 = cvt4(v[k][3]);
      const float e[4][4] = {{f0.x, f0.y, f0.z, f0.w}, {f1.x, f1.y, f1.z, f1.w}, {f2.x, f2.y, f2.z, f2.w}, {f3.x, f3.y, f3.z, f3.w}};
      if (ROW) {
        unsigned* Rh = base; unsigned* Rl = Rh + kRow;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned h0, l0, h1, l1;
          split_pair(e[i][0], e[i][1], h0, l0);
          split_pair(e[i][2], e[i][3], h1, l1);
          *reinterpret_cast<uint2*>(Rh + (4 * mg + i) * PR + 2 * cq) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Rl + (4 * mg + i) * PR + 2 * cq) = make_uint2(l0, l1);
        }
      }
      if (TR) {
        unsigned* Th = base + kT0; unsigned* Tl = Th + kCol;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          unsigned h0, l0, h1, l1;
          split_pair(e[0][j], e[1][j], h0, l0);
          split_pair(e[2][j], e[3][j], h1, l1);
          *reinterpret_cast<uint2*>(Th + (4 * cq + j) * PT + 2 * mg) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(Tl + (4 * cq + j) * PT + 2 * mg) = make_uint2(l0, l1);
        }
      }
    }
  }
  // A operand, rows = the tile's rows, contraction = channels [16 ks, 16 ks + 16)
  static __device__ __forceinline__ void row_frag(const unsigned* __restrict__ base, int ks, int jl, int hh, u32x4v& hi, u32x4v& lo) {
    hi = *reinterpret_cast<const u32x4v*>(base + jl * PR + 8 * ks + 4 * hh);
    lo = *reinterpret_cast<const u32x4v*>(base + kRow + jl * PR + 8 * ks + 4 * hh);
  }
  // operand whose contraction runs over the tile's rows in accumulator order: slot (hh, j) of step s <-> row rho(8 s + j, hh),
  // i.e. the 4-row runs 16 s + 4 hh .. +3 and 16 s + 8 + 4 hh .. +3; lane = channel 32 ct + jl
  static __device__ __forceinline__ void col_frag(const unsigned* __restrict__ base, int ct, int s, int jl, int hh, u32x4v& hi, u32x4v& lo) {
    const unsigned* Th = base + kT0 + (32 * ct + jl) * PT + 8 * s + 2 * hh;
    const uint2 a = *reinterpret_cast<const uint2*>(Th), b = *reinterpret_cast<const uint2*>(Th + 4);
    const uint2 c = *reinterpret_cast<const uint2*>(Th + kCol), d = *reinterpret_cast<const uint2*>(Th + kCol + 4);
    hi = (u32x4v){a.x, a.y, b.x, b.y};
    lo = (u32x4v){c.x, c.y, d.x, d.y};
  }
  // the same operand read out of the ROW-major planes (no transposed planes staged): eight 2-byte reads per plane, lane = channel ch
  static __device__ __forceinline__ void col_from_rows(const unsigned* __restrict__ base, int ch, int s, int hh, u32x4v& hi, u32x4v& lo) {
    const unsigned short* Rh = reinterpret_cast<const unsigned short*>(base) + ch;
    const unsigned short* Rl = Rh + 2 * kRow;
    unsigned h[4], l[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r0 = rho(8 * s + 2 * j, hh), r1 = rho(8 * s + 2 * j + 1, hh);
      h[j] = (unsigned)Rh[r0 * 2 * PR] | ((unsigned)Rh[r1 * 2 * PR] << 16);
      l[j] = (unsigned)Rl[r0 * 2 * PR] | ((unsigned)Rl[r1 * 2 * PR] << 16);
    }
    hi = (u32x4v){h[0], h[1], h[2], h[3]};
    lo = (u32x4v){l[0], l[1], l[2], l[3]};
  }
};

// The workgroup's own 128 rows (wave w: rows row0 + 32 w .. +31) as B-operand fragments over the channels -- lane (jl, hh) holds
// channels 16 ks + 8 hh .. +7 of row 32 w + jl for every step ks.  Every lane reads its 32-byte pieces itself; staging the rows
// through LDS as lane-quad segments instead (one L1 lookup per lane quad rather than per lane) measured the same (31.3 vs 31.1 us
// forward, 121.7 vs 123.3 us backward): this phase waits for HBM, not for the L1 pipe.
template <int DT, typename ET>
__device__ __forceinline__ void own_rows(const ET* __restrict__ src, int ld, int row0, int rows_total,
                                         u32x4v (&hi)[2 * DT], u32x4v (&lo)[2 * DT]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, jl = lane & 31, hh = lane >> 5;
  const int row = row0 + 32 * wv + jl;
  float4 x[2 * DT], y[2 * DT];
#pragma unroll
  for (int ks = 0; ks < 2 * DT; ++ks) {
    x[ks] = y[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows_total) {
      x[ks] = ld4(src + (size_t)row * ld + 16 * ks + 8 * hh);
      y[ks] = ld4(src + (size_t)row * ld + 16 * ks + 8 * hh + 4);
    }
  }
#pragma unroll
  for (int ks = 0; ks < 2 * DT; ++ks) {
    const float e[8] = {x[ks].x, x[ks].y, x[ks].z, x[ks].w, y[ks].x, y[ks].y, y[ks].z, y[ks].w};
    split8(e, hi[ks], lo[ks]);
  }
}

// own_rows of dO that also returns this lane's part of delta[row] = <dO[row], O[row]> over the head's channels (the lane's 8-channel
// pieces; lane ^ 32 holds the others): the delta pre-pass folded into the kernel that reads dO anyway.
template <int DT, typename ET>
__device__ __forceinline__ float own_rows_delta(const ET* __restrict__ src, const ET* __restrict__ osrc, int ld, int row0, int rows_total,
                                                u32x4v (&hi)[2 * DT], u32x4v (&lo)[2 * DT]) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, jl = lane & 31, hh = lane >> 5;
  const int row = row0 + 32 * wv + jl;
  float4 x[2 * DT], y[2 * DT];
  float acc = 0.f;
#pragma unroll
  for (int ks = 0; ks < 2 * DT; ++ks) {
    x[ks] = y[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows_total) {
      x[ks] = ld4(src + (size_t)row * ld + 16 * ks + 8 * hh);
      y[ks] = ld4(src + (size_t)row * ld + 16 * ks + 8 * hh + 4);
      const float4 ox = ld4(osrc + (size_t)row * ld + 16 * ks + 8 * hh);
      const float4 oy = ld4(osrc + (size_t)row * ld + 16 * ks + 8 * hh + 4);
      acc = fmaf(x[ks].x, ox.x, fmaf(x[ks].y, ox.y, fmaf(x[ks].z, ox.z, fmaf(x[ks].w, ox.w, acc))));
      acc = fmaf(y[ks].x, oy.x, fmaf(y[ks].y, oy.y, fmaf(y[ks].z, oy.z, fmaf(y[ks].w, oy.w, acc))));
    }
  }
#pragma unroll
  for (int ks = 0; ks < 2 * DT; ++ks) {
    const float e[8] = {x[ks].x, x[ks].y, x[ks].z, x[ks].w, y[ks].x, y[ks].y, y[ks].z, y[ks].w};
    split8(e, hi[ks], lo[ks]);
  }
  return acc;
}

// Operand whose contraction runs over the wave's own 32 rows in accumulator order (slot (hh, j) of step s <-> row rho(8 s + j, hh)),
// gathered from global memory: lane = channel c.  For a fixed j the 32 lanes of a half read 128 contiguous bytes of one row.
template <typename ET>
__device__ __forceinline__ void gather_frag(const ET* __restrict__ rows, int ld, int valid, int s, int hh, int c, u32x4v& hi, u32x4v& lo) {
  float e[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = rho(8 * s + j, hh);
    e[j] = r < valid ? ld1(rows + (size_t)r * ld + c) : 0.f;
  }
  split8(e, hi, lo);
}

// (A) dK, dV.  grid = B * H * ceil(Tk / 128); wave w owns keys kb + 32 w .. +31.
// WDS (round 6; one key block, Tk <= the head width): the dS tile this kernel forms anyway is also WRITTEN -- dS[b][q][head's columns: key] -- into
// the dQ buffer (a [Tq x Tk] block per head fits the head's [Tq x 32 DT] columns), and dQ = dS K becomes a small product over it
// (mha_bwd_dq_from_ds_kernel) instead of a second kernel that re-reads Q, K, V, dO and forms S, P, dP, dS again.
template <int DT, bool DROP, typename ET, bool WDS = false>
__global__ __launch_bounds__(256) void mha_bwd_split_dkv_kernel(
    const ET* __restrict__ Q, const ET* __restrict__ K, const ET* __restrict__ V, const ET* __restrict__ dO,
    const float* __restrict__ LSE, const float* __restrict__ delta, ET* __restrict__ dK, ET* __restrict__ dV, ET* __restrict__ dS,
    int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, DropCfg dc) {
  drop_resolve(dc);
#ifdef TSG_K2_TIMING
  unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tm0 = __builtin_amdgcn_s_memtime(), tm1 = 0;
#define K2_TICK(i) { tm1 = __builtin_amdgcn_s_memtime(); tph[i] += tm1 - tm0; tm0 = tm1; }
#else
#define K2_TICK(i) {}
#endif
  using ST = SplitTile<DT, true, true, ET>;
  extern __shared__ __align__(16) unsigned lds_u[];                      // [2 buffers][Q tile | dO tile] + lse / delta rows
  constexpr int kBuf = 2 * ST::kDwords;
  float* rows = reinterpret_cast<float*>(lds_u + 2 * kBuf);              // [2][2][32]: lse, delta of the tile's queries
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int kblocks = (Tk + 127) / 128;
  const int b = blockIdx.x / (H * kblocks), hd = (blockIdx.x / kblocks) % H, kb = (blockIdx.x % kblocks) * 128;
  const ET* Qb = Q + (size_t)b * Tq * dk + hd * 32 * DT;
  const ET* Gb = dO + (size_t)b * Tq * dk + hd * 32 * DT;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const float* dlt = delta + ((size_t)b * H + hd) * Tq;
  const int key = kb + 32 * wv + jl;                                     // this lane's key (the S / dP column)

  ST tq(Qb, dk), tg(Gb, dk);
  float lse_n = 0.f, dl_n = 0.f;
  auto request = [&](int q0) {
    tq.request(q0, Tq); tg.request(q0, Tq);
    if (tid < 32) { lse_n = q0 + tid < Tq ? lse[q0 + tid] : 0.f; dl_n = q0 + tid < Tq ? dlt[q0 + tid] : 0.f; }
  };
  auto stage = [&](int buf) {
    tq.stage(lds_u + buf * kBuf); tg.stage(lds_u + buf * kBuf + ST::kDwords);
    if (tid < 32) { rows[(buf * 2 + 0) * 32 + tid] = lse_n; rows[(buf * 2 + 1) * 32 + tid] = dl_n; }
  };
  request(0);                                                            // in flight together with the workgroup's own K / V rows
  u32x4v kh[2 * DT], kl[2 * DT], vh[2 * DT], vl[2 * DT];
  own_rows<DT>(K + (size_t)b * Tk * dk + hd * 32 * DT, dk, kb, Tk, kh, kl);
  own_rows<DT>(V + (size_t)b * Tk * dk + hd * 32 * DT, dk, kb, Tk, vh, vl);
  f32x16 dkt[DT], dvt[DT];
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[ct][r] = 0.f; dvt[ct][r] = 0.f; }
  stage(0);
  __syncthreads();
  K2_TICK(0)
  const int ntiles = (Tq + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1, q0 = 32 * t;
    request(min(q0 + 32, 32 * (ntiles - 1)));                            // the last iteration re-requests its own tile (unused)
    K2_TICK(1)
    const unsigned* Qt = lds_u + buf * kBuf; const unsigned* Gt = Qt + ST::kDwords;
    // S = Q K^T, dP = dO V^T over the head's channels: rows = the tile's queries, columns = this wave's keys
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2 * DT; ++ks) {
      u32x4v ah, al;
      ST::row_frag(Qt, ks, jl, hh, ah, al);
      sp = mfma3(ah, al, kh[ks], kl[ks], sp);
      ST::row_frag(Gt, ks, jl, hh, ah, al);
      dp = mfma3(ah, al, vh[ks], vl[ks], dp);
    }
    K2_TICK(2)
    // P (dropout-scaled) and dS in the accumulator layout: register r <-> query rho(r, hh)
    float pm[16], ds[16], lsev[16], dlv[16];
    const float* lsel = rows + (buf * 2 + 0) * 32; const float* dl = rows + (buf * 2 + 1) * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {                                        // rho(4 i + j, hh) = 8 i + 4 hh + j: four b128 reads each, one wait
      const float4 a = *reinterpret_cast<const float4*>(lsel + 8 * i + 4 * hh), c = *reinterpret_cast<const float4*>(dl + 8 * i + 4 * hh);
      lsev[4 * i] = a.x; lsev[4 * i + 1] = a.y; lsev[4 * i + 2] = a.z; lsev[4 * i + 3] = a.w;
      dlv[4 * i] = c.x; dlv[4 * i + 1] = c.y; dlv[4 * i + 2] = c.z; dlv[4 * i + 3] = c.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + rho(r, hh);
      float sv = sp[r];
      if (causal && key > q) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (key < Tk && q < Tq) ? __expf(sv - lsev[r]) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
      pm[r] = p * mk;
      ds[r] = p * (dp[r] * mk - dlv[r]) * inv_scale;
    }
    if constexpr (WDS) {                                                 // (kb = 0; keys beyond Tk carry p = 0: zeros are stored and contracted)
      if (32 * wv + jl < 32 * DT) {
        ET* dsb = dS + (size_t)b * Tq * dk + hd * 32 * DT + 32 * wv + jl;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int q = q0 + rho(r, hh);
          if (q < Tq) st1(dsb + (size_t)q * dk, ds[r]);
        }
      }
    }
    K2_TICK(3)
    // dV[key][c] += sum_q P[q][key] dO[q][c];  dK[key][c] += sum_q dS[q][key] Q[q][c]   (A = the registers above, B = transposed planes)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v ph, pl, sh, sl;
      const float pa[8] = {pm[8 * s], pm[8 * s + 1], pm[8 * s + 2], pm[8 * s + 3], pm[8 * s + 4], pm[8 * s + 5], pm[8 * s + 6], pm[8 * s + 7]};
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(pa, ph, pl);
      split8(sa, sh, sl);
#pragma unroll
      for (int ct = 0; ct < DT; ++ct) {
        u32x4v bh, bl;
        ST::col_frag(Gt, ct, s, jl, hh, bh, bl);
        dvt[ct] = mfma3(ph, pl, bh, bl, dvt[ct]);
        ST::col_frag(Qt, ct, s, jl, hh, bh, bl);
        dkt[ct] = mfma3(sh, sl, bh, bl, dkt[ct]);
      }
    }
    K2_TICK(4)
    stage(buf ^ 1);
    K2_TICK(5)
    __syncthreads();
    K2_TICK(6)
  }
  // accumulators: rows = keys rho(r, hh) of this wave, column = channel 32 ct + jl: 128-byte row segments
  ET* dKb = dK + (size_t)b * Tk * dk + hd * 32 * DT;
  ET* dVb = dV + (size_t)b * Tk * dk + hd * 32 * DT;
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = kb + 32 * wv + rho(r, hh);
      if (k < Tk) {
        st1(dKb + (size_t)k * dk + 32 * ct + jl, dkt[ct][r]);
        st1(dVb + (size_t)k * dk + 32 * ct + jl, dvt[ct][r]);
      }
    }
#ifdef TSG_K2_TIMING
  K2_TICK(7)
  __syncthreads();
  if (blockIdx.x == 0 && lane == 0)
    for (int i = 0; i < 8; ++i) dK[wv * 8 + i] = (float)tph[i];
#endif
}

// (B') dQ = dS K from the dS the WDS kernel left in the dQ buffer, IN PLACE: grid = B * H * ceil(Tq / 128); wave w owns queries qb + 32 w .. +31,
// reads its 32 dS rows (the head's columns) into A-operand fragments before anything is stored, streams the head's K rows through LDS as
// transposed bf16 planes (32 keys per tile) and writes its 32 dQ rows over the dS rows it read.  2 DT x DT x 3 MFMAs per tile and wave; no S, no
// softmax, no dP; reads dS + K, writes dQ.
template <int DT, typename ET>
__global__ __launch_bounds__(256) void mha_bwd_dq_from_ds_kernel(const ET* __restrict__ K, ET* __restrict__ dQ, int B, int Tq, int Tk, int dk, int H) {
  using SK = SplitTile<DT, true, false, ET>;                             // transposed planes only
  extern __shared__ __align__(16) unsigned lds_u[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int qblocks = (Tq + 127) / 128;
  const int b = blockIdx.x / (H * qblocks), hd = (blockIdx.x / qblocks) % H, qb = (blockIdx.x % qblocks) * 128;
  SK tk(K + (size_t)b * Tk * dk + hd * 32 * DT, dk);
  tk.request(0, Tk);
  ET* dQb = dQ + (size_t)b * Tq * dk + hd * 32 * DT;
  u32x4v sh[2 * DT], sl[2 * DT];
  own_rows<DT>(static_cast<const ET*>(dQb), dk, qb, Tq, sh, sl);          // dS[q][keys 16 ks + 8 hh .. +7], q = qb + 32 wv + jl
  f32x16 dqt[DT];
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[ct][r] = 0.f;
  tk.stage(lds_u);
  __syncthreads();
  const int ntiles = (Tk + 31) / 32;                                     // <= DT (host-checked: Tk <= 32 DT)
#pragma unroll
  for (int t = 0; t < DT; ++t) {                                         // (unrolled: 2 t + s selects among the register-resident dS fragments)
    if (t < ntiles) {                                                    // workgroup-uniform
      const int buf = t & 1;
      tk.request(min(32 * t + 32, 32 * (ntiles - 1)), Tk);
      const unsigned* Th = lds_u + buf * SK::kDwords;                    // [channel][32 keys as 16 dwords (+4)], hi plane; lo plane kCol dwords on
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ct = 0; ct < DT; ++ct) {
          const unsigned* pk = Th + (32 * ct + jl) * SK::PT + 8 * s + 4 * hh;      // keys 16 s + 8 hh .. +7 of the tile, channel 32 ct + jl
          const u32x4v bh = *reinterpret_cast<const u32x4v*>(pk), bl = *reinterpret_cast<const u32x4v*>(pk + SK::kCol);
          dqt[ct] = mfma3(sh[2 * t + s], sl[2 * t + s], bh, bl, dqt[ct]);
        }
      tk.stage(lds_u + (buf ^ 1) * SK::kDwords);
      __syncthreads();
    }
  }
  // accumulators: rows = queries rho(r, hh) of this wave, column = channel 32 ct + jl: 128-byte row segments
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = qb + 32 * wv + rho(r, hh);
      if (q < Tq) st1(dQb + (size_t)q * dk + 32 * ct + jl, dqt[ct][r]);
    }
}

// delta[b][h][q] = <dO[b,q,head], O[b,q,head]> for either storage type (one wave per (b, q) row)
template <typename ET>
__global__ __launch_bounds__(256) void mha_bwd_delta_t_kernel(const ET* __restrict__ O, const ET* __restrict__ dO, float* __restrict__ delta,
                                                              int B, int Tq, int dv, int H) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wv;
  if (row >= (long)B * Tq) return;
  const int b = (int)(row / Tq), q = (int)(row % Tq), dvh = dv / H;
  const ET* o = O + (size_t)row * dv; const ET* g = dO + (size_t)row * dv;
  for (int hd = 0; hd < H; ++hd) {
    float acc = 0.f;
    for (int c = lane * 4; c < dvh; c += 256) {
      const float4 x = ld4(o + hd * dvh + c), y = ld4(g + hd * dvh + c);
      acc = fmaf(x.x, y.x, fmaf(x.y, y.y, fmaf(x.z, y.z, fmaf(x.w, y.w, acc))));
    }
    acc = wave_allsum(acc);
    if (lane == 0) delta[((size_t)b * H + hd) * Tq + q] = acc;
  }
}

// (B) dQ.  grid = B * H * ceil(Tq / 128); wave w owns queries qb + 32 w .. +31.
template <int DT, bool DROP, typename ET>
__global__ __launch_bounds__(256) void mha_bwd_split_dq_kernel(
    const ET* __restrict__ Q, const ET* __restrict__ K, const ET* __restrict__ V, const ET* __restrict__ dO,
    const float* __restrict__ LSE, const ET* __restrict__ O, float* __restrict__ delta, ET* __restrict__ dQ,
    int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, DropCfg dc) {
  drop_resolve(dc);
  using SK = SplitTile<DT, true, true, ET>;                              // K tile: row-major + transposed planes
  using SV = SplitTile<DT, false, true, ET>;                             // V tile: row-major planes only
  extern __shared__ __align__(16) unsigned lds_u[];
  constexpr int kBuf = SK::kDwords + SV::kDwords;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int qblocks = (Tq + 127) / 128;
  const int b = blockIdx.x / (H * qblocks), hd = (blockIdx.x / qblocks) % H, qb = (blockIdx.x % qblocks) * 128;
  const ET* Kb = K + (size_t)b * Tk * dk + hd * 32 * DT;
  const ET* Vb = V + (size_t)b * Tk * dk + hd * 32 * DT;
  const int q = qb + 32 * wv + jl;                                       // this lane's query (the S^T / dP^T column)
  const float lse_q = q < Tq ? LSE[((size_t)b * H + hd) * Tq + q] : 0.f;

  SK tk(Kb, dk); SV tv(Vb, dk);
  auto request = [&](int k0) { tk.request(k0, Tk); tv.request(k0, Tk); };
  auto stage = [&](int buf) { tk.stage(lds_u + buf * kBuf); tv.stage(lds_u + buf * kBuf + SK::kDwords); };
  request(0);
  u32x4v qh[2 * DT], ql[2 * DT], gh[2 * DT], gl[2 * DT];
  own_rows<DT>(Q + (size_t)b * Tq * dk + hd * 32 * DT, dk, qb, Tq, qh, ql);
  // delta[q] = <dO[q], O[q]> is formed here, from the dO rows this wave reads anyway, and written for the dK/dV kernel that follows
  const float dl_q = xhalf_sum(own_rows_delta<DT>(dO + (size_t)b * Tq * dk + hd * 32 * DT, O + (size_t)b * Tq * dk + hd * 32 * DT, dk, qb, Tq, gh, gl));
  if (hh == 0 && q < Tq) delta[((size_t)b * H + hd) * Tq + q] = dl_q;
  f32x16 dqt[DT];
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[ct][r] = 0.f;
  stage(0);
  __syncthreads();
  const int ntiles = (Tk + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1, k0 = 32 * t;
    request(min(k0 + 32, 32 * (ntiles - 1)));
    const unsigned* Kt = lds_u + buf * kBuf; const unsigned* Vt = Kt + SK::kDwords;
    // S^T = K Q^T, dP^T = V dO^T: rows = the tile's keys, columns = this wave's queries
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2 * DT; ++ks) {
      u32x4v ah, al;
      SK::row_frag(Kt, ks, jl, hh, ah, al);
      sp = mfma3(ah, al, qh[ks], ql[ks], sp);
      SV::row_frag(Vt, ks, jl, hh, ah, al);
      dp = mfma3(ah, al, gh[ks], gl[ks], dp);
    }
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + rho(r, hh);
      float sv = sp[r];
      if (causal && key > q) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (key < Tk && q < Tq) ? __expf(sv - lse_q) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
      ds[r] = p * (dp[r] * mk - dl_q) * inv_scale;
    }
    // dQ^T[c][q] += sum_key K^T[c][key] dS^T[key][q]   (A = transposed K planes, B = the registers above)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v sh, sl;
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(sa, sh, sl);
#pragma unroll
      for (int ct = 0; ct < DT; ++ct) {
        u32x4v ah, al;
        SK::col_frag(Kt, ct, s, jl, hh, ah, al);
        dqt[ct] = mfma3(ah, al, sh, sl, dqt[ct]);
      }
    }
    stage(buf ^ 1);
    __syncthreads();
  }
  // dQ^T accumulators (rows = channels 32 ct + rho(r, hh), column = this lane's query) -> the wave's [32][dh] block in LDS -> rows out
  constexpr int OP = 32 * DT + 4;                                        // floats per query row
  float* Ol = reinterpret_cast<float*>(lds_u) + wv * 32 * OP;
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ol[jl * OP + 32 * ct + rho(r, hh)] = dqt[ct][r];
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");                 // the block is private to this wave
  __builtin_amdgcn_wave_barrier();
  ET* dQb = dQ + (size_t)b * Tq * dk + hd * 32 * DT;
  for (int idx = lane; idx < 32 * 8 * DT; idx += 64) {
    const int r = idx / (8 * DT), c = (idx % (8 * DT)) * 4;
    const int qq = qb + 32 * wv + r;
    if (qq < Tq) st4(dQb + (size_t)qq * dk + c, *reinterpret_cast<const float4*>(Ol + r * OP + c));
  }
}

// ---- wide heads (160 .. 256 channels) in the backward: at width 256 a 32-row wave's own rows are 128 registers per tensor, and two
// such tensors plus the accumulators do not fit (the kernels above with the accumulators taken in passes spilled 80-380 registers).
// Here TWO waves share a 32-row group and split the CHANNELS: wave (g, c) holds channels [128 c, 128 c + 128) of its rows (64
// registers per tensor), forms the partial S / dP over them, the two partial tiles meet in LDS (one exchange per tile, both waves
// then hold the full tile and compute P / dS redundantly), and each accumulates its own channel half of dV / dK (dQ).  Workgroup =
// 64 rows (2 groups x 2 halves).  The operand contracted over the tile's rows is read out of the ROW-major planes with eight 2-byte
// LDS reads per plane (gathering the fp32 rows from global memory and splitting them again per use measured 511 instead of
// 360 us), so the tiles are staged as row-major planes only: one LDS buffer (the next tile waits in registers) + the exchange
// block = 100 KiB.
template <int NKS>
__device__ __forceinline__ void own_rows_half(const float* __restrict__ src, int ld, int row, int rows_total, int nks, int hh,
                                              u32x4v (&hi)[NKS], u32x4v (&lo)[NKS]) {
  float4 x[NKS], y[NKS];
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    x[ks] = y[ks] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < rows_total && ks < nks) {
      x[ks] = *reinterpret_cast<const float4*>(src + (size_t)row * ld + 16 * ks + 8 * hh);
      y[ks] = *reinterpret_cast<const float4*>(src + (size_t)row * ld + 16 * ks + 8 * hh + 4);
    }
  }
#pragma unroll
  for (int ks = 0; ks < NKS; ++ks) {
    const float e[8] = {x[ks].x, x[ks].y, x[ks].z, x[ks].w, y[ks].x, y[ks].y, y[ks].z, y[ks].w};
    split8(e, hi[ks], lo[ks]);
  }
}

// partial S / dP tiles of the two channel halves -> the full tiles in both waves of the pair.  X: [4 waves][2][16][64] floats.
__device__ __forceinline__ void exchange_halves(float* __restrict__ X, int wv, int lane, f32x16& sp, f32x16& dp) {
#pragma unroll
  for (int r = 0; r < 16; ++r) { X[((wv * 2 + 0) * 16 + r) * 64 + lane] = sp[r]; X[((wv * 2 + 1) * 16 + r) * 64 + lane] = dp[r]; }
  __syncthreads();
  const int pw = wv ^ 1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { sp[r] += X[((pw * 2 + 0) * 16 + r) * 64 + lane]; dp[r] += X[((pw * 2 + 1) * 16 + r) * 64 + lane]; }
}

template <int DT, bool DROP>
__global__ __launch_bounds__(256) void mha_bwd_split_dq_wide_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO,
    const float* __restrict__ LSE, const float* __restrict__ O, float* __restrict__ delta, float* __restrict__ dQ,
    int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, DropCfg dc) {
  using ET = float;                                                      // the wide-head backward is fp32 storage only
  drop_resolve(dc);
  using SR = SplitTile<DT, false>;
  extern __shared__ __align__(16) unsigned lds_u[];
  float* X = reinterpret_cast<float*>(lds_u + 2 * SR::kDwords);          // exchange block behind the K | V tile
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int g = wv >> 1, c = wv & 1;                                     // 32-query group, channel half
  const int nt = c ? DT - 4 : 4, nks = 2 * nt, ks0 = 8 * c;             // channel tiles / contraction steps of this half
  const int qblocks = (Tq + 63) / 64;
  const int b = blockIdx.x / (H * qblocks), hd = (blockIdx.x / qblocks) % H, qb = (blockIdx.x % qblocks) * 64;
  const ET* Kb = K + (size_t)b * Tk * dk + hd * 32 * DT;
  const ET* Vb = V + (size_t)b * Tk * dk + hd * 32 * DT;
  const int q = qb + 32 * g + jl;
  const float lse_q = q < Tq ? LSE[((size_t)b * H + hd) * Tq + q] : 0.f;
  SR tk(Kb, dk), tv(Vb, dk);
  tk.request(0, Tk); tv.request(0, Tk);
  u32x4v qh[8], ql[8], gh[8], gl[8];
  const float* Qr = Q + (size_t)b * Tq * dk + hd * 32 * DT + 128 * c;
  const float* Gr = dO + (size_t)b * Tq * dk + hd * 32 * DT + 128 * c;
  const float* Or = O + (size_t)b * Tq * dk + hd * 32 * DT + 128 * c;
  own_rows_half<8>(Qr, dk, q, Tq, nks, hh, qh, ql);
  own_rows_half<8>(Gr, dk, q, Tq, nks, hh, gh, gl);
  // delta[q] = <dO[q], O[q]> over ALL channels: this lane's pieces of its half, + lane ^ 32, + the partner wave's half (through X)
  float dpart = 0.f;
  if (q < Tq) {
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < nks) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const float4 a = *reinterpret_cast<const float4*>(Gr + (size_t)q * dk + 16 * ks + 8 * hh + 4 * u);
          const float4 o = *reinterpret_cast<const float4*>(Or + (size_t)q * dk + 16 * ks + 8 * hh + 4 * u);
          dpart = fmaf(a.x, o.x, fmaf(a.y, o.y, fmaf(a.z, o.z, fmaf(a.w, o.w, dpart))));
        }
      }
  }
  dpart = xhalf_sum(dpart);
  if (hh == 0) X[wv * 32 + jl] = dpart;
  tk.stage(lds_u); tv.stage(lds_u + SR::kDwords);
  __syncthreads();
  const float dl_q = dpart + X[(wv ^ 1) * 32 + jl];
  if (c == 0 && hh == 0 && q < Tq) delta[((size_t)b * H + hd) * Tq + q] = dl_q;
  __syncthreads();                                                       // X is reused by the tile loop
  f32x16 dqt[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) dqt[ct][r] = 0.f;
  const unsigned* Kt = lds_u; const unsigned* Vt = lds_u + SR::kDwords;
  const int ntiles = (Tk + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = 32 * t;
    const int kn = min(k0 + 32, 32 * (ntiles - 1));
    tk.request(kn, Tk); tv.request(kn, Tk);
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < nks) {
        u32x4v ah, al;
        SR::row_frag(Kt, ks0 + ks, jl, hh, ah, al);
        sp = mfma3(ah, al, qh[ks], ql[ks], sp);
        SR::row_frag(Vt, ks0 + ks, jl, hh, ah, al);
        dp = mfma3(ah, al, gh[ks], gl[ks], dp);
      }
    exchange_halves(X, wv, lane, sp, dp);
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + rho(r, hh);
      float sv = sp[r];
      if (causal && key > q) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (key < Tk && q < Tq) ? __expf(sv - lse_q) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
      ds[r] = p * (dp[r] * mk - dl_q) * inv_scale;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v sh, sl;
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(sa, sh, sl);
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct < nt) {
          u32x4v ah, al;                                                 // K^T[c][key]: the key tile's rows at this lane's channel
          SR::col_from_rows(Kt, 128 * c + 32 * ct + jl, s, hh, ah, al);
          dqt[ct] = mfma3(ah, al, sh, sl, dqt[ct]);
        }
    }
    __syncthreads();                                                     // the tile and the exchange block have been read
    tk.stage(lds_u); tv.stage(lds_u + SR::kDwords);
    __syncthreads();
  }
  constexpr int OP = 132;                                                // floats per query row of a wave's [32][128] block
  float* Ol = reinterpret_cast<float*>(lds_u) + wv * 32 * OP;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
    if (ct < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) Ol[jl * OP + 32 * ct + rho(r, hh)] = dqt[ct][r];
    }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* dQb = dQ + (size_t)b * Tq * dk + hd * 32 * DT + 128 * c;
  for (int idx = lane; idx < 32 * 8 * nt; idx += 64) {
    const int r = idx / (8 * nt), cc = (idx % (8 * nt)) * 4;
    const int qq = qb + 32 * g + r;
    if (qq < Tq) *reinterpret_cast<float4*>(dQb + (size_t)qq * dk + cc) = *reinterpret_cast<const float4*>(Ol + r * OP + cc);
  }
}

template <int DT, bool DROP>
__global__ __launch_bounds__(256) void mha_bwd_split_dkv_wide_kernel(
    const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V, const float* __restrict__ dO,
    const float* __restrict__ LSE, const float* __restrict__ delta, float* __restrict__ dK, float* __restrict__ dV,
    int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, DropCfg dc) {
  using ET = float;
  drop_resolve(dc);
  using SR = SplitTile<DT, false>;
  extern __shared__ __align__(16) unsigned lds_u[];
  float* X = reinterpret_cast<float*>(lds_u + 2 * SR::kDwords);
  float* rows = X + 4 * 2 * 16 * 64;                                     // [2][32]: lse, delta of the tile's queries
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int g = wv >> 1, c = wv & 1;
  const int nt = c ? DT - 4 : 4, nks = 2 * nt, ks0 = 8 * c;
  const int kblocks = (Tk + 63) / 64;
  const int b = blockIdx.x / (H * kblocks), hd = (blockIdx.x / kblocks) % H, kb = (blockIdx.x % kblocks) * 64;
  const ET* Qb = Q + (size_t)b * Tq * dk + hd * 32 * DT;
  const ET* Gb = dO + (size_t)b * Tq * dk + hd * 32 * DT;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const float* dlt = delta + ((size_t)b * H + hd) * Tq;
  const int key = kb + 32 * g + jl;
  SR tq(Qb, dk), tg(Gb, dk);
  float lse_n = 0.f, dl_n = 0.f;
  auto request = [&](int q0) {
    tq.request(q0, Tq); tg.request(q0, Tq);
    if (tid < 32) { lse_n = q0 + tid < Tq ? lse[q0 + tid] : 0.f; dl_n = q0 + tid < Tq ? dlt[q0 + tid] : 0.f; }
  };
  auto stage = [&]() {
    tq.stage(lds_u); tg.stage(lds_u + SR::kDwords);
    if (tid < 32) { rows[tid] = lse_n; rows[32 + tid] = dl_n; }
  };
  request(0);
  u32x4v kh[8], kl[8], vh[8], vl[8];
  own_rows_half<8>(K + (size_t)b * Tk * dk + hd * 32 * DT + 128 * c, dk, key, Tk, nks, hh, kh, kl);
  own_rows_half<8>(V + (size_t)b * Tk * dk + hd * 32 * DT + 128 * c, dk, key, Tk, nks, hh, vh, vl);
  f32x16 dkt[4], dvt[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dkt[ct][r] = 0.f; dvt[ct][r] = 0.f; }
  stage();
  __syncthreads();
  const unsigned* Qt = lds_u; const unsigned* Gt = lds_u + SR::kDwords;
  const int ntiles = (Tq + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int q0 = 32 * t;
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
      if (ks < nks) {
        u32x4v ah, al;
        SR::row_frag(Qt, ks0 + ks, jl, hh, ah, al);
        sp = mfma3(ah, al, kh[ks], kl[ks], sp);
        SR::row_frag(Gt, ks0 + ks, jl, hh, ah, al);
        dp = mfma3(ah, al, vh[ks], vl[ks], dp);
      }
    exchange_halves(X, wv, lane, sp, dp);
    float pm[16], ds[16], lsev[16], dlv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 a = *reinterpret_cast<const float4*>(rows + 8 * i + 4 * hh), d4 = *reinterpret_cast<const float4*>(rows + 32 + 8 * i + 4 * hh);
      lsev[4 * i] = a.x; lsev[4 * i + 1] = a.y; lsev[4 * i + 2] = a.z; lsev[4 * i + 3] = a.w;
      dlv[4 * i] = d4.x; dlv[4 * i + 1] = d4.y; dlv[4 * i + 2] = d4.z; dlv[4 * i + 3] = d4.w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = q0 + rho(r, hh);
      float sv = sp[r];
      if (causal && key > q) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (key < Tk && q < Tq) ? __expf(sv - lsev[r]) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, key) : 1.f;
      pm[r] = p * mk;
      ds[r] = p * (dp[r] * mk - dlv[r]) * inv_scale;
    }
    u32x4v ph[2], pl[2], sh[2], sl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float pa[8] = {pm[8 * s], pm[8 * s + 1], pm[8 * s + 2], pm[8 * s + 3], pm[8 * s + 4], pm[8 * s + 5], pm[8 * s + 6], pm[8 * s + 7]};
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(pa, ph[s], pl[s]);
      split8(sa, sh[s], sl[s]);
    }
    __builtin_amdgcn_sched_barrier(0);                                   // the next tile's rows are requested only now: while S / dP and
    request(min(q0 + 32, 32 * (ntiles - 1)));                            // the softmax were live every register was taken
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (ct < nt) {
          u32x4v bh, bl;
          SR::col_from_rows(Gt, 128 * c + 32 * ct + jl, s, hh, bh, bl);
          dvt[ct] = mfma3(ph[s], pl[s], bh, bl, dvt[ct]);
          SR::col_from_rows(Qt, 128 * c + 32 * ct + jl, s, hh, bh, bl);
          dkt[ct] = mfma3(sh[s], sl[s], bh, bl, dkt[ct]);
        }
    __syncthreads();
    stage();
    __syncthreads();
  }
  float* dKb = dK + (size_t)b * Tk * dk + hd * 32 * DT + 128 * c;
  float* dVb = dV + (size_t)b * Tk * dk + hd * 32 * DT + 128 * c;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct)
    if (ct < nt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = kb + 32 * g + rho(r, hh);
        if (k < Tk) {
          dKb[(size_t)k * dk + 32 * ct + jl] = dkt[ct][r];
          dVb[(size_t)k * dk + 32 * ct + jl] = dvt[ct][r];
        }
      }
    }
}

template <int DT, bool DROP>
int launch_bwd_split_wide(const char* fn, const float* Q, const float* K, const float* V, const float* O, const float* dO, const float* lse,
                          float* delta, float* dQ, float* dK, float* dV, int B, int Tq, int Tk, int dk, int H, float inv_scale,
                          int causal, const DropCfg& dc, hipStream_t st) {
  using SR = SplitTile<DT, false>;
  size_t lds = sizeof(unsigned) * (size_t)(2 * SR::kDwords) + sizeof(float) * (4 * 2 * 16 * 64 + 64);
  const size_t outb = sizeof(float) * 4 * 32 * 132;
  if (outb > lds) lds = outb;
  auto ka = mha_bwd_split_dkv_wide_kernel<DT, DROP>;
  auto kq = mha_bwd_split_dq_wide_kernel<DT, DROP>;
  hipError_t e = allow_lds(ka, lds);                       // (device, kernel)-keyed table: cheap when already set
  if (e == hipSuccess) e = allow_lds(kq, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  hipLaunchKernelGGL(kq, dim3(B * H * cdiv(Tq, 64)), dim3(256), lds, st, Q, K, V, dO, lse, O, delta, dQ, B, Tq, Tk, dk, H, inv_scale, causal, dc);
  int rc = check_launch(fn);
  if (rc) return rc;
  hipLaunchKernelGGL(ka, dim3(B * H * cdiv(Tk, 64)), dim3(256), lds, st, Q, K, V, dO, lse, (const float*)delta, dK, dV, B, Tq, Tk, dk, H, inv_scale, causal, dc);
  return check_launch(fn);
}

// Forward in split precision (dtype TSG_F32S, no A_forward side outputs; same shapes as the backward): the dQ kernel's structure with
// an online softmax.  Workgroup = (b, head, 128 queries), wave = 32 queries held as B-operand fragments; per 32-key tile
// S^T = K Q^T (rows = keys in the accumulator registers, column = the lane's query), running max / sum per query (16 registers
// + one exchange with lane ^ 32, which holds the other 16 keys of the same query), O^T[c][q] += V^T[c][key] P^T[key][q] with the
// P^T registers as the B operand and the transposed V planes as A.  O leaves through LDS as whole rows.
template <int DT, bool DROP, typename ET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(DT <= 4 ? 2 : 1, DT <= 4 ? 2 : 1))) void mha_fwd_split_kernel(
    const ET* __restrict__ Q, const ET* __restrict__ K, const ET* __restrict__ V, ET* __restrict__ O,
    float* __restrict__ LSE, int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, DropCfg dc) {
  drop_resolve(dc);
  using SK = SplitTile<DT, false, true, ET>;                             // K tile: row-major planes
  using SV = SplitTile<DT, true, false, ET>;                             // V tile: transposed planes only
  extern __shared__ __align__(16) unsigned lds_u[];
  constexpr int kBuf = SK::kDwords + SV::kDwords;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int qblocks = (Tq + 127) / 128;
  const int b = blockIdx.x / (H * qblocks), hd = (blockIdx.x / qblocks) % H, qb = (blockIdx.x % qblocks) * 128;
  const int q = qb + 32 * wv + jl;
  SK tk(K + (size_t)b * Tk * dk + hd * 32 * DT, dk); SV tv(V + (size_t)b * Tk * dk + hd * 32 * DT, dk);
  auto request = [&](int k0) { tk.request(k0, Tk); tv.request(k0, Tk); };
  auto stage = [&](int buf) { tk.stage(lds_u + buf * kBuf); tv.stage(lds_u + buf * kBuf + SK::kDwords); };
  request(0);
  u32x4v qh[2 * DT], ql[2 * DT];
  own_rows<DT>(Q + (size_t)b * Tq * dk + hd * 32 * DT, dk, qb, Tq, qh, ql);
  f32x16 ot[DT];
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[ct][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;                                  // l_run: this lane's 16 keys per tile only (pair-summed at the end)
  stage(0);
  __syncthreads();
  const int ntiles = (Tk + 31) / 32;
  for (int t = 0; t < ntiles; ++t) {
    const int buf = t & 1, k0 = 32 * t;
    request(min(k0 + 32, 32 * (ntiles - 1)));
    const unsigned* Kt = lds_u + buf * kBuf; const unsigned* Vt = Kt + SK::kDwords;
    f32x16 sp;
#pragma unroll
    for (int r = 0; r < 16; ++r) sp[r] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 2 * DT; ++ks) {
      u32x4v ah, al;
      SK::row_frag(Kt, ks, jl, hh, ah, al);
      sp = mfma3(ah, al, qh[ks], ql[ks], sp);
    }
    float sv[16], mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + rho(r, hh);
      float x = sp[r];
      if (causal && key > q) x -= 1e10f;
      x = key < Tk ? x * inv_scale : -INFINITY;
      sv[r] = x;
      mx = fmaxf(mx, x);
    }
    mx = xhalf_max(mx);                                                  // both lanes of a query agree on the tile maximum
    const float m_new = fmaxf(m_run, mx);
    const float alpha = __expf(m_run - m_new);                           // 0 on the first tile (m_run = -inf, m_new finite: key k0 is valid)
    m_run = m_new;
    float pm[16], ls = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float p = __expf(sv[r] - m_new);
      ls += p;
      pm[r] = DROP ? p * drop_scale(dc, b, H, hd, Tq, q, Tk, k0 + rho(r, hh)) : p;
    }
    l_run = l_run * alpha + ls;
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[ct][r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v ph, pl;
      const float pa[8] = {pm[8 * s], pm[8 * s + 1], pm[8 * s + 2], pm[8 * s + 3], pm[8 * s + 4], pm[8 * s + 5], pm[8 * s + 6], pm[8 * s + 7]};
      split8(pa, ph, pl);
#pragma unroll
      for (int ct = 0; ct < DT; ++ct) {
        u32x4v ah, al;
        SV::col_frag(Vt, ct, s, jl, hh, ah, al);
        ot[ct] = mfma3(ah, al, ph, pl, ot[ct]);
      }
    }
    stage(buf ^ 1);
    __syncthreads();
  }
  const float l_tot = xhalf_sum(l_run);
  const float inv_l = 1.f / l_tot;
  if (hh == 0 && q < Tq) LSE[((size_t)b * H + hd) * Tq + q] = m_run + __logf(l_tot);
  constexpr int OP = 32 * DT + 4;
  float* Ol = reinterpret_cast<float*>(lds_u) + wv * 32 * OP;
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ol[jl * OP + 32 * ct + rho(r, hh)] = ot[ct][r] * inv_l;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  ET* Ob = O + (size_t)b * Tq * dk + hd * 32 * DT;
  for (int idx = lane; idx < 32 * 8 * DT; idx += 64) {
    const int r = idx / (8 * DT), c = (idx % (8 * DT)) * 4;
    const int qq = qb + 32 * wv + r;
    if (qq < Tq) st4(Ob + (size_t)qq * dk + c, *reinterpret_cast<const float4*>(Ol + r * OP + c));
  }
}

template <int DT, bool DROP, typename ET = float>
int launch_fwd_split(const char* fn, const ET* Q, const ET* K, const ET* V, ET* O, float* lse, int B, int Tq, int Tk, int dk,
                     int H, float inv_scale, int causal, const DropCfg& dc, hipStream_t st) {
  using SK = SplitTile<DT, false>;
  using SV = SplitTile<DT, true, false>;
  size_t lds = sizeof(unsigned) * (size_t)(2 * (SK::kDwords + SV::kDwords));
  const size_t outb = sizeof(float) * 4 * 32 * (32 * DT + 4);
  if (outb > lds) lds = outb;
  auto kf = mha_fwd_split_kernel<DT, DROP, ET>;
  hipError_t e = allow_lds(kf, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  hipLaunchKernelGGL(kf, dim3(B * H * cdiv(Tq, 128)), dim3(256), lds, st, Q, K, V, O, lse, B, Tq, Tk, dk, H, inv_scale, causal, dc);
  return check_launch(fn);
}

// (C) backward for ONE key tile (Tk <= 32: cross attention over the T_word <= 32 words of a query) -- one kernel, every input read
// once from HBM.  Workgroup = (b, head, 128 queries), wave = 32 queries whose Q / dO rows are register fragments; the K / V tile is
// staged once.  The wave forms S and dP twice, in both orientations (the matrix work is negligible: 168 MFMAs per wave):
//   * with the queries on the lanes (S^T = K Q^T): dS^T registers are the B operand of dQ^T = K^T dS^T -> dQ through LDS;
//   * with the keys on the lanes (S = Q K^T): P / dS registers are the A operand of the wave's PARTIAL dV = P^T dO, dK = dS^T Q over
//     its 32 queries, whose other operand is gathered from global memory row by row (128-byte segments, no transposed staging).
// The four waves' partial dK / dV tiles are added through LDS; with more than one query block per head they are added to the
// (zero-filled) outputs with float atomics, with one block (Tq <= 128) they are stored.  Two workgroups per CU.
template <int DT, bool DROP, typename ET>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void mha_bwd_split_cross_kernel(
    const ET* __restrict__ Q, const ET* __restrict__ K, const ET* __restrict__ V, const ET* __restrict__ dO,
    const float* __restrict__ LSE, const ET* __restrict__ O, ET* __restrict__ dQ, ET* __restrict__ dK,
    ET* __restrict__ dV, int B, int Tq, int Tk, int dk, int H, float inv_scale, int causal, int atomic_out, DropCfg dc) {
  drop_resolve(dc);
  using SK = SplitTile<DT, true, true, ET>;                              // K tile: row-major + transposed planes
  using SV = SplitTile<DT, false, true, ET>;                             // V tile: row-major planes
  extern __shared__ __align__(16) unsigned lds_u[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, jl = lane & 31, hh = lane >> 5;
  const int qblocks = (Tq + 127) / 128;
  const int b = blockIdx.x / (H * qblocks), hd = (blockIdx.x / qblocks) % H, qb = (blockIdx.x % qblocks) * 128;
  const ET* Qb = Q + (size_t)b * Tq * dk + hd * 32 * DT;
  const ET* Gb = dO + (size_t)b * Tq * dk + hd * 32 * DT;
  const float* lse = LSE + ((size_t)b * H + hd) * Tq;
  const int qw0 = qb + 32 * wv, q = qw0 + jl;                            // the wave's first query; this lane's query / key
  const int key = jl;
  {
    SK tk(K + (size_t)b * Tk * dk + hd * 32 * DT, dk); SV tv(V + (size_t)b * Tk * dk + hd * 32 * DT, dk);
    tk.request(0, Tk); tv.request(0, Tk);
    tk.stage(lds_u); tv.stage(lds_u + SK::kDwords);
  }
  u32x4v qh[2 * DT], ql[2 * DT], gh[2 * DT], gl[2 * DT];
  own_rows<DT>(Qb, dk, qb, Tq, qh, ql);
  // delta[q] = <dO[q], O[q]> of this lane's query: the two lanes of a query hold disjoint channel pieces
  const float dl_q = xhalf_sum(own_rows_delta<DT>(Gb, O + (size_t)b * Tq * dk + hd * 32 * DT, dk, qb, Tq, gh, gl));
  float* dls = reinterpret_cast<float*>(lds_u + SK::kDwords + SV::kDwords);      // [128]: delta of the workgroup's queries
  if (hh == 0) dls[32 * wv + jl] = dl_q;
  __syncthreads();
  const unsigned* Kt = lds_u; const unsigned* Vt = lds_u + SK::kDwords;

  // ---- keys on the lanes: S = Q K^T, dP = dO V^T (rows = the wave's queries rho(r, hh)) -> P, dS packed as A operands
  u32x4v ph[2], pl[2], sh[2], sl[2];
  {
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2 * DT; ++ks) {
      u32x4v bh, bl;
      SK::row_frag(Kt, ks, jl, hh, bh, bl);
      sp = mfma3(qh[ks], ql[ks], bh, bl, sp);
      SV::row_frag(Vt, ks, jl, hh, bh, bl);
      dp = mfma3(gh[ks], gl[ks], bh, bl, dp);
    }
    float lsev[16], dlv[16];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float4 dd = *reinterpret_cast<const float4*>(dls + 32 * wv + 8 * i + 4 * hh);      // rho(4 i + j, hh) = 8 i + 4 hh + j
      dlv[4 * i] = dd.x; dlv[4 * i + 1] = dd.y; dlv[4 * i + 2] = dd.z; dlv[4 * i + 3] = dd.w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int qq = qw0 + 8 * i + 4 * hh + j;
        lsev[4 * i + j] = qq < Tq ? lse[qq] : 0.f;
      }
    }
    float pm[16], ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qq = qw0 + rho(r, hh);
      float sv = sp[r];
      if (causal && key > qq) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (key < Tk && qq < Tq) ? __expf(sv - lsev[r]) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, qq, Tk, key) : 1.f;
      pm[r] = p * mk;
      ds[r] = p * (dp[r] * mk - dlv[r]) * inv_scale;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float pa[8] = {pm[8 * s], pm[8 * s + 1], pm[8 * s + 2], pm[8 * s + 3], pm[8 * s + 4], pm[8 * s + 5], pm[8 * s + 6], pm[8 * s + 7]};
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(pa, ph[s], pl[s]);
      split8(sa, sh[s], sl[s]);
    }
  }
  // ---- queries on the lanes: S^T = K Q^T, dP^T = V dO^T (rows = keys rho(r, hh)) -> dS^T -> dQ^T = K^T dS^T
  f32x16 dqt[DT];
  {
    f32x16 sp, dp;
#pragma unroll
    for (int r = 0; r < 16; ++r) { sp[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
    for (int ks = 0; ks < 2 * DT; ++ks) {
      u32x4v ah, al;
      SK::row_frag(Kt, ks, jl, hh, ah, al);
      sp = mfma3(ah, al, qh[ks], ql[ks], sp);
      SV::row_frag(Vt, ks, jl, hh, ah, al);
      dp = mfma3(ah, al, gh[ks], gl[ks], dp);
    }
    const float lse_q = q < Tq ? lse[q] : 0.f;
    float ds[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = rho(r, hh);
      float sv = sp[r];
      if (causal && kk > q) sv -= 1e10f;
      sv *= inv_scale;
      const float p = (kk < Tk && q < Tq) ? __expf(sv - lse_q) : 0.f;
      const float mk = DROP ? drop_scale(dc, b, H, hd, Tq, q, Tk, kk) : 1.f;
      ds[r] = p * (dp[r] * mk - dl_q) * inv_scale;
    }
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) dqt[ct][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      u32x4v th, tl;
      const float sa[8] = {ds[8 * s], ds[8 * s + 1], ds[8 * s + 2], ds[8 * s + 3], ds[8 * s + 4], ds[8 * s + 5], ds[8 * s + 6], ds[8 * s + 7]};
      split8(sa, th, tl);
#pragma unroll
      for (int ct = 0; ct < DT; ++ct) {
        u32x4v ah, al;
        SK::col_frag(Kt, ct, s, jl, hh, ah, al);
        dqt[ct] = mfma3(ah, al, th, tl, dqt[ct]);
      }
    }
  }
  __syncthreads();                                                       // every wave is done with the K / V planes: the LDS is reused
  constexpr int OP = 32 * DT + 4;                                        // floats per row of a wave's [32][dh] block
  float* Ol = reinterpret_cast<float*>(lds_u) + wv * 32 * OP;
  auto wave_sync = [&]() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); };
  // dQ: rows = channels rho(r, hh) + 32 ct, column = this lane's query -> the wave's block -> whole rows out
#pragma unroll
  for (int ct = 0; ct < DT; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ol[jl * OP + 32 * ct + rho(r, hh)] = dqt[ct][r];
  wave_sync();
  ET* dQb = dQ + (size_t)b * Tq * dk + hd * 32 * DT;
  for (int idx = lane; idx < 32 * 8 * DT; idx += 64) {
    const int r = idx / (8 * DT), c = (idx % (8 * DT)) * 4;
    if (qw0 + r < Tq) st4(dQb + (size_t)(qw0 + r) * dk + c, *reinterpret_cast<const float4*>(Ol + r * OP + c));
  }
  wave_sync();
  // partial dV / dK of this wave's 32 queries: A = P / dS registers (row = key on the lane), B gathered from the wave's dO / Q rows;
  // accumulator: rows = keys rho(r, hh), column = channel 32 ct + jl.  Folded over the four waves through LDS.
  const int valid = Tq - qw0;                                            // rows of this wave that exist (may be <= 0)
  ET* dst[2] = {dV + (size_t)b * Tk * dk + hd * 32 * DT, dK + (size_t)b * Tk * dk + hd * 32 * DT};
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const ET* rows = (which == 0 ? Gb : Qb) + (size_t)qw0 * dk;
    f32x16 acc[DT];
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < DT; ++ct) {
        u32x4v bh, bl;
        gather_frag(rows, dk, valid, s, hh, 32 * ct + jl, bh, bl);
        acc[ct] = which == 0 ? mfma3(ph[s], pl[s], bh, bl, acc[ct]) : mfma3(sh[s], sl[s], bh, bl, acc[ct]);
      }
#pragma unroll
    for (int ct = 0; ct < DT; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ol[rho(r, hh) * OP + 32 * ct + jl] = acc[ct][r];
    __syncthreads();
    const float* all = reinterpret_cast<const float*>(lds_u);
    for (int idx = tid; idx < 32 * 8 * DT; idx += 256) {
      const int r = idx / (8 * DT), c = (idx % (8 * DT)) * 4;
      if (r < Tk) {
        float4 t = *reinterpret_cast<const float4*>(all + r * OP + c);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 u = *reinterpret_cast<const float4*>(all + (w * 32 + r) * OP + c);
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        ET* o = dst[which] + (size_t)r * dk + c;
        if constexpr (storage_is_bf16<ET>::value) {
          st4(o, t);                                       // bf16 outputs: one query block per head only (host-checked): no atomics
        } else {
          if (atomic_out) { atomicAdd(o, t.x); atomicAdd(o + 1, t.y); atomicAdd(o + 2, t.z); atomicAdd(o + 3, t.w); }
          else *reinterpret_cast<float4*>(o) = t;
        }
      }
    }
    __syncthreads();
  }
}

template <int DT, bool DROP, typename ET = float>
int launch_bwd_split(const char* fn, const ET* Q, const ET* K, const ET* V, const ET* O, const ET* dO, const float* lse,
                     float* delta, ET* dQ, ET* dK, ET* dV, int B, int Tq, int Tk, int dk, int H, float inv_scale,
                     int causal, const DropCfg& dc, hipStream_t st) {
  using ST = SplitTile<DT, true>;
  using SV = SplitTile<DT, false>;
  // one key tile: the fused kernel (bf16 outputs cannot take the float atomics several query blocks add with: those run the pair)
  if (Tk <= 32 && (!storage_is_bf16<ET>::value || Tq <= 128)) {
    size_t lds_c = sizeof(unsigned) * (size_t)(ST::kDwords + SV::kDwords) + sizeof(float) * 128;
    const size_t out_c = sizeof(float) * 4 * 32 * (32 * DT + 4);
    if (out_c > lds_c) lds_c = out_c;
    const int qblocks = cdiv(Tq, 128);
    auto kc = mha_bwd_split_cross_kernel<DT, DROP, ET>;
    hipError_t e = allow_lds(kc, lds_c);          // table hit (no runtime call) after the first, eager, launch on this device
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    if (qblocks > 1) {                                                     // several query blocks add into dK / dV
      const size_t nb = sizeof(float) * (size_t)B * Tk * dk;
      e = zero_async(dK, nb, st);
      if (e == hipSuccess) e = zero_async(dV, nb, st);
      if (e != hipSuccess) return set_error((int)e, "%s: zero fill: %s", fn, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kc, dim3(B * H * qblocks), dim3(256), lds_c, st, Q, K, V, dO, lse, O, dQ, dK, dV, B, Tq, Tk, dk, H, inv_scale,
                       causal, qblocks > 1 ? 1 : 0, dc);
    return check_launch(fn);
  }
  const size_t lds_a = sizeof(unsigned) * (size_t)(2 * 2 * ST::kDwords) + sizeof(float) * 128;
  // one key block no wider than the head (self-attention over T <= 128 clips at head width 128: the north-star shape): dS is written once and
  // dQ = dS K is a small product over it -- Q, K, V, dO are read ONCE (round-5 review item 6).  TSG_K2_BWD=pair keeps the two full kernels (A/B timing).
  static const bool pair_only = [] { const char* e = getenv("TSG_K2_BWD"); return e && e[0] == 'p'; }();
  if (Tk <= 128 && Tk <= 32 * DT && !pair_only) {
    using SKT = SplitTile<DT, true, false, ET>;
    const size_t lds_q = sizeof(unsigned) * (size_t)(2 * SKT::kDwords);
    auto kd = mha_bwd_delta_t_kernel<ET>;
    auto kw = mha_bwd_split_dkv_kernel<DT, DROP, ET, true>;
    auto kx = mha_bwd_dq_from_ds_kernel<DT, ET>;
    hipError_t e = allow_lds(kw, lds_a);
    if (e == hipSuccess) e = allow_lds(kx, lds_q);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kd, dim3(cdiv(B * Tq, 4)), dim3(256), 0, st, O, dO, delta, B, Tq, dk, H);
    hipLaunchKernelGGL(kw, dim3(B * H), dim3(256), lds_a, st, Q, K, V, dO, lse, (const float*)delta, dK, dV, dQ, B, Tq, Tk, dk, H, inv_scale, causal, dc);
    int rc = check_launch(fn);
    if (rc) return rc;
    hipLaunchKernelGGL(kx, dim3(B * H * cdiv(Tq, 128)), dim3(256), lds_q, st, K, dQ, B, Tq, Tk, dk, H);
    return check_launch(fn);
  }
  size_t lds_b = sizeof(unsigned) * (size_t)(2 * (ST::kDwords + SV::kDwords));
  const size_t out_b = sizeof(float) * 4 * 32 * (32 * DT + 4);
  if (out_b > lds_b) lds_b = out_b;
  auto ka = mha_bwd_split_dkv_kernel<DT, DROP, ET>;
  auto kq = mha_bwd_split_dq_kernel<DT, DROP, ET>;
  hipError_t e = allow_lds(ka, lds_a);
  if (e == hipSuccess) e = allow_lds(kq, lds_b);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  // dQ first: it also writes delta = <dO, O>, which the dK/dV kernel reads
  hipLaunchKernelGGL(kq, dim3(B * H * cdiv(Tq, 128)), dim3(256), lds_b, st, Q, K, V, dO, lse, O, delta, dQ, B, Tq, Tk, dk, H, inv_scale, causal, dc);
  int rc = check_launch(fn);
  if (rc) return rc;
  hipLaunchKernelGGL(ka, dim3(B * H * cdiv(Tk, 128)), dim3(256), lds_a, st, Q, K, V, dO, lse, (const float*)delta, dK, dV, (ET*)nullptr, B, Tq, Tk, dk, H, inv_scale, causal, dc);
  return check_launch(fn);
}

int check(const char* fn, int B, int Tq, int Tk, int dk, int dv, int H, int dtype) {
  if (dtype != TSG_F32 && dtype != TSG_F32S && dtype != TSG_BF16)
    return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (TSG_F32, TSG_F32S or TSG_BF16)", fn, dtype);
  if (B <= 0 || Tq <= 0 || Tk <= 0 || dk <= 0 || dv <= 0 || H <= 0)
    return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d Tq=%d Tk=%d dk=%d dv=%d heads=%d", fn, B, Tq, Tk, dk, dv, H);
  if (dk % H || dv % H) return set_error(TSG_E_SHAPE, "%s: d_key=%d / d_value=%d not divisible by n_heads=%d", fn, dk, dv, H);
  if ((dk / H) % 4 || (dv / H) % 4)
    return set_error(TSG_E_ALIGN, "%s: head widths %d / %d must be multiples of 4", fn, dk / H, dv / H);
  if (dv / H > MAXVC * CC) return set_error(TSG_E_SHAPE, "%s: value head width %d > %d not supported", fn, dv / H, MAXVC * CC);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

static int make_drop(const char* fn, float p_drop, uint64_t seed, uint64_t offset, const void* rng_dev, tsg::DropCfg* dc) {
  dc->rng = static_cast<const unsigned long long*>(rng_dev);
  if (!(p_drop >= 0.f) || p_drop >= 1.f) return tsg::set_error(TSG_E_SHAPE, "%s: dropout probability %g outside [0, 1)", fn, p_drop);
  const double t = (double)p_drop * 4294967296.0;
  dc->thresh = p_drop > 0.f ? (unsigned)(t < 1.0 ? 1.0 : (t > 4294967295.0 ? 4294967295.0 : t)) : 0u;
  dc->inv_keep = 1.f / (1.f - p_drop);
  dc->k0 = (unsigned)seed ^ ((unsigned)offset * 0x9E3779B1u);
  dc->k1 = (unsigned)(seed >> 32) ^ ((unsigned)(offset >> 32) * 0x85EBCA77u + 0x165667B1u);
  return 0;
}

static int mha_fwd_impl(const void* Q, const void* K, const void* V, void* O, void* A_sum, void* S_sum, void* lse,
                        int B, int Tq, int Tk, int d_key, int d_value, int n_heads, float scale, int causal,
                        float p_drop, uint64_t seed, uint64_t offset, const void* rng_dev, int dtype, void* stream) {
  const char* fn = "tsg_mha_fwd";
  for (const void* p : {Q, K, V, (const void*)O, (const void*)lse}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = check(fn, B, Tq, Tk, d_key, d_value, n_heads, dtype);
  if (rc) return rc;
  DropCfg dc;
  rc = make_drop(fn, p_drop, seed, offset, rng_dev, &dc);
  if (rc) return rc;
  if (!(scale > 0.f)) return set_error(TSG_E_SHAPE, "%s: scale must be positive", fn);
  if (dtype == TSG_BF16 && (A_sum || S_sum || d_key != d_value || (d_key / n_heads) % 32 || d_key / n_heads > 256))     // before any launch
    return set_error(TSG_E_SHAPE, "%s: dtype TSG_BF16 needs d_key == d_value, head widths 32 .. 256 in steps of 32 and no A_sum / S_sum "
                     "outputs (head width %d / %d)", fn, d_key / n_heads, d_value / n_heads);
  auto st = static_cast<hipStream_t>(stream);
  const size_t map_bytes = sizeof(float) * (size_t)B * Tq * Tk;
  if (A_sum) { hipError_t e = zero_async(A_sum, map_bytes, st); if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e)); }
  if (S_sum) { hipError_t e = zero_async(S_sum, map_bytes, st); if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e)); }
  const int dh = d_key / n_heads, dvh = d_value / n_heads;
  static int split_on = -1;                                         // TSG_MHA_SPLIT=0: exact-fp32 kernels also for TSG_F32S (A/B)
  if (split_on < 0) { const char* e = getenv("TSG_MHA_SPLIT"); split_on = e ? atoi(e) : 1; }
  if (dtype == TSG_BF16) {                       // bf16 storage of Q, K, V, O: the split kernels with 2-byte elements (lo planes zero)
    if (A_sum || S_sum || dh != dvh || dh % 32 || dh > 256)
      return set_error(TSG_E_SHAPE, "%s: dtype TSG_BF16 needs d_key == d_value, head widths 32 .. 256 in steps of 32 and no A_sum / S_sum "
                       "outputs (head width %d / %d)", fn, dh, dvh);
    const bf16_t* q = (const bf16_t*)Q; const bf16_t* k = (const bf16_t*)K; const bf16_t* v = (const bf16_t*)V;
    const float is = 1.f / scale;
#define TSG_SPLIT_CASE(DT) \
    return dc.thresh ? launch_fwd_split<DT, true, bf16_t>(fn, q, k, v, (bf16_t*)O, (float*)lse, B, Tq, Tk, d_key, n_heads, is, causal, dc, st) \
                     : launch_fwd_split<DT, false, bf16_t>(fn, q, k, v, (bf16_t*)O, (float*)lse, B, Tq, Tk, d_key, n_heads, is, causal, dc, st)
    switch (dh / 32) {
      case 1: TSG_SPLIT_CASE(1);
      case 2: TSG_SPLIT_CASE(2);
      case 3: TSG_SPLIT_CASE(3);
      case 4: TSG_SPLIT_CASE(4);
      case 5: TSG_SPLIT_CASE(5);
      case 6: TSG_SPLIT_CASE(6);
      case 7: TSG_SPLIT_CASE(7);
      default: TSG_SPLIT_CASE(8);
    }
#undef TSG_SPLIT_CASE
  }
  if (dtype == TSG_F32S && split_on && !A_sum && !S_sum && dh == dvh && dh % 32 == 0 && dh <= 256) {     // split precision
    const float* q = (const float*)Q; const float* k = (const float*)K; const float* v = (const float*)V;
    const float is = 1.f / scale;
#define TSG_SPLIT_CASE(DT) \
    return dc.thresh ? launch_fwd_split<DT, true>(fn, q, k, v, (float*)O, (float*)lse, B, Tq, Tk, d_key, n_heads, is, causal, dc, st) \
                     : launch_fwd_split<DT, false>(fn, q, k, v, (float*)O, (float*)lse, B, Tq, Tk, d_key, n_heads, is, causal, dc, st)
    switch (dh / 32) {
      case 1: TSG_SPLIT_CASE(1);
      case 2: TSG_SPLIT_CASE(2);
      case 3: TSG_SPLIT_CASE(3);
      case 4: TSG_SPLIT_CASE(4);
      case 5: TSG_SPLIT_CASE(5);
      case 6: TSG_SPLIT_CASE(6);
      case 7: TSG_SPLIT_CASE(7);
      default: TSG_SPLIT_CASE(8);
    }
#undef TSG_SPLIT_CASE
  }
  if (!A_sum && !S_sum && dh <= DHMAX && dvh <= DHMAX) {            // MFMA path
    const int KS = roundup(dh, 64) + 2, VS = roundup(dvh, 32) + 4, qblocks = cdiv(Tq, QB);
    size_t lds = sizeof(float) * (size_t)32 * (KS + VS);
    const size_t qstage = sizeof(float) * (size_t)QB * KS, ostage = sizeof(float) * (size_t)QB * VS;
    if (qstage > lds) lds = qstage;
    if (ostage > lds) lds = ostage;
    const bool full = dh == DHMAX && dvh == DHMAX;
    auto kern = dc.thresh ? (full ? mha_fwd_mfma_kernel<true, true> : mha_fwd_mfma_kernel<true, false>)
                          : (full ? mha_fwd_mfma_kernel<false, true> : mha_fwd_mfma_kernel<false, false>);
    if (lds > 64 * 1024) {
      hipError_t e = allow_lds(kern, lds);
      if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B * qblocks * n_heads), dim3(kMfmaThreads), lds, st, (const float*)Q, (const float*)K,
                       (const float*)V, (float*)O, (float*)lse, B, Tq, Tk, d_key, d_value, n_heads, 1.f / scale, causal,
                       qblocks, KS, VS, dc);
    return check_launch(fn);
  }
  if (!A_sum && !S_sum && dh <= DHW && dvh <= DHW) {                // MFMA path for wide heads (channels split over the waves)
    const int KS = DHW + 2, VS = DHW + 4, qblocks = cdiv(Tq, QBW);
    size_t lds = sizeof(float) * ((size_t)32 * (KS + VS) + 2 * 4 * 16 * 64);
    const size_t qstage = sizeof(float) * (size_t)QBW * KS, ostage = sizeof(float) * (size_t)QBW * VS;
    if (qstage > lds) lds = qstage;
    if (ostage > lds) lds = ostage;
    auto kern = dc.thresh ? mha_fwd_mfma_wide_kernel<true> : mha_fwd_mfma_wide_kernel<false>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * qblocks * n_heads), dim3(kWideThreads), lds, st, (const float*)Q, (const float*)K,
                       (const float*)V, (float*)O, (float*)lse, B, Tq, Tk, d_key, d_value, n_heads, 1.f / scale, causal,
                       qblocks, KS, VS, dc);
    return check_launch(fn);
  }
  const int qtiles = cdiv(Tq, TQ);
  hipLaunchKernelGGL(mha_fwd_kernel, dim3(B * qtiles), dim3(kThreads), 0, st, (const float*)Q, (const float*)K,
                     (const float*)V, (float*)O, (float*)A_sum, (float*)S_sum, (float*)lse, B, Tq, Tk, d_key, d_value,
                     n_heads, 1.f / scale, causal, qtiles, dc);
  return check_launch(fn);
}

extern "C" int tsg_mha_fwd(const void* Q, const void* K, const void* V, void* O, void* A_sum, void* S_sum, void* lse,
                           int B, int Tq, int Tk, int d_key, int d_value, int n_heads, float scale, int causal,
                           float p_drop, uint64_t seed, uint64_t offset, int dtype, void* stream) {
  return mha_fwd_impl(Q, K, V, O, A_sum, S_sum, lse, B, Tq, Tk, d_key, d_value, n_heads, scale, causal, p_drop, seed, offset, nullptr, dtype, stream);
}
extern "C" int tsg_mha_fwd_rng(const void* Q, const void* K, const void* V, void* O, void* A_sum, void* S_sum, void* lse,
                               int B, int Tq, int Tk, int d_key, int d_value, int n_heads, float scale, int causal,
                               float p_drop, const void* rng_dev, int dtype, void* stream) {
  if (p_drop > 0.f && !rng_dev) return set_error(TSG_E_NULL, "tsg_mha_fwd_rng: rng_dev is NULL");
  return mha_fwd_impl(Q, K, V, O, A_sum, S_sum, lse, B, Tq, Tk, d_key, d_value, n_heads, scale, causal, p_drop, 0, 0, rng_dev, dtype, stream);
}

static int mha_bwd_impl(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* lse,
                        void* dQ, void* dK, void* dV, void* delta_ws, int B, int Tq, int Tk, int d_key, int d_value,
                        int n_heads, float scale, int causal, float p_drop, uint64_t seed, uint64_t offset, const void* rng_dev,
                        int dtype, void* stream) {
  const char* fn = "tsg_mha_bwd";
  for (const void* p : {Q, K, V, O, dO, lse, (const void*)dQ, (const void*)dK, (const void*)dV}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  int rc = check(fn, B, Tq, Tk, d_key, d_value, n_heads, dtype);
  if (rc) return rc;
  DropCfg dc;
  rc = make_drop(fn, p_drop, seed, offset, rng_dev, &dc);
  if (rc) return rc;
  const int dh = d_key / n_heads, dvh = d_value / n_heads;
  static int split_on = -1;                                         // TSG_MHA_SPLIT=0: exact-fp32 kernels also for TSG_F32S (A/B)
  if (split_on < 0) { const char* e = getenv("TSG_MHA_SPLIT"); split_on = e ? atoi(e) : 1; }
  if (dtype == TSG_BF16) {                       // bf16 storage of Q, K, V, O, dO, dQ, dK, dV (head widths up to 128)
    if (!delta_ws || dh != dvh || dh % 32 || dh > 128)
      return set_error(TSG_E_SHAPE, "%s: dtype TSG_BF16 needs delta_ws, d_key == d_value and head widths 32 .. 128 in steps of 32 "
                       "(head width %d / %d)", fn, dh, dvh);
    auto st = static_cast<hipStream_t>(stream);
    typedef bf16_t E;
    const float is = 1.f / scale;
#define TSG_SPLIT_CASE(DT) \
    return dc.thresh ? launch_bwd_split<DT, true, E>(fn, (const E*)Q, (const E*)K, (const E*)V, (const E*)O, (const E*)dO, (const float*)lse, (float*)delta_ws, \
                                                     (E*)dQ, (E*)dK, (E*)dV, B, Tq, Tk, d_key, n_heads, is, causal, dc, st) \
                     : launch_bwd_split<DT, false, E>(fn, (const E*)Q, (const E*)K, (const E*)V, (const E*)O, (const E*)dO, (const float*)lse, (float*)delta_ws, \
                                                      (E*)dQ, (E*)dK, (E*)dV, B, Tq, Tk, d_key, n_heads, is, causal, dc, st)
    switch (dh / 32) {
      case 1: TSG_SPLIT_CASE(1);
      case 2: TSG_SPLIT_CASE(2);
      case 3: TSG_SPLIT_CASE(3);
      default: TSG_SPLIT_CASE(4);
    }
#undef TSG_SPLIT_CASE
  }
  // split-precision path (one key tile: the fused kernel, else the dK/dV + dQ pair)
  if (dtype == TSG_F32S && split_on && delta_ws && dh == dvh && dh % 32 == 0 && dh <= 128) {
    auto st = static_cast<hipStream_t>(stream);
    const float* q = (const float*)Q; const float* k = (const float*)K; const float* v = (const float*)V; const float* g = (const float*)dO;
    const float* o = (const float*)O; const float* l = (const float*)lse; float* dl = (float*)delta_ws;
    float* dq = (float*)dQ; float* dk_ = (float*)dK; float* dv_ = (float*)dV;
    const float is = 1.f / scale;
#define TSG_SPLIT_CASE(DT) \
    return dc.thresh ? launch_bwd_split<DT, true>(fn, q, k, v, o, g, l, dl, dq, dk_, dv_, B, Tq, Tk, d_key, n_heads, is, causal, dc, st) \
                     : launch_bwd_split<DT, false>(fn, q, k, v, o, g, l, dl, dq, dk_, dv_, B, Tq, Tk, d_key, n_heads, is, causal, dc, st)
    switch (dh / 32) {
      case 1: TSG_SPLIT_CASE(1);
      case 2: TSG_SPLIT_CASE(2);
      case 3: TSG_SPLIT_CASE(3);
      default: TSG_SPLIT_CASE(4);
    }
#undef TSG_SPLIT_CASE
  }
  if (dtype == TSG_F32S && split_on && delta_ws && dh == dvh && dh % 32 == 0 && dh > 128 && dh <= 256) {    // wide heads: channel halves per wave pair
    auto st = static_cast<hipStream_t>(stream);
    const float* q = (const float*)Q; const float* k = (const float*)K; const float* v = (const float*)V; const float* g = (const float*)dO;
    const float* o = (const float*)O; const float* l = (const float*)lse; float* dl = (float*)delta_ws;
    float* dq = (float*)dQ; float* dk_ = (float*)dK; float* dv_ = (float*)dV;
    const float is = 1.f / scale;
#define TSG_SPLIT_CASE(DT) \
    return dc.thresh ? launch_bwd_split_wide<DT, true>(fn, q, k, v, o, g, l, dl, dq, dk_, dv_, B, Tq, Tk, d_key, n_heads, is, causal, dc, st) \
                     : launch_bwd_split_wide<DT, false>(fn, q, k, v, o, g, l, dl, dq, dk_, dv_, B, Tq, Tk, d_key, n_heads, is, causal, dc, st)
    switch (dh / 32) {
      case 5: TSG_SPLIT_CASE(5);
      case 6: TSG_SPLIT_CASE(6);
      case 7: TSG_SPLIT_CASE(7);
      default: TSG_SPLIT_CASE(8);
    }
#undef TSG_SPLIT_CASE
  }
  if (delta_ws && dh <= DHMAX && dvh <= DHMAX) {                    // MFMA path
    auto st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mha_bwd_delta_kernel, dim3(cdiv(B * Tq, 4)), dim3(256), 0, st, (const float*)O, (const float*)dO,
                       (float*)delta_ws, B, Tq, d_value, n_heads);
    rc = check_launch(fn);
    if (rc) return rc;
    const int KS = roundup(dh, 64) + 2, VS2 = roundup(dvh, 64) + 2;
    const size_t lds = sizeof(float) * ((size_t)64 * (KS + VS2) + 2 * 4 * 16 * 64 + 32 * 66 + 64);
    const bool full = dh == DHMAX && dvh == DHMAX;
    auto kern = dc.thresh ? (full ? mha_bwd_mfma_kernel<true, true> : mha_bwd_mfma_kernel<true, false>)
                          : (full ? mha_bwd_mfma_kernel<false, true> : mha_bwd_mfma_kernel<false, false>);
    if (lds > 64 * 1024) {
      hipError_t e = allow_lds(kern, lds);
      if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(kMfmaThreads), lds, st, (const float*)Q, (const float*)K, (const float*)V,
                       (const float*)dO, (const float*)lse, (const float*)delta_ws, (float*)dQ, (float*)dK, (float*)dV,
                       B, Tq, Tk, d_key, d_value, n_heads, 1.f / scale, causal, KS, VS2, dc);
    return check_launch(fn);
  }
  if (delta_ws && dh <= DHW && dvh <= DHW) {                        // MFMA path for wide heads
    auto st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(mha_bwd_delta_kernel, dim3(cdiv(B * Tq, 4)), dim3(256), 0, st, (const float*)O, (const float*)dO,
                       (float*)delta_ws, B, Tq, d_value, n_heads);
    rc = check_launch(fn);
    if (rc) return rc;
    const int KS = DHW + 2, VS2 = DHW + 2;
    const size_t lds = sizeof(float) * ((size_t)32 * (2 * KS + VS2) + 2 * 4 * 16 * 64 + 32 * 66 + 64);
    auto kern = dc.thresh ? mha_bwd_mfma_wide_kernel<true> : mha_bwd_mfma_wide_kernel<false>;
    hipError_t e = allow_lds(kern, lds);
    if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(B * n_heads), dim3(kMfmaThreads), lds, st, (const float*)Q, (const float*)K, (const float*)V,
                       (const float*)dO, (const float*)lse, (const float*)delta_ws, (float*)dQ, (float*)dK, (float*)dV,
                       B, Tq, Tk, d_key, d_value, n_heads, 1.f / scale, causal, KS, VS2, dc);
    return check_launch(fn);
  }
  hipLaunchKernelGGL(mha_bwd_kernel, dim3(B * n_heads), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     (const float*)Q, (const float*)K, (const float*)V, (const float*)O, (const float*)dO,
                     (const float*)lse, (float*)dQ, (float*)dK, (float*)dV, B, Tq, Tk, d_key, d_value, n_heads,
                     1.f / scale, causal, dc);
  return check_launch(fn);
}

extern "C" int tsg_mha_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* lse,
                           void* dQ, void* dK, void* dV, void* delta_ws, int B, int Tq, int Tk, int d_key, int d_value,
                           int n_heads, float scale, int causal, float p_drop, uint64_t seed, uint64_t offset, int dtype,
                           void* stream) {
  return mha_bwd_impl(Q, K, V, O, dO, lse, dQ, dK, dV, delta_ws, B, Tq, Tk, d_key, d_value, n_heads, scale, causal, p_drop, seed, offset,
                      nullptr, dtype, stream);
}
extern "C" int tsg_mha_bwd_rng(const void* Q, const void* K, const void* V, const void* O, const void* dO, const void* lse,
                               void* dQ, void* dK, void* dV, void* delta_ws, int B, int Tq, int Tk, int d_key, int d_value,
                               int n_heads, float scale, int causal, float p_drop, const void* rng_dev, int dtype, void* stream) {
  if (p_drop > 0.f && !rng_dev) return set_error(TSG_E_NULL, "tsg_mha_bwd_rng: rng_dev is NULL");
  return mha_bwd_impl(Q, K, V, O, dO, lse, dQ, dK, dV, delta_ws, B, Tq, Tk, d_key, d_value, n_heads, scale, causal, p_drop, 0, 0, rng_dev,
                      dtype, stream);
}
