// Split-precision projection GEMM with the operands converted ON LOAD ("f32s" arithmetic, no operand planes in memory):
//     Y[M,N] = X[M,K] W[N,K]^T (+ bias[N])                       fp32 in, fp32 out, both operands contraction-contiguous
// i.e. torch.nn.Linear's layout -- the forward of the path's Linears and of nn.LSTM's input projection (reference:
// networks/attention.py:104-106,112-113; components/VideoEncoder.py:59; networks/RNN.py:31,42), and their input gradients
// dX = dY W with the weight passed transposed.  Every operand element x is split into hi = rne_bf16(x), lo = rne_bf16(x - hi) in
// registers and  hi*hi + hi*lo + lo*hi  is accumulated in fp32 on v_mfma_f32_32x32x16_bf16: the arithmetic of
// tsg_split_bf16x3 + a bf16 GEMM over the 3x longer contraction (functional._mm), without the split passes (fp32 read, three bf16
// planes written, read again) in front of it.
//
// Tiling.  Workgroup = 256 x 256 output tile, 8 waves as 2 (M) x 4 (N), each wave 128 x 64 = 4 x 2 MFMA tiles (128 accumulator
// registers); persistent workgroups walk the tiles (grid = min(tiles, CUs)).  K advances in chunks of 32 fp32 columns.  A chunk's X
// and W row tiles (256 rows x 128 bytes each) are requested into registers (8 float4 per thread) one chunk ahead, split ONCE by
// the thread that loaded them, and written to LDS as four bf16 planes (X hi, X lo, W hi, W lo; 64 bytes per row; 64 KiB per chunk,
// double-buffered): a fragment is then one ds_read_b128 per plane and costs no conversion at the eight waves that share it.
// (First version: LDS-DMA of the raw fp32 tiles and a split per fragment read -- every X fragment was converted by four waves, every
// W fragment by two: 116 us of the 523 us launch at [16384 x 1024] x [4096 x 1024] was conversion VALU.)
// LDS image: the four 16-byte pieces of a plane row are XOR-swizzled by (row >> 2) & 3, on the write and on the read: a fragment
// read (32 rows x the same piece) touches 16 distinct bank quads per 16 lanes (SQ_LDS_BANK_CONFLICT = 0).
// One LDS-only barrier per chunk; behind it the registers (chunk ks + 1) are converted into the other buffer and chunk ks + 2 is
// requested; the steady-state loop body is one basic block (the tail chunks are instantiated separately), the next X fragment is
// read while the current one is multiplied.
// Measured (tools/gemm_f32s_time.py, one MI355X; "library" = tsg_split_bf16x3 of both operands + hipBLASLt's bf16 GEMM over 3K):
//   [16384 x 1024] x [1024 x 1024]   92-97 us vs 115-117      [16384 x 1024] x [2048 x 1024]  179-186 vs 198-205
//   [16384 x 2048] x [ 512 x 2048]  133-146 vs 171-174        [16384 x 4096] x [1024 x 4096]  380-420 vs 425-450
//   [16384 x 1024] x [4096 x 1024]  450-455 vs 436-454 (tie)  [ 2560 x 1024] x [1024 x 1024]   63-72 vs 41 (40 tiles: the library path stays)
// = 0.9-1.15 PFLOP/s of bf16 matrix work; PMC: the matrix pipe is busy 54 % of the launch (the clock under this load is 1.62 GHz, so
// the sustained peak is 1.7 PFLOP/s), the waves wait on their global requests 12-14 % of it (one chunk = 1.6 us of look-ahead; a
// second register set spills 35 registers, an L2 touch-prefetch by inline asm cost more than it hid), LDS waits 5 %.
// Requirements: M % 256 == 0, N % 256 == 0, K % 32 == 0, 16-byte aligned rows (the host code falls back to the planes + library
// path otherwise).
#include "tsg_common.h"
#include <cstdlib>
#include <type_traits>

// ---- Round 4: one kernel template, three epilogues -------------------------------------------------------------------------------
//   * the M tile is a template parameter (TM = 256 / 128 / 64 rows; the N tile stays 256 columns, 8 waves as 2 (M) x 4 (N)): a head GEMM
//     with few output columns ([8192 x 1024] x [512 x 1024]^T: 64 tiles of 256 x 256) still covers the chip with 64-row tiles;
//   * X and W take row strides (ldx, ldw) and W may come as TWO row segments (w0: columns < nseg, w1: the rest): the video half of a
//     head's first Linear is the column slice W[:, :Dv] of the [Hm, Dv + Ds] parameter, and the boundary head stacks two such
//     parameters (start | end) -- both are read in place, no sliced or concatenated copies;
//   * the epilogue is a functor on the accumulator tile:
//       EpiStore     Y = acc + bias                                                     (tsg_gemm_f32s, as before)
//       EpiMatch     K5, the matching head (DistributionAlign.py:83-118):  logits[row] = w2 . act(acc + cs[b,:]) + b2
//       EpiBoundary  K3, the boundary head (SpanPredictor.py:71-85, gate SpanGroundMatchDisc.py:86):
//                    l[branch][row] = w2 . tanh(gate[row] (acc + cs[b,:]) + b1) + b2[branch], mask_logits; softmax over T by the
//                    existing boundary_softmax kernel
//     The head epilogues reduce over the tile's 256 columns in registers (DPP) and LDS, in a fixed order; a head wider than one N tile
//     (K5: 1024 hidden columns = 4 tiles) publishes per-tile partial rows (write-through stores, vmcnt(0), barrier, one relaxed ticket
//     per M tile: the K3-backward / K1-backward recipe) and the last arrival adds them in tile order: run-to-run identical, no float
//     atomics.  `y` (the pre-activation GEMM output the backward kernels read) is written only when the caller asks for it
//     (training); under no_grad it never exists.

namespace tsg {
namespace {

typedef float g_f32x16 __attribute__((ext_vector_type(16)));
typedef float g_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 g_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned g_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kGT = 512;                       // threads
constexpr int kTN = 256, kBK = 32;             // N tile, fp32 columns per chunk
constexpr int kPlaneW = kTN * kBK / 2;          // dwords of one bf16 plane of the W tile: 256 rows x 16 dwords (16 KiB)
template <int TM> constexpr int plane_x() { return TM * kBK / 2; }
template <int TM> constexpr size_t gemm_lds() { return sizeof(unsigned) * 2 * (2 * plane_x<TM>() + 2 * kPlaneW); }   // [2 buffers][X hi | X lo | W hi | W lo]

__device__ __forceinline__ unsigned g_pk(float a, float b) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector((g_f32x2){a, b}, g_bf16x2));
}
__device__ __forceinline__ void g_split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = g_pk(a, b);
  lo = g_pk(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ g_f32x16 g_mfma(g_u32x4 a, g_u32x4 b, g_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(g_bf16x8, a), __builtin_bit_cast(g_bf16x8, b), c, 0, 0, 0);
}

// Sum over the 32 lanes of each wave half (lanes 0..31 / 32..63); every lane of a half receives its half's total.
__device__ __forceinline__ float half_allsum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // xor 16
  return v;
}
__device__ __forceinline__ void g_store_agent(float* p, float v) {          // write-through: visible to every XCD once acknowledged
  asm volatile("global_store_dword %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ float4 g_load_agent_x4(const float* p0, const float* p1, const float* p2, const float* p3) {
  float4 v;                                                               // four independent device-coherent loads, one wait
  asm volatile("global_load_dword %0, %4, off sc1\n\tglobal_load_dword %1, %5, off sc1\n\t"
               "global_load_dword %2, %6, off sc1\n\tglobal_load_dword %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(v.x), "=&v"(v.y), "=&v"(v.z), "=&v"(v.w) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
  return v;
}
__device__ __forceinline__ float g_tanh(float x) {        // 1 - 2/(exp(2x)+1), as K3's tanh_fast
  const float e = fast_exp2(clampf(x, -44.f, 44.f) * k2Log2e);
  return 1.f - 2.f * fast_rcp(e + 1.f);
}
template <int ACT> __device__ __forceinline__ float g_act(float z) {          // 0 relu, 1 tanh, 2 sigmoid (as K5's act_f)
  if (ACT == 0) return fmaxf(z, 0.f);
  if (ACT == 1) return g_tanh(z);
  return fast_rcp(1.f + fast_exp2(-z * kLog2e));
}

// Where a lane's accumulator elements sit in the workgroup tile (32x32x16 C/D map): element r of tile (i, j) is
// row wm + 32 i + 4 kg + (r & 3) + 8 (r >> 2), column wn + 32 j + jl.
struct TilePos { int m0, n0, wm, wn, jl, kg, tid; };
__device__ __forceinline__ int acc_row(const TilePos& p, int i, int r) { return p.wm + 32 * i + 4 * p.kg + (r & 3) + 8 * (r >> 2); }

// ---- epilogue: plain store ------------------------------------------------------------------------------------------------------
struct EpiStore {
  const float* bias; float* Y; long long ldy;
  int accumulate = 0;                              // 1: Y += acc (+ bias): a second gradient contribution lands on the first one's buffer, no add kernel
  static constexpr bool kUsesLds = false;
  template <int MI>
  __device__ __forceinline__ void run(const g_f32x16 (&acc)[MI][2], const TilePos& p, float* /*lds*/) const {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = p.n0 + p.wn + 32 * j + p.jl;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        float* yp = Y + (size_t)(p.m0 + p.wm + 32 * i + 4 * p.kg) * ldy + col;
        if (accumulate) {
          float prev[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) prev[r] = yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy];
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = prev[r] + (acc[i][j][r] + bv);
        } else {
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = acc[i][j][r] + bv;
        }
      }
    }
  }
};

// ---- epilogue: a head.  HEAD = 0: matching head (K5), HEAD = 1: boundary head (K3).
// Rows are (batch item b = row / T, clip t = row % T); cs [Bn, N] is the per-item sentence half of the first Linear.
// lds (reused after the K loop, behind a barrier): red [4 N-waves][TM] partial row sums + one flag word.
struct HeadArgs {
  const float* cs;       // [M / T, N]
  const float* b1;       // K3: [N] first-Linear bias (K5 carries it inside cs); else NULL
  const float* w2;       // [N]
  const float* b2;       // K5: [1]; K3: [2]
  const float* gate;     // K3: [M] or NULL
  const int* mask;       // K3: [M] or NULL
  float* Y;              // [M, N] pre-activation GEMM output for the backward, or NULL
  float* out0;           // K5: logits [M]; K3: start logits [M]
  float* out1;           // K3: end logits [M]
  float* part;           // [tiles per head][M] partial rows (only when a head spans several N tiles)
  unsigned* cnt;         // [M / TM] tickets, zero at launch
  int M, N, T, Hm;       // Hm: columns per head (K5: N; K3: N / 2), a multiple of 256
  float invT;
};

template <int HEAD, int ACT>
struct EpiHead {
  HeadArgs a;
  static constexpr bool kUsesLds = true;
  template <int MI>
  __device__ __forceinline__ void run(const g_f32x16 (&acc)[MI][2], const TilePos& p, float* lds) const {
    constexpr int TM = 64 * MI;
    float* red = lds;                       // [4][TM]
    unsigned* flag = reinterpret_cast<unsigned*>(lds + 4 * TM);
    const int colg[2] = {p.n0 + p.wn + p.jl, p.n0 + p.wn + 32 + p.jl};
    float w2v[2], b1v[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) { w2v[j] = a.w2[colg[j]]; b1v[j] = (HEAD == 1 && a.b1) ? a.b1[colg[j]] : 0.f; }
#pragma unroll
    for (int i = 0; i < MI; ++i) {
      const int r_first = p.m0 + p.wm + 32 * i;                              // first row of this 32-row MFMA tile (wave-uniform)
      const int bA = (int)(((float)r_first + 0.5f) * a.invT), bB = (int)(((float)(r_first + 31) + 0.5f) * a.invT);
      const bool two = bB - bA <= 1;                                          // the tile spans at most two batch items (T >= 32)
      float cA[2], cB[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        cA[j] = a.cs[(size_t)bA * a.N + colg[j]];
        cB[j] = a.cs[(size_t)(two ? bB : bA) * a.N + colg[j]];
      }
      const int boundary = (bA + 1) * a.T;                                    // first row of item bA + 1
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int rl = acc_row(p, i, r), row = p.m0 + rl;
        float c0, c1;
        if (two) { const bool hi = row >= boundary; c0 = hi ? cB[0] : cA[0]; c1 = hi ? cB[1] : cA[1]; }
        else { const int b = (int)(((float)row + 0.5f) * a.invT); c0 = a.cs[(size_t)b * a.N + colg[0]]; c1 = a.cs[(size_t)b * a.N + colg[1]]; }
        const float y0 = acc[i][0][r], y1 = acc[i][1][r];
        if (a.Y) { a.Y[(size_t)row * a.N + colg[0]] = y0; a.Y[(size_t)row * a.N + colg[1]] = y1; }
        float v;
        if (HEAD == 0) {
          v = w2v[0] * g_act<ACT>(y0 + c0) + w2v[1] * g_act<ACT>(y1 + c1);
        } else {
          const float g = a.gate ? a.gate[row] : 1.f;
          v = w2v[0] * g_tanh(fmaf(g, y0 + c0, b1v[0])) + w2v[1] * g_tanh(fmaf(g, y1 + c1, b1v[1]));
        }
        v = half_allsum(v);                                                    // over the wave's 64 columns of this row
        if (p.jl == 0) red[(p.wn >> 6) * TM + rl] = v;
      }
    }
    __syncthreads();
    const int tiles_h = a.Hm / kTN;                                           // N tiles per head
    const int head = p.n0 / a.Hm, q = (p.n0 % a.Hm) / kTN;                    // which head this tile belongs to, which of its tiles
    float* out = (HEAD == 1 && head == 1) ? a.out1 : a.out0;
    const float bias2 = a.b2[HEAD == 1 ? head : 0];
    auto finish = [&](float s, int row) {
      s += bias2;
      if (HEAD == 1 && a.mask) { const float m = (float)a.mask[row]; s = s * m + (-1e30f) * (1.f - m); }   // mask_logits, attention.py:129-133
      out[row] = s;
    };
    if (tiles_h == 1) {
      if (p.tid < TM) finish(red[p.tid] + red[TM + p.tid] + red[2 * TM + p.tid] + red[3 * TM + p.tid], p.m0 + p.tid);
    } else {
      float* mine = a.part + ((size_t)(head * tiles_h + q)) * a.M;
      if (p.tid < TM) g_store_agent(mine + p.m0 + p.tid, red[p.tid] + red[TM + p.tid] + red[2 * TM + p.tid] + red[3 * TM + p.tid]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                        // the partial rows are acknowledged before the ticket moves
      __syncthreads();
      if (p.tid == 0)
        *flag = __hip_atomic_fetch_add(a.cnt + (size_t)head * (a.M / TM) + p.m0 / TM, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(tiles_h - 1);
      __syncthreads();
      if (*flag && p.tid < TM) {                                              // last arrival: the head's tiles in order
        // four device-coherent loads in flight per wait (one dependent round trip per group instead of one per tile: the serial
        // form cost the last arriver 4 x 1-2 us at K5's four tiles)
        const float* base = a.part + (size_t)head * tiles_h * a.M + p.m0 + p.tid;
        float s = 0.f;
        for (int t = 0; t < tiles_h; t += 4) {
          const float4 v = g_load_agent_x4(base + (size_t)min(t, tiles_h - 1) * a.M, base + (size_t)min(t + 1, tiles_h - 1) * a.M,
                                           base + (size_t)min(t + 2, tiles_h - 1) * a.M, base + (size_t)min(t + 3, tiles_h - 1) * a.M);
          s += v.x;
          if (t + 1 < tiles_h) s += v.y;
          if (t + 2 < tiles_h) s += v.z;
          if (t + 3 < tiles_h) s += v.w;
        }
        finish(s, p.m0 + p.tid);
      }
    }
    __syncthreads();                                                           // red / flag are free again before the next tile's chunks land
  }
};

// WT = true: the right operand is stored CONTRACTION-major -- W[K][N] row-major (row stride ldw), y = x W -- as the weight of a Linear is
// for its input gradient dX = dY W ([N_out][K_in]: the contraction index N_out is the row).  Its 32 (k) x 256 (n) chunk is loaded
// row-wise as 4 x 4 blocks and TRANSPOSED on the way into the same LDS image (four 8-byte pieces per thread and plane: the staging of
// the weight-gradient kernel, csrc/wgrad_split.hip); nothing else changes.  It replaces a transposed copy of the weight per call
// (16 MB for an LSTM layer's W_ih, 4 MB for a d x d projection).  In this mode the two segments (W0 | W1, nseg) split the ROWS.
template <int TM, typename Epi, bool WT = false>
__global__ __launch_bounds__(kGT) void gemm_nt_f32s_kernel(const float* __restrict__ X, long long ldx, const float* __restrict__ W0,
                                                           const float* __restrict__ W1, int nseg, long long ldw, Epi epi,
                                                           int M, int N, int K, int tiles_n) {
  constexpr int MI = TM / 64, PX = TM / 64;                         // MFMA row tiles per wave; staging passes over the X tile
  constexpr int kPlaneX = plane_x<TM>();
  constexpr int kBuf = 2 * kPlaneX + 2 * kPlaneW;
  extern __shared__ __align__(16) unsigned lds[];                   // [2 buffers][X hi | X lo | W hi | W lo]
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // The SIMD's arbiter serves its OLDER wave first (cdna_hip_programming.md T5, static form; seen in K1g this round as 7.7 k vs 11 k ticks per row): the
  // second-dispatched half of an 8-wave workgroup gets priority 1 once, so the two waves of a SIMD advance together -- 1-2.5 % at the step's shapes
  // (profiles/r6/gemm_young_half_priority_ab_v1.txt)
  if (wv >= 4) __builtin_amdgcn_s_setprio(1);
  const int wm = (wv >> 2) * (TM / 2), wn = (wv & 3) * 64;
  const int jl = lane & 31, kg = lane >> 5;
  const int ntiles = (M / TM) * tiles_n;
  // Persistent workgroups: workgroup w walks the tiles w, w + grid, ...: the fp32 stores of a finished tile drain under the next
  // tile's chunks (a 256 x 256 tile is 256 KiB of output: 15-20 us of store tail per tile when nothing runs beside it).
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
  // Placement: workgroup i runs on XCD i % 8, each XCD has its own L2.  An XCD's 32 concurrent tiles are 8 (M) x 4 (N) neighbours
  // when the shape allows it: per chunk its L2 serves 8 X pieces to 4 readers each and 4 W pieces to 8 readers each.  (Until round 4:
  // all N tiles of one M tile on an XCD = 2 x 16 at N = 4096, every XCD streaming all of W every round: 478-488 us vs 450 us at
  // [16384 x 1024] x [4096 x 1024]^T, profiles/r4/gemm_wgrad_flags_ab_v1.txt.)
  int m0, n0;
  if (tiles_n % 4 == 0 && (ntiles / tiles_n) % 8 == 0 && ntiles % 256 == 0) {
    const int b = xcd_remap(tile, ntiles, 32), gq = b >> 5, r = b & 31, gn = tiles_n >> 2;
    m0 = ((gq / gn) * 8 + (r >> 2)) * TM; n0 = ((gq % gn) * 4 + (r & 3)) * kTN;
  } else {
    const int b = xcd_remap(tile, ntiles, tiles_n);                 // the N tiles of one M tile share an XCD (X rows stay in its L2)
    m0 = (b / tiles_n) * TM; n0 = (b % tiles_n) * kTN;
  }

  // staging role: 8 threads per row (8 float4 = one 128-byte row segment), 64 rows per pass
  const int sr = tid >> 3, sq = tid & 7;
  const float* xsrc = X + (size_t)(m0 + sr) * ldx + 4 * sq;
  const float* wbase = n0 < nseg ? W0 + (size_t)n0 * ldw : W1 + (size_t)(n0 - nseg) * ldw;     // a tile lies inside one segment (nseg % 256 == 0)
  const float* wsrc = wbase + (size_t)sr * ldw + 4 * sq;
  const size_t passx = (size_t)64 * ldx, passw = (size_t)64 * ldw;
  // WT staging role: a 4 (k) x 4 (n) block; lane bits [1:0] = column quad inside a 64-byte segment, [4:2] = row group, [5..8] = segment
  const int tmg = (tid >> 2) & 7, tc4 = (tid & 3) + 4 * (tid >> 5);
  float4 rx[PX], rw[4];
  auto request = [&](int k0) {
#pragma unroll
    for (int p = 0; p < PX; ++p) rx[p] = *reinterpret_cast<const float4*>(xsrc + p * passx + k0);
    if constexpr (WT) {
      const int kr = k0 + 4 * tmg;                                    // a chunk lies inside one row segment (nseg % 32 == 0)
      const float* wt = (kr < nseg ? W0 + (size_t)kr * ldw : W1 + (size_t)(kr - nseg) * ldw) + n0 + 4 * tc4;
#pragma unroll
      for (int p = 0; p < 4; ++p) rw[p] = *reinterpret_cast<const float4*>(wt + (size_t)p * ldw);
    } else {
#if !(defined(TSG_GEMM_ABL) && (TSG_GEMM_ABL & 2))
#pragma unroll
      for (int p = 0; p < 4; ++p) rw[p] = *reinterpret_cast<const float4*>(wsrc + p * passw + k0);
#endif
    }
  };
  // split once, here, and write the bf16 planes: row r, 16-byte piece (sq >> 1) ^ ((r >> 2) & 3), 8-byte half sq & 1
  auto write_planes = [&](int buf) {
    unsigned* base = lds + buf * kBuf;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int r = sr + 64 * p;
      const int o = r * 16 + (((sq >> 1) ^ ((r >> 2) & 3)) << 2) + ((sq & 1) << 1);
      unsigned h0, l0, h1, l1;
      if (p < PX) {
        g_split_pair(rx[p < PX ? p : 0].x, rx[p < PX ? p : 0].y, h0, l0); g_split_pair(rx[p < PX ? p : 0].z, rx[p < PX ? p : 0].w, h1, l1);
        *reinterpret_cast<uint2*>(base + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + kPlaneX + o) = make_uint2(l0, l1);
      }
      if constexpr (WT) {
        // column j = p of the thread's 4 x 4 block = plane row 4 tc4 + p; its k = 4 tmg .. + 3 = half (tmg & 1) of 16-byte piece tmg >> 1
        const int rt = 4 * tc4 + p;
        const int ot = rt * 16 + ((((tmg >> 1) ^ ((rt >> 2) & 3))) << 2) + ((tmg & 1) << 1);
        const float e0 = p == 0 ? rw[0].x : p == 1 ? rw[0].y : p == 2 ? rw[0].z : rw[0].w, e1 = p == 0 ? rw[1].x : p == 1 ? rw[1].y : p == 2 ? rw[1].z : rw[1].w;
        const float e2 = p == 0 ? rw[2].x : p == 1 ? rw[2].y : p == 2 ? rw[2].z : rw[2].w, e3 = p == 0 ? rw[3].x : p == 1 ? rw[3].y : p == 2 ? rw[3].z : rw[3].w;
        g_split_pair(e0, e1, h0, l0); g_split_pair(e2, e3, h1, l1);
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + ot) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + kPlaneW + ot) = make_uint2(l0, l1);
      } else {
#if defined(TSG_GEMM_ABL) && (TSG_GEMM_ABL & 2)
        // ablation: no W staging at all (stale planes)
#elif defined(TSG_GEMM_ABL) && (TSG_GEMM_ABL & 1)
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + o) = make_uint2(__float_as_uint(rw[p].x), __float_as_uint(rw[p].y));     // ablation: no conversion
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + kPlaneW + o) = make_uint2(__float_as_uint(rw[p].z), __float_as_uint(rw[p].w));
#else
        g_split_pair(rw[p].x, rw[p].y, h0, l0); g_split_pair(rw[p].z, rw[p].w, h1, l1);
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(base + 2 * kPlaneX + kPlaneW + o) = make_uint2(l0, l1);
#endif
      }
    }
  };
  // fragment of k step s: 8 consecutive k (one 16-byte piece 2 s + kg, swizzled) of row r, from the hi and the lo plane
  auto frag = [&](const unsigned* hi_plane, int plane, int r, int s, g_u32x4& hi, g_u32x4& lo) {
    const int o = r * 16 + (((2 * s + kg) ^ ((r >> 2) & 3)) << 2);
    hi = *reinterpret_cast<const g_u32x4*>(hi_plane + o);
    lo = *reinterpret_cast<const g_u32x4*>(hi_plane + plane + o);
  };

  g_f32x16 acc[MI][2];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nk = K / kBK;
  request(0);
  write_planes(0);                                                   // chunk 0 -> buffer 0
  if (nk > 1) request(kBK);
  // chunk ks: buffer ks & 1 is multiplied; the registers hold chunk ks + 1, which is converted into the other buffer right after the
  // barrier, and chunk ks + 2 is requested into them.  WRITE / REQ are compile-time so that the steady-state body is one basic block
  // (a branch inside it pins every fragment read in front of its own MFMAs).
  auto chunk = [&](int ks, auto w_tag, auto r_tag) {
    constexpr bool WRITE = decltype(w_tag)::value, REQ = decltype(r_tag)::value;
    lds_barrier();                                                   // chunk ks is in LDS; nobody reads the other buffer any more
    __builtin_amdgcn_sched_barrier(0);                               // nothing of this chunk is scheduled in front of the barrier (2-4 %)
    if (WRITE) write_planes((ks + 1) & 1);
    if (REQ) request((ks + 2) * kBK);
    const unsigned* xt = lds + (ks & 1) * kBuf;
    const unsigned* wt = xt + 2 * kPlaneX;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      g_u32x4 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) frag(wt, kPlaneW, wn + 32 * j + jl, s, bh[j], bl[j]);
      g_u32x4 ah, al, nh, nl;
      frag(xt, kPlaneX, wm + jl, s, ah, al);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        if (i < MI - 1) frag(xt, kPlaneX, wm + 32 * (i + 1) + jl, s, nh, nl);     // the next X tile's fragment flies under this tile's MFMAs
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = g_mfma(ah, bh[j], acc[i][j]);
          acc[i][j] = g_mfma(ah, bl[j], acc[i][j]);
          acc[i][j] = g_mfma(al, bh[j], acc[i][j]);
        }
        if (i < MI - 1) { ah = nh; al = nl; }
      }
    }
  };
  typedef std::true_type Y_; typedef std::false_type N_;
  int ks = 0;
#pragma unroll 1
  for (; ks + 2 < nk; ++ks) chunk(ks, Y_{}, Y_{});
  if (ks + 1 < nk) { chunk(ks, Y_{}, N_{}); ++ks; }
  chunk(ks, N_{}, N_{});
  if (Epi::kUsesLds) __syncthreads();                                // every wave is done with the last chunk's buffer: the epilogue reuses LDS
  const TilePos pos{m0, n0, wm, wn, jl, kg, tid};
  epi.template run<MI>(acc, pos, reinterpret_cast<float*>(lds));
  if (!Epi::kUsesLds) lds_barrier();                                 // (the head epilogues end with their own barrier)
  }
}

// (Round 4, measured and removed: STAGGERED SIMD partners -- waves 4..7 run each barrier interval in the other order (MFMAs first, the
// next chunk's conversion behind them; MI355X_MICROARCH.md "Two waves per SIMD" item 9): step 14.13 / 14.06 / 14.09 ms vs 14.11 / 14.08 /
// 14.07 ms lock-step, stand-alone 454-459 vs 452-455 us at [16384 x 1024] x [4096 x 1024]^T: profiles/r4/gemm_f32s_stagger_ab_v1.txt.
// hipcc already interleaves the conversion with the MFMAs inside the one basic block of the steady-state chunk.)
// (Round 4, measured and removed: the same kernel on v_mfma_f32_16x16x32_bf16 -- MI355X_MICROARCH.md reports a higher sustained clock for
// that shape in bare MFMA loops.  Here, with the conversion VALU beside the MFMAs, it is a tie: 387-431 vs 396-438 us at
// [16384 x 1024] x [4096 x 1024]^T, 94-96 vs 94-101 us at [16384 x 1024] x [1024 x 1024]^T, two processes each, alternating:
// profiles/r4/gemm_f32s_mfma_shape_ab_v1.txt.  The LDS image for its fragment reads: piece p of row r at slot p ^ h((r >> 2) & 3),
// h = {0, 3, 2, 1}, conflict-free for both the b128 reads and the staging writes.)

template <int TM, typename Epi, bool WT = false>
int launch_gemm(const char* fn, const float* x, long long ldx, const float* w0, const float* w1, int nseg, long long ldw, const Epi& epi,
                int M, int N, int K, hipStream_t st) {
  auto kern = gemm_nt_f32s_kernel<TM, Epi, WT>;
  constexpr size_t lds = gemm_lds<TM>();
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute(%zu): %s", fn, lds, hipGetErrorString(e));
  const int tiles_n = N / kTN;
  const int tiles = (M / TM) * tiles_n, cus = device_cu_count();
  const int grid = tiles < cus ? tiles : cus;
  hipLaunchKernelGGL(kern, dim3(grid), dim3(kGT), lds, st, x, ldx, w0, w1, nseg, ldw, epi, M, N, K, tiles_n);
  return check_launch(fn);
}

// M tile: the one of 256 / 128 / 64 rows (dividing M) with the smallest estimated time = full-chip rounds x time of one tile.  A tile's
// time per 1024 columns of K, measured at one tile per CU (tools/gemm_small_time.py): 73 / 46 / 31 us -- the W tile (256 x 32 per chunk)
// is converted by every workgroup whatever its M, so small tiles pay off only while they save a round or fill idle CUs:
// [2560 x 1024] x [1024 x 1024]^T 70 / 47 / 33 us, [8192 x 1024] x [1024 x 1024]^T 77 / 57 / 69 us, [1280 x 1024] x [4096 x 1024]^T 74 / 53 / 67 us.
inline int pick_tm(int M, int N) {
  const int cus = device_cu_count(), tiles_n = N / kTN;
  int best = 0; float best_cost = 0.f;
  for (int tm : {256, 128, 64}) {
    if (M % tm) continue;
    const int tiles = (M / tm) * tiles_n;
    const float cost = (float)((tiles + cus - 1) / cus) * (17.f + 0.22f * tm);
    if (!best || cost < best_cost) { best = tm; best_cost = cost; }
  }
  return best;
}

inline long long head_ws_bytes(int M, int N, int heads) {
  const int tiles_h = N / heads / kTN;
  const long long cnt = roundup((long long)heads * (M / 64), 4);              // tickets for the smallest M tile
  return (long long)sizeof(float) * (cnt + (tiles_h > 1 ? (long long)(N / kTN) * M : 0));
}

int check_head(const char* fn, std::initializer_list<const void*> ptrs, int M, int T, int N, int K, long long ldx, long long ldw, int Hm) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (M <= 0 || T <= 0 || N <= 0 || K <= 0 || M % T) return set_error(TSG_E_SHAPE, "%s: bad dimensions M=%d T=%d N=%d K=%d (M must be B*T)", fn, M, T, N, K);
  if (M % 64 || N % kTN || K % kBK || Hm % kTN || M > (1 << 22))   // (1 << 22: the epilogue derives the batch item of a row as (row + 0.5) * (1 / T) in fp32, exact below 2^22)
    return set_error(TSG_E_SHAPE, "%s: M=%d must be a multiple of 64, N=%d and the head width %d of 256, K=%d of 32", fn, M, N, Hm, K);
  if (ldx < K || ldw < K || ldx % 4 || ldw % 4) return set_error(TSG_E_ALIGN, "%s: ldx=%lld / ldw=%lld must be >= K and multiples of 4", fn, ldx, ldw);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

// Y[M,N] = X[M,K] W[N,K]^T (+ bias, may be NULL) in the split-precision arithmetic, operands converted on load.
// M % 256 == 0, N % 256 == 0, K % 32 == 0 (TSG_E_SHAPE otherwise: the caller uses tsg_split_bf16x3 + a bf16 GEMM).
extern "C" int tsg_gemm_f32s(const void* x, const void* w, const void* bias, void* y, int M, int N, int K, void* stream) {
  return tsg_gemm_f32s_ld(x, K, w, K, bias, y, N, M, N, K, stream);
}

extern "C" int tsg_gemm_f32s_ld(const void* x, long long ldx, const void* w, long long ldw, const void* bias, void* y, long long ldy,
                                int M, int N, int K, void* stream) {
  const char* fn = "tsg_gemm_f32s";
  for (const void* p : {x, w, (const void*)y}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (M <= 0 || N <= 0 || K <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension M=%d N=%d K=%d", fn, M, N, K);
  if (M % 64 || N % kTN || K % kBK)
    return set_error(TSG_E_SHAPE, "%s: M=%d must be a multiple of 64, N=%d of 256 and K=%d of 32", fn, M, N, K);
  if (ldx < K || ldw < K || ldy < N || ldx % 4 || ldw % 4) return set_error(TSG_E_ALIGN, "%s: leading dimensions ldx=%lld ldw=%lld ldy=%lld", fn, ldx, ldw, ldy);
  const EpiStore epi{(const float*)bias, (float*)y, ldy};
  auto st = static_cast<hipStream_t>(stream);
  // M tile: 256 rows when that still gives every CU a tile; with few rows (the sentence side: 2560 = 128 x 20) 128 or 64, as the heads
  static const int force_tm = getenv("TSG_GEMM_TM") ? atoi(getenv("TSG_GEMM_TM")) : 0;       // developer override (A/B timing)
  const int tm = force_tm && M % force_tm == 0 ? force_tm : pick_tm(M, N);
  if (tm == 256) return launch_gemm<256>(fn, (const float*)x, ldx, (const float*)w, (const float*)w, N, ldw, epi, M, N, K, st);
  if (tm == 128) return launch_gemm<128>(fn, (const float*)x, ldx, (const float*)w, (const float*)w, N, ldw, epi, M, N, K, st);
  return launch_gemm<64>(fn, (const float*)x, ldx, (const float*)w, (const float*)w, N, ldw, epi, M, N, K, st);
}

// y[M,N] = x[M,K] w[K,N] (+ bias): the right operand contraction-major, optionally as two ROW segments (w0: rows < kseg, w1: the rest;
// kseg = K and w1 = NULL for one matrix) -- the input gradient dX = dY W of a Linear with its weight as it is stored.
static int gemm_nn_impl(const char* fn, const void* x, long long ldx, const void* w0, const void* w1, int kseg, long long ldw, const void* bias,
                        void* y, long long ldy, int M, int N, int K, int accumulate, void* stream) {
  for (const void* p : {x, w0, (const void*)y}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (M <= 0 || N <= 0 || K <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension M=%d N=%d K=%d", fn, M, N, K);
  if (M % 256 || N % kTN || K % kBK) return set_error(TSG_E_SHAPE, "%s: M=%d, N=%d must be multiples of 256 and K=%d of 32", fn, M, N, K);
  if (kseg <= 0 || kseg > K || kseg % kBK || (kseg < K && (!w1 || !aligned16(w1))))
    return set_error(TSG_E_SHAPE, "%s: kseg=%d must be a multiple of 32 in (0, K] and w1 given when kseg < K", fn, kseg);
  if (ldx < K || ldw < N || ldy < N || ldx % 4 || ldw % 4) return set_error(TSG_E_ALIGN, "%s: leading dimensions ldx=%lld ldw=%lld ldy=%lld", fn, ldx, ldw, ldy);
  const EpiStore epi{(const float*)bias, (float*)y, ldy, accumulate};
  return launch_gemm<256, EpiStore, true>(fn, (const float*)x, ldx, (const float*)w0, (const float*)(w1 ? w1 : w0), kseg, ldw, epi, M, N, K,
                                          static_cast<hipStream_t>(stream));
}

extern "C" int tsg_gemm_f32s_nn(const void* x, long long ldx, const void* w0, const void* w1, int kseg, long long ldw, const void* bias,
                                void* y, long long ldy, int M, int N, int K, void* stream) {
  return gemm_nn_impl("tsg_gemm_f32s_nn", x, ldx, w0, w1, kseg, ldw, bias, y, ldy, M, N, K, 0, stream);
}

// y[M,N] += x[M,K] w[K,N]: the same kernel with a read-add-store epilogue.  A tensor with two consumers gets two input gradients; the second
// consumer's dX = dY W lands on the first one's buffer instead of in a tensor of its own that autograd then adds (a 3 x 64 MB elementwise
// kernel per [128, 128, 1024] activation: 30 us each, four per GMD step).
extern "C" int tsg_gemm_f32s_nn_acc(const void* x, long long ldx, const void* w0, const void* w1, int kseg, long long ldw,
                                    void* y, long long ldy, int M, int N, int K, void* stream) {
  return gemm_nn_impl("tsg_gemm_f32s_nn_acc", x, ldx, w0, w1, kseg, ldw, nullptr, y, ldy, M, N, K, 1, stream);
}

extern "C" long long tsg_head_gemm_ws_bytes(int M, int N, int heads) {
  if (M <= 0 || N <= 0 || heads < 1 || heads > 2 || N % (heads * kTN) || M % 64) return -1;
  return head_ws_bytes(M, N, heads);
}

extern "C" int tsg_match_head_gemm(const void* x, long long ldx, const void* w, long long ldw, const void* cs, const void* w2, const void* b2,
                                   void* y, void* logits, void* ws, long long ws_bytes, int M, int T, int N, int K, int activation,
                                   void* stream) {
  const char* fn = "tsg_match_head_gemm";
  int rc = check_head(fn, {x, w, cs, w2, b2, (const void*)logits, (const void*)ws}, M, T, N, K, ldx, ldw, N);
  if (rc) return rc;
  if (y && !aligned16(y)) return set_error(TSG_E_ALIGN, "%s: y not 16-byte aligned", fn);
  if (activation < 0 || activation > 2) return set_error(TSG_E_SHAPE, "%s: activation %d (0 relu, 1 tanh, 2 sigmoid)", fn, activation);
  if (ws_bytes < head_ws_bytes(M, N, 1)) return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_head_gemm_ws_bytes)", fn, ws_bytes, head_ws_bytes(M, N, 1));
  auto st = static_cast<hipStream_t>(stream);
  const int tm = pick_tm(M, N);
  const long long cntw = roundup((long long)(M / 64), 4);
  unsigned* cnt = static_cast<unsigned*>(ws);
  float* part = static_cast<float*>(ws) + cntw;
  if (N / kTN > 1) {
    hipError_t e = zero_async(cnt, sizeof(unsigned) * cntw, st);
    if (e != hipSuccess) return set_error((int)e, "%s: zero fill: %s", fn, hipGetErrorString(e));
  }
  const HeadArgs a{(const float*)cs, nullptr, (const float*)w2, (const float*)b2, nullptr, nullptr, (float*)y, (float*)logits, nullptr,
                   part, cnt, M, N, T, N, 1.f / (float)T};
  const float* xf = (const float*)x; const float* wf = (const float*)w;
#define TSG_MH(TMV, ACTV) launch_gemm<TMV>(fn, xf, ldx, wf, wf, N, ldw, EpiHead<0, ACTV>{a}, M, N, K, st)
#define TSG_MH_TM(ACTV) (tm == 256 ? TSG_MH(256, ACTV) : (tm == 128 ? TSG_MH(128, ACTV) : TSG_MH(64, ACTV)))
  if (activation == 0) return TSG_MH_TM(0);
  if (activation == 1) return TSG_MH_TM(1);
  return TSG_MH_TM(2);
#undef TSG_MH_TM
#undef TSG_MH
}

extern "C" int tsg_boundary_head_gemm(const void* x, long long ldx, const void* w_start, const void* w_end, long long ldw, const void* cs,
                                      const void* b1, const void* w2, const void* b2, const void* gate, const int32_t* mask, void* y,
                                      void* p_start, void* p_end, void* ws, long long ws_bytes, int B, int T, int Hm, int K, void* stream) {
  const char* fn = "tsg_boundary_head_gemm";
  if (B <= 0 || T <= 0 || (long long)B * T > (1 << 22)) return set_error(TSG_E_SHAPE, "%s: bad B=%d T=%d", fn, B, T);
  const int M = B * T, N = 2 * Hm;
  int rc = check_head(fn, {x, w_start, w_end, cs, b1, w2, b2, (const void*)p_start, (const void*)p_end, (const void*)ws}, M, T, N, K, ldx, ldw, Hm);
  if (rc) return rc;
  if (y && !aligned16(y)) return set_error(TSG_E_ALIGN, "%s: y not 16-byte aligned", fn);
  if (T > 8192) return set_error(TSG_E_SHAPE, "%s: T=%d > 8192 not supported", fn, T);
  if (ws_bytes < head_ws_bytes(M, N, 2)) return set_error(TSG_E_SHAPE, "%s: workspace of %lld B < %lld B (tsg_head_gemm_ws_bytes)", fn, ws_bytes, head_ws_bytes(M, N, 2));
  auto st = static_cast<hipStream_t>(stream);
  const int tm = pick_tm(M, N);
  const long long cntw = roundup((long long)2 * (M / 64), 4);
  unsigned* cnt = static_cast<unsigned*>(ws);
  float* part = static_cast<float*>(ws) + cntw;
  if (Hm / kTN > 1) {
    hipError_t e = zero_async(cnt, sizeof(unsigned) * cntw, st);
    if (e != hipSuccess) return set_error((int)e, "%s: zero fill: %s", fn, hipGetErrorString(e));
  }
  const HeadArgs a{(const float*)cs, (const float*)b1, (const float*)w2, (const float*)b2, (const float*)gate, mask, (float*)y,
                   (float*)p_start, (float*)p_end, part, cnt, M, N, T, Hm, 1.f / (float)T};
  const EpiHead<1, 1> epi{a};
  const float* xf = (const float*)x;
  if (tm == 256) rc = launch_gemm<256>(fn, xf, ldx, (const float*)w_start, (const float*)w_end, Hm, ldw, epi, M, N, K, st);
  else if (tm == 128) rc = launch_gemm<128>(fn, xf, ldx, (const float*)w_start, (const float*)w_end, Hm, ldw, epi, M, N, K, st);
  else rc = launch_gemm<64>(fn, xf, ldx, (const float*)w_start, (const float*)w_end, Hm, ldw, epi, M, N, K, st);
  if (rc) return rc;
  return tsg_boundary_softmax(p_start, p_end, B, T, stream);       // softmax over T, in place (csrc/boundary_head.hip)
}
