// Weight-gradient products of the path in the split-precision mode ("f32s"), operands converted ON LOAD:
//     C[g][n][k] = sum_m A[m][g*a_gs + n] * B_g[m][k]            (dW = dY^T X: contraction over the T*B rows)
// Reference sites: the weight gradients autograd forms for the path's Linears (networks/attention.py:105-106,113-114,
// components/SpanPredictor.py:62-72, components/DistributionAlign.py:88-94) and for nn.LSTM's W_ih / W_hh (networks/RNN.py:31,42).
//
// Why a hand-written kernel here: the output is small (<= 2048 x 1536) and the contraction long (16384 rows), and BOTH operands
// are stored contraction-major (the contraction index is the row of a row-major fp32 matrix).  The library path needs a
// transposing operand pass per operand (fp32 read, three bf16 planes written, read again by the GEMM) and then runs a shape its
// bf16 kernels are slow at (few output tiles); its fp32 GEMM is at the fp32 MFMA peak but that peak is 16x lower.  This kernel
// reads the fp32 rows once, splits every element into hi = rne_bf16(x), lo = rne_bf16(x - hi) in registers, transposes through
// LDS into m-contiguous bf16 fragments and accumulates  hi*hi + hi*lo + lo*hi  on v_mfma_f32_32x32x16_bf16 (fp32 accumulate):
// the arithmetic of the split-precision GEMMs around it (split_bf16.hip), without the operand planes ever touching HBM.
//
// Workgroup = 512 threads = 8 waves (4 along n x 2 along k), output tile 256 (n) x 128 (k), each wave 64 x 64 = 2 x 2 MFMA tiles
// (64 accumulator registers).  The contraction is walked in chunks of 32 rows; per chunk and wave: 24 MFMAs on the current LDS
// buffer, 6 float4 row requests TWO chunks ahead (an L2 round trip under load is longer than one chunk), and the split of the next
// chunk's rows (requested an iteration ago) into the other LDS buffer; one barrier per chunk.  The three are independent and are laid
// out as ONE interleaved instruction stream (sched_group_barrier: after every MFMA a fragment read / a load / an LDS write and 3-4
// conversion instructions), so a wave hides its own memory and VALU work in the 24 of 32 cycles an MFMA leaves the issue port free.
// Loads: four neighbouring lanes read 64 contiguous bytes of a row (one L1 tag lookup per lane quad; lanes running along the rows
// cost 64 lookups per instruction and made the L1 pipe the bottleneck: 3 900 -> 3 000 cycles per chunk), then the lanes run along
// the rows, which keeps the transposing LDS writes conflict-free.  LDS image per operand plane: [column][32 m] bf16 on an 80-byte
// pitch -- conflict-free for the b128 fragment reads (16 lanes = 16 distinct bank quads).
// Where the time goes at [1024 x 16384] x [16384 x 1024] (ablation builds -DTSG_WGRAD_ABL, tools/wgrad_abl.py): 111 us by events =
// 26 us fixed (launches, 32 MB of partial tiles out and back, the reduce) + 65 us for fragment reads + MFMAs alone (41 us at the
// 2.5 PFLOP/s peak: LDS fragment traffic, 128 KiB per chunk and CU, is the co-bottleneck) + 20 us of loads and conversion that
// still do not overlap.  Against it: 290 us for the library's fp32 GEMM, 340 us for operand planes + its bf16 GEMM.
// Long contraction, few tiles: the rows are cut into `splits` ranges, one per workgroup; partial tiles go to a workspace and a
// second kernel adds them in split order (deterministic).  Workgroups of one split are neighbours in XCD-major order, so an XCD's
// L2 streams one row range of A and B once.
// B can have two column segments: [0,K0) from B0, [K0,K0+K1) from B1 with a ROW SHIFT (source row m - shift, zeros outside the
// row's sequence; group 1 uses -shift): the h_{t-1} / h_{t+1} operand of dW_hh straight from the LSTM output, no shifted copy.
// (Round 4, measured and removed: a 256 x 256 tile variant with 128 x 64 wave tiles -- the main loop of tsg_gemm_f32s behind this kernel's
// transposing stage: 32 instead of 24 converted elements and 24 instead of 16 fragment reads per thread and chunk for 48 instead of 24
// MFMAs.  Slower: the LSTM layer's [2][2048 x 16384] x [16384 x 1536] 796-805 vs 776-777 us, [1024 x 16384] x [16384 x 1024] 126-128 vs
// 121-124 us, the step 14.60-14.65 vs 14.36-14.43 ms (profiles/r4/wgrad_tile256_ab_v1.txt): fewer, larger tiles need twice the row
// ranges (partial-tile traffic) and lose the sched_group_barrier interleave below, which is worth more than the geometry.)
#include "tsg_common.h"
#include <atomic>
#include <cstdlib>

namespace tsg {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int BM = 32;                                  // rows of the contraction per chunk
constexpr int P = 20;                                   // LDS pitch of one column in dwords: 16 (32 bf16 rows) + 4

constexpr int kMaxContrib = 3;                          // stream-K: workgroups that may contribute to one output tile
struct WgradArgs {                                      // A / B0 / B1: fp32 (tsg_wgrad_f32s) or bf16 (tsg_wgrad_bf16) elements; strides in elements
  const void* A; long lda; long a_gs;
  const void* B0; long ldb0; int K0;
  const void* B1; long ldb1; long b1_gs; int K1; long shift; long period;
  float* C; long ldc; long c_gs;                        // output (splits == 1) ...
  float* C1; long ldc1; long c1_gs;                     // optional second output for the columns k >= K0 (the B1 segment), else NULL: C1[g][n][k - K0]
  float* ws;                                            // ... or partials [splits][groups][N][K0+K1]
  long M; int N; int groups; int splits; int cps;       // cps = chunks (of 32 rows) per split
  unsigned per_magic; int per_sh;                       // row / period = (row * per_magic) >> per_sh  (rows < 2^31; host: period_division)
  // stream-K (round 5; ipw > 0): the grid's workgroups cut the flattened (tile, chunk) space into equal pieces of ipw chunks -- a workgroup
  // covers the tail of one tile and the head of the next -- instead of `splits` whole row ranges per tile.  A tile that several workgroups
  // contribute to (at most kMaxContrib) is finished by its LAST arriver: see finish_tile.
  long ipw;                                             // chunks per workgroup (0 = the split / reduce-kernel scheme above)
  unsigned* tickets;                                    // [tiles] arrival counters, zero at launch, zero again at exit
  float* part;                                          // [tiles][kMaxContrib][TN * TK] partial accumulators in register order
};

typedef float w_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_agent_x4(float* p, w_f32x4 v) {       // write-through (agent scope): visible to an sc1 load on another XCD
  asm volatile("global_store_dwordx4 %0, %1, off sc1" :: "v"(p), "v"(v) : "memory");
}

__device__ __forceinline__ unsigned pk_bf16(float a, float b) {          // (rne(a), rne(b)) packed, a in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ void split_pair(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pk_bf16(a, b);
  lo = pk_bf16(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xffff0000u));
}
__device__ __forceinline__ f32x16 mfma(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ int xcd_major(int bid, int g) {               // position when the grid is ordered by (XCD, arrival)
  const int x = bid & 7, s = bid >> 3;
  return x * (g >> 3) + min(x, g & 7) + s;
}
// The shifted segment (B1): source row m - shift exists iff the position t = m % period of row m in its sequence has 0 <= t - shift <
// period (the host passes period = M for "one sequence").  Until round 4 this was `cond ? load : zero` around a `%`: a divergent
// branch in the chunk loop; behind it the selects of the landed rows were scheduled into the half that had just REQUESTED them
// (each behind a s_waitcnt vmcnt(0)), and the LSTM layer's [2][2048 x 16384] x [16384 x 1536] took 777-825 us against 596 us with
// shift = 0 (profiles/r4/wgrad_shift_ab_v1.txt).  Now: the chunk's first position by one multiply-high on the scalar unit, one add
// and one unsigned compare per row, an UNCONDITIONAL load (of row m itself when row m - shift does not exist), a select on the
// landed registers when they are split, and a scheduling fence behind the workgroup barrier: 632 us.
__device__ __forceinline__ unsigned div_period(unsigned r, unsigned magic, int sh) {
  return (unsigned)(((unsigned long long)r * magic) >> sh);
}
__device__ __forceinline__ float4 keep_if(bool ok, float4 v) { return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f); }
__device__ __forceinline__ uint2 keep_if(bool ok, uint2 v) { return make_uint2(ok ? v.x : 0u, ok ? v.y : 0u); }

// Geometry of one kernel variant: WN x WK waves, each a 64 x 64 output block.
template <int WN, int WK>
struct Geo {
  static constexpr int TN = 64 * WN, TK = 64 * WK, NT = 64 * WN * WK;
  static constexpr int RA = 8 * TN / NT, RB = 8 * TK / NT;               // operand rows per thread and chunk (float4 each): 2 or 4
  static constexpr int kPlaneA = TN * P, kPlaneB = TK * P;               // dwords per LDS plane
  static constexpr int kBuf = 2 * kPlaneA + 2 * kPlaneB;                 // A_hi, A_lo, B_hi, B_lo
  static constexpr size_t kLds = sizeof(unsigned) * 2 * kBuf;            // double-buffered
  static_assert((RA == 2 || RA == 4) && (RB == 2 || RB == 4), "thread roles");
};

// a thread's rows of one operand tile: row group mg (4 rows), column quad c4, and for 2-row roles which half of the group
template <int R, int W>
struct Role {
  int mg, c4, sub;
  // lane bits [1:0] = column quad within a 64-byte segment, [4:2] = row group, [5..] = further segments: four neighbouring lanes
  // read 64 contiguous bytes (one L1 tag lookup per lane quad -- with the lanes running along the rows first every lane was its
  // own lookup: 64 per load instruction, and the L1 pipe, not the MFMA, set the pace), and the transposing LDS writes of one
  // instruction still spread over all banks: column offset 80 dwords = 16 banks per quad step, 2 dwords per row group.
  __device__ __forceinline__ explicit Role(int tid)
      : mg((tid >> 2) & 7), c4((tid & 3) + 4 * ((tid >> 5) % (W / 16))), sub((tid >> 5) / (W / 16)) {}
  __device__ __forceinline__ int row0() const { return 4 * mg + R * sub; }
  // split the R x 4 block and write it m-contiguous: column 4*c4 + j, dwords (4*mg + R*sub) / 2 ..
  // bf16 operands (tsg_wgrad_bf16): the R x 4 block arrives as R uint2 of four bf16; the transposition is a 16-bit shuffle, there is
  // one plane and nothing to convert
  __device__ __forceinline__ void write(const uint2 (&v)[R], unsigned* hi, unsigned*) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned e[R];
#pragma unroll
      for (int i = 0; i < R; ++i) {
        const unsigned d = j < 2 ? v[i].x : v[i].y;
        e[i] = (j & 1) ? (d >> 16) : (d & 0xffffu);
      }
      const int o = (4 * c4 + j) * P + 2 * mg + (R == 2 ? sub : 0);
      if constexpr (R == 4) *reinterpret_cast<uint2*>(hi + o) = make_uint2(e[0] | (e[1] << 16), e[2] | (e[3] << 16));
      else hi[o] = e[0] | (e[1] << 16);
    }
  }
  __device__ __forceinline__ void write(const float4 (&v)[R], unsigned* hi, unsigned* lo) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float e[R];
#pragma unroll
      for (int i = 0; i < R; ++i) e[i] = j == 0 ? v[i].x : j == 1 ? v[i].y : j == 2 ? v[i].z : v[i].w;
      const int o = (4 * c4 + j) * P + 2 * mg + (R == 2 ? sub : 0);
      if constexpr (R == 4) {
        unsigned h0, l0, h1, l1;
        split_pair(e[0], e[1], h0, l0);
        split_pair(e[2], e[3], h1, l1);
        *reinterpret_cast<uint2*>(hi + o) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(lo + o) = make_uint2(l0, l1);
      } else {
        unsigned h, l;
        split_pair(e[0], e[1], h, l);
        hi[o] = h; lo[o] = l;
      }
    }
  }
};

// Epilogue shared by the kernels of this file: the wave's 2 x 2 accumulator blocks (row = n, column = k on the lanes) go
//   * straight to C / C1 (the workgroup computed the whole contraction of the tile),
//   * to this row range's partial tile in the workspace (the split scheme; wgrad_reduce_kernel adds them), or
//   * stream-K: to the tile's partial slot `me` of `ncontrib` (write-through stores in register order, s_waitcnt vmcnt(0), barrier, one relaxed
//     agent-scope ticket -- the K3 / K1 backward's publication recipe, under the ISA gate of tests/test_isa_cpu.py); the LAST arriver reads the
//     other slots with device-coherent loads, adds the contributions in slot order -- its own from registers -- and writes C.  The sum order
//     is fixed by the slot index, so the result does not depend on who arrives last; no float atomics, no second kernel, and a tile's partial
//     data (128 KiB per slot) is written and read once instead of `splits` times through a reduce launch.
// a tile lies inside one column segment (K0 % 128 == 0); with a second output the B1 segment's tiles go to C1, columns from 0.
__device__ __forceinline__ void store_tile(const WgradArgs& a, const f32x16 (&acc)[2][2], float* out, long ldo, int n0, int k0, int wn, int wk, int lane) {
  const int r = lane & 31, hh = lane >> 5;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int n = n0 + 64 * wn + 32 * i + (q & 3) + 8 * (q >> 2) + 4 * hh;
        const int k = k0 + 64 * wk + 32 * j + r;
        out[(size_t)n * ldo + k] = acc[i][j][q];
      }
}
template <int NT, int TNK>                              // NT threads, TNK = TN * TK elements per tile
__device__ __forceinline__ void finish_tile(const WgradArgs& a, f32x16 (&acc)[2][2], int tile, int g, int n0, int k0, int wn, int wk,
                                            int me, int ncontrib, unsigned* flag) {
  const int tid = threadIdx.x, lane = tid & 63, K = a.K0 + a.K1;
  const bool to1 = a.C1 && k0 >= a.K0;
  float* out = to1 ? a.C1 + g * a.c1_gs - a.K0 : a.C + g * a.c_gs;
  const long ldo = to1 ? a.ldc1 : a.ldc;
  (void)K;
  if (ncontrib == 1) { store_tile(a, acc, out, ldo, n0, k0, wn, wk, lane); return; }
  // slot layout: [block ij (4)][float4 q4 (4)][thread][4] -> a wave instruction writes / reads 1 KiB contiguous; blocks q4 are NT * 16 bytes apart
  float* slot0 = a.part + (size_t)tile * kMaxContrib * TNK;
  float* mine = slot0 + (size_t)me * TNK + (size_t)tid * 4;
#pragma unroll
  for (int ij = 0; ij < 4; ++ij)
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
      const f32x16& c = acc[ij >> 1][ij & 1];
      st_agent_x4(mine + (size_t)(ij * 4 + q4) * NT * 4, (w_f32x4){c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]});
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                            // the partial tile is acknowledged before the ticket moves
  __syncthreads();
  if (tid == 0) *flag = __hip_atomic_fetch_add(a.tickets + tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(ncontrib - 1);
  __syncthreads();
  const bool last = *flag != 0u;
  __syncthreads();                                                             // (flag may be rewritten by the workgroup's next segment)
  if (!last) return;
#pragma unroll
  for (int ij = 0; ij < 4; ++ij) {
    f32x16& c = acc[ij >> 1][ij & 1];
    w_f32x4 sum[4];
    bool have = false;
    for (int j = 0; j < ncontrib; ++j) {                                       // slot order; slot `me` from the registers
      w_f32x4 v[4];
      if (j == me) {
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) v[q4] = (w_f32x4){c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]};
      } else {
        const float* src = slot0 + (size_t)j * TNK + (size_t)tid * 4 + (size_t)(ij * 4) * NT * 4;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)                                         // four device-coherent loads in flight, one wait
          asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[q4]) : "v"(src + (size_t)q4 * NT * 4) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]) :: "memory");
      }
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) sum[q4] = have ? sum[q4] + v[q4] : v[q4];
      have = true;
    }
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) { c[4 * q4] = sum[q4][0]; c[4 * q4 + 1] = sum[q4][1]; c[4 * q4 + 2] = sum[q4][2]; c[4 * q4 + 3] = sum[q4][3]; }
  }
  store_tile(a, acc, out, ldo, n0, k0, wn, wk, lane);
  if (tid == 0) __hip_atomic_store(a.tickets + tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the workspace leaves as it came: zero tickets
}

template <int WN, int WK, int SHIFTED, typename ET>     // SHIFTED: 0 no; 1 period % 32 == 0 (a chunk lies inside one sequence); 2 any period
__device__ __forceinline__ void wgrad_tile(const WgradArgs& a, unsigned* lds, long c_begin, int nc, int g, int n0, int k0, f32x16 (&acc)[2][2]) {
  using G = Geo<WN, WK>;
  constexpr bool BF = storage_is_bf16<ET>::value;          // bf16 operands: one plane per operand, one MFMA per product
  typedef typename Raw4T<ET>::type Raw;
  struct Staged { Raw a[G::RA]; Raw b[G::RB]; bool ok[G::RB]; };   // one chunk's operand rows of a thread, in flight / waiting for the split (ok: the B row exists)
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  // chunks [c_begin, c_begin + nc) of the contraction (nc may be <= 0)

  // load roles (see Role)
  const Role<G::RA, G::TN> ra(tid);
  const Role<G::RB, G::TK> rb(tid);
  const bool seg1 = k0 >= a.K0;
  const long ldb = seg1 ? a.ldb1 : a.ldb0;
  const int shift = SHIFTED ? (int)(g ? -a.shift : a.shift) : 0;
  const unsigned brow0 = (unsigned)(c_begin * BM) + rb.row0();
  const ET* pa = static_cast<const ET*>(a.A) + g * a.a_gs + n0 + 4 * ra.c4 + (c_begin * BM + ra.row0()) * a.lda;
  const ET* bcol = (seg1 ? static_cast<const ET*>(a.B1) + g * a.b1_gs + (k0 - a.K0) : static_cast<const ET*>(a.B0) + k0) + 4 * rb.c4;
  const ET* pb = bcol + ((long)brow0 - shift) * ldb;
  const unsigned per = (unsigned)a.period;
  const long soff = (long)shift * ldb;

  auto request = [&](Staged& r, int c) {                   // chunk c of the range (clamped: the tail re-requests the last chunk)
    c = min(c, nc - 1);
    const ET* qa = pa + (long)c * BM * a.lda;
    const ET* qb = pb + (long)c * BM * ldb;
#pragma unroll
    for (int i = 0; i < G::RA; ++i) r.a[i] = ldraw4(qa + i * a.lda);
    unsigned tb = 0;
    if (SHIFTED == 1) {                                     // position of the chunk's first row in its sequence: scalar unit
      const unsigned m0 = __builtin_amdgcn_readfirstlane((unsigned)(c_begin + c) * BM);
      tb = m0 - div_period(m0, a.per_magic, a.per_sh) * per;
    }
#pragma unroll
    for (int i = 0; i < G::RB; ++i) {
      if (SHIFTED) {
        unsigned ts;                                        // position of the SOURCE row in the sequence; >= period (as unsigned) = outside
        if (SHIFTED == 1) ts = tb + (unsigned)(rb.row0() + i - shift);          // (period % 32 == 0: a chunk lies inside one sequence)
        else { const unsigned m = brow0 + (unsigned)c * BM + i; ts = m - div_period(m, a.per_magic, a.per_sh) * per - (unsigned)shift; }
        const bool ok = ts < per;
        r.b[i] = ldraw4(qb + i * ldb + (ok ? 0L : soff));   // a row that does not exist: read row m itself (always there)
        r.ok[i] = ok;
      } else {
        r.b[i] = ldraw4(qb + i * ldb);
      }
    }
  };
  auto stage = [&](const Staged& r, int buf) {             // registers -> split -> LDS planes of buffer `buf`
    unsigned* Ahi = lds + buf * G::kBuf; unsigned* Alo = Ahi + G::kPlaneA;
    unsigned* Bhi = Alo + G::kPlaneA;    unsigned* Blo = Bhi + G::kPlaneB;
    ra.write(r.a, Ahi, Alo);
    if (SHIFTED) {
      Raw b[G::RB];
#pragma unroll
      for (int i = 0; i < G::RB; ++i) b[i] = keep_if(r.ok[i], r.b[i]);
      rb.write(b, Bhi, Blo);
    } else {
      rb.write(r.b, Bhi, Blo);
    }
  };

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int wn = wv / WK, wk = wv % WK, r = lane & 31, hh = lane >> 5;
  const int ao = (64 * wn + r) * P + 4 * hh, bo = (64 * wk + r) * P + 4 * hh;
  auto compute = [&](int buf) {                            // 2 m-steps x (8 fragment reads, 12 MFMAs)
    const unsigned* Ahi = lds + buf * G::kBuf; const unsigned* Alo = Ahi + G::kPlaneA;
    const unsigned* Bhi = Alo + G::kPlaneA;    const unsigned* Blo = Bhi + G::kPlaneB;
#pragma unroll
    for (int ms = 0; ms < 2; ++ms) {
      u32x4 fa_h[2], fa_l[2], fb_h[2], fb_l[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa_h[t] = *reinterpret_cast<const u32x4*>(Ahi + ao + 32 * t * P + 8 * ms);
        fb_h[t] = *reinterpret_cast<const u32x4*>(Bhi + bo + 32 * t * P + 8 * ms);
        if constexpr (!BF) {
          fa_l[t] = *reinterpret_cast<const u32x4*>(Alo + ao + 32 * t * P + 8 * ms);
          fb_l[t] = *reinterpret_cast<const u32x4*>(Blo + bo + 32 * t * P + 8 * ms);
        }
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma(fa_h[i], fb_h[j], acc[i][j]);
          if constexpr (!BF) {
            acc[i][j] = mfma(fa_h[i], fb_l[j], acc[i][j]);
            acc[i][j] = mfma(fa_l[i], fb_h[j], acc[i][j]);
          }
        }
    }
  };

#ifdef TSG_WGRAD_TIMING
  unsigned long long tph[4] = {0, 0, 0, 0}, tm0 = __builtin_amdgcn_s_memtime(), tm1 = 0;
#define TSG_TICK(i) { tm1 = __builtin_amdgcn_s_memtime(); tph[i] += tm1 - tm0; tm0 = tm1; }
#else
#define TSG_TICK(i) {}
#endif
  auto interleave = [&]() {
    if constexpr (BF) return;                              // (the recipe below is counted for the 24-MFMA fp32 chunk)
#ifndef TSG_WGRAD_NO_SGB
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                    // fragments of the first m-step
#pragma unroll
    for (int i = 0; i < 8; ++i) {                                        // MFMAs 0-7: the second m-step's fragments
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {                                        // MFMAs 8-13: the requests two chunks ahead
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, SHIFTED ? 7 : 4, 0);     // (shifted rows: position, validity and address select per row)
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) {                                       // MFMAs 14-23: the next chunk's planes go to LDS
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, SHIFTED ? 5 : 4, 0);
    }
#endif
  };
  if (nc > 0) {
    // two chunks in flight: the rows requested at the top of an iteration are split and written to LDS at the end of the NEXT
    // one (an L2 round trip under load is longer than one chunk's 24 MFMAs per wave).  No conditionals in the loop: the tail
    // re-requests the last chunk and stages it into a buffer nobody reads any more.
    Staged s0, s1;
    request(s0, 0);
    request(s1, 1);
    stage(s0, 0);
    __syncthreads();
    // One basic block per chunk: 24 MFMAs on the current buffer, the requests two chunks ahead and the split of the next chunk
    // into the other buffer are independent, and sched_group_barrier lays them out as ONE interleaved stream -- after every
    // MFMA (8 issue cycles of its 32) a fragment read / a load / an LDS write and 3-4 conversion instructions -- so that a wave
    // hides its own memory and VALU work under its own MFMAs.  (With the three pieces one after the other the two waves of a
    // SIMD, which meet at every barrier, were in the same phase at the same time: MFMA time + VALU time + L1 issue time per
    // chunk, matrix pipe 30 % busy.)
#ifndef TSG_WGRAD_ABL
#define TSG_WGRAD_ABL 0
#endif
#define ABL_REQ(x) { if (!(TSG_WGRAD_ABL & 2)) { x; } }
#define ABL_CMP(x) { if (!(TSG_WGRAD_ABL & 8)) { x; } }
#define ABL_STG(x) { if (!(TSG_WGRAD_ABL & 4)) { x; } }
// (sched_barrier behind the workgroup barrier: nothing of the next half may be scheduled into this one.  Without it the selects of the
// shifted segment -- ready as soon as their load is ISSUED -- filled VALU slots of the half that requested them, each behind a
// s_waitcnt vmcnt(0): a full memory round trip per chunk in the B1 tiles.)
#define ABL_BAR() { if (!(TSG_WGRAD_ABL & 1)) __syncthreads(); __builtin_amdgcn_sched_barrier(0); }
    for (int c = 0; c + 1 < nc; c += 2) {
      ABL_REQ(request(s0, c + 2)) TSG_TICK(0)
      ABL_CMP(compute(0)) TSG_TICK(1)
      ABL_STG(stage(s1, 1)) TSG_TICK(2)
      interleave();
      ABL_BAR() TSG_TICK(3)
      ABL_REQ(request(s1, c + 3)) TSG_TICK(0)
      ABL_CMP(compute(1)) TSG_TICK(1)
      ABL_STG(stage(s0, 0)) TSG_TICK(2)
      interleave();
      ABL_BAR() TSG_TICK(3)
    }
    if (nc & 1) compute(0);
  }

#ifdef TSG_WGRAD_TIMING
  __syncthreads();
  if (blockIdx.x == 0 && lane == 0)                        // cycles per chunk and phase: request, compute, stage, barrier
    for (int i = 0; i < 4; ++i) a.C[wv * 4 + i] = (float)(tph[i] / (unsigned long long)max(nc, 1));
#endif
}

template <int WN, int WK, typename ET>
__device__ __forceinline__ void wgrad_segment(const WgradArgs& a, unsigned* lds, long c_begin, int nc, int g, int n0, int k0, f32x16 (&acc)[2][2]) {
  if (k0 >= a.K0 && a.shift != 0) {                                                            // workgroup-uniform
    if (a.period % BM == 0) wgrad_tile<WN, WK, 1, ET>(a, lds, c_begin, nc, g, n0, k0, acc);
    else wgrad_tile<WN, WK, 2, ET>(a, lds, c_begin, nc, g, n0, k0, acc);
  } else {
    wgrad_tile<WN, WK, 0, ET>(a, lds, c_begin, nc, g, n0, k0, acc);
  }
}

template <int WN, int WK, typename ET>
__global__ __launch_bounds__(64 * WN * WK) void wgrad_split_kernel(const WgradArgs a) {
  extern __shared__ __align__(16) unsigned lds[];
  __shared__ unsigned s_flag;
  using G = Geo<WN, WK>;
  const int K = a.K0 + a.K1, tiles_k = K / G::TK, tpg = (a.N / G::TN) * tiles_k, tps = tpg * a.groups;
  const int v = xcd_major(blockIdx.x, gridDim.x);
  const long chunks = a.M / BM;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wn = wv / WK, wk = wv % WK;
  if (__builtin_amdgcn_readfirstlane(wv) >= 4) __builtin_amdgcn_s_setprio(1);     // static priority for the younger wave of each SIMD (see gemm_f32s.hip)
  f32x16 acc[2][2];
  // The workgroup's piece of the flattened (tile, chunk) space.  stream-K: [v * ipw, (v + 1) * ipw) = at most kMaxContrib - 1 tile segments;
  // split scheme: ONE segment, row range `split` of tile v % tps (possibly empty: its partial tile is still written, as zeros).
  long cur, end;
  bool once = a.ipw == 0;
  if (a.ipw > 0) {
    cur = (long)v * a.ipw;
    end = min((long)tps * chunks, cur + a.ipw);
  } else {
    const int split = v / tps;
    cur = (long)(v % tps) * chunks + min(chunks, (long)split * a.cps);
    end = (long)(v % tps) * chunks + min(chunks, (long)(split + 1) * a.cps);
  }
  while (cur < end || once) {                                                                  // workgroup-uniform
    once = false;
    const int tl = a.ipw > 0 ? (int)(cur / chunks) : v % tps;                                   // (split scheme: the tile is fixed, also for an empty row range)
    const long c0 = cur - (long)tl * chunks;
    const int nc = (int)max(0L, min(chunks - c0, end - cur));
    const int g = tl / tpg, tile = tl % tpg;
    const int n0 = (tile / tiles_k) * G::TN, k0 = (tile % tiles_k) * G::TK;
    wgrad_segment<WN, WK, ET>(a, lds, c0, nc, g, n0, k0, acc);
    if (a.ipw > 0) {
      const int first = (int)(((long)tl * chunks) / a.ipw), lastw = (int)(((long)(tl + 1) * chunks - 1) / a.ipw);
      __syncthreads();                                                                         // every wave has left the operand buffers (and s_flag)
      finish_tile<G::NT, G::TN * G::TK>(a, acc, tl, g, n0, k0, wn, wk, v - first, lastw - first + 1, &s_flag);
    } else {
      const bool to1 = a.splits == 1 && a.C1 && k0 >= a.K0;
      float* out = a.splits == 1 ? (to1 ? a.C1 + g * a.c1_gs - a.K0 : a.C + g * a.c_gs) : a.ws + ((size_t)(v / tps) * a.groups + g) * (size_t)a.N * K;
      store_tile(a, acc, out, a.splits == 1 ? (to1 ? a.ldc1 : a.ldc) : K, n0, k0, wn, wk, lane);
    }
    cur += max(nc, 1);
  }
}

// C[g][n][k] = sum_s ws[s][g][n][k], float4 per thread
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ C, long ldc, long c_gs,
                                                           int N, int K, int groups, int splits, float* __restrict__ C1, long ldc1,
                                                           long c1_gs, int K0) {
  const long per = (long)N * K / 4, total = per * groups;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += gridDim.x * 256L) {
    const int g = (int)(i / per);
    const long e = (i % per) * 4, n = e / K, k = e % K;
    float4 s = *reinterpret_cast<const float4*>(ws + (size_t)g * N * K + e);
    for (int p = 1; p < splits; ++p) {
      const float4 t = *reinterpret_cast<const float4*>(ws + ((size_t)p * groups + g) * (size_t)N * K + e);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
    if (C1 && k >= K0) *reinterpret_cast<float4*>(C1 + g * c1_gs + n * ldc1 + (k - K0)) = s;      // the B1 segment's columns: second output
    else *reinterpret_cast<float4*>(C + g * c_gs + n * ldc + k) = s;
  }
}

// Kernel variant.  0 (default): 4 x 2 waves, 256 x 128 tile, one workgroup per CU (LDS 120 KiB); 1: 2 x 2 waves, 128 x 128 tile,
// two workgroups per CU (80 KiB each: tools/ubench/lds_occupancy.hip -- 2 x 81 920 B is the most two workgroups can hold).
// Measured at [1024 x 16384] x [16384 x 1024]: 121.6 us vs 134.8 us -- the smaller tile moves a third more operand bytes through
// L1 and LDS per MFMA and converts a third more elements.  TSG_WGRAD_CFG selects (A/B).
int variant() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("TSG_WGRAD_CFG"); v = e ? atoi(e) : 0; if (v < 0 || v > 1) v = 0; }
  return v;
}
struct Plan { int tn, tk, slots, tiles, splits; long long ws; long ipw; int grid; long long ticket_bytes; };
std::atomic<int> g_stream_k{-2};           // -2: not decided yet (TSG_WGRAD_SK); -1 auto, 0 never, 1 always; tsg_wgrad_set_stream_k overrides
int stream_k_mode() {
  int v = g_stream_k.load(std::memory_order_relaxed);
  if (v == -2) { const char* e = getenv("TSG_WGRAD_SK"); v = e ? (atoi(e) != 0) : -1; g_stream_k.store(v, std::memory_order_relaxed); }
  return v;
}

// number of row ranges: the smallest power of two (<= 16, >= 4 chunks per range) that minimises the number of full-chip rounds
// per unit of work
int choose_splits(long chunks, int tiles, int slots) {
  int best = 1; double best_cost = 1e30;
  for (int s = 1; s <= 16; s *= 2) {
    if (chunks / s < 4 && s > 1) break;
    const double rounds = (double)((tiles * s + slots - 1) / slots);
    const double cost = rounds / s + (s > 1 ? 0.02 : 0.0) + 0.002 * s;      // + the reduce pass / prologue per workgroup
    if (cost < best_cost - 1e-9) { best_cost = cost; best = s; }
  }
  return best;
}

int check_args(const char* fn, long long M, int N, int K0, int K1, int groups) {
  if (M <= 0 || N <= 0 || K0 < 0 || K1 < 0 || K0 + K1 <= 0 || groups < 1 || groups > 2)
    return set_error(TSG_E_SHAPE, "%s: M=%lld N=%d K0=%d K1=%d groups=%d", fn, M, N, K0, K1, groups);
  if (M % BM || N % 256 || K0 % 128 || K1 % 128)
    return set_error(TSG_E_SHAPE, "%s: needs M %% %d == 0, N %% 256 == 0, K0 and K1 %% 128 == 0 (M=%lld N=%d K0=%d K1=%d)", fn, BM,
                     M, N, K0, K1);
  return 0;
}

Plan make_plan(long long M, int N, int K, int groups, bool bf) {
  Plan p;
  const int v = variant();
  p.tn = v == 0 ? 256 : 128; p.tk = 128; p.slots = v == 0 ? 256 : 512;
  p.tiles = groups * (N / p.tn) * (K / p.tk);
  p.splits = choose_splits(M / BM, p.tiles, p.slots);
  p.ws = p.splits == 1 ? 0 : (long long)sizeof(float) * p.splits * groups * N * K;
  p.ipw = 0; p.grid = p.tiles * p.splits; p.ticket_bytes = 0;
  // stream-K (round 5) where the tile count does not fill the chip evenly but is at least half of it (then a tile has at most kMaxContrib
  // contributors): the LSTM layer's [2][2048 x 16384] x [16384 x 1536] = 192 tiles ran as 4 row ranges = 768 workgroups in 3 rounds with 100 MB
  // of partial tiles out and back through a reduce launch; as ONE round of 256 workgroups of 384 chunks each it writes / reads the 128 KiB
  // slots of the ~190 tile boundaries once.  TSG_WGRAD_SK=0: the split scheme (A/B).
  // Measured (profiles/r5/wgrad_stream_k_ab_v1.txt, one MI355X, alternating processes): bf16 operands 302-311 -> 288-296 us at the LSTM layer's
  // shape (the 3-round split scheme's fixed costs are 5 % of a 300 us kernel); fp32 operands (f32s) 707-719 -> 716-727 us -- there the reduce
  // pass was never the bottleneck (the 3-product chunk loop is), and the two-segment workgroups' second prologue costs what the saved traffic
  // gains.  Default: stream-K for bf16 operands only.  TSG_WGRAD_SK=1 / 0 or tsg_wgrad_set_stream_k: always / never (A/B, tests).
  const int sk_mode = stream_k_mode();
  const bool sk_on = sk_mode >= 0 ? sk_mode != 0 : bf;
  const long chunks = (long)(M / BM);
  if (sk_on && p.tn == 256 && p.tiles % p.slots != 0 && 2 * p.tiles >= p.slots && chunks >= 16) {
    p.grid = p.slots;
    p.ipw = ((long)p.tiles * chunks + p.grid - 1) / p.grid;
    p.splits = 1;
    p.ticket_bytes = roundup((long long)p.tiles * 4, 256);
    p.ws = p.ticket_bytes + (long long)sizeof(float) * p.tiles * kMaxContrib * p.tn * p.tk;
  }
  return p;
}

// ---- bf16 operands, round 4: LDS-DMA staging + transposed fragment reads ------------------------------------------------------
// With bf16 operands (tsg_wgrad_bf16) the kernel above is bound by its operand path, not by the MFMAs (one product instead of three:
// 8 MFMAs per wave and chunk; ablation at the LSTM layer's shape: 303 us, 208 without the loads, 252 without the MFMAs).  bf16 rows need
// no conversion, so they can go global -> LDS by DMA (global_load_lds_dwordx4: no registers, no VALU, no ds_write) as they are stored --
// [32 rows m][128 columns] images with 256-byte rows -- and the m-contiguous MFMA fragments come out of ds_read_b64_tr_b16 (gfx950's
// transposing LDS read: per 16 lanes a 4-row x 16-column block, delivered column-major), two per fragment.
// Image = cdna_hip_programming.md T10 (b): 16-byte chunk ch of row r at 256 r + 16 (ch ^ (((r & 3) << 2) | ((r >> 2) & 3))); the DMA writes
// LDS lane-linearly, so the XOR is applied to the SOURCE chunk each lane fetches.  Tile 256 (n) x 128 (k) as above = three images per
// chunk (A columns 0-127, A columns 128-255, B); ring of three chunk buffers, DMA two chunks ahead, one raw s_barrier per chunk behind a
// counted s_waitcnt vmcnt (a wave waits for ITS three DMA instructions of the chunk, the barrier covers the others').
// Rows of the shifted segment that do not exist are fetched from a page of zeros.  Requires period % 32 == 0 (else the kernel above).
constexpr int kTrImg = 8192, kTrSub = 3 * kTrImg;                      // one 32-row sub-chunk: A columns 0-127 | A columns 128-255 | B
__device__ const uint4 g_wgrad_zero_page[16] = {};                     // 256 bytes of zeros

__device__ __forceinline__ int tr_swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }
// The DMA as inline asm: hipcc puts a s_waitcnt vmcnt(0) in front of every LDS read while a compiler-visible LDS-DMA is in flight (it
// cannot tell the ring buffers apart), which would drain the two chunks of look-ahead in every iteration.  It does not see this one;
// the kernel waits for it itself (counted s_waitcnt vmcnt + s_barrier).  The fragment reads stay compiler-visible (their lgkmcnt waits
// are the compiler's).  lds_addr: wave-uniform LDS byte address; the hardware adds 16 bytes per lane.
__device__ __forceinline__ void tr_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_addr) : "memory", "m0");
}
typedef short tr_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr_read(const char* img, int off) {
  const tr_s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr_s16x4*)(img + off));
  return __builtin_bit_cast(uint2, v);
}

template <int SUB, int NB>                      // SUB = 32-row sub-chunks per barrier interval (1 or 2), NB = ring depth (DMA runs NB - 1 intervals ahead)
__device__ __forceinline__ void tr_segment(const WgradArgs& a, char* ring, long c_begin, int nc, int g, int n0, int k0, f32x16 (&acc)[2][2]) {
  constexpr int kTrBuf = SUB * kTrSub, AHEAD = NB - 1;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  // chunks [c_begin, c_begin + nc) of the contraction (nc may be <= 0)
  const bool seg1 = k0 >= a.K0;
  const bool shifted = seg1 && a.shift != 0;
  const long ldb = seg1 ? a.ldb1 : a.ldb0;
  const int shift = shifted ? (int)(g ? -a.shift : a.shift) : 0;
  const unsigned per = (unsigned)a.period;

  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;     // LDS byte address of the ring
  // DMA role: the wave moves rows 4 wv .. 4 wv + 3 of the three images (one 1 KiB instruction each); lane -> (row, 16-byte slot)
  const int drow = 4 * wv + (lane >> 4), dch = (lane & 15) ^ tr_swz(drow);
  const bf16_t* pa = static_cast<const bf16_t*>(a.A) + g * a.a_gs + n0 + 8 * dch + (c_begin * BM + drow) * a.lda;
  const bf16_t* pb = (seg1 ? static_cast<const bf16_t*>(a.B1) + g * a.b1_gs + (k0 - a.K0) : static_cast<const bf16_t*>(a.B0) + k0) + 8 * dch +
                     (c_begin * BM + drow - shift) * ldb;
  const bf16_t* zero = reinterpret_cast<const bf16_t*>(g_wgrad_zero_page) + 8 * (lane & 15);
  auto dma = [&](int c, int buf) {                                     // interval c = sub-chunks SUB c .. SUB c + SUB - 1 of this row range
#pragma unroll
    for (int sc = 0; sc < SUB; ++sc) {
      const int cc = SUB * c + sc;
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)(buf * kTrBuf + sc * kTrSub + 4 * wv * 256));   // wave-uniform
      const bf16_t* qa = pa + (long)cc * BM * a.lda;
      const bf16_t* qb = pb + (long)cc * BM * ldb;
      if (shifted) {                                                     // workgroup-uniform
        const unsigned m0 = (unsigned)(c_begin + cc) * BM;
        const unsigned tb = m0 - div_period(m0, a.per_magic, a.per_sh) * per;
        const unsigned ts = tb + (unsigned)(drow - shift);               // position of the source row in its sequence; >= period: outside
        qb = ts < per ? qb : zero;
      }
      const bool beyond = SUB > 1 && cc >= nc;                           // a sub-chunk past the range (odd count): zeros contribute nothing
      tr_dma16(beyond ? zero : qa, dst);
      tr_dma16(beyond ? zero : qa + 128, dst + kTrImg);
      tr_dma16(beyond ? zero : qb, dst + 2 * kTrImg);
    }
  };

  // fragment addresses: lane = (16-lane group: column half, k half; q = row of the 4 x 16 block; p = 8-byte piece of the row's 32 bytes)
  const int wn = wv >> 1, wk = wv & 1;
  const int g16 = lane >> 4, chalf = g16 & 1, kg = g16 >> 1, fq = (lane & 15) >> 2, fp = lane & 3;
  int offA[2][2][2], offB[2][2][2];                                      // [tile][m step][half of the 8 rows]
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ms = 0; ms < 2; ++ms)
#pragma unroll
      for (int hf = 0; hf < 2; ++hf) {
        const int row = 16 * ms + 8 * kg + 4 * hf + fq, sw = tr_swz(row);
        const int cA = 64 * wn + 32 * t, cB = 64 * wk + 32 * t;
        offA[t][ms][hf] = (cA >> 7) * kTrImg + 256 * row + 16 * (((((cA & 127) + 16 * chalf) >> 3) + (fp >> 1)) ^ sw) + 8 * (fp & 1);
        offB[t][ms][hf] = 2 * kTrImg + 256 * row + 16 * ((((cB + 16 * chalf) >> 3) + (fp >> 1)) ^ sw) + 8 * (fp & 1);
      }

#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  const int ni = (nc + SUB - 1) / SUB;                                  // barrier intervals
  if (ni > 0) {
#pragma unroll
    for (int i = 0; i < AHEAD; ++i)
      if (i < ni) dma(i, i);
#pragma unroll 1
    for (int c = 0; c < ni; ++c) {
      // this wave's DMA instructions of interval c have landed: 3 SUB per interval, up to AHEAD - 1 later intervals stay in flight
      const int later = min(ni - 1 - c, AHEAD - 1);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * 3 * SUB) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(3 * SUB) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                      // ... and everybody else's; nobody reads buffer (c + AHEAD) % NB any more
      if (c + AHEAD < ni) dma(c + AHEAD, (c + AHEAD) % NB);
      const char* img = ring + (c % NB) * kTrBuf;
#pragma unroll
      for (int sc = 0; sc < SUB; ++sc)
#pragma unroll
        for (int ms = 0; ms < 2; ++ms) {
          u32x4 fa[2], fb[2];
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            const uint2 a0 = tr_read(img + sc * kTrSub, offA[t][ms][0]), a1 = tr_read(img + sc * kTrSub, offA[t][ms][1]);
            const uint2 b0 = tr_read(img + sc * kTrSub, offB[t][ms][0]), b1 = tr_read(img + sc * kTrSub, offB[t][ms][1]);
            fa[t] = (u32x4){a0.x, a0.y, a1.x, a1.y};
            fb[t] = (u32x4){b0.x, b0.y, b1.x, b1.y};
          }
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = mfma(fa[i], fb[j], acc[i][j]);
        }
    }
  }

}

template <int SUB, int NB>
__global__ __launch_bounds__(512) void wgrad_bf16_tr_kernel(const WgradArgs a) {
  extern __shared__ __align__(16) unsigned lds[];
  __shared__ unsigned s_flag;
  char* ring = reinterpret_cast<char*>(lds);
  const int K = a.K0 + a.K1, tiles_k = K / 128, tpg = (a.N / 256) * tiles_k, tps = tpg * a.groups;
  const int v = xcd_major(blockIdx.x, gridDim.x);
  const long chunks = a.M / BM;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wn = wv >> 1, wk = wv & 1;
  f32x16 acc[2][2];
  long cur, end;                                                           // the workgroup's piece of the (tile, chunk) space: see wgrad_split_kernel
  bool once = a.ipw == 0;
  if (a.ipw > 0) {
    cur = (long)v * a.ipw;
    end = min((long)tps * chunks, cur + a.ipw);
  } else {
    const int split = v / tps;
    cur = (long)(v % tps) * chunks + min(chunks, (long)split * a.cps);
    end = (long)(v % tps) * chunks + min(chunks, (long)(split + 1) * a.cps);
  }
  while (cur < end || once) {
    once = false;
    const int tl = a.ipw > 0 ? (int)(cur / chunks) : v % tps;
    const long c0 = cur - (long)tl * chunks;
    const int nc = (int)max(0L, min(chunks - c0, end - cur));
    const int g = tl / tpg, tile = tl % tpg;
    const int n0 = (tile / tiles_k) * 256, k0 = (tile % tiles_k) * 128;
    tr_segment<SUB, NB>(a, ring, c0, nc, g, n0, k0, acc);
    if (a.ipw > 0) {
      const int first = (int)(((long)tl * chunks) / a.ipw), lastw = (int)(((long)(tl + 1) * chunks - 1) / a.ipw);
      __syncthreads();                                                     // every wave has left the ring before the next segment's DMA
      finish_tile<512, 256 * 128>(a, acc, tl, g, n0, k0, wn, wk, v - first, lastw - first + 1, &s_flag);
    } else {
      const bool to1 = a.splits == 1 && a.C1 && k0 >= a.K0;
      float* out = a.splits == 1 ? (to1 ? a.C1 + g * a.c1_gs - a.K0 : a.C + g * a.c_gs) : a.ws + ((size_t)(v / tps) * a.groups + g) * (size_t)a.N * K;
      store_tile(a, acc, out, a.splits == 1 ? (to1 ? a.ldc1 : a.ldc) : K, n0, k0, wn, wk, lane);
    }
    cur += max(nc, 1);
  }
}

template <int WN, int WK, typename ET>
int launch(const char* fn, const WgradArgs& a, int grid, hipStream_t st) {
  using G = Geo<WN, WK>;
  hipError_t e = allow_lds(wgrad_split_kernel<WN, WK, ET>, G::kLds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e));
  hipLaunchKernelGGL((wgrad_split_kernel<WN, WK, ET>), dim3(grid), dim3(G::NT), G::kLds, st, a);
  return check_launch(fn);
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_wgrad_set_stream_k(int mode) { g_stream_k.store(mode < 0 ? -1 : (mode != 0), std::memory_order_relaxed); return 0; }

extern "C" long long tsg_wgrad_f32s_ws_bytes(long long M, int N, int K0, int K1, int groups) {
  if (check_args("tsg_wgrad_f32s_ws_bytes", M, N, K0, K1, groups)) return -1;
  const long long a = make_plan(M, N, K0 + K1, groups, false).ws, b = make_plan(M, N, K0 + K1, groups, true).ws;   // either operand type
  return a > b ? a : b;
}

static int wgrad_impl(const char* fn, bool bf, const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                      const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift,
                      long long period, void* C, long long ldc, long long c_group_stride, void* ws, long long ws_bytes,
                      long long M, int N, int groups, void* stream, void* C1 = nullptr, long long ldc1 = 0, long long c1_group_stride = 0) {
  int rc = check_args(fn, M, N, K0, K1, groups);
  if (C1 && (K1 <= 0 || (ldc1 & 3) || (c1_group_stride & 3) || ldc1 < K1 || ldc < K0 || !aligned16(C1)))
    return set_error(TSG_E_SHAPE, "%s: second output needs K1 > 0, ldc1 >= K1, ldc >= K0, strides %% 4 == 0, 16-byte alignment", fn);
  if (rc) return rc;
  if (!A || !C || (K0 > 0 && !B0) || (K1 > 0 && !B1)) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  const int K = K0 + K1;
  if ((lda & 3) || (ldb0 & 3) || (ldb1 & 3) || (ldc & 3) || (a_group_stride & 3) || (b1_group_stride & 3) || (c_group_stride & 3) ||
      lda < (groups - 1) * a_group_stride + N || (K0 > 0 && ldb0 < K0) || (K1 > 0 && ldb1 < (groups - 1) * b1_group_stride + K1) ||
      ldc < (C1 ? K0 : K) || period < 0)
    return set_error(TSG_E_SHAPE, "%s: leading dimensions / group strides must be multiples of 4 and cover the operands", fn);
  if (period > 0 && M % period) return set_error(TSG_E_SHAPE, "%s: M=%lld is not a multiple of period=%lld", fn, M, period);
  if (M >= (1LL << 31) || shift >= (1LL << 31) || shift <= -(1LL << 31) || period >= (1LL << 31))
    return set_error(TSG_E_SHAPE, "%s: M, shift and period must fit 31 bits", fn);
  if (!aligned16(A) || !aligned16(C) || (B0 && !aligned16(B0)) || (B1 && !aligned16(B1)))
    return set_error(TSG_E_ALIGN, "%s: operands must be 16-byte aligned", fn);
  const Plan p = make_plan(M, N, K, groups, bf);
  if (p.ws > 0 && (!ws || ws_bytes < p.ws || !aligned16(ws)))
    return set_error(TSG_E_SHAPE, "%s: workspace of %lld bytes (16-byte aligned) required, got %lld", fn, p.ws, ws_bytes);
  auto st = static_cast<hipStream_t>(stream);
  const long chunks = M / BM;
  WgradArgs a;
  a.A = A; a.lda = lda; a.a_gs = a_group_stride;
  a.B0 = B0; a.ldb0 = ldb0; a.K0 = K0;
  a.B1 = B1; a.ldb1 = ldb1; a.b1_gs = b1_group_stride; a.K1 = K1; a.shift = shift; a.period = period > 0 ? period : M;   // 0 = one sequence of M rows
  {                                                       // floor(r / period) for r < 2^31: (r * ceil(2^(31+l) / period)) >> (31 + l), 2^l >= period
    int l = 0;
    while ((1LL << l) < a.period) ++l;
    a.per_sh = 31 + l;
    a.per_magic = (unsigned)((((unsigned __int128)1 << a.per_sh) + a.period - 1) / a.period);
  }
  a.C = (float*)C; a.ldc = ldc; a.c_gs = c_group_stride; a.ws = (float*)ws;
  a.C1 = (float*)C1; a.ldc1 = ldc1; a.c1_gs = c1_group_stride;
  a.M = M; a.N = N; a.groups = groups; a.splits = p.splits; a.cps = (int)((chunks + p.splits - 1) / p.splits);
  a.ipw = p.ipw; a.tickets = (unsigned*)ws; a.part = p.ipw > 0 ? (float*)((char*)ws + p.ticket_bytes) : nullptr;
  if (p.ipw > 0) {                                          // the tickets start at zero (the kernel leaves them at zero, but the caller's buffer is fresh)
    hipError_t e = zero_async(ws, (size_t)p.ticket_bytes, st);
    if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
  }
  static const bool tr_ok = !(getenv("TSG_WGRAD_BF16_TR") && atoi(getenv("TSG_WGRAD_BF16_TR")) == 0);    // A/B switch: 0 = the register-staged kernel
  if (bf && tr_ok && p.tn == 256 && (shift == 0 || K1 == 0 || a.period % BM == 0) && (lda & 7) == 0 && (ldb0 & 7) == 0 && (ldb1 & 7) == 0 &&
      (a_group_stride & 7) == 0 && (b1_group_stride & 7) == 0) {
    static const int cfg = getenv("TSG_WGRAD_TR_CFG") ? atoi(getenv("TSG_WGRAD_TR_CFG")) : 0;      // developer A/B: (sub-chunks, ring depth)
#define TSG_TR_LAUNCH(SUBV, NBV) { auto kern = wgrad_bf16_tr_kernel<SUBV, NBV>; const size_t lb = (size_t)NBV * SUBV * kTrSub;          \
      hipError_t e = allow_lds(kern, lb); if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute: %s", fn, hipGetErrorString(e)); \
      hipLaunchKernelGGL(kern, dim3(p.grid), dim3(512), lb, st, a); }
    if (cfg == 1) TSG_TR_LAUNCH(1, 4) else if (cfg == 2) TSG_TR_LAUNCH(2, 3) else if (cfg == 3) TSG_TR_LAUNCH(2, 2) else TSG_TR_LAUNCH(1, 3)
#undef TSG_TR_LAUNCH
    rc = check_launch(fn);
  } else if (bf) rc = p.tn == 256 ? launch<4, 2, bf16_t>(fn, a, p.grid, st) : launch<2, 2, bf16_t>(fn, a, p.grid, st);
  else rc = p.tn == 256 ? launch<4, 2, float>(fn, a, p.grid, st) : launch<2, 2, float>(fn, a, p.grid, st);
#ifdef TSG_WGRAD_TIMING
  return rc;
#endif
  if (rc || p.splits == 1) return rc;
  const long total = (long)groups * N * K / 4;
  const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid), dim3(256), 0, st, (const float*)ws, (float*)C, (long)ldc, (long)c_group_stride,
                     N, K, groups, p.splits, (float*)C1, (long)ldc1, (long)c1_group_stride, K0);
  return check_launch(fn);
}

extern "C" int tsg_wgrad_f32s(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                              const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift,
                              long long period, void* C, long long ldc, long long c_group_stride, void* ws, long long ws_bytes,
                              long long M, int N, int groups, void* stream) {
  return wgrad_impl("tsg_wgrad_f32s", false, A, lda, a_group_stride, B0, ldb0, K0, B1, ldb1, b1_group_stride, K1, shift, period, C, ldc,
                    c_group_stride, ws, ws_bytes, M, N, groups, stream);
}

// Two outputs: the columns of the B0 segment to C (C[g][n][k], k < K0, row stride ldc >= K0), those of the B1 segment to C1
// (C1[g][n][k - K0], row stride ldc1 >= K1) -- the LSTM layer's dW_ih and dW_hh leave the kernel as the two parameter-shaped tensors
// autograd wants, instead of one [2][4h][I + h] block the host then slices and copies (4 x 2 copies of 16 + 8 MB per step).
extern "C" int tsg_wgrad_f32s_out2(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                                   const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift,
                                   long long period, void* C, long long ldc, long long c_group_stride, void* C1, long long ldc1,
                                   long long c1_group_stride, void* ws, long long ws_bytes, long long M, int N, int groups, void* stream) {
  if (!C1) return set_error(TSG_E_NULL, "tsg_wgrad_f32s_out2: NULL second output");
  return wgrad_impl("tsg_wgrad_f32s_out2", false, A, lda, a_group_stride, B0, ldb0, K0, B1, ldb1, b1_group_stride, K1, shift, period, C, ldc,
                    c_group_stride, ws, ws_bytes, M, N, groups, stream, C1, ldc1, c1_group_stride);
}

// The same product for the bf16 storage mode: A, B0, B1 are bf16 matrices (strides in elements; rows 8-byte aligned), C and the
// workspace fp32 -- one bf16 MFMA per product, operands transposed through LDS with 16-bit shuffles, nothing converted.
extern "C" int tsg_wgrad_bf16(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                              const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift,
                              long long period, void* C, long long ldc, long long c_group_stride, void* ws, long long ws_bytes,
                              long long M, int N, int groups, void* stream) {
  return wgrad_impl("tsg_wgrad_bf16", true, A, lda, a_group_stride, B0, ldb0, K0, B1, ldb1, b1_group_stride, K1, shift, period, C, ldc,
                    c_group_stride, ws, ws_bytes, M, N, groups, stream);
}

// tsg_wgrad_f32s_out2 for bf16 operands (the bf16 storage mode's LSTM layers): dW_ih / dW_hh as two fp32 outputs from one launch.
extern "C" int tsg_wgrad_bf16_out2(const void* A, long long lda, long long a_group_stride, const void* B0, long long ldb0, int K0,
                                   const void* B1, long long ldb1, long long b1_group_stride, int K1, long long shift,
                                   long long period, void* C, long long ldc, long long c_group_stride, void* C1, long long ldc1,
                                   long long c1_group_stride, void* ws, long long ws_bytes, long long M, int N, int groups, void* stream) {
  if (!C1) return set_error(TSG_E_NULL, "tsg_wgrad_bf16_out2: NULL second output");
  return wgrad_impl("tsg_wgrad_bf16_out2", true, A, lda, a_group_stride, B0, ldb0, K0, B1, ldb1, b1_group_stride, K1, shift, period, C, ldc,
                    c_group_stride, ws, ws_bytes, M, N, groups, stream, C1, ldc1, c1_group_stride);
}
