// Dense projection GEMMs of the bf16 STORAGE mode (dtype TSG_BF16; BASELINE configs 2 / 4 name bf16):
//     Y[m][n] = sum_k X[m][k] W[n][k] (+ bias[n])          X [M,K] bf16, W [N,K] bf16 (an nn.Linear weight as stored), fp32 accumulate,
//     Y bf16 (activations) or fp32 (logits / pre-activations the fp32 heads read)
// Reference sites: the Linears of the matching path -- W_s / W_a (networks/attention.py:104-113), sent_linear (components/VideoEncoder.py:59),
// the heads' first Linear (components/SpanPredictor.py:71-85, components/DistributionAlign.py:83-118), nn.LSTM's input projection and its
// input gradient (networks/RNN.py:31,42).  Round-4 review, "missing" #1: in the bf16 mode these products were torch.mm -> hipBLASLt.
//
// The split-precision kernel (gemm_f32s.hip) converts fp32 rows to (hi, lo) bf16 planes in registers on their way into LDS.  bf16 operands need
// no conversion, so nothing here touches a VGPR between HBM and the MFMA operand read:
//   * global -> LDS by DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes = 16 rows x 64 bytes of a 32-deep K chunk per instruction),
//     issued three chunks ahead into a ring of four buffers (TM x 64 B of X + 256 x 64 B of W per buffer);
//   * both operands are K-contiguous ("NT"), so an MFMA fragment (8 consecutive k of one row) is ONE ds_read_b128 of the row image; rows are
//     64 bytes = 16 banks, so the 16-byte piece p of row r sits in slot p ^ ((r >> 2) & 3) -- sixteen lanes of a b128 read then cover all 64
//     banks once.  The DMA writes LDS lane-linearly, so the XOR is applied to the SOURCE piece each lane fetches;
//   * one raw s_barrier per chunk behind a counted s_waitcnt vmcnt (a wave waits for ITS DMA instructions of the chunk, the barrier covers the
//     others'); 16 v_mfma_f32_32x32x16_bf16 per wave and chunk (tile 256 x 256: 8 waves as 2 x 4, each 128 x 64 = 4 x 2 MFMA tiles).
// Tile geometry, XCD-aware tile order and the persistent tile walk are those of gemm_nt_f32s_kernel.
#include <cstdlib>
#include <type_traits>

#include "tsg_common.h"

namespace tsg {
namespace {

typedef float b_f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 b_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned b_u32x4 __attribute__((ext_vector_type(4)));

constexpr int kBT = 512;                         // threads
constexpr int kBN = 256;                         // N tile
// KC = K elements per chunk (32: 64-byte rows, ring of 4; 64: 128-byte rows, ring of 2 -- half the barriers, one chunk of look-ahead)
template <int KC> constexpr int ring_depth() { return KC == 32 ? 4 : 2; }
template <int TM, int KC> constexpr int buf_bytes() { return (TM + kBN) * KC * 2; }
template <int TM, int KC> constexpr size_t bgemm_lds() { return (size_t)ring_depth<KC>() * buf_bytes<TM, KC>(); }
#ifndef TSG_BGEMM_ABL
#define TSG_BGEMM_ABL 0                          // timing-only ablations: 1 no DMA, 2 no MFMAs (fragment reads kept), 4 no fragment reads either
#endif

__device__ __forceinline__ b_f32x16 b_mfma(b_u32x4 a, b_u32x4 b, b_f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b_bf16x8, a), __builtin_bit_cast(b_bf16x8, b), c, 0, 0, 0);
}
// LDS-DMA as inline asm (the compiler would drain vmcnt in front of every LDS read while a DMA it can see is in flight: see wgrad_split.hip).
// lds_addr: wave-uniform LDS byte address; the hardware adds 16 bytes per lane.
__device__ __forceinline__ void b_dma16(const void* src, unsigned lds_addr) {
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" :: "v"(src), "s"(lds_addr) : "memory", "m0");
}

template <int TM, int KC, bool OUT_BF16>
__global__ __launch_bounds__(kBT) void gemm_nt_bf16_kernel(const bf16_t* __restrict__ X, long long ldx, const bf16_t* __restrict__ W, long long ldw,
                                                           const float* __restrict__ bias, void* __restrict__ Yv, long long ldy,
                                                           int M, int N, int K, int tiles_n) {
  static_assert(TM == 256 || TM == 128, "every wave issues the same number of DMA instructions per chunk");
  static_assert(KC == 32 || KC == 64, "chunk depth");
  constexpr int MI = TM / 64;                                          // MFMA row tiles per wave
  constexpr int ROWB = KC * 2;                                         // bytes of one row of a chunk image
  constexpr int RPI = 1024 / ROWB;                                     // rows per DMA instruction (64 lanes x 16 bytes): 16 or 8
  constexpr int PPR = ROWB / 16;                                       // 16-byte pieces per row: 4 or 8
  constexpr int XI = TM / RPI / 8, WI = kBN / RPI / 8;                 // DMA instructions per wave and chunk for the X / W image (8 waves)
  constexpr int PER = XI + WI;
  constexpr int kNB = ring_depth<KC>(), kAhead = kNB - 1;              // the DMA runs kAhead chunks in front of the MFMAs
  constexpr int kBuf = buf_bytes<TM, KC>();
  extern __shared__ __align__(16) unsigned lds[];
  char* ring = reinterpret_cast<char*>(lds);
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)ring;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = (wv >> 2) * (TM / 2), wn = (wv & 3) * 64;
  const int jl = lane & 31, kg = lane >> 5;
  const int ntiles = (M / TM) * tiles_n, nk = K / KC;
  // piece p of row r sits in slot p ^ swz(r): 16 consecutive rows reading the same logical piece cover all 64 banks once
  auto swz = [](int r) { return KC == 32 ? (r >> 2) & 3 : (r >> 1) & 7; };
  // DMA role: instruction j of a wave moves rows RPI (wv + 8 j) .. + RPI - 1 of an image; lane -> (row, LDS slot); the slot holds source piece slot ^ swz(row)
  const int drow = lane / PPR, dslot = lane % PPR;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int m0, n0;
    if (tiles_n % 4 == 0 && (ntiles / tiles_n) % 8 == 0 && ntiles % 256 == 0) {      // an XCD's 32 concurrent tiles = 8 (M) x 4 (N) neighbours
      const int b = xcd_remap(tile, ntiles, 32), gq = b >> 5, r = b & 31, gn = tiles_n >> 2;
      m0 = ((gq / gn) * 8 + (r >> 2)) * TM; n0 = ((gq % gn) * 4 + (r & 3)) * kBN;
    } else {
      const int b = xcd_remap(tile, ntiles, tiles_n);
      m0 = (b / tiles_n) * TM; n0 = (b % tiles_n) * kBN;
    }
    const bf16_t* xsrc[XI]; const bf16_t* wsrc[WI];
#pragma unroll
    for (int j = 0; j < XI; ++j) {
      const int r = RPI * (wv + 8 * j) + drow;
      xsrc[j] = X + (size_t)(m0 + r) * ldx + 8 * (dslot ^ swz(r));
    }
#pragma unroll
    for (int j = 0; j < WI; ++j) {
      const int r = RPI * (wv + 8 * j) + drow;
      wsrc[j] = W + (size_t)(n0 + r) * ldw + 8 * (dslot ^ swz(r));
    }
    auto dma = [&](int c, int buf) {
      if (TSG_BGEMM_ABL & 1) return;
      const unsigned base = lds0 + (unsigned)(buf * kBuf);
#pragma unroll
      for (int j = 0; j < XI; ++j) b_dma16(xsrc[j] + (size_t)c * KC, __builtin_amdgcn_readfirstlane(base + 1024u * (unsigned)(wv + 8 * j)));
#pragma unroll
      for (int j = 0; j < WI; ++j) b_dma16(wsrc[j] + (size_t)c * KC, __builtin_amdgcn_readfirstlane(base + (unsigned)(TM * ROWB) + 1024u * (unsigned)(wv + 8 * j)));
    };
    auto frag = [&](const char* img, int r, int s) {                   // 8 consecutive k (k step s, half kg) of row r
      return *reinterpret_cast<const b_u32x4*>(img + r * ROWB + 16 * ((2 * s + kg) ^ swz(r)));
    };

    b_f32x16 acc[MI][2];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

#pragma unroll
    for (int c = 0; c < kAhead; ++c)
      if (c < nk) dma(c, c);
#pragma unroll 1
    for (int c = 0; c < nk; ++c) {
      // this wave's DMA instructions of chunk c have landed: PER per chunk, up to kAhead - 1 later chunks stay in flight
      const int later = min(nk - 1 - c, kAhead - 1);
      if (later >= 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * PER) : "memory");
      else if (later == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();                                    // ... and everybody else's; nobody reads buffer (c - 1) % kNB any more
      if (c + kAhead < nk) dma(c + kAhead, (c + kAhead) % kNB);
      const char* xt = ring + (c % kNB) * kBuf;
      const char* wt = xt + TM * ROWB;
      if (!(TSG_BGEMM_ABL & 4)) {
        // software-pipelined fragment reads: the (k step, row tile) pair after the current one is requested while the current pair's two MFMAs
        // run; the W fragments of the next k step ride along with the first two row tiles.  (In program order "reads, then their MFMAs" every
        // k step exposed an LDS round trip in front of its first MFMA.)
        constexpr int KS = KC / 16;
        b_u32x4 bc[2], bn[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) bc[j] = frag(wt, wn + 32 * j + jl, 0);
        b_u32x4 a = frag(xt, wm + jl, 0), an = a;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            if (i + 1 < MI) an = frag(xt, wm + 32 * (i + 1) + jl, s);
            else if (s + 1 < KS) an = frag(xt, wm + jl, s + 1);
            if (s + 1 < KS && i < 2) bn[i] = frag(wt, wn + 32 * i + jl, s + 1);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              if (TSG_BGEMM_ABL & 2) { acc[i][j][0] += __uint_as_float(a[0] ^ bc[j][1]); acc[i][j][1] += __uint_as_float(a[2] ^ bc[j][3]); }
              else acc[i][j] = b_mfma(a, bc[j], acc[i][j]);
            }
            a = an;
          }
          if (s + 1 < KS) { bc[0] = bn[0]; bc[1] = bn[1]; }
        }
#ifndef TSG_BGEMM_NO_SGB
        // lay the block out as written: three reads, then [two MFMAs | the reads of the pair after next] -- hipcc otherwise sinks every read
        // to just in front of its first use (lgkmcnt(0) in front of half the MFMAs)
        if (!(TSG_BGEMM_ABL & 2)) {
          __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
#pragma unroll
          for (int q = 0; q < KS * MI; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
          }
        }
#endif
      }
    }
    // epilogue.  Element r of tile (i, j): row m0 + wm + 32 i + 4 kg + (r & 3) + 8 (r >> 2), column n0 + wn + 32 j + jl.
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + 32 * j + jl;
      const float bv = bias ? bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const size_t row0 = (size_t)(m0 + wm + 32 * i + 4 * kg);
        if constexpr (OUT_BF16) {
          // neighbouring lanes hold neighbouring columns: the even lane takes the pair's rows r = 0, 2, ..., the odd lane r = 1, 3, ... and each
          // writes 4-byte (col, col + 1) pairs -- 8 dword stores per tile and lane instead of 16 two-byte ones
          bf16_t* Y = static_cast<bf16_t*>(Yv);
          const bool odd = jl & 1;
#pragma unroll
          for (int q = 0; q < 8; ++q) {
            const float mine0 = acc[i][j][2 * q] + bv, mine1 = acc[i][j][2 * q + 1] + bv;
            const float got = dpp_mov<0xB1>(odd ? mine0 : mine1);     // quad_perm [1,0,3,2]: the neighbour's value of the row THIS lane stores
            const int r = 2 * q + (odd ? 1 : 0);
            const unsigned pk = odd ? pack_bf16x2(got, mine1) : pack_bf16x2(mine0, got);
            *reinterpret_cast<unsigned*>(Y + (row0 + (r & 3) + 8 * (r >> 2)) * ldy + (col & ~1)) = pk;
          }
        } else {
          float* yp = static_cast<float*>(Yv) + row0 * ldy + col;
#pragma unroll
          for (int r = 0; r < 16; ++r) yp[(size_t)((r & 3) + 8 * (r >> 2)) * ldy] = acc[i][j][r] + bv;
        }
      }
    }
    __builtin_amdgcn_s_barrier();                                      // every wave has left the ring before the next tile's DMA lands in it
  }
}

template <int TM, int KC, bool OUT_BF16>
int launch_bgemm(const char* fn, const void* x, long long ldx, const void* w, long long ldw, const float* bias, void* y, long long ldy,
                 int M, int N, int K, hipStream_t st) {
  auto kern = gemm_nt_bf16_kernel<TM, KC, OUT_BF16>;
  constexpr size_t lds = bgemm_lds<TM, KC>();
  hipError_t e = allow_lds(kern, lds);
  if (e != hipSuccess) return set_error((int)e, "%s: hipFuncSetAttribute(%zu): %s", fn, lds, hipGetErrorString(e));
  const int tiles_n = N / kBN, tiles = (M / TM) * tiles_n, cus = device_cu_count();
  hipLaunchKernelGGL(kern, dim3(tiles < cus ? tiles : cus), dim3(kBT), lds, st, (const bf16_t*)x, ldx, (const bf16_t*)w, ldw, bias, y, ldy, M, N, K, tiles_n);
  return check_launch(fn);
}

}  // namespace
}  // namespace tsg

using namespace tsg;

// y [M,N] (bf16 if out_dtype == TSG_BF16, fp32 if TSG_F32; row stride ldy elements) = x [M,K] (bf16, row stride ldx) . w [N,K]^T (bf16, row stride
// ldw) + bias [N] (fp32 or NULL).  M % 128 == 0, N % 256 == 0, K % 32 == 0; ldx, ldw multiples of 8 (16-byte rows), ldy even; 16-byte aligned.
extern "C" int tsg_gemm_bf16(const void* x, long long ldx, const void* w, long long ldw, const void* bias, void* y, long long ldy,
                             int M, int N, int K, int out_dtype, void* stream) {
  const char* fn = "tsg_gemm_bf16";
  for (const void* p : {x, w, (const void*)y}) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  if (out_dtype != TSG_BF16 && out_dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: out_dtype %d (TSG_BF16 or TSG_F32)", fn, out_dtype);
  if (M <= 0 || N <= 0 || K <= 0 || M % 128 || N % kBN || K % 32)
    return set_error(TSG_E_SHAPE, "%s: needs M %% 128 == 0, N %% 256 == 0, K %% 32 == 0 (M=%d N=%d K=%d)", fn, M, N, K);
  if (ldx < K || ldw < K || ldy < N || (ldx & 7) || (ldw & 7) || (ldy & 1))
    return set_error(TSG_E_ALIGN, "%s: ldx=%lld / ldw=%lld must be >= K and multiples of 8, ldy=%lld >= N and even", fn, ldx, ldw, ldy);
  auto st = static_cast<hipStream_t>(stream);
  const float* b = static_cast<const float*>(bias);
  // the 128-row tile where it saves a full-chip round or fills idle CUs (the W image is staged per workgroup whatever its M: see pick_tm in gemm_f32s.hip)
  const int cus = device_cu_count(), tiles_n = N / kBN;
  bool tm256 = M % 256 == 0;
  if (tm256 && M % 128 == 0) {
    const int t256 = (M / 256) * tiles_n, t128 = (M / 128) * tiles_n;
    const float c256 = (float)((t256 + cus - 1) / cus) * (17.f + 0.22f * 256), c128 = (float)((t128 + cus - 1) / cus) * (17.f + 0.22f * 128);
    tm256 = c256 <= c128;
  }
  const bool obf = out_dtype == TSG_BF16;
  // chunk depth: 64 (half the barriers) where K allows it; TSG_BGEMM_KC=32 / 64 forces one (A/B)
  static const int kc_env = [] { const char* e = getenv("TSG_BGEMM_KC"); return e ? atoi(e) : 0; }();
  const bool kc64 = K % 64 == 0 && kc_env != 32;
#define TSG_BG(TM_, KC_) (obf ? launch_bgemm<TM_, KC_, true>(fn, x, ldx, w, ldw, b, y, ldy, M, N, K, st) : launch_bgemm<TM_, KC_, false>(fn, x, ldx, w, ldw, b, y, ldy, M, N, K, st))
  if (tm256) return kc64 ? TSG_BG(256, 64) : TSG_BG(256, 32);
  return kc64 ? TSG_BG(128, 64) : TSG_BG(128, 32);
#undef TSG_BG
}
