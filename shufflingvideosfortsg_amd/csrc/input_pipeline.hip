// Device-side input pipeline and span decoding of the grounding path for gfx950.
//
// The reference builds every batch on the host with numpy, one sample at a time, in 8 DataLoader workers feeding ONE GPU
// (grounding/dataset/charades.py, charades_pair_aug.py, data_augment.py); eight ranks on one host would contend for that.
// Here the per-batch work runs on the GPU that consumes it:
//   tsg_pool_clips        adjacent-pair mean pooling of the raw i3d clips + zero pad + nfeats + frame stamps
//                         (CharadesDataSentence.generate_video_fts_data, dataset/charades.py:177-196)
//   tsg_sequence_masks    video / temporal / fore / back masks from (nfeats, span)
//                         (Sequence_mask, charades.py:12-18, as combined at charades.py:167-170 / charades_pair_aug.py:96-107)
//   tsg_moment_translate  the shuffling augmentation as an index gather (DataAugmentForTSG.gt_moment_translate,
//                         dataset/data_augment.py:135-156), insert position given or drawn from a counter-based hash
//   tsg_span_pred         argmax_{i,j} of the zero-filled upper-triangular start_i + end_j matrix, first maximum wins
//                         (span_pred, grounding/loss.py:53-70)
// All of it is integer / byte work bound by HBM (the gather and the pooling move B*T*D*4 bytes each way) or by latency
// (masks, decode).  Index outputs are bit-exact with the reference; feature outputs are copies / one rounded fp32 mean.
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kThreads = 256;
constexpr int kRowsPerWg = 8;                 // feature rows per workgroup: 8 x D floats, all loads issued before the stores

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

// insert position of the shuffling augmentation when the caller does not provide one: uniform over [0, wo_len]
// (random.randint(0, wo_len), data_augment.py:149) from a counter-based hash of (seed, sample index): seedable, order-free
__device__ __forceinline__ int draw_cropin(unsigned long long seed, int b, int wo_len) {
  const unsigned long long h = splitmix64(seed ^ (0xD1B54A32D192ED03ull * (unsigned long long)(b + 1)));
  return (int)(((h >> 32) * (unsigned long long)(wo_len + 1)) >> 32);
}

struct Translate { int s, n, c, nf; bool noop; };

__device__ __forceinline__ Translate translate_of(const int* __restrict__ spans, const int* __restrict__ nfeats,
                                                  const int* __restrict__ cropin, unsigned long long seed, int b) {
  Translate tr;
  const int s = spans[2 * b], e = spans[2 * b + 1];
  tr.s = s; tr.n = e - s + 1; tr.nf = nfeats[b];
  tr.noop = tr.n <= 1 || tr.n >= tr.nf;                       // data_augment.py:138-139
  tr.c = tr.noop ? s : (cropin ? cropin[b] : draw_cropin(seed, b, tr.nf - tr.n));
  return tr;
}

// source row of output row t (-1 = zero row): [0,c) gap-closed rows, [c,c+n) the moment, then the rest of the gap-closed rows
__device__ __forceinline__ int translate_src(const Translate& tr, int t) {
  if (tr.noop) return t;
  if (t >= tr.nf) return -1;
  if (t >= tr.c && t < tr.c + tr.n) return tr.s + (t - tr.c);
  const int i = t < tr.c ? t : t - tr.n;                      // index in the sequence without the moment
  return i < tr.s ? i : i + tr.n;
}

__global__ __launch_bounds__(kThreads) void moment_translate_kernel(
    const float* __restrict__ video, const int* __restrict__ spans, const int* __restrict__ nfeats,
    const int* __restrict__ cropin, unsigned long long seed, float* __restrict__ out, int* __restrict__ new_spans,
    int B, int T, int D, int tiles) {
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * kRowsPerWg;
  const Translate tr = translate_of(spans, nfeats, cropin, seed, b);
  if (blockIdx.x % tiles == 0 && threadIdx.x == 0) {
    new_spans[2 * b] = tr.c;                                  // no-op: the span itself (c = s, n = e - s + 1)
    new_spans[2 * b + 1] = tr.c + tr.n - 1;
  }
  const int d4 = D / 4, per_wg = kRowsPerWg * d4;
  const float4* vb = reinterpret_cast<const float4*>(video) + (size_t)b * T * d4;
  float4* ob = reinterpret_cast<float4*>(out) + (size_t)b * T * d4;
  for (int base = threadIdx.x; base < per_wg; base += 8 * kThreads) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads, r = idx / d4, k = idx % d4, t = t0 + r;
      v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (idx < per_wg && t < T) {
        const int src = translate_src(tr, t);
        if (src >= 0) v[u] = vb[(size_t)src * d4 + k];
      }
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int idx = base + u * kThreads, r = idx / d4, k = idx % d4, t = t0 + r;
      if (idx < per_wg && t < T) ob[(size_t)t * d4 + k] = v[u];
    }
  }
}

__global__ __launch_bounds__(kThreads) void pool_clips_kernel(
    const float* __restrict__ raw, const long long* __restrict__ offsets, const double* __restrict__ timestamps,
    float* __restrict__ out, int* __restrict__ nfeats, int* __restrict__ framestps, int B, int T, int D, int tiles) {
  const int b = blockIdx.x / tiles, t0 = (blockIdx.x % tiles) * kRowsPerWg;
  const long long r0 = offsets[b];
  const int n = (int)(offsets[b + 1] - r0);
  const int nf = min((n + 1) / 2, T);                        // one output row per even clip index, at most T (charades.py:185-193)
  if (blockIdx.x % tiles == 0 && threadIdx.x == 0) {
    nfeats[b] = nf;
    if (timestamps && framestps) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {                           // int(x) if int(x) < SAMPLE_LEN else SAMPLE_LEN - 1 (charades.py:178)
        const int f = (int)timestamps[2 * b + q];
        framestps[2 * b + q] = f < T ? f : T - 1;
      }
    }
  }
  const int d4 = D / 4, per_wg = kRowsPerWg * d4;
  const float4* rb = reinterpret_cast<const float4*>(raw) + (size_t)r0 * d4;
  float4* ob = reinterpret_cast<float4*>(out) + (size_t)b * T * d4;
  for (int base = threadIdx.x; base < per_wg; base += 4 * kThreads) {
    float4 x[4], y[4];
    bool two[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * kThreads, r = idx / d4, k = idx % d4, t = t0 + r;
      x[u] = y[u] = make_float4(0.f, 0.f, 0.f, 0.f);
      two[u] = false;
      if (idx < per_wg && t < nf) {
        x[u] = rb[(size_t)(2 * t) * d4 + k];
        two[u] = 2 * t + 1 <= n - 1;
        if (two[u]) y[u] = rb[(size_t)(2 * t + 1) * d4 + k];
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + u * kThreads, r = idx / d4, k = idx % d4, t = t0 + r;
      if (idx < per_wg && t < T) {
        // np.mean over two float32 rows = fl(fl(a + b) / 2); a lone last clip is copied (charades.py:186-191)
        const float4 m = two[u] ? make_float4((x[u].x + y[u].x) * 0.5f, (x[u].y + y[u].y) * 0.5f, (x[u].z + y[u].z) * 0.5f,
                                              (x[u].w + y[u].w) * 0.5f) : x[u];
        ob[(size_t)t * d4 + k] = m;
      }
    }
  }
}

// Sequence_mask(T, [st, et]): ones on [max(0,st), min(et, T-1)] inclusive (charades.py:12-18)
__device__ __forceinline__ int seq_mask(int t, int st, int et, int T) { return (t >= max(0, st) && t <= min(et, T - 1)) ? 1 : 0; }

__global__ __launch_bounds__(kThreads) void sequence_masks_kernel(
    const int* __restrict__ nfeats, const int* __restrict__ spans, int* __restrict__ vm, int* __restrict__ tl,
    int* __restrict__ fm, int* __restrict__ bm, int B, int T) {
  const int idx = blockIdx.x * kThreads + threadIdx.x;
  if (idx >= B * T) return;
  const int b = idx / T, t = idx % T;
  const int nf = nfeats[b], s = spans[2 * b], e = spans[2 * b + 1];
  if (vm) vm[idx] = seq_mask(t, 0, nf, T);                    // video_mask     = [0, nfeats]   (charades.py:167)
  if (tl) tl[idx] = seq_mask(t, s, e, T);                     // temporal_labels = framestamps  (:168)
  if (fm) fm[idx] = seq_mask(t, 0, s, T);                     // fore_mask      = [0, start]    (:169)
  if (bm) bm[idx] = seq_mask(t, e, nf, T);                    // back_mask      = [end, nfeats] (:170)
}

// span_pred: row i of the matrix is m[i][j] = j >= i ? fl(start_i + end_j) : 0 (triu zero-fills the lower triangle, which
// takes part in the max).  torch.max returns the FIRST maximal index along a dimension: strict '>' in increasing j, then in
// increasing i.  The sums are formed exactly as the reference forms them (one fp32 add per (i, j)): ties created by rounding
// must resolve the same way, so the O(T^2) walk is kept -- 64 x 128^2 adds is nothing.
__global__ __launch_bounds__(kThreads) void span_pred_kernel(const float* __restrict__ start, const float* __restrict__ end,
                                                            long long* __restrict__ pred, float* __restrict__ score, int B, int T) {
  extern __shared__ float sh[];                               // end row [T]
  __shared__ float wbest[kThreads / kWave];
  __shared__ int wi[kThreads / kWave], wj[kThreads / kWave];
  const int b = blockIdx.x, tid = threadIdx.x;
  for (int j = tid; j < T; j += kThreads) sh[j] = end[(size_t)b * T + j];
  __syncthreads();
  float best = -INFINITY; int bi = 0x7fffffff, bj = 0;
  for (int i = tid; i < T; i += kThreads) {                   // rows in increasing i per thread
    const float si = start[(size_t)b * T + i];
    float rb = i == 0 ? si + sh[0] : 0.f; int rj = 0;
    for (int j = 1; j < T; ++j) {
      const float v = j >= i ? si + sh[j] : 0.f;
      if (v > rb) { rb = v; rj = j; }
    }
    if (rb > best) { best = rb; bi = i; bj = rj; }
  }
  // (value, row) argmax with the smaller row winning ties: across the wave, then across the waves
#pragma unroll
  for (int off = 1; off < kWave; off <<= 1) {
    const float ob = __shfl_xor(best, off, kWave); const int oi = __shfl_xor(bi, off, kWave), oj = __shfl_xor(bj, off, kWave);
    if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; bj = oj; }
  }
  if ((tid & (kWave - 1)) == 0) { wbest[tid / kWave] = best; wi[tid / kWave] = bi; wj[tid / kWave] = bj; }
  __syncthreads();
  if (tid == 0) {
    for (int u = 1; u < kThreads / kWave; ++u)
      if (wbest[u] > best || (wbest[u] == best && wi[u] < bi)) { best = wbest[u]; bi = wi[u]; bj = wj[u]; }
    pred[2 * b] = bi; pred[2 * b + 1] = bj; score[b] = best;
  }
}

int check_ptrs(const char* fn, std::initializer_list<const void*> ptrs) {
  for (const void* p : ptrs) {
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
    if (!aligned16(p)) return set_error(TSG_E_ALIGN, "%s: pointer %p is not 16-byte aligned", fn, p);
  }
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_moment_translate(const void* video, const int32_t* spans, const int32_t* nfeats, const int32_t* cropin,
                                    uint64_t seed, void* out, int32_t* new_spans, int B, int T, int D, int dtype, void* stream) {
  const char* fn = "tsg_moment_translate";
  int rc = check_ptrs(fn, {video, spans, nfeats, out, new_spans});
  if (rc) return rc;
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (B <= 0 || T <= 0 || D <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d D=%d", fn, B, T, D);
  if (D % 4) return set_error(TSG_E_ALIGN, "%s: D=%d must be a multiple of 4", fn, D);
  if (video == out) return set_error(TSG_E_SHAPE, "%s: in-place operation is not supported", fn);
  const int tiles = cdiv(T, kRowsPerWg);
  hipLaunchKernelGGL(moment_translate_kernel, dim3(B * tiles), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     (const float*)video, spans, nfeats, cropin, (unsigned long long)seed, (float*)out, new_spans, B, T, D, tiles);
  return check_launch(fn);
}

extern "C" int tsg_pool_clips(const void* raw, const int64_t* offsets, const double* timestamps, void* out, int32_t* nfeats,
                              int32_t* framestps, int B, int T, int D, int dtype, void* stream) {
  const char* fn = "tsg_pool_clips";
  int rc = check_ptrs(fn, {raw, offsets, out, nfeats});
  if (rc) return rc;
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (B <= 0 || T <= 0 || D <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d D=%d", fn, B, T, D);
  if (D % 4) return set_error(TSG_E_ALIGN, "%s: D=%d must be a multiple of 4", fn, D);
  if ((timestamps == nullptr) != (framestps == nullptr)) return set_error(TSG_E_NULL, "%s: timestamps and framestps go together", fn);
  const int tiles = cdiv(T, kRowsPerWg);
  hipLaunchKernelGGL(pool_clips_kernel, dim3(B * tiles), dim3(kThreads), 0, static_cast<hipStream_t>(stream), (const float*)raw,
                     (const long long*)offsets, timestamps, (float*)out, nfeats, framestps, B, T, D, tiles);
  return check_launch(fn);
}

extern "C" int tsg_sequence_masks(const int32_t* nfeats, const int32_t* spans, int32_t* video_mask, int32_t* temporal_labels,
                                  int32_t* fore_mask, int32_t* back_mask, int B, int T, void* stream) {
  const char* fn = "tsg_sequence_masks";
  int rc = check_ptrs(fn, {nfeats, spans});
  if (rc) return rc;
  if (B <= 0 || T <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d", fn, B, T);
  hipLaunchKernelGGL(sequence_masks_kernel, dim3(cdiv(B * T, kThreads)), dim3(kThreads), 0, static_cast<hipStream_t>(stream),
                     nfeats, spans, video_mask, temporal_labels, fore_mask, back_mask, B, T);
  return check_launch(fn);
}

extern "C" int tsg_span_pred(const void* start, const void* end, int64_t* pred, void* score, int B, int T, int dtype, void* stream) {
  const char* fn = "tsg_span_pred";
  int rc = check_ptrs(fn, {start, end, pred, score});
  if (rc) return rc;
  if (dtype != TSG_F32) return set_error(TSG_E_DTYPE, "%s: dtype %d not supported (fp32 only)", fn, dtype);
  if (B <= 0 || T <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d", fn, B, T);
  if (T > 16384) return set_error(TSG_E_LDS, "%s: T=%d > 16384", fn, T);
  hipLaunchKernelGGL(span_pred_kernel, dim3(B), dim3(kThreads), sizeof(float) * T, static_cast<hipStream_t>(stream),
                     (const float*)start, (const float*)end, (long long*)pred, (float*)score, B, T);
  return check_launch(fn);
}
