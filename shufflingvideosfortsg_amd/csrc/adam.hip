// Adam(lr, L2 weight decay, eps) over ALL parameters of the model in ONE launch -- the optimizer the reference builds (grounding/train.py:367-371:
// torch.optim.Adam(lr=1e-3, weight_decay=1e-4 (L2, added to the gradient -- not AdamW), eps=1e-6)), torch's single-tensor formulation in fp32:
//     g' = g * grad_scale + wd * p ;  m = m + (1 - b1) (g' - m) ;  v = b2 v + (1 - b2) g'^2
//     p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)          t = the step count AFTER this update (1, 2, ...)
// Round-4 review, item 7: torch's fused Adam was the last foreign kernel family of size in the step (6 launches of multi_tensor_apply, 0.34 ms:
// its pointer tables are chunked over several launches).  Here the tables of up to 64 tensors per launch (two launches for GMD's 80) travel in the kernel
// arguments, a workgroup finds its tensor by a binary search over the chunk prefix, and every element is read / written once with 16-byte
// accesses: 7 x 4 bytes per parameter = HBM-bound, 1.3 GB per step at d = 1024.
//   * `skip` (device float, may be NULL): non-zero = leave parameters, moments and the step count untouched (the guard of
//     engine.optimizer_step: a non-finite loss or an expired bounded wait anywhere in the step, reduced over the ranks);
//   * `step` (device float): the count of updates so far; read by every workgroup at its start, advanced by the LAST workgroup to finish (a
//     ticket: everybody has read it by then) unless the update is skipped -- no separate "step += 1" launch, graph-replay safe;
//   * grad_scale folds the 1 / world of a SUM all-reduce into the update (1 when RCCL already averaged).
#include "tsg_common.h"
#include <cmath>

namespace tsg {
namespace {

constexpr int kAdamMaxTensors = 64;               // per launch: the tables travel in the kernel arguments (2.6 KB)
constexpr int kAdamChunk = 8192;                  // elements per workgroup (256 threads x 8 float4)
struct AdamTable {
  float* p[kAdamMaxTensors]; const float* g[kAdamMaxTensors]; float* m[kAdamMaxTensors]; float* v[kAdamMaxTensors];
  unsigned short* pb[kAdamMaxTensors];             // bf16 shadow of p (the bf16 storage mode's GEMM operand), or NULL: rewritten with every update
  unsigned first_chunk[kAdamMaxTensors + 1];      // prefix of chunks per tensor
  unsigned numel[kAdamMaxTensors];
  int n;
};

__global__ __launch_bounds__(256) void adam_kernel(const AdamTable tb, float lr, float b2, float omb1, float omb2, float logb1, float logb2,
                                                   float eps, float wd, float grad_scale,
                                                   float* __restrict__ step, const float* __restrict__ skip, unsigned* __restrict__ ticket, int advance) {
  const bool skipped = skip && *skip != 0.f;
  if (!skipped) {
    // the workgroup's tensor: the last t with first_chunk[t] <= blockIdx.x
    int lo = 0, hi = tb.n - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (tb.first_chunk[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const unsigned base = (blockIdx.x - tb.first_chunk[lo]) * (unsigned)kAdamChunk, n = tb.numel[lo];
    float* __restrict__ p = tb.p[lo]; const float* __restrict__ g = tb.g[lo]; float* __restrict__ m = tb.m[lo]; float* __restrict__ v = tb.v[lo];
    unsigned short* __restrict__ pb = tb.pb[lo];
    const float t = *step + 1.f;
    // 1 - b^t = -expm1(t log b): formed as 1 - exp(..) the fp32 difference loses three digits at b2 = 0.999, t = 1 (1 - 0.999 = 1e-3 from
    // two numbers known to 6e-8).  log b and 1 - b arrive from the host, formed in DOUBLE from the double hyper-parameters as torch forms them
    // (1 - 0.999f in fp32 is 1.0000467e-3: 5e-5 off)
    const float bc1 = -expm1f(t * logb1);
    const float bc2 = -expm1f(t * logb2);
    const float step_size = lr / bc1, rs2 = 1.f / sqrtf(bc2);
    auto upd = [&](float& pp, float gg, float& mm, float& vv) {
      gg = fmaf(gg, grad_scale, wd * pp);
      mm = fmaf(omb1, gg - mm, mm);
      vv = fmaf(b2, vv, omb2 * gg * gg);
      pp -= step_size * mm / fmaf(sqrtf(vv), rs2, eps);
    };
    const bool al16 = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15u) == 0 &&
                      (reinterpret_cast<uintptr_t>(pb) & 7u) == 0;
    if (al16 && base + kAdamChunk <= n) {                                // a whole chunk of 16-byte aligned tensors (views into a flat gradient buffer need not be)
#pragma unroll
      for (int i = 0; i < kAdamChunk / 1024; ++i) {
        const unsigned e = base + 1024u * i + 4u * threadIdx.x;
        float4 pq = *reinterpret_cast<const float4*>(p + e), mq = *reinterpret_cast<const float4*>(m + e), vq = *reinterpret_cast<const float4*>(v + e);
        const float4 gq = *reinterpret_cast<const float4*>(g + e);
        upd(pq.x, gq.x, mq.x, vq.x); upd(pq.y, gq.y, mq.y, vq.y); upd(pq.z, gq.z, mq.z, vq.z); upd(pq.w, gq.w, mq.w, vq.w);
        *reinterpret_cast<float4*>(p + e) = pq; *reinterpret_cast<float4*>(m + e) = mq; *reinterpret_cast<float4*>(v + e) = vq;
        if (pb) *reinterpret_cast<uint2*>(pb + e) = make_uint2(pack_bf16x2(pq.x, pq.y), pack_bf16x2(pq.z, pq.w));     // (16-byte aligned tensors: 8-byte aligned here)
      }
    } else {
      for (unsigned e = base + threadIdx.x; e < n && e < base + kAdamChunk; e += 256u) {
        float pp = p[e], mm = m[e], vv = v[e];
        upd(pp, g[e], mm, vv);
        p[e] = pp; m[e] = mm; v[e] = vv;
        if (pb) pb[e] = (unsigned short)(pack_bf16x2(pp, 0.f) & 0xffffu);
      }
    }
  }
  // the last workgroup to finish advances the step count (every workgroup has read it) and re-arms the ticket
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (arrived == gridDim.x - 1) {
      if (!skipped && advance) *step = *step + 1.f;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The guard's gradient check: *flag = 1 when any gradient element is NaN or +-inf.  One read of the gradients (186 MB at d = 1024,
// ~35 us) in front of the update; the update kernel above reads `skip` at its start, so the check cannot live inside it.
struct GradTable {
  const float* g[kAdamMaxTensors];
  unsigned first_chunk[kAdamMaxTensors + 1];
  unsigned numel[kAdamMaxTensors];
  int n;
};
constexpr int kCheckChunk = 32768;                // elements per workgroup
__global__ __launch_bounds__(256) void grads_nonfinite_kernel(const GradTable tb, float* __restrict__ flag) {
  int lo = 0, hi = tb.n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tb.first_chunk[mid] <= blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const unsigned base = (blockIdx.x - tb.first_chunk[lo]) * (unsigned)kCheckChunk, n = tb.numel[lo];
  const float* __restrict__ g = tb.g[lo];
  unsigned bad = 0;                                // an exponent field of all ones = inf or NaN
  auto look = [&](float x) { bad |= (unsigned)((__float_as_uint(x) & 0x7f800000u) == 0x7f800000u); };
  if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0 && base + kCheckChunk <= n) {
#pragma unroll 8
    for (int i = 0; i < kCheckChunk / 1024; ++i) {
      const float4 q = *reinterpret_cast<const float4*>(g + base + 1024u * i + 4u * threadIdx.x);
      look(q.x); look(q.y); look(q.z); look(q.w);
    }
  } else {
    for (unsigned e = base + threadIdx.x; e < n && e < base + kCheckChunk; e += 256u) look(g[e]);
  }
  if (__any(bad) && (threadIdx.x & 63) == 0) *flag = 1.f;       // (every writer stores the same value)
}

}  // namespace
}  // namespace tsg

using namespace tsg;

// n tensors (host arrays of device pointers, fp32, contiguous; 16-byte aligned tensors take the float4 path): params / grads / exp_avg / exp_avg_sq,
// numel[i] elements each (< 2^31).  state: device buffer of 2 words {float step count, unsigned ticket (zero between calls)} owned by the caller
// (zero-initialised once).  skip: device float or NULL.  More than 64 tensors run as consecutive launches that read the same step count; the
// LAST launch advances it.
static int adam_impl(const char* fn, int n, const void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                     void* const* shadow, const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                     void* state, const void* skip, void* stream) {
  if (n <= 0 || !params || !grads || !exp_avg || !exp_avg_sq || !numel || !state) return set_error(TSG_E_NULL, "%s: NULL argument or n=%d", fn, n);
  if (!(beta1 > 0. && beta1 < 1. && beta2 > 0. && beta2 < 1.)) return set_error(TSG_E_SHAPE, "%s: betas (%g, %g) outside (0, 1)", fn, beta1, beta2);
  for (int i = 0; i < n; ++i) {
    if (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]) return set_error(TSG_E_NULL, "%s: NULL tensor %d", fn, i);
    if (numel[i] <= 0 || numel[i] >= (1LL << 31)) return set_error(TSG_E_SHAPE, "%s: tensor %d has %lld elements", fn, i, numel[i]);
    if ((reinterpret_cast<uintptr_t>(params[i]) | reinterpret_cast<uintptr_t>(grads[i]) | reinterpret_cast<uintptr_t>(exp_avg[i]) |
         reinterpret_cast<uintptr_t>(exp_avg_sq[i])) & 3u)
      return set_error(TSG_E_ALIGN, "%s: tensor %d is not 4-byte aligned", fn, i);
  }
  for (int i0 = 0; i0 < n; i0 += kAdamMaxTensors) {
    const int cnt = n - i0 < kAdamMaxTensors ? n - i0 : kAdamMaxTensors;
    AdamTable tb;
    tb.n = cnt;
    unsigned chunks = 0;
    for (int i = 0; i < cnt; ++i) {
      tb.p[i] = (float*)params[i0 + i]; tb.g[i] = (const float*)grads[i0 + i]; tb.m[i] = (float*)exp_avg[i0 + i]; tb.v[i] = (float*)exp_avg_sq[i0 + i];
      tb.pb[i] = shadow ? (unsigned short*)shadow[i0 + i] : nullptr;
      tb.numel[i] = (unsigned)numel[i0 + i];
      tb.first_chunk[i] = chunks;
      chunks += (unsigned)((numel[i0 + i] + kAdamChunk - 1) / kAdamChunk);
    }
    tb.first_chunk[cnt] = chunks;
    hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(256), 0, static_cast<hipStream_t>(stream), tb, (float)lr, (float)beta2, (float)(1. - beta1), (float)(1. - beta2),
                       (float)log(beta1), (float)log(beta2), (float)eps, (float)weight_decay, (float)grad_scale,
                       (float*)state, (const float*)skip, (unsigned*)state + 1, i0 + cnt >= n ? 1 : 0);
  }
  return check_launch(fn);
}

extern "C" int tsg_adam_step(int n, const void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                             const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay, double grad_scale,
                             void* state, const void* skip, void* stream) {
  return adam_impl("tsg_adam_step", n, params, grads, exp_avg, exp_avg_sq, nullptr, numel, lr, beta1, beta2, eps, weight_decay, grad_scale, state, skip, stream);
}

// The same update that also rewrites a bf16 SHADOW of every parameter that has one (shadow[i] may be NULL; bf16 = rne of the updated fp32 value, what
// `.to(torch.bfloat16)` gives): the bf16 storage mode's GEMMs read the shadows, so the per-step fp32 -> bf16 casts of the weights (eleven launches,
// 0.11 ms of a 7 ms step) disappear.  A skipped update leaves the shadows as they are (they match the untouched parameters).
extern "C" int tsg_adam_step_shadow(int n, const void* const* params, const void* const* grads, void* const* exp_avg, void* const* exp_avg_sq,
                                    void* const* shadow, const long long* numel, double lr, double beta1, double beta2, double eps, double weight_decay,
                                    double grad_scale, void* state, const void* skip, void* stream) {
  if (shadow)
    for (int i = 0; i < n; ++i)
      if (shadow[i] && (reinterpret_cast<uintptr_t>(shadow[i]) & 1u)) return set_error(TSG_E_ALIGN, "tsg_adam_step_shadow: shadow %d is not 2-byte aligned", i);
  return adam_impl("tsg_adam_step_shadow", n, params, grads, exp_avg, exp_avg_sq, shadow, numel, lr, beta1, beta2, eps, weight_decay, grad_scale, state, skip, stream);
}

// *flag (device float, caller-owned, NOT cleared here) = 1 when any of the n gradients holds a NaN or an infinity: the part of the optimizer guard that
// looks at the gradients themselves (finite garbage is not detected by anything; non-finite values are what the K1 backward poisons an expired exchange
// with, and what an overflow produces).  engine.TsgAdam runs it in front of a guarded update, into the update's own skip flag.
extern "C" int tsg_grads_nonfinite(int n, const void* const* grads, const long long* numel, void* flag, void* stream) {
  if (n <= 0 || !grads || !numel || !flag) return set_error(TSG_E_NULL, "tsg_grads_nonfinite: NULL argument or n=%d", n);
  for (int i = 0; i < n; ++i) {
    if (!grads[i]) return set_error(TSG_E_NULL, "tsg_grads_nonfinite: NULL tensor %d", i);
    if (numel[i] <= 0 || numel[i] >= (1LL << 31)) return set_error(TSG_E_SHAPE, "tsg_grads_nonfinite: tensor %d has %lld elements", i, numel[i]);
    if (reinterpret_cast<uintptr_t>(grads[i]) & 3u) return set_error(TSG_E_ALIGN, "tsg_grads_nonfinite: tensor %d is not 4-byte aligned", i);
  }
  for (int i0 = 0; i0 < n; i0 += kAdamMaxTensors) {
    const int cnt = n - i0 < kAdamMaxTensors ? n - i0 : kAdamMaxTensors;
    GradTable tb;
    tb.n = cnt;
    unsigned chunks = 0;
    for (int i = 0; i < cnt; ++i) {
      tb.g[i] = (const float*)grads[i0 + i];
      tb.numel[i] = (unsigned)numel[i0 + i];
      tb.first_chunk[i] = chunks;
      chunks += (unsigned)((numel[i0 + i] + kCheckChunk - 1) / kCheckChunk);
    }
    tb.first_chunk[cnt] = chunks;
    hipLaunchKernelGGL(grads_nonfinite_kernel, dim3(chunks), dim3(256), 0, static_cast<hipStream_t>(stream), tb, (float*)flag);
  }
  return check_launch("tsg_grads_nonfinite");
}
