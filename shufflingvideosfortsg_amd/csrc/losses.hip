// K4: the four GMD training losses in one launch each way (reference grounding/loss.py:6-51 as train.py:142-165 combines
// them).  As torch ops they are ~100 launches of a few microseconds forward + backward, a launch-gap-bound stretch at the
// turn of every step.  One workgroup per clip-query pair b (all tensors fp32 [B,T] unless noted):
//
//   span   L0 = 1/B  sum_b ( -log ps[b, s1_b] - log pe[b, e1_b] )                                  loss.py:22-28
//   match  L1 = ( sum_{b,t} bce(om, tl) vm + sum_{b,t} bce(pm, ptl) vm ) / ( sum vm + 1e-4 )       loss.py:30-36, twice
//          bce(x, y) = max(x,0) - x y + log(1 + exp(-|x|))
//   KL     L2 = 1/B  sum_b sum_{k < len_b} p1[i1] log( (p1[i1] + 1e-4) / (p2[i2] + 1e-4) )         loss.py:38-51
//          p1 = masked_softmax(om, tl), p2 = masked_softmax(pm, ptl):  exp(x) m / (sum_t exp(x) m + 1e-4), no max-subtraction
//          (networks/attention.py:123-127);  len = max(e1 - s1 + 1, 0), i1 = min(s1 + k, T-1), i2 = min(s2 + k, T-1), k < T
//   order  L3 = 1/(2B) sum_b ( -log softmax(od[b])[0] - log softmax(pd[b])[1] )                    loss.py:6-20
//
// (s1, e1) = fs[b], (s2, .) = pfs[b] (int64 [B,2]); od / pd are the [B,2] discriminator logits.  The per-pair partial sums
// go to six accumulators with atomics; the last workgroup to arrive (a ticket counter) turns them into out[4].  The lambda
// weights of train.py:146-160 are applied by the caller.  The backward recomputes the per-pair quantities and writes the
// gradients of ps, pe (one non-zero per row), om, pm, od, pd given dL[4].
#include "tsg_common.h"

namespace tsg {
namespace {

constexpr int kLossThreads = 128;
constexpr int kLossMaxT = 2048;          // p1 / p2 / dp1 / dp2 rows in LDS: 4 * T floats
constexpr float kEps = 1e-4f;

__device__ __forceinline__ float block_sum(float v, float* red) {      // sum over the workgroup, result in every thread
  v = wave_allsum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < kLossThreads / 64; ++i) t += red[i];
  return t;
}

__device__ __forceinline__ float bce_logits(float x, float y) { return fmaxf(x, 0.f) - x * y + log1pf(__expf(-fabsf(x))); }

// ws: [0..5] accumulators (span, bce, vm, kl, ce, -), [6] ticket counter (as unsigned) -- zeroed by the host before the launch
__global__ __launch_bounds__(kLossThreads) void gmd_losses_fwd_kernel(
    const float* __restrict__ ps, const float* __restrict__ pe, const float* __restrict__ om, const float* __restrict__ pm,
    const float* __restrict__ od, const float* __restrict__ pd, const long long* __restrict__ fs, const long long* __restrict__ pfs,
    const float* __restrict__ tl, const float* __restrict__ ptl, const float* __restrict__ vm,
    float* __restrict__ ws, float* __restrict__ out, int B, int T, float lam1, float lam2, float lam3) {
  extern __shared__ float sm[];
  float* p1 = sm; float* p2 = sm + T;
  __shared__ float red[kLossThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* omb = om + (size_t)b * T; const float* pmb = pm + (size_t)b * T;
  const float* tlb = tl + (size_t)b * T; const float* ptlb = ptl + (size_t)b * T; const float* vmb = vm + (size_t)b * T;
  float bce = 0.f, vms = 0.f, z1 = 0.f, z2 = 0.f;
  for (int t = tid; t < T; t += kLossThreads) {
    const float x1 = omb[t], x2 = pmb[t], y1 = tlb[t], y2 = ptlb[t], m = vmb[t];
    bce += (bce_logits(x1, y1) + bce_logits(x2, y2)) * m;
    vms += m;
    const float e1 = __expf(x1) * y1, e2 = __expf(x2) * y2;
    p1[t] = e1; p2[t] = e2;
    z1 += e1; z2 += e2;
  }
  bce = block_sum(bce, red); vms = block_sum(vms, red);
  z1 = block_sum(z1, red) + kEps; z2 = block_sum(z2, red) + kEps;      // (block_sum's barriers also publish p1 / p2)
  const long long s1 = fs[2 * b], e1 = fs[2 * b + 1], s2 = pfs[2 * b];
  long long len = e1 - s1 + 1; if (len < 0) len = 0; if (len > T) len = T;
  const int gs = (int)(s1 < 0 ? 0 : (s1 > T - 1 ? T - 1 : s1)), ge = (int)(e1 < 0 ? 0 : (e1 > T - 1 ? T - 1 : e1));   // gather rows (in range
  float kl = 0.f;                                                                                                   // by contract)
  for (int k = tid; k < (int)len; k += kLossThreads) {
    long long i1 = s1 + k, i2 = s2 + k;
    if (i1 > T - 1) i1 = T - 1;
    if (i2 > T - 1) i2 = T - 1;
    if (i1 < 0) i1 = 0;
    if (i2 < 0) i2 = 0;
    const float a = p1[i1] / z1, c = p2[i2] / z2;
    kl += a * __logf((a + kEps) / (c + kEps));
  }
  kl = block_sum(kl, red);
  if (tid == 0) {
    const float span = -(__logf(ps[(size_t)b * T + gs]) + __logf(pe[(size_t)b * T + ge]));
    const float o0 = od[2 * b], o1 = od[2 * b + 1], q0 = pd[2 * b], q1 = pd[2 * b + 1];
    const float mo = fmaxf(o0, o1), mq = fmaxf(q0, q1);
    const float ce = (mo + __logf(__expf(o0 - mo) + __expf(o1 - mo)) - o0) + (mq + __logf(__expf(q0 - mq) + __expf(q1 - mq)) - q1);
    atomicAdd(ws + 0, span); atomicAdd(ws + 1, bce); atomicAdd(ws + 2, vms); atomicAdd(ws + 3, kl); atomicAdd(ws + 4, ce);
    __threadfence();
    const unsigned ticket = atomicAdd(reinterpret_cast<unsigned*>(ws + 6), 1u);
    if (ticket == (unsigned)gridDim.x - 1) {                      // last workgroup: every partial sum has been added
      __threadfence();
      const float a0 = atomicAdd(ws + 0, 0.f), a1 = atomicAdd(ws + 1, 0.f), a2 = atomicAdd(ws + 2, 0.f), a3 = atomicAdd(ws + 3, 0.f),
                  a4 = atomicAdd(ws + 4, 0.f);
      const float l0 = a0 / B, l1 = a1 / (a2 + kEps), l2 = a3 / B, l3 = a4 / (2.f * B);
      out[0] = l0; out[1] = l1; out[2] = l2; out[3] = l3;
      out[4] = l0 + lam1 * l1 + lam2 * l2 + lam3 * l3;          // the training loss of train.py:160
    }
  }
}

// ws[2] = sum of the mask from the forward.  dL[4] = gradient of the four losses, dtot[1] = gradient of their weighted sum (either may be NULL).
__global__ __launch_bounds__(kLossThreads) void gmd_losses_bwd_kernel(
    const float* __restrict__ ps, const float* __restrict__ pe, const float* __restrict__ om, const float* __restrict__ pm,
    const float* __restrict__ od, const float* __restrict__ pd, const long long* __restrict__ fs, const long long* __restrict__ pfs,
    const float* __restrict__ tl, const float* __restrict__ ptl, const float* __restrict__ vm, const float* __restrict__ ws,
    const float* __restrict__ dL, const float* __restrict__ dtot, float* __restrict__ dps, float* __restrict__ dpe,
    float* __restrict__ dom, float* __restrict__ dpm, float* __restrict__ dod, float* __restrict__ dpd, int B, int T,
    float lam1, float lam2, float lam3) {
  extern __shared__ float sm[];
  float* p1 = sm; float* p2 = sm + T; float* g1 = sm + 2 * T; float* g2 = sm + 3 * T;    // probabilities and dKL/dp
  __shared__ float red[kLossThreads / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const size_t row = (size_t)b * T;
  // gradient of loss i = (gradient of out[i], if given) + (gradient of the weighted total, if given) * lambda_i
  const float gt = dtot ? dtot[0] : 0.f;
  const float gsp = ((dL ? dL[0] : 0.f) + gt) / B, gbce = ((dL ? dL[1] : 0.f) + gt * lam1) / (ws[2] + kEps),
              gkl = ((dL ? dL[2] : 0.f) + gt * lam2) / B, gce = ((dL ? dL[3] : 0.f) + gt * lam3) / (2.f * B);
  float z1 = 0.f, z2 = 0.f;
  for (int t = tid; t < T; t += kLossThreads) {
    const float e1 = __expf(om[row + t]) * tl[row + t], e2 = __expf(pm[row + t]) * ptl[row + t];
    p1[t] = e1; p2[t] = e2; g1[t] = 0.f; g2[t] = 0.f;
    z1 += e1; z2 += e2;
  }
  z1 = block_sum(z1, red) + kEps; z2 = block_sum(z2, red) + kEps;
  for (int t = tid; t < T; t += kLossThreads) { p1[t] /= z1; p2[t] /= z2; }
  __syncthreads();
  const long long s1 = fs[2 * b], e1 = fs[2 * b + 1], s2 = pfs[2 * b];
  long long len = e1 - s1 + 1; if (len < 0) len = 0; if (len > T) len = T;
  const int gs = (int)(s1 < 0 ? 0 : (s1 > T - 1 ? T - 1 : s1)), ge = (int)(e1 < 0 ? 0 : (e1 > T - 1 ? T - 1 : e1));
  // dKL/dp1[i1] += log((a+eps)/(c+eps)) + a/(a+eps);  dKL/dp2[i2] += -a/(c+eps)   (clamped indices can repeat: LDS atomics)
  for (int k = tid; k < (int)len; k += kLossThreads) {
    long long i1 = s1 + k, i2 = s2 + k;
    if (i1 > T - 1) i1 = T - 1;
    if (i2 > T - 1) i2 = T - 1;
    if (i1 < 0) i1 = 0;
    if (i2 < 0) i2 = 0;
    const float a = p1[i1], c = p2[i2];
    atomicAdd(g1 + i1, __logf((a + kEps) / (c + kEps)) + a / (a + kEps));
    atomicAdd(g2 + i2, -a / (c + kEps));
  }
  __syncthreads();
  float d1 = 0.f, d2 = 0.f;                                       // sum_j dKL/dp[j] p[j]
  for (int t = tid; t < T; t += kLossThreads) { d1 += g1[t] * p1[t]; d2 += g2[t] * p2[t]; }
  d1 = block_sum(d1, red); d2 = block_sum(d2, red);
  for (int t = tid; t < T; t += kLossThreads) {
    const float x1 = om[row + t], x2 = pm[row + t], m = vm[row + t];
    const float sg1 = 1.f / (1.f + __expf(-x1)), sg2 = 1.f / (1.f + __expf(-x2));
    dom[row + t] = gbce * (sg1 - tl[row + t]) * m + gkl * p1[t] * (g1[t] - d1);
    dpm[row + t] = gbce * (sg2 - ptl[row + t]) * m + gkl * p2[t] * (g2[t] - d2);
    dps[row + t] = (t == gs) ? -gsp / ps[row + t] : 0.f;
    dpe[row + t] = (t == ge) ? -gsp / pe[row + t] : 0.f;
  }
  if (tid == 0) {
    const float o0 = od[2 * b], o1 = od[2 * b + 1], q0 = pd[2 * b], q1 = pd[2 * b + 1];
    const float so = 1.f / (1.f + __expf(o0 - o1)), sq = 1.f / (1.f + __expf(q0 - q1));   // softmax(.)[1]
    dod[2 * b] = gce * ((1.f - so) - 1.f); dod[2 * b + 1] = gce * so;                      // label 0
    dpd[2 * b] = gce * (1.f - sq); dpd[2 * b + 1] = gce * (sq - 1.f);                      // label 1
  }
}

int loss_check(const char* fn, int B, int T) {
  if (B <= 0 || T <= 0) return set_error(TSG_E_SHAPE, "%s: non-positive dimension B=%d T=%d", fn, B, T);
  if (T > kLossMaxT) return set_error(TSG_E_SHAPE, "%s: T=%d > %d not supported", fn, T, kLossMaxT);
  return 0;
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_gmd_losses_fwd(const void* ps, const void* pe, const void* om, const void* pm, const void* od, const void* pd,
                                  const void* fs, const void* pfs, const void* tl, const void* ptl, const void* vm,
                                  void* ws, void* out, int B, int T, float lam_match, float lam_kl, float lam_disc, void* stream) {
  const char* fn = "tsg_gmd_losses_fwd";
  for (const void* p : {ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, (const void*)ws, (const void*)out})
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  int rc = loss_check(fn, B, T);
  if (rc) return rc;
  auto st = static_cast<hipStream_t>(stream);
  hipError_t e = zero_async(ws, 32, st);
  if (e != hipSuccess) return set_error((int)e, "%s: memset: %s", fn, hipGetErrorString(e));
  hipLaunchKernelGGL(gmd_losses_fwd_kernel, dim3(B), dim3(kLossThreads), sizeof(float) * 2 * T, st, (const float*)ps, (const float*)pe,
                     (const float*)om, (const float*)pm, (const float*)od, (const float*)pd, (const long long*)fs,
                     (const long long*)pfs, (const float*)tl, (const float*)ptl, (const float*)vm, (float*)ws, (float*)out, B, T,
                     lam_match, lam_kl, lam_disc);
  return check_launch(fn);
}

extern "C" int tsg_gmd_losses_bwd(const void* ps, const void* pe, const void* om, const void* pm, const void* od, const void* pd,
                                  const void* fs, const void* pfs, const void* tl, const void* ptl, const void* vm,
                                  const void* ws, const void* dL, const void* dtotal, void* dps, void* dpe, void* dom, void* dpm,
                                  void* dod, void* dpd, int B, int T, float lam_match, float lam_kl, float lam_disc, void* stream) {
  const char* fn = "tsg_gmd_losses_bwd";
  for (const void* p : {ps, pe, om, pm, od, pd, fs, pfs, tl, ptl, vm, ws, (const void*)dps, (const void*)dpe, (const void*)dom,
                        (const void*)dpm, (const void*)dod, (const void*)dpd})
    if (!p) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!dL && !dtotal) return set_error(TSG_E_NULL, "%s: dL and dtotal are both NULL", fn);
  int rc = loss_check(fn, B, T);
  if (rc) return rc;
  hipLaunchKernelGGL(gmd_losses_bwd_kernel, dim3(B), dim3(kLossThreads), sizeof(float) * 4 * T, static_cast<hipStream_t>(stream),
                     (const float*)ps, (const float*)pe, (const float*)om, (const float*)pm, (const float*)od, (const float*)pd,
                     (const long long*)fs, (const long long*)pfs, (const float*)tl, (const float*)ptl, (const float*)vm,
                     (const float*)ws, (const float*)dL, (const float*)dtotal, (float*)dps, (float*)dpe, (float*)dom, (float*)dpm,
                     (float*)dod, (float*)dpd, B, T, lam_match, lam_kl, lam_disc);
  return check_launch(fn);
}
