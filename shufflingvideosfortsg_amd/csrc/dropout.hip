// Element-wise dropout without a stored mask: y = keep(i) ? x / (1 - p) : 0, where keep(i) is a counter-based hash of the element
// index and of (seed, offset).  Where the path needs it: the dropout BETWEEN the layers of the video / sentence BiLSTMs (reference
// networks/RNN.py:27-31, nn.LSTM(dropout=...): applied to the output of every layer but the last, in training).  torch's
// native_dropout writes a byte mask next to the output and its backward reads it again: 150 MB at [128, 128, 1024] fp32, 55 + 50 us
// per use.  Here the backward is the SAME launch on the gradient -- the mask is regenerated from the keys -- so each direction is one
// read and one write of the tensor (134 MB).  The hash is the one of the attention dropout (csrc/mha.hip).
// Keys: from (seed, offset) given by the host (key_mode 0); or derived in the kernel from a device-resident (seed, offset) pair and
// written to `keys_io` for the backward (key_mode 1: launches captured into a HIP graph, whose replays advance the offset with a
// captured add); or read from `keys_io` (key_mode 2: the backward of such a launch).
#include "tsg_common.h"

namespace tsg {
namespace {

__device__ __forceinline__ unsigned dmix32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ float keep_scale(unsigned long long idx, unsigned k0, unsigned k1, unsigned thresh, float inv_keep) {
  const unsigned h = dmix32(dmix32((unsigned)idx + k0) ^ (unsigned)(idx >> 32) ^ k1);
  return h >= thresh ? inv_keep : 0.f;
}

struct DropArgs { unsigned thresh; float inv_keep; unsigned k0, k1; const unsigned long long* rng; unsigned* keys; int key_mode; };

// VEC elements per thread and iteration (16 bytes: 4 floats or 8 bf16); the tail (n % VEC elements) is handled by the last thread
template <typename ST, int VEC>
__global__ __launch_bounds__(256) void dropout_kernel(const ST* __restrict__ x, ST* __restrict__ y, long long n, DropArgs a) {
  unsigned k0 = a.k0, k1 = a.k1;
  if (a.key_mode == 1) {
    const unsigned long long seed = a.rng[0], off = a.rng[1];
    k0 = (unsigned)seed ^ ((unsigned)off * 0x9E3779B1u);
    k1 = (unsigned)(seed >> 32) ^ ((unsigned)(off >> 32) * 0x85EBCA77u + 0x165667B1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.keys[0] = k0; a.keys[1] = k1; }
  } else if (a.key_mode == 2) {
    k0 = a.keys[0]; k1 = a.keys[1];
  }
  const long long nv = n / VEC, stride = (long long)gridDim.x * blockDim.x;
  for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < nv; v += stride) {
    const long long i = v * VEC;
    float f[VEC];
    if constexpr (VEC == 4) {
      const float4 q = ld4(x + i);
      f[0] = q.x; f[1] = q.y; f[2] = q.z; f[3] = q.w;
    } else {
      const float4 q0 = ld4(x + i), q1 = ld4(x + i + 4);
      f[0] = q0.x; f[1] = q0.y; f[2] = q0.z; f[3] = q0.w; f[4] = q1.x; f[5] = q1.y; f[6] = q1.z; f[7] = q1.w;
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) f[j] *= keep_scale((unsigned long long)(i + j), k0, k1, a.thresh, a.inv_keep);
    st4(y + i, make_float4(f[0], f[1], f[2], f[3]));
    if constexpr (VEC == 8) st4(y + i + 4, make_float4(f[4], f[5], f[6], f[7]));
  }
  if (blockIdx.x == gridDim.x - 1 && threadIdx.x == blockDim.x - 1)
    for (long long i = nv * VEC; i < n; ++i) st1(y + i, ld1(x + i) * keep_scale((unsigned long long)i, k0, k1, a.thresh, a.inv_keep));
}

}  // namespace
}  // namespace tsg

using namespace tsg;

extern "C" int tsg_dropout(const void* x, void* y, long long n, float p, uint64_t seed, uint64_t offset, const void* rng_dev, void* keys_io,
                           int key_mode, int dtype, void* stream) {
  const char* fn = "tsg_dropout";
  if (!x || !y) return set_error(TSG_E_NULL, "%s: NULL pointer argument", fn);
  if (!aligned16(x) || !aligned16(y)) return set_error(TSG_E_ALIGN, "%s: operands must be 16-byte aligned", fn);
  if (n <= 0) return set_error(TSG_E_SHAPE, "%s: n=%lld", fn, n);
  if (!(p >= 0.f) || p >= 1.f) return set_error(TSG_E_SHAPE, "%s: dropout probability %g outside [0, 1)", fn, p);
  if (dtype != TSG_F32 && dtype != TSG_F32S && dtype != TSG_BF16) return set_error(TSG_E_DTYPE, "%s: dtype %d", fn, dtype);
  if (key_mode < 0 || key_mode > 2) return set_error(TSG_E_SHAPE, "%s: key_mode %d", fn, key_mode);
  if (key_mode == 1 && (!rng_dev || !keys_io)) return set_error(TSG_E_NULL, "%s: key_mode 1 needs rng_dev and keys_io", fn);
  if (key_mode == 2 && !keys_io) return set_error(TSG_E_NULL, "%s: key_mode 2 needs keys_io", fn);
  DropArgs a{};
  const double t = (double)p * 4294967296.0;
  a.thresh = p > 0.f ? (unsigned)(t < 1.0 ? 1.0 : (t > 4294967295.0 ? 4294967295.0 : t)) : 0u;
  a.inv_keep = 1.f / (1.f - p);
  a.k0 = (unsigned)seed ^ ((unsigned)offset * 0x9E3779B1u);
  a.k1 = (unsigned)(seed >> 32) ^ ((unsigned)(offset >> 32) * 0x85EBCA77u + 0x165667B1u);
  a.rng = (const unsigned long long*)rng_dev; a.keys = (unsigned*)keys_io; a.key_mode = key_mode;
  const bool bf = dtype == TSG_BF16;
  const long long nv = n / (bf ? 8 : 4);
  long long blocks = (nv + 255) / 256;
  const long long cap = (long long)device_cu_count() * 16;          // grid-stride beyond 16 workgroups per CU
  if (blocks > cap) blocks = cap;
  if (blocks < 1) blocks = 1;
  auto st = static_cast<hipStream_t>(stream);
  if (bf) hipLaunchKernelGGL((dropout_kernel<bf16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, a);
  else hipLaunchKernelGGL((dropout_kernel<float, 4>), dim3((unsigned)blocks), dim3(256), 0, st, (const float*)x, (float*)y, n, a);
  return check_launch(fn);
}
