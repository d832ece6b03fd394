// Shared host/device helpers for libtsg_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/tsg_hip.h"

namespace tsg {

constexpr int kWave = 64;
constexpr int kLdsBytes = 160 * 1024;          // LDS per CU on MI355X
constexpr float k2Log2e = 2.8853900817779268f; // exp(2x) = exp2(x * 2*log2(e))
constexpr float kLog2e = 1.4426950408889634f;

// ---- host side ---------------------------------------------------------------------------
int set_error(int code, const char* fmt, ...);   // stores a thread-local message, returns code
int check_launch(const char* what);              // hipGetLastError() -> 0 or positive hipError_t
unsigned* error_sink();                          // the registered error sink (tsg_error_sink), or nullptr

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int roundup(int a, int b) { return cdiv(a, b) * b; }

// Allow a kernel to use more than the default 64 KiB of dynamic LDS.
template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                             hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(bytes));
}

// ---- device side -------------------------------------------------------------------------
#if defined(__HIPCC__)

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// DPP cross-lane move (bit pattern preserved).
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Sum over the 64 lanes of a wave; every lane receives the total.
// 4 DPP steps inside each 16-lane row (xor1, xor2, half-mirror, mirror), then xor16 / xor32
// through the LDS crossbar (ds_swizzle / ds_bpermute; no LDS memory is touched).
__device__ __forceinline__ float wave_allsum(float v) {
  v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
  v += dpp_mov<0x141>(v);   // row_half_mirror
  v += dpp_mov<0x140>(v);   // row_mirror
  v += __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F));  // xor 16 (bit mode)
  v += __shfl_xor(v, 32, 64);
  return v;
}

__device__ __forceinline__ float wave_allmax(float v) {
  v = fmaxf(v, dpp_mov<0xB1>(v));
  v = fmaxf(v, dpp_mov<0x4E>(v));
  v = fmaxf(v, dpp_mov<0x141>(v));
  v = fmaxf(v, dpp_mov<0x140>(v));
  v = fmaxf(v, __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), 0x401F)));
  v = fmaxf(v, __shfl_xor(v, 32, 64));
  return v;
}

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. it waits
// for every global store the wave still has in flight -- exactly what a kernel that overlaps its
// output stores with the next tile's compute must not do.
__device__ __forceinline__ void lds_barrier() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Blocks are dealt round-robin over the 8 XCDs (block b and b+8 share an L2).  Remap a linear
// block id so that `group` consecutive logical ids (e.g. the tiles that share one batch item's
// word features) land on ONE XCD.  Speed only -- any placement is correct.
__device__ __forceinline__ int xcd_remap(int bid, int nblocks, int group) {
  const int per_round = 8 * group;
  const int full = (nblocks / per_round) * per_round;
  if (bid >= full) return bid;                 // ragged tail keeps its natural order
  const int round = bid / per_round, r = bid % per_round;
  const int xcd = r % 8, slot = r / 8;         // slot-th block this XCD receives in the round
  return round * per_round + xcd * group + slot;
}

// Zero-fill of a small accumulator / sync block as an ordinary kernel node.  (hipMemsetAsync is avoided on purpose: the
// library's launches must behave identically when the caller's stream is being captured into a HIP graph and replayed, and a
// plain kernel is the one node type every capture / replay path treats as ordered, re-executed work.)
static __global__ void zero_words_kernel(unsigned* __restrict__ p, size_t words) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = 0u;
}
inline hipError_t zero_async(void* p, size_t bytes, hipStream_t st) {      // p 4-byte aligned, bytes a multiple of 4
  const size_t words = bytes / 4;
  if (words == 0) return hipSuccess;
  const int blocks = (int)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024);
  hipLaunchKernelGGL(zero_words_kernel, dim3(blocks), dim3(256), 0, st, static_cast<unsigned*>(p), words);
  return hipGetLastError();
}

#endif  // __HIPCC__
}  // namespace tsg
