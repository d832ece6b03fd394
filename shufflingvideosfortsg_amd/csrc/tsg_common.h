// Shared host/device helpers for libtsg_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
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
// Where a kernel with a bounded wait reports an expiry: `host` = the registered sink (tsg_error_sink: host-mapped memory the
// host polls without synchronising), `dev` = the registered device word (tsg_error_word: device memory the optimizer's
// skip-the-update guard reads on the device, so a corrupted backward never reaches the parameters -- also under graph replay,
// where no host code runs between the launches).  Either may be nullptr.
struct ErrSink { unsigned* host; unsigned* dev; };
ErrSink error_sink();

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int roundup(int a, int b) { return cdiv(a, b) * b; }

// Allow a kernel to use more than the default 64 KiB of dynamic LDS.  The attribute is per (device, kernel): ensure_lds keeps a
// process-wide table keyed by both (a mutex-protected map in tsg_api.hip) and calls hipFuncSetAttribute only when the requested
// size exceeds what that device has been told for that kernel -- so call it on EVERY launch path (a table hit costs ~50 ns); the
// call sites keep no one-shot flags of their own (such flags were not keyed per device).
hipError_t ensure_lds(const void* kernel, size_t bytes);
template <typename K>
inline hipError_t allow_lds(K kernel, size_t bytes) { return ensure_lds(reinterpret_cast<const void*>(kernel), bytes); }
int device_cu_count();                           // CUs of the current device (cached per device)
// tsg_time_next_launch: true (once) when the calling thread armed a slot; the slot's event pair then brackets the launch
bool take_launch_events(hipEvent_t* start, hipEvent_t* stop);

// ---- device side -------------------------------------------------------------------------
#if defined(__HIPCC__)

// Launch `kern`; when the thread armed tsg_time_next_launch, with that slot's events around this kernel alone.
template <typename K, typename... A>
inline void launch_timed(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
  hipEvent_t e0, e1;
  if (take_launch_events(&e0, &e1)) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, args...);
  else hipLaunchKernelGGL(kern, grid, block, lds, st, args...);
}

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ float clampf(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }

// ---- storage types.  Kernels compute in fp32; `ST` is what the activations are stored as in HBM: float (TSG_F32 / TSG_F32S) or
// bf16_t (TSG_BF16: 2 bytes per element, converted on load, rounded to nearest-even on store with v_cvt_pk_bf16_f32, which keeps
// a NaN a NaN).  ldN / stN move N consecutive elements (N*sizeof(ST)-byte aligned).
struct bf16_t { unsigned short bits; };
template <typename ST> struct storage_is_bf16 { static constexpr bool value = false; };
template <> struct storage_is_bf16<bf16_t> { static constexpr bool value = true; };
typedef __bf16 tsg_bf16x2 __attribute__((ext_vector_type(2)));
typedef float tsg_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {      // (rne(lo), rne(hi)), lo in the low half
  return __builtin_bit_cast(unsigned, __builtin_convertvector((tsg_f32x2){lo, hi}, tsg_bf16x2));
}
__device__ __forceinline__ float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __uint_as_float((unsigned)p->bits << 16); }
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const bf16_t* p) {
  const unsigned u = *reinterpret_cast<const unsigned*>(p);
  return make_float2(bf16_lo(u), bf16_hi(u));
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 ld4(const bf16_t* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y));
}
// Raw (unconverted) 4-element pieces: a prefetch must keep what the load instruction returned -- converting right away puts the
// s_waitcnt for the load next to the load and the prefetch hides nothing.  cvt4() at the point of use.
template <typename ST> struct Raw4T { typedef float4 type; };
template <> struct Raw4T<bf16_t> { typedef uint2 type; };
__device__ __forceinline__ float4 ldraw4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ uint2 ldraw4(const bf16_t* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ float4 cvt4(float4 v) { return v; }
__device__ __forceinline__ float4 cvt4(uint2 u) { return make_float4(bf16_lo(u.x), bf16_hi(u.x), bf16_lo(u.y), bf16_hi(u.y)); }
__device__ __forceinline__ float4 zero_raw4(const float*) { return make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ uint2 zero_raw4(const bf16_t*) { return make_uint2(0u, 0u); }
template <typename ST> struct Raw2T { typedef float2 type; };
template <> struct Raw2T<bf16_t> { typedef unsigned type; };
__device__ __forceinline__ float2 ldraw2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ unsigned ldraw2(const bf16_t* p) { return *reinterpret_cast<const unsigned*>(p); }
__device__ __forceinline__ float2 cvt2(float2 v) { return v; }
__device__ __forceinline__ float2 cvt2(unsigned u) { return make_float2(bf16_lo(u), bf16_hi(u)); }

__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16_t* p, float v) { p->bits = (unsigned short)(pack_bf16x2(v, 0.f) & 0xffffu); }
__device__ __forceinline__ void st2(float* p, float2 v) { *reinterpret_cast<float2*>(p) = v; }
__device__ __forceinline__ void st2(bf16_t* p, float2 v) { *reinterpret_cast<unsigned*>(p) = pack_bf16x2(v.x, v.y); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void st4(bf16_t* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}

__device__ __forceinline__ void report_expiry(ErrSink es) {
  if (es.host) __hip_atomic_store(es.host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (es.dev) __hip_atomic_store(es.dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

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

// Up to three small buffers zeroed by ONE kernel node (each launch costs ~4.5 us of device time plus the gap in front of it).
static __global__ void zero_words3_kernel(unsigned* __restrict__ p0, size_t n0, unsigned* __restrict__ p1, size_t n1,
                                          unsigned* __restrict__ p2, size_t n2) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n0 + n1 + n2; i += stride) {
    if (i < n0) p0[i] = 0u;
    else if (i < n0 + n1) p1[i - n0] = 0u;
    else p2[i - n0 - n1] = 0u;
  }
}
inline hipError_t zero3_async(void* p0, size_t b0, void* p1, size_t b1, void* p2, size_t b2, hipStream_t st) {   // pointers 4-byte aligned,
  const size_t n0 = p0 ? b0 / 4 : 0, n1 = p1 ? b1 / 4 : 0, n2 = p2 ? b2 / 4 : 0, words = n0 + n1 + n2;          // bytes multiples of 4
  if (words == 0) return hipSuccess;
  const int blocks = (int)((words + 255) / 256 < 1024 ? (words + 255) / 256 : 1024);
  hipLaunchKernelGGL(zero_words3_kernel, dim3(blocks), dim3(256), 0, st, static_cast<unsigned*>(p0), n0, static_cast<unsigned*>(p1), n1,
                     static_cast<unsigned*>(p2), n2);
  return hipGetLastError();
}

#endif  // __HIPCC__
}  // namespace tsg
