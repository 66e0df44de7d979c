// Shared helpers for libgd4d.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <utility>

#include "gd4d.h"

#define GD4D_WAVE 64

namespace gd4d {

void set_last_hip_error(hipError_t e);

// Record and translate a launch failure (never synchronises).
inline int check_launch() {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_last_hip_error(e);
    return GD4D_ELAUNCH;
  }
  return GD4D_OK;
}

// Allow `bytes` of dynamic LDS for `kern` on the CURRENT device.  hipFuncAttributeMaxDynamicSharedMemorySize is a
// per-device attribute, so what has been granted is remembered per (kernel, device id) - a process driving several GPUs
// configures each of them (a process-wide "configured" flag would leave every device but the first at 64 KB).
inline bool allow_dynamic_lds(const void* kern, int bytes) {
  static std::mutex mu;
  static std::map<std::pair<const void*, int>, int> granted;
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> guard(mu);
  int& have = granted[std::make_pair(kern, dev)];
  if (have >= bytes) return true;
  if (hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) != hipSuccess) return false;
  have = bytes;
  return true;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

__device__ __forceinline__ float bf16_to_f32(uint16_t v) {
  return __uint_as_float(static_cast<uint32_t>(v) << 16);
}

// round-to-nearest-even f32 -> bf16 bits (NaN kept quiet)
__device__ __forceinline__ uint16_t f32_to_bf16(float f) {
  uint32_t u = __float_as_uint(f);
  if ((u & 0x7fffffffu) > 0x7f800000u) return static_cast<uint16_t>((u >> 16) | 0x40u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return static_cast<uint16_t>(u >> 16);
}

// inverse_sigmoid of the reference (deform3d_cross_attn.py:16-31), eps = 1e-5
__device__ __forceinline__ float inv_sigmoid(float x) {
  x = fminf(fmaxf(x, 0.f), 1.f);
  const float a = fminf(fmaxf(x, 1e-5f), 1.f), b = fminf(fmaxf(1.f - x, 1e-5f), 1.f);
  return logf(a / b);
}

// ---- dev tracing (gd4d_trace_enable): block 0 of a kernel stamps s_memrealtime (100 MHz) at entry / exit -----------
// buffer: [0] entries written, [1] capacity (entries), then {id, time} pairs.  One device-global pointer per translation
// unit (no relocatable device code); nullptr = off: one scalar load per kernel.
#define GD4D_TRACE_UNIT(tag)                                                                          \
  __device__ unsigned long long* g_trace_##tag = nullptr;                                              \
  void trace_set_##tag(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_trace_##tag), &p, sizeof(p)); }
__device__ __forceinline__ void trace_mark_if(unsigned long long* t, unsigned long long id, bool stamping_block) {
  if (t && stamping_block && threadIdx.x == 0) {
    const unsigned long long slot = atomicAdd(t, 1ull);
    if (slot < t[1]) { t[2 + 2 * slot] = id; t[3 + 2 * slot] = __builtin_amdgcn_s_memrealtime(); }
  }
}
__device__ __forceinline__ void trace_mark(unsigned long long* t, unsigned long long id) {
  trace_mark_if(t, id, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0);
}

}  // namespace gd4d
