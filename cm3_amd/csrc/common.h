// Shared host/device helpers for libcm3_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/cm3_amd.h"

// Timeline instrumentation for the diagnostic probes under tools/probes only (never defined in the product build).
#ifdef CM3_STAMPS
extern __device__ long long *cm3_stamp_buf;
#define CM3_STAMP(slot, drain)                                                                        \
  do {                                                                                                \
    __builtin_amdgcn_sched_barrier(0);                                                                \
    if (drain) __builtin_amdgcn_s_waitcnt(0);                                                         \
    const long long _t = clock64();                                                                   \
    if ((threadIdx.x & 63) == 0)                                                                      \
      cm3_stamp_buf[((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16 + (slot)] = _t; \
    __builtin_amdgcn_sched_barrier(0);                                                                \
  } while (0)
#else
#define CM3_STAMP(slot, drain) \
  do {                         \
  } while (0)
#endif

namespace cm3 {

// Asks for kernel-argument fields NOW: fields first used late in a kernel are otherwise fetched right before that use, and the
// wave then waits for the scalar load in the middle of its critical path (measured in the pair kernel: three such fetches).
#define CM3_FETCH_EARLY(...) cm3_fetch_early(__VA_ARGS__)
template <typename T> __device__ __forceinline__ void cm3_fetch_one(const T &v) { asm volatile("" ::"s"(v)); }
template <typename... T> __device__ __forceinline__ void cm3_fetch_early(const T &...v) { (cm3_fetch_one(v), ...); }

// element at <uniform base> + <32-bit byte offset>: selects the scalar-base addressing mode of global loads / stores
template <typename T> __device__ __forceinline__ T *at32(const void *base, uint32_t byte_offset) {
  return reinterpret_cast<T *>(const_cast<char *>(reinterpret_cast<const char *>(base)) + byte_offset);
}

// ---- error plumbing ------------------------------------------------------------------------
char *last_error_buf();  // thread-local, 512 bytes
int fail(int code, const char *fmt, ...);

#define CM3_HIP_CHECK(expr)                                                                  \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return ::cm3::fail(CM3_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                         __FILE__, __LINE__);                                                \
  } while (0)

#define CM3_REQUIRE(cond, ...)                                \
  do {                                                        \
    if (!(cond)) return ::cm3::fail(CM3_ERR_INVALID, __VA_ARGS__); \
  } while (0)

// ---- vector types per real -------------------------------------------------------------------
template <typename R> struct Vec;
template <> struct Vec<float> {
  using v2 = float2;
  using v4 = float4;
};
template <> struct Vec<double> {
  using v2 = double2;
  using v4 = double4;
};

constexpr int kWave = 64;

}  // namespace cm3
