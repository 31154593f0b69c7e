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

// XCD-aware block order of the shared-env step kernels.  Workgroup b of a launch runs on XCD b % 8 (observed; HIP promises nothing,
// so this is for speed only -- results do not depend on it): with the plain order the cache lines of the per-ENV arrays (done: 128
// envs per line, reward / collisions / episode: 32, meta: 16) are written in pieces by workgroups on eight different XCDs, each of
// which keeps, and at the end of the launch writes back, its own partial copy.  Here the workgroups that are dispatched together
// (a tile of 8 G consecutive block ids; G = 32: one per CU) cover 8 G CONSECUTIVE env blocks, and XCD x takes G consecutive ones
// of them: logical block = tile base + (b % 8) G + (b / 8) % G.  The mode travels in the top byte of the kernels' leading `flags`
// argument (a preloaded SGPR: reading gridDim instead put a scalar load in front of the first global loads and lost what the order
// gains); launches of fewer than CM3_XCD_MIN_BLOCKS workgroups keep the plain order.  Launchers round the grid up with
// cm3_xcd_grid(); a logical block beyond the batch finds no env of its own (the kernels clamp and store nothing).
// Measured (profiles/r03_xcd_block_order.txt): C5 4.83 -> 4.41 us per tick, C3 3.64 -> 3.56, C2 2.54 -> 2.49.
// Launches of up to 256 workgroups (all in flight together) simply give XCD x the x-th eighth: logical block = (b % 8) G + b / 8 with
// G = ceil(blocks / 8) -- every XCD gets work whatever the count (a 256-tile with 128 or 192 blocks would leave XCDs idle: measured
// +11 .. 18 % on such launches).  The mode travels in the top byte of `flags`: 0 plain order, 1 .. 32 = G of an eighths launch,
// kXcdTiles = tiles of 256.
constexpr uint32_t kFlagXcdShift = 24, kXcdTiles = 63u;   // internal launch flag bits
#ifndef CM3_XCD_MIN_BLOCKS
#define CM3_XCD_MIN_BLOCKS 64u    // (macro: build variant for the comparison)
#endif
__device__ __forceinline__ uint32_t cm3_xcd_block(uint32_t flags) {
  const uint32_t b = blockIdx.x, v = flags >> kFlagXcdShift;
  const uint32_t tiled = (b & ~255u) | ((b & 7u) << 5) | ((b >> 3) & 31u);
  const uint32_t eighth = (b & 7u) * v + (b >> 3);
  return v == 0u ? b : (v == kXcdTiles ? tiled : eighth);
}
// host: the flag bits and the grid for a launch of `blocks` workgroups
static inline uint32_t cm3_xcd_flags(unsigned blocks) {
#ifdef CM3_NO_XCD_ORDER
  (void)blocks;
  return 0u;
#else
  if (blocks < CM3_XCD_MIN_BLOCKS) return 0u;
  return (blocks <= 256u ? (blocks + 7u) / 8u : kXcdTiles) << kFlagXcdShift;
#endif
}
static inline unsigned cm3_xcd_grid(unsigned blocks) {
  const uint32_t v = cm3_xcd_flags(blocks) >> kFlagXcdShift;
  return v == 0u ? blocks : (v == kXcdTiles ? (blocks + 255u) / 256u * 256u : 8u * v);
}

// Kernel-span instrumentation (build variant -DCM3_SPAN_STAMPS -> libcm3_hip_span.so; never defined in the product build):
// exactly TWO time stamps per wave -- the first instruction of the wave and, after its last store has been acknowledged, its
// last -- each the constant 100 MHz s_memrealtime counter plus the shader clock (s_memtime).  Lane 0 of every wave writes the
// four values to record <wave index in the launch> (128 bytes each) of the launch's slot (cm3_span_config / cm3::span_next_slot, util.hip);
// tools/kernel_span.py reduces them per launch to first-wave-in / last-wave-out (span) and start-to-start.  The entry stamps
// stay in SGPRs until the exit store, so nothing waits for them on the way in.
#ifdef CM3_SPAN_STAMPS
namespace cm3 { long long *span_next_slot(); }
#define CM3_SPAN_FIELD long long *span;
#define CM3_SPAN_IN()                                                         \
  const unsigned long long _span_rt0 = __builtin_amdgcn_s_memrealtime();     \
  const unsigned long long _span_ck0 = __builtin_amdgcn_s_memtime();         \
  unsigned long long _span_mk[8] = {0, 0, 0, 0, 0, 0, 0, 0};                 \
  __builtin_amdgcn_sched_barrier(0)
// optional intermediate shader-clock marks (at most 8; -DCM3_SPAN_MARKS builds only: they pin the instruction schedule around them
// and so lengthen the kernel a little -- the two-stamp record is taken WITHOUT them).  drain: first wait for every outstanding
// memory operation (only where the code waits for all of them anyway, e.g. right behind the initial loads).
#ifdef CM3_SPAN_MARKS
#define CM3_SPAN_STORE_MARKS(r) for (int _k = 0; _k < 8; ++_k) (r)[4 + _k] = _span_mk[_k]
#define CM3_SPAN_MARK(k, drain)                          \
  do {                                                   \
    __builtin_amdgcn_sched_barrier(0);                   \
    if (drain) __builtin_amdgcn_s_waitcnt(0);            \
    _span_mk[k] = __builtin_amdgcn_s_memtime();          \
    __builtin_amdgcn_sched_barrier(0);                   \
  } while (0)
#else
#define CM3_SPAN_STORE_MARKS(r) (void)_span_mk   /* the two-stamp record stores two 16-byte vectors per wave and nothing else */
#define CM3_SPAN_MARK(k, drain) do { } while (0)
#endif
#define CM3_SPAN_OUT(slot)                                                                                  \
  do {                                                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                      \
    __builtin_amdgcn_s_waitcnt(0);                                                                          \
    const unsigned long long _rt1 = __builtin_amdgcn_s_memrealtime(), _ck1 = __builtin_amdgcn_s_memtime();  \
    if ((slot) && (threadIdx.x & 63) == 0) {                                                                \
      unsigned long long *_r = reinterpret_cast<unsigned long long *>(slot) +                               \
                               ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;           \
      _r[0] = _span_rt0; _r[1] = _span_ck0; _r[2] = _rt1; _r[3] = _ck1;                                      \
      CM3_SPAN_STORE_MARKS(_r);                                                                             \
    }                                                                                                       \
  } while (0)
#define CM3_SPAN_SET(p) (p).span = ::cm3::span_next_slot()
#else
#define CM3_SPAN_FIELD
#define CM3_SPAN_IN() do { } while (0)
#define CM3_SPAN_MARK(k, drain) do { } while (0)
#define CM3_SPAN_OUT(slot) do { } while (0)
#define CM3_SPAN_SET(p) do { } while (0)
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

// ---- which instantiation ran (cm3_last_kernel_variant; tests/test_gpu_dispatch_sizes.py) ------
// The launchers of the env kernels choose among size-gated builds of one template (mapping, waves per workgroup, store policy,
// live-state, the max-ILP translation unit ...).  Every launcher records its choice in a thread-local POD -- a handful of host
// stores per launch, formatted only when asked for -- so that a test can assert WHICH build it just checked.
struct KernelVariant {
  const char *kernel;  // template name
  int real_bytes;      // 4 / 8 (0: not a template parameter)
  int n, waves, fused, sp, live, tu, early, g;
};
KernelVariant &last_variant();  // thread-local (util.hip)
static inline void note_variant(const char *kernel, int real_bytes, int n, int waves, int fused, int sp, int live, int tu,
                                int early = 0, int g = 0) {
  KernelVariant &v = last_variant();
  v.kernel = kernel;
  v.real_bytes = real_bytes;
  v.n = n;
  v.waves = waves;
  v.fused = fused;
  v.sp = sp;
  v.live = live;
  v.tu = tu;
  v.early = early;
  v.g = g;
}

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
