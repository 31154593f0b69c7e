// Error plumbing, device queries, the HBM read-bandwidth probe, hipGraph capture and HIP events.
#include <stdarg.h>

#include "common.h"

namespace cm3 {

char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

KernelVariant &last_variant() {
  static thread_local KernelVariant v = {"none", 0, 0, 0, 0, 0, 0, 0, 0, 0};
  return v;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

#ifdef CM3_SPAN_STAMPS
// Kernel-span stamps (common.h): launches take consecutive slots of a caller-provided device buffer, in enqueue order (a
// captured graph node keeps the slot it was given at capture time, so a replay overwrites the same slots).
static long long *g_span_base = nullptr;
static int64_t g_span_slots = 0, g_span_slot_bytes = 0, g_span_next = 0;
long long *span_next_slot() {
  if (!g_span_base || g_span_slots <= 0) return nullptr;
  long long *r = reinterpret_cast<long long *>(reinterpret_cast<char *>(g_span_base) + (g_span_next % g_span_slots) * g_span_slot_bytes);
  ++g_span_next;
  return r;
}
#endif

// ---- streaming read / copy probes: the measured roofline denominator ---------------
// Grid-stride 16-byte loads, U independent loads in flight per lane per iteration (all issued before the XOR fold
// consumes them); NT selects non-temporal loads (the stream is read once: no reason to keep it in L2 / MALL).  One
// word per workgroup is written so the kernel has an observable result.  cm3_hbm_read_bench runs the configuration
// that measured best on MI355X (tools/hbm_probe_sweep.py -> profiles/r02_hbm_probe_sweep.txt); the _cfg entry sweeps.
constexpr int kBenchBlock = 256;
constexpr int kBenchMaxGrid = 256 * 32;  // up to 32 workgroups per CU
// measured (profiles/r02_hbm_probe_sweep.txt, 3 x 20 repetitions): unroll 1, 8 WG/CU, nt -> 7.06 TB/s; the a-priori guess
// (8 loads in flight, 16 WG/CU, nt) gave 6.4 TB/s, plain loads 5.6-6.4 TB/s
constexpr int kBenchUnroll = 1, kBenchWgPerCu = 8, kBenchNt = 1;

typedef uint32_t bench_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 bench_load(const uint4 *p, bool nt) {
  if (nt) {
    const bench_u4 t = __builtin_nontemporal_load(reinterpret_cast<const bench_u4 *>(p));
    return make_uint4(t.x, t.y, t.z, t.w);
  }
  return *p;
}

template <int U, bool NT>
__global__ void __launch_bounds__(kBenchBlock) k_hbm_read(const uint4 *__restrict__ src, size_t n_vec,
                                                          uint32_t *__restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * kBenchBlock;
  size_t i = (size_t)blockIdx.x * kBenchBlock + threadIdx.x;
  uint32_t acc = 0;
  for (; i + (U - 1) * stride < n_vec; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = bench_load(src + i + u * stride, NT);
#pragma unroll
    for (int u = 0; u < U; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  for (; i < n_vec; i += stride) {
    const uint4 a = src[i];
    acc ^= a.x ^ a.y ^ a.z ^ a.w;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc ^= __shfl_xor(acc, off, 64);
  __shared__ uint32_t part[kBenchBlock / 64];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t r = 0;
    for (int w = 0; w < kBenchBlock / 64; ++w) r ^= part[w];
    sink[blockIdx.x] = r;
  }
}


// float4 copy (read + write streams together): the guide's 6.29 TB/s figure is this pattern
template <int U, bool NT>
__global__ void __launch_bounds__(kBenchBlock) k_hbm_copy(uint4 *__restrict__ dst, const uint4 *__restrict__ src, size_t n_vec) {
  const size_t stride = (size_t)gridDim.x * kBenchBlock;
  size_t i = (size_t)blockIdx.x * kBenchBlock + threadIdx.x;
  for (; i + (U - 1) * stride < n_vec; i += U * stride) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = bench_load(src + i + u * stride, NT);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (NT) {
        uint4 *q = dst + i + u * stride;
        __builtin_nontemporal_store(v[u].x, &q->x);
        __builtin_nontemporal_store(v[u].y, &q->y);
        __builtin_nontemporal_store(v[u].z, &q->z);
        __builtin_nontemporal_store(v[u].w, &q->w);
      } else {
        dst[i + u * stride] = v[u];
      }
    }
  }
  for (; i < n_vec; i += stride) dst[i] = src[i];
}

template <int U> static void launch_read(bool nt, int grid, hipStream_t s, const uint4 *src, size_t n, uint32_t *sink) {
  if (nt)
    hipLaunchKernelGGL((k_hbm_read<U, true>), dim3(grid), dim3(kBenchBlock), 0, s, src, n, sink);
  else
    hipLaunchKernelGGL((k_hbm_read<U, false>), dim3(grid), dim3(kBenchBlock), 0, s, src, n, sink);
}
template <int U> static void launch_copy(bool nt, int grid, hipStream_t s, uint4 *dst, const uint4 *src, size_t n) {
  if (nt)
    hipLaunchKernelGGL((k_hbm_copy<U, true>), dim3(grid), dim3(kBenchBlock), 0, s, dst, src, n);
  else
    hipLaunchKernelGGL((k_hbm_copy<U, false>), dim3(grid), dim3(kBenchBlock), 0, s, dst, src, n);
}

// Launch-structure floor of a step launch: the same grid reads `n_read` 16-byte vectors (all loads first), then writes
// `n_write` vectors whose value depends on everything it read -- load -> (no arithmetic) -> store, nothing else.  What a
// one-launch-per-tick kernel with this traffic cannot go below.
#ifdef CM3_SPAN_STAMPS
#define CM3_FLOOR_SPAN_PARAM , long long *span
#define CM3_FLOOR_SPAN_ARG , ::cm3::span_next_slot()
#else
#define CM3_FLOOR_SPAN_PARAM
#define CM3_FLOOR_SPAN_ARG
#endif
__global__ void __launch_bounds__(1024) k_traffic_floor(const uint4 *__restrict__ src, size_t n_read, uint4 *__restrict__ dst,
                                                        size_t n_write CM3_FLOOR_SPAN_PARAM) {
  CM3_SPAN_IN();
  const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x, total = (size_t)gridDim.x * blockDim.x;
  uint4 acc = make_uint4(0u, 0u, 0u, 0u);
  for (size_t i = gid; i < n_read; i += total) {
    const uint4 v = src[i];
    acc.x ^= v.x;
    acc.y ^= v.y;
    acc.z ^= v.z;
    acc.w ^= v.w;
  }
  for (size_t i = gid; i < n_write; i += total) dst[i] = acc;
  CM3_SPAN_OUT(span);
}
}  // namespace cm3

extern "C" {

int cm3_abi_version(void) { return CM3_ABI_VERSION; }

#ifndef CM3_SOURCE_ID
#define CM3_SOURCE_ID "unknown"
#endif
const char *cm3_source_id(void) { return CM3_SOURCE_ID; }

const char *cm3_last_error(void) { return cm3::last_error_buf(); }

const char *cm3_last_kernel_variant(void) {
  static thread_local char buf[160];
  const cm3::KernelVariant &v = cm3::last_variant();
  static const char *const sp[] = {"plain", "nt", "wt"};
  snprintf(buf, sizeof(buf), "%s<%s,N=%d,waves=%d,fused=%d,sp=%s,live=%d,early=%d,g=%d,tu=%s>", v.kernel,
           v.real_bytes == 4 ? "f32" : (v.real_bytes == 8 ? "f64" : "-"), v.n, v.waves, v.fused, sp[v.sp >= 0 && v.sp <= 2 ? v.sp : 0],
           v.live, v.early, v.g, v.tu ? "ilp" : "default");
  return buf;
}

int cm3_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n;
}

int cm3_device_name(int dev, char *name, int len) {
  hipDeviceProp_t prop;
  CM3_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
  if (name && len > 0) {
    snprintf(name, (size_t)len, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
  }
  return CM3_OK;
}

#ifdef CM3_SPAN_STAMPS
// (span build only; not part of the product ABI)  buf: n_slots x slot_bytes device bytes, slot_bytes >= 128 x waves per launch;
// NULL switches the stamps off.  Resets the slot counter.  Returns the number of slots handed out before the call.
int64_t cm3_span_config(void *buf, int64_t n_slots, int64_t slot_bytes) {
  const int64_t used = cm3::g_span_next;
  cm3::g_span_base = (long long *)buf;
  cm3::g_span_slots = n_slots;
  cm3::g_span_slot_bytes = slot_bytes;
  cm3::g_span_next = 0;
  return used;
}
#endif

int cm3_hbm_bench_sink_words(void) { return cm3::kBenchMaxGrid; }

int cm3_traffic_floor_bench(const void *src, size_t read_bytes, void *dst, size_t write_bytes, int32_t blocks,
                            int32_t threads, void *stream) {
  CM3_REQUIRE(src && dst, "null buffer");
  CM3_REQUIRE(read_bytes % 16 == 0 && write_bytes % 16 == 0, "byte counts must be multiples of 16");
  CM3_REQUIRE(blocks >= 1 && threads >= 64 && threads <= 1024 && threads % 64 == 0, "bad launch shape");
  hipLaunchKernelGGL(cm3::k_traffic_floor, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream,
                     (const uint4 *)src, read_bytes / 16, (uint4 *)dst, write_bytes / 16 CM3_FLOOR_SPAN_ARG);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_hbm_read_bench_cfg(const void *buf, size_t bytes, void *sink, int32_t unroll, int32_t wg_per_cu, int32_t nt,
                           void *stream) {
  CM3_REQUIRE(buf && sink, "null buffer");
  CM3_REQUIRE(bytes >= 16 && bytes % 16 == 0, "bytes must be a positive multiple of 16");
  CM3_REQUIRE(wg_per_cu >= 1 && wg_per_cu <= 32, "wg_per_cu must be in 1..32");
  const int grid = 256 * wg_per_cu;
  hipStream_t s = (hipStream_t)stream;
  const uint4 *src = (const uint4 *)buf;
  const size_t n = bytes / 16;
  switch (unroll) {
    case 1: cm3::launch_read<1>(nt != 0, grid, s, src, n, (uint32_t *)sink); break;
    case 2: cm3::launch_read<2>(nt != 0, grid, s, src, n, (uint32_t *)sink); break;
    case 4: cm3::launch_read<4>(nt != 0, grid, s, src, n, (uint32_t *)sink); break;
    case 8: cm3::launch_read<8>(nt != 0, grid, s, src, n, (uint32_t *)sink); break;
    case 16: cm3::launch_read<16>(nt != 0, grid, s, src, n, (uint32_t *)sink); break;
    default: return cm3::fail(CM3_ERR_INVALID, "unroll must be 1, 2, 4, 8 or 16");
  }
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_hbm_read_bench(const void *buf, size_t bytes, void *sink, void *stream) {
  return cm3_hbm_read_bench_cfg(buf, bytes, sink, cm3::kBenchUnroll, cm3::kBenchWgPerCu, cm3::kBenchNt, stream);
}

int cm3_hbm_copy_bench_cfg(void *dst, const void *src, size_t bytes, int32_t unroll, int32_t wg_per_cu, int32_t nt,
                           void *stream) {
  CM3_REQUIRE(dst && src, "null buffer");
  CM3_REQUIRE(bytes >= 16 && bytes % 16 == 0, "bytes must be a positive multiple of 16");
  CM3_REQUIRE(wg_per_cu >= 1 && wg_per_cu <= 32, "wg_per_cu must be in 1..32");
  const int grid = 256 * wg_per_cu;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = bytes / 16;
  switch (unroll) {
    case 1: cm3::launch_copy<1>(nt != 0, grid, s, (uint4 *)dst, (const uint4 *)src, n); break;
    case 2: cm3::launch_copy<2>(nt != 0, grid, s, (uint4 *)dst, (const uint4 *)src, n); break;
    case 4: cm3::launch_copy<4>(nt != 0, grid, s, (uint4 *)dst, (const uint4 *)src, n); break;
    case 8: cm3::launch_copy<8>(nt != 0, grid, s, (uint4 *)dst, (const uint4 *)src, n); break;
    default: return cm3::fail(CM3_ERR_INVALID, "unroll must be 1, 2, 4 or 8");
  }
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_hbm_copy_bench(void *dst, const void *src, size_t bytes, void *stream) {
  return cm3_hbm_copy_bench_cfg(dst, src, bytes, 1, 4, 1, stream);  // best of the sweep: 5.56 TB/s
}

// ---- hipGraph capture ---------------------------------------------------------------------------------
int cm3_graph_begin(void *stream) {
  CM3_HIP_CHECK(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
  return CM3_OK;
}

int cm3_graph_end(void *stream, void **graph_exec) {
  CM3_REQUIRE(graph_exec, "null graph_exec");
  hipGraph_t graph = nullptr;
  CM3_HIP_CHECK(hipStreamEndCapture((hipStream_t)stream, &graph));
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return cm3::fail(CM3_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  *graph_exec = (void *)exec;
  return CM3_OK;
}

int cm3_graph_launch(void *graph_exec, void *stream) {
  CM3_REQUIRE(graph_exec, "null graph_exec");
  CM3_HIP_CHECK(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream));
  return CM3_OK;
}

int cm3_graph_destroy(void *graph_exec) {
  if (graph_exec) CM3_HIP_CHECK(hipGraphExecDestroy((hipGraphExec_t)graph_exec));
  return CM3_OK;
}

// ---- events ------------------------------------------------------------------------------------------
int cm3_event_create(void **event) {
  CM3_REQUIRE(event, "null event");
  hipEvent_t ev;
  CM3_HIP_CHECK(hipEventCreate(&ev));
  *event = (void *)ev;
  return CM3_OK;
}
int cm3_event_record(void *event, void *stream) {
  CM3_HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
  return CM3_OK;
}
int cm3_event_synchronize(void *event) {
  CM3_HIP_CHECK(hipEventSynchronize((hipEvent_t)event));
  return CM3_OK;
}
int cm3_event_elapsed_ms(void *start, void *stop, float *ms) {
  CM3_REQUIRE(ms, "null ms");
  CM3_HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
  return CM3_OK;
}
int cm3_event_destroy(void *event) {
  if (event) CM3_HIP_CHECK(hipEventDestroy((hipEvent_t)event));
  return CM3_OK;
}
int cm3_stream_synchronize(void *stream) {
  CM3_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return CM3_OK;
}
}
