// Error plumbing, device queries, the HBM read-bandwidth probe, hipGraph capture and HIP events.
#include <stdarg.h>

#include "common.h"

namespace cm3 {

char *last_error_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(last_error_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

// ---- streaming read probe: the measured roofline denominator ----------------------
// Grid-stride 16-byte loads, 4 independent loads in flight per lane per iteration; the XOR fold keeps
// the loads live.  One word per workgroup is written so the kernel has an observable result.
constexpr int kBenchBlock = 256;
constexpr int kBenchGrid = 256 * 8;  // 8 workgroups per CU

__global__ void __launch_bounds__(kBenchBlock) k_hbm_read(const uint4 *__restrict__ src, size_t n_vec,
                                                          uint32_t *__restrict__ sink) {
  const size_t stride = (size_t)gridDim.x * kBenchBlock;
  size_t i = (size_t)blockIdx.x * kBenchBlock + threadIdx.x;
  uint32_t acc = 0;
  for (; i + 3 * stride < n_vec; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
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


// Launch-structure floor of a step launch: the same grid reads `n_read` 16-byte vectors (all loads first), then writes
// `n_write` vectors whose value depends on everything it read -- load -> (no arithmetic) -> store, nothing else.  What a
// one-launch-per-tick kernel with this traffic cannot go below.
__global__ void __launch_bounds__(1024) k_traffic_floor(const uint4 *__restrict__ src, size_t n_read, uint4 *__restrict__ dst,
                                                        size_t n_write) {
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
}
}  // namespace cm3

extern "C" {

int cm3_abi_version(void) { return CM3_ABI_VERSION; }

const char *cm3_last_error(void) { return cm3::last_error_buf(); }

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

int cm3_hbm_bench_sink_words(void) { return cm3::kBenchGrid; }

int cm3_traffic_floor_bench(const void *src, size_t read_bytes, void *dst, size_t write_bytes, int32_t blocks,
                            int32_t threads, void *stream) {
  CM3_REQUIRE(src && dst, "null buffer");
  CM3_REQUIRE(read_bytes % 16 == 0 && write_bytes % 16 == 0, "byte counts must be multiples of 16");
  CM3_REQUIRE(blocks >= 1 && threads >= 64 && threads <= 1024 && threads % 64 == 0, "bad launch shape");
  hipLaunchKernelGGL(cm3::k_traffic_floor, dim3((unsigned)blocks), dim3((unsigned)threads), 0, (hipStream_t)stream,
                     (const uint4 *)src, read_bytes / 16, (uint4 *)dst, write_bytes / 16);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_hbm_read_bench(const void *buf, size_t bytes, void *sink, void *stream) {
  CM3_REQUIRE(buf && sink, "null buffer");
  CM3_REQUIRE(bytes >= 16 && bytes % 16 == 0, "bytes must be a positive multiple of 16");
  hipLaunchKernelGGL(cm3::k_hbm_read, dim3(cm3::kBenchGrid), dim3(cm3::kBenchBlock), 0, (hipStream_t)stream,
                     (const uint4 *)buf, bytes / 16, (uint32_t *)sink);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
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
