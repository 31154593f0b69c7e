// Diagnostic: effective shader clock / issue rate at different occupancies (is the chip in a low DPM state
// while a stream of tiny kernels runs?).
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int ILP> __global__ void chain(float *out, long long *clk, int iters) {
  float x[ILP];
  for (int k = 0; k < ILP; ++k) x[k] = threadIdx.x * 1e-3f + k;
  const float y = 1.0001f;
  long long c0 = clock64(), w0 = wall_clock64();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int k = 0; k < ILP; ++k) x[k] = __builtin_fmaf(x[k], y, 1e-7f);
  }
  long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int k = 0; k < ILP; ++k) s += x[k];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}
template <int ILP> void run(hipStream_t s, float *out, long long *clk, int blocks, int threads, int iters, int launches) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a, s);
  for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(chain<ILP>, dim3(blocks), dim3(threads), 0, s, out, clk, iters);
  hipEventRecord(b, s); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  long long h[2]; (void)hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
  double ns_per_fma = (double)h[1] * 10.0 / ((double)iters * ILP);
  printf("ILP=%d blocks=%5d threads=%4d iters=%7d launches=%4d: %10.3f us/launch; %.3f ns per fma per wave (%.2f cyc @2.4GHz)\n",
         ILP, blocks, threads, iters, launches, ms * 1e3 / launches, ns_per_fma, ns_per_fma * 2.4);
}
int main() {
  float *out; long long *clk;
  (void)hipMalloc(&out, (size_t)8192 * 1024 * 4); (void)hipMalloc(&clk, 8192 * 16);
  hipStream_t s; (void)hipStreamCreate(&s);
  for (int rep = 0; rep < 2; ++rep) {
    run<1>(s, out, clk, 64, 64, 2000, 1000);
    run<1>(s, out, clk, 1024, 256, 200000, 3);
    run<1>(s, out, clk, 2048, 1024, 200000, 3);
    run<8>(s, out, clk, 64, 64, 2000, 1000);
    run<8>(s, out, clk, 1024, 256, 100000, 3);
    run<8>(s, out, clk, 2048, 1024, 100000, 3);
  }
  return 0;
}
