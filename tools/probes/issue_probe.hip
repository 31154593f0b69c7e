// Diagnostic: VALU issue interval seen by ONE wave as a function of the number of waves resident on its SIMD and of the
// independence of its instructions (fully unrolled straight-line code, no loop branch in the timed region).
//   chain<ILP>: 512 v_fma_f32 in ILP independent dependency chains, timed with the shader clock by lane 0 of every wave.
// Launch shapes: waves per workgroup x workgroups chosen so that every SIMD holds 1, 2, 4 or 8 waves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int ILP, bool TRANS> __global__ void __launch_bounds__(1024) chain(float *out, long long *clk) {
  float x[ILP];
#pragma unroll
  for (int k = 0; k < ILP; ++k) x[k] = threadIdx.x * 1e-3f + k + 1.0f;
  const float y = 1.0001f;
  __builtin_amdgcn_sched_barrier(0);
  const long long c0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 512 / ILP; ++i) {
#pragma unroll
    for (int k = 0; k < ILP; ++k) {
      if constexpr (TRANS) x[k] = __builtin_amdgcn_rcpf(x[k]);
      else x[k] = __builtin_fmaf(x[k], y, 1e-7f);
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long c1 = clock64();
  float s = 0;
#pragma unroll
  for (int k = 0; k < ILP; ++k) s += x[k];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) clk[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
}
// integer multiplies (Philox): dependent chains of v_mad_u64_u32 / v_mul_hi_u32 / v_mul_lo_u32 / ds_bpermute_b32
template <int KIND> __global__ void __launch_bounds__(1024) ichain(unsigned *out, long long *clk) {
  unsigned x = threadIdx.x * 2654435761u + 12345u;
  unsigned long long acc = x;
  __builtin_amdgcn_sched_barrier(0);
  const long long c0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 256; ++i) {
    if constexpr (KIND == 0) { acc = (unsigned long long)(unsigned)acc * 0xD2511F53u + (acc >> 32); }          // v_mad_u64_u32
    else if constexpr (KIND == 1) { x = __umulhi(x, 0xD2511F53u) ^ 0x9E3779B9u; }                             // v_mul_hi_u32
    else if constexpr (KIND == 2) { x = x * 0xCD9E8D57u ^ 0x9E3779B9u; }                                      // v_mul_lo_u32
    else if constexpr (KIND == 3) { x = (unsigned)__builtin_amdgcn_ds_bpermute((int)((x & 63u) << 2), (int)x) + 1u; }  // shuffle
    else if constexpr (KIND == 4) { x = __builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, true) + 1u; }   // DPP row_shr:1
    else { x = ((x & 0xffffffu) * 0xD2511Fu) ^ 0x9E3779B9u; }                                        // v_mul_u32_u24
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long c1 = clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + (unsigned)acc + (unsigned)(acc >> 32);
  if ((threadIdx.x & 63) == 0) clk[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
}
template <int KIND> void irun(const char *name, float *out, long long *clk, int blocks, int threads) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((ichain<KIND>), dim3(blocks), dim3(threads), 0, 0, (unsigned *)out, clk);
  (void)hipDeviceSynchronize();
  const int nw = blocks * (threads / 64);
  std::vector<long long> h(nw);
  (void)hipMemcpy(h.data(), clk, nw * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long v : h) sum += v;
  printf("%-28s %4d WG x %4d threads (%d waves/SIMD): %.2f clocks per dependent step per wave\n", name, blocks, threads,
         blocks * (threads / 64) / 1024, sum / nw / 256.0);
}
template <int ILP, bool TRANS> void run(float *out, long long *clk, int blocks, int threads) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((chain<ILP, TRANS>), dim3(blocks), dim3(threads), 0, 0, out, clk);
  (void)hipDeviceSynchronize();
  const int nw = blocks * (threads / 64);
  std::vector<long long> h(nw);
  (void)hipMemcpy(h.data(), clk, nw * 8, hipMemcpyDeviceToHost);
  double sum = 0; long long mn = h[0], mx = h[0];
  for (long long v : h) { sum += v; mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
  printf("%s ILP=%d  %4d WG x %4d threads (%2d waves/SIMD if spread evenly): %.2f clocks per instruction per wave (min %.2f max %.2f)\n",
         TRANS ? "v_rcp_f32" : "v_fma_f32", ILP, blocks, threads, blocks * (threads / 64) / 1024, sum / nw / 512.0, mn / 512.0, mx / 512.0);
}
int main() {
  float *out; long long *clk;
  (void)hipMalloc(&out, (size_t)4096 * 1024 * 4); (void)hipMalloc(&clk, (size_t)4096 * 16 * 8);
  for (int w = 1; w <= 8; w *= 2) {
    run<1, false>(out, clk, 256, 256 * w);
    run<4, false>(out, clk, 256, 256 * w);
    if (w > 4) break;
  }
  run<1, false>(out, clk, 512, 1024);
  run<4, false>(out, clk, 512, 1024);
  for (int w = 1; w <= 4; w *= 2) {
    run<1, true>(out, clk, 256, 256 * w);
    run<4, true>(out, clk, 256, 256 * w);
  }
  for (int w = 1; w <= 2; w *= 2) {
    irun<0>("v_mad_u64_u32", out, clk, 256, 256 * w);
    irun<1>("v_mul_hi_u32 + v_xor", out, clk, 256, 256 * w);
    irun<2>("v_mul_lo_u32 + v_xor", out, clk, 256, 256 * w);
    irun<3>("v_and+v_lshl+ds_bpermute+v_add", out, clk, 256, 256 * w);
    irun<4>("v_mov_dpp row_shr:1 + v_add", out, clk, 256, 256 * w);
    irun<5>("v_mul_u32_u24 + v_xor", out, clk, 256, 256 * w);
  }
  return 0;
}
