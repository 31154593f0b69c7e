// Diagnostic: what a VALU -> SGPR -> SALU -> VALU round trip costs a lone wave (one wave per SIMD), against staying in the VALU.
// Dependent chains of 128 steps, fully unrolled, timed with the shader clock by lane 0 of every wave (like issue_probe.hip).
//   0  v_cmp -> vcc -> v_cndmask                         (VALU only, condition through VCC)
//   1  v_cmp -> s[..] -> s_and_b64 -> v_cndmask          (one SALU op between the compare and its consumer)
//   2  v_cmp -> s[..] (ballot) -> s_bcnt1 -> v_add       (wave ballot + scalar popcount fed back to the VALU)
//   3  v_readfirstlane -> s_add -> v_add
//   4  integer-mask form of 1: v_sub, v_lshrrev 31, v_and, v_mul/or ... no SGPR
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
template <int KIND> __global__ void __launch_bounds__(256) chain(unsigned *out, long long *clk, unsigned seed) {
  unsigned x = threadIdx.x * 2654435761u + seed, y = x ^ 0x5bd1e995u;
  __builtin_amdgcn_sched_barrier(0);
  const long long c0 = clock64();
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int i = 0; i < 128; ++i) {
    if constexpr (KIND == 0) {
      x = (x > 0x80000000u) ? x + 3u : x * 5u + 1u;
    } else if constexpr (KIND == 1) {
      const bool a = x > 0x80000000u, b = y > 0x40000000u;
      x = (a & b) ? x + 3u : x * 5u + 1u;
      y += x;
    } else if constexpr (KIND == 2) {
      const unsigned long long m = __ballot(x > 0x80000000u);
      x = x * 5u + (unsigned)__popcll(m);
    } else if constexpr (KIND == 3) {
      const unsigned s = (unsigned)__builtin_amdgcn_readfirstlane((int)x) + 7u;
      x = x * 5u + s;
    } else {
      const unsigned a = (0x80000000u - x) >> 31, b = (0x40000000u - y) >> 31;   // 0 / 1 masks in VGPRs
      const unsigned m = 0u - (a & b);
      x = ((x + 3u) & m) | ((x * 5u + 1u) & ~m);
      y += x;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long c1 = clock64();
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = x + y;
  if ((threadIdx.x & 63) == 0) clk[(size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = c1 - c0;
}
template <int KIND> void run(const char *name, unsigned *out, long long *clk) {
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((chain<KIND>), dim3(256), dim3(256), 0, 0, out, clk, 12345u + rep);
  (void)hipDeviceSynchronize();
  std::vector<long long> h(1024);
  (void)hipMemcpy(h.data(), clk, 1024 * 8, hipMemcpyDeviceToHost);
  double sum = 0;
  for (long long v : h) sum += v;
  printf("%-52s %.1f clocks per dependent step\n", name, sum / 1024 / 128.0);
}
int main() {
  unsigned *out; long long *clk;
  (void)hipMalloc(&out, 65536 * 4); (void)hipMalloc(&clk, 1024 * 8);
  run<0>("0 v_cmp -> vcc -> v_cndmask (+ add, mul-add)", out, clk);
  run<1>("1 two v_cmp -> s_and_b64 -> v_cndmask (+ add, mad, add)", out, clk);
  run<2>("2 v_cmp -> ballot -> s_bcnt1 -> v_mad", out, clk);
  run<3>("3 v_readfirstlane -> s_add -> v_mad", out, clk);
  run<4>("4 integer masks in VGPRs (no SGPR), same logic as 1", out, clk);
  return 0;
}
