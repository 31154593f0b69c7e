// Does a matrix instruction's A / B operand survive being reloaded from LDS right behind it, when several waves share a SIMD?
// (profiles/r04_policy_head.txt (9): the particle actor's second layer produced wrong rows with the activations as the B operand,
// reloaded inside the k loop, once two workgroups shared a CU.)
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-mfma-vgpr-form=1 -o mfma_operand_reload_probe mfma_operand_reload_probe.hip
// Every wave accumulates S k-steps: x_s (all halves = v_s, read from LDS into the SAME registers every step) times W (all ones), as
// the A or as the B operand.  Expected accumulator: 32 * sum_s v_s in every element, exactly.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int S = 96;
template <bool ACT_AS_B, int PAD> __global__ void __launch_bounds__(256) k(unsigned *bad, int reps) {
  __shared__ __attribute__((aligned(16))) f16x8 xs[4][8][64];   // eight distinct steps, cycled
  __shared__ float pad[PAD];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  if (PAD > 1 && reps < 0) pad[lane] = 1.0f;
  float want = 0.0f;
  for (int s = 0; s < S; ++s) {
    const _Float16 v = (_Float16)(float)(((s & 7) * 5 + w) % 7 + 1);
    f16x8 t;
    for (int q = 0; q < 8; ++q) t[q] = v;
    if (s < 8) xs[w][s][lane] = t;
    want += 32.0f * (float)v;
  }
  __syncthreads();
  f16x8 ones;
  for (int q = 0; q < 8; ++q) ones[q] = (_Float16)1.0f;
  unsigned nbad = 0;
  for (int r = 0; r < reps; ++r) {
    f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f}, acc2 = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 1
    for (int s = 0; s < S; s += 2) {
      const f16x8 x0 = xs[w][s & 7][lane], x1 = xs[w][(s + 1) & 7][lane];
      if (ACT_AS_B) {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, x0, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, x1, acc2, 0, 0, 0);
      } else {
        acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, ones, acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, ones, acc2, 0, 0, 0);
      }
    }
    for (int q = 0; q < 4; ++q) nbad += (acc[q] + acc2[q] != want);
  }
  if (nbad) atomicAdd(bad, nbad);
}
template <bool B, int PAD> static void run(const char *what, int blocks, int reps) {
  unsigned *bad, h = 0;
  hipMalloc((void **)&bad, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((k<B, PAD>), dim3(blocks), dim3(256), 0, 0, bad, reps);
  hipDeviceSynchronize();
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("%-44s blocks %5d: %u wrong accumulator elements of %llu\n", what, blocks, h, (unsigned long long)blocks * 256 * 4 * reps);
  hipFree(bad);
}
int main() {
  const int reps = 200;
  run<false, 1>("activations = A operand, CU shared", 2048, reps);
  run<true, 1>("activations = B operand, CU shared", 2048, reps);
  run<false, 22000>("activations = A operand, alone on the CU", 256, reps);
  run<true, 22000>("activations = B operand, alone on the CU", 256, reps);
  return 0;
}
