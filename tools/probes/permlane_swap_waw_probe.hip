// Is a VALU write to the SECOND operand of v_permlane32_swap_b32 / v_permlane16_swap_b32 (the instruction writes both operands)
// safe right behind the swap?  The sequence the compiler produced in the withdrawn build of the policy rollout was
//     v_permlane32_swap_b32 vA, vB ; <one VALU> ; v_mov_b32 vB, 0        (vB re-used for a fresh variable)
// and rows came out wrong once a second wave shared the SIMD (profiles/r04_policy_head.txt (9)).  Here: the same sequence in inline
// assembly with GAP other VALU instructions in between, many waves per SIMD, every wave also issuing matrix instructions.
//   hipcc --offload-arch=gfx950 -O3 -o permlane_swap_waw_probe permlane_swap_waw_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int GAP, bool S32> __global__ void __launch_bounds__(256) k(unsigned *bad, int reps) {
  const unsigned lane = threadIdx.x & 63;
  f16x8 ones;
  for (int q = 0; q < 8; ++q) ones[q] = (_Float16)1.0f;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  unsigned nbad = 0;
  for (int r = 0; r < reps; ++r) {
    unsigned a = lane * 7u + (unsigned)r + 1u, b, z;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, ones, acc, 0, 0, 0);
    if (S32) {
      if (GAP == 0) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=v"(z));
      if (GAP == 1) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_add_u32 %2, %0, %0\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=&v"(z));
      if (GAP == 4) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 3\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=v"(z));
    } else {
      if (GAP == 0) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=v"(z));
      if (GAP == 1) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_add_u32 %2, %0, %0\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=&v"(z));
      if (GAP == 4) asm volatile("v_mov_b32 %1, %0\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 3\n\tv_mov_b32 %1, 0\n\ts_nop 7\n\ts_nop 7\n\tv_mov_b32 %2, %1" : "+v"(a), "=&v"(b), "=v"(z));
    }
    nbad += (z != 0u);
  }
  if (nbad || acc[0] < 0.0f) atomicAdd(bad, nbad);
}
template <int GAP, bool S32> static void run(int blocks, int reps) {
  unsigned *bad, h = 0;
  hipMalloc((void **)&bad, 4); hipMemset(bad, 0, 4);
  hipLaunchKernelGGL((k<GAP, S32>), dim3(blocks), dim3(256), 0, 0, bad, reps);
  hipDeviceSynchronize();
  hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
  printf("v_permlane%d_swap, %d instruction(s) before the overwrite of its second operand, %5d blocks: %u lanes read back non-zero of %llu\n",
         S32 ? 32 : 16, GAP, blocks, h, (unsigned long long)blocks * 256 * reps);
  hipFree(bad);
}
int main() {
  for (int blocks : {256, 2048}) {
    run<0, true>(blocks, 20000); run<1, true>(blocks, 20000); run<4, true>(blocks, 20000);
    run<0, false>(blocks, 20000); run<1, false>(blocks, 20000); run<4, false>(blocks, 20000);
  }
  return 0;
}
