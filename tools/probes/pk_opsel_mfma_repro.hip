// Round 5: STANDALONE REPRODUCER of the fault behind round 4's withdrawn policy layer (profiles/r05_policy_fault.txt).
//
// On MI355X (gfx950) a packed float32 VALU instruction whose LOW result takes the HIGH dword of a source register pair
// (`op_sel:[0,1]`: v_pk_mul_f32 vD, vA, vB op_sel:[0,1] computes vD.lo = vA.lo * vB.hi, vD.hi = vA.hi * vB.hi) returns a wrong LOW
// result -- as if vB.hi had been read as 0 -- in lanes 48..63, intermittently, WHILE ANOTHER WAVE ON THE SAME SIMD EXECUTES A
// float16 MATRIX INSTRUCTION (v_mfma_f32_16x16x32_f16).  The high result is never wrong; lanes 0..47 are never wrong; wait states
// around the instruction do not help; the wave that computes wrongly needs no matrix instruction of its own.
//
// Layout: one workgroup of 512 lanes per CU.  Waves 0..3 are VICTIMS (a loop over the packed instruction on pseudo-random operands,
// every result compared with two v_mul_f32), waves 4..7 AGGRESSORS (a loop of matrix instructions); wave w and wave w + 4 of a
// workgroup share SIMD w.  Without aggressors (second argument 0) nothing is ever wrong.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_opsel_mfma_repro pk_opsel_mfma_repro.hip && ./pk_opsel_mfma_repro [victim] [aggressor] [iters]
//     victim     0 v_pk_mul_f32 op_sel:[0,1]      (lo <- A.lo * B.hi; the form the compiler's SLP vectoriser produced)
//                1 v_pk_mul_f32 op_sel_hi:[1,0]   (hi <- A.hi * B.lo: the cross select on the HIGH half)
//                2 v_pk_mul_f32                   (no cross select)
//                3 v_pk_add_f32 op_sel:[0,1]
//                4 v_pk_fma_f32 op_sel:[0,1,0]
//                5 v_pk_mul_f32 op_sel:[1,0]      (lo <- A.hi * B.lo: the cross select on src0)
//                6 v_pk_mul_f32 op_sel:[0,1] with s_nop 7 in front of and behind it
//     aggressor  0 none   1 / 2 / 3 / 4: v_mfma_f32_16x16x32_f16 / 16x16x4_f32 / 16x16x32_bf16 / 32x32x16_f16 BACK TO BACK (the pipe never idles)
//                5 no matrix instruction: packed float32 FMAs (a busy vector wave)
//                6 ONE v_mfma_f32_16x16x32_f16 every ~64 cycles   7 / 8 / 9 the same rhythm with 16x16x4_f32 / 16x16x32_bf16 / 32x32x16_f16
//                10 one every ~16 cycles   11 four back to back, then ~64 idle cycles   12 the rhythm of 6 with a vector instruction instead
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

template <int V> __device__ __forceinline__ void victim(f32x2 a, f32x2 b, f32x2 c, f32x2 &r, f32x2 &want) {
  if constexpr (V == 0) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[0] * b[1], a[1] * b[1]}; }
  if constexpr (V == 1) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[0] * b[0], a[1] * b[0]}; }
  if constexpr (V == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[0] * b[0], a[1] * b[1]}; }
  if constexpr (V == 3) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[0] + b[1], a[1] + b[1]}; }
  if constexpr (V == 4) { asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(a), "v"(b), "v"(c)); want = f32x2{__builtin_fmaf(a[0], b[1], c[0]), __builtin_fmaf(a[1], b[1], c[1])}; }
  if constexpr (V == 5) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[1] * b[0], a[1] * b[1]}; }
  if constexpr (V == 6) { asm volatile("s_nop 7\n\tv_pk_mul_f32 %0, %1, %2 op_sel:[0,1]\n\ts_nop 7" : "=v"(r) : "v"(a), "v"(b)); want = f32x2{a[0] * b[1], a[1] * b[1]}; }
}

template <int V, int A> __global__ void __launch_bounds__(512) k_repro(unsigned *counts, float *sink, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  if (w < 4) {   // ---- victims ----------------------------------------------------------------------------------------------
    unsigned bad_lo = 0, bad_hi = 0, zero_lo = 0;
    for (int it = 0; it < iters; ++it) {
      const uint32_t h = mix((uint32_t)(blockIdx.x * 512 + tid) * 2654435761u + (uint32_t)it), h2 = mix(h ^ 0x9e3779b9u);
      f32x2 a, b, c, r = {0.0f, 0.0f}, want = {0.0f, 0.0f};
      a[0] = ((float)(int)(h & 0xffff) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      a[1] = ((float)(int)(h >> 16) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      b[0] = 1.5f + (float)(h2 & 0xffff) * (1.0f / 65536.0f);
      b[1] = 1e-3f + (float)(h2 >> 16) * (0.05f / 65536.0f);
      c[0] = a[1] * 0.25f;
      c[1] = a[0] * 0.5f;
      victim<V>(a, b, c, r, want);
      // (the expected values come from scalar float instructions: -ffp-contract is off for this file's purposes because every
      // product above is a single operation)
      if (__float_as_uint(r[0]) != __float_as_uint(want[0])) { bad_lo += 1; if (r[0] == 0.0f || r[0] == a[0] * 0.0f) zero_lo += 1; }
      if (__float_as_uint(r[1]) != __float_as_uint(want[1])) bad_hi += 1;
    }
    if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);
    if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);
    if (zero_lo) atomicAdd(&counts[8], zero_lo);
  } else {       // ---- aggressors -------------------------------------------------------------------------------------------
    f16x8 x, y;
    bf16x8 xb, yb;
    for (int j = 0; j < 8; ++j) {
      x[j] = (_Float16)(0.01f * (float)((lane + j) % 7)); y[j] = (_Float16)(0.02f * (float)((lane + 3 * j) % 5));
      xb[j] = (__bf16)(0.01f * (float)((lane + j) % 7)); yb[j] = (__bf16)(0.02f * (float)((lane + 3 * j) % 5));
    }
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x16 big = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    f32x2 p = {1.0f, 2.0f}, q = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
      if constexpr (A == 1) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k], 0, 0, 0);
      } else if constexpr (A == 2) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)x[k], (float)y[k], acc[k], 0, 0, 0);
      } else if constexpr (A == 3) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, acc[k], 0, 0, 0);
      } else if constexpr (A == 4) {
        big = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, big, 0, 0, 0);
      } else if constexpr (A == 5) {
#pragma unroll
        for (int k = 0; k < 8; ++k) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(q));
      } else if constexpr (A == 6) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[0], 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      } else if constexpr (A == 7) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)x[0], (float)y[0], acc[0], 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      } else if constexpr (A == 8) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xb, yb, acc[0], 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      } else if constexpr (A == 9) {
        big = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, big, 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      } else if constexpr (A == 10) {
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[0], 0, 0, 0);
        asm volatile("s_nop 15" ::: "memory");
      } else if constexpr (A == 11) {
#pragma unroll
        for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, acc[k], 0, 0, 0);
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      } else if constexpr (A == 12) {   // no matrix instruction: a vector instruction every ~64 cycles
        asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(p) : "v"(q), "v"(q));
        asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
      }
    }
    float v = p[0] + p[1] + big[0] + big[5];
    for (int k = 0; k < 4; ++k) v += acc[k][0] + acc[k][3];
    if (v == 123.456f) sink[tid] = v;
  }
}

template <int V> static void launch(int a, unsigned *c, float *s, int iters) {
  switch (a) {
    case 0: hipLaunchKernelGGL((k_repro<V, 0>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 1: hipLaunchKernelGGL((k_repro<V, 1>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 2: hipLaunchKernelGGL((k_repro<V, 2>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 3: hipLaunchKernelGGL((k_repro<V, 3>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 4: hipLaunchKernelGGL((k_repro<V, 4>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 5: hipLaunchKernelGGL((k_repro<V, 5>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 6: hipLaunchKernelGGL((k_repro<V, 6>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 7: hipLaunchKernelGGL((k_repro<V, 7>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 8: hipLaunchKernelGGL((k_repro<V, 8>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 9: hipLaunchKernelGGL((k_repro<V, 9>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 10: hipLaunchKernelGGL((k_repro<V, 10>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    case 11: hipLaunchKernelGGL((k_repro<V, 11>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
    default: hipLaunchKernelGGL((k_repro<V, 12>), dim3(256), dim3(512), 0, 0, c, s, iters); break;
  }
}

int main(int argc, char **argv) {
  const int vic = argc > 1 ? atoi(argv[1]) : 0, agg = argc > 2 ? atoi(argv[2]) : 1, iters = argc > 3 ? atoi(argv[3]) : 20000;
  unsigned *counts;
  float *sink;
  CHECK(hipMalloc(&counts, 64));
  CHECK(hipMalloc(&sink, 4096));
  CHECK(hipMemset(counts, 0, 64));
  switch (vic) {
    case 0: launch<0>(agg, counts, sink, iters); break;
    case 1: launch<1>(agg, counts, sink, iters); break;
    case 2: launch<2>(agg, counts, sink, iters); break;
    case 3: launch<3>(agg, counts, sink, iters); break;
    case 4: launch<4>(agg, counts, sink, iters); break;
    case 5: launch<5>(agg, counts, sink, iters); break;
    default: launch<6>(agg, counts, sink, iters); break;
  }
  CHECK(hipGetLastError());
  CHECK(hipDeviceSynchronize());
  unsigned h[16];
  CHECK(hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost));
  const double per_row = 1024.0 * 16.0 * (double)iters;
  printf("victim %d aggressor %d: wrong LOW results in lanes 0-15 / 16-31 / 32-47 / 48-63: %u %u %u %u (%.3f %% of the last row's, %u of them exactly the product with 0)   wrong HIGH: %u %u %u %u\n",
         vic, agg, h[0], h[1], h[2], h[3], 100.0 * h[3] / per_row, h[8], h[4], h[5], h[6], h[7]);
  return 0;
}
