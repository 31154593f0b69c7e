// Round 5: FIRST reproducer attempts for the k_policy_rollout fault (profiles/r05_policy_fault.txt) -- superseded by
// pk_opsel_mfma_repro.hip, which reproduces it; kept because its null results are part of the record: aggressor waves that keep
// the matrix pipe BUSY never trigger the fault, the alternating-phase variant (PROBE_TICKS=1) does.
//
// In the failing builds ONE instruction computes a wrong value, in lanes 48..63 only, a handful of times per launch:
//     v_div_fixup_f32 v84, v83, v71, v84
//     v_pk_mul_f32    v[80:81], v[84:85], v[80:81] op_sel:[0,1]        ; (qx * pen, qy * pen): pen is the HIGH word of src1 = dst
//     s_or_b64        exec, exec, s[2:3]
// low result = +-0 (sign of qx) instead of qx * pen, high result right -- and only while the SIMD's other wave runs the transposed
// float16 second layer (v_mfma_f32_16x16x32_f16 with ds_read_b128 reloads).  Replacing that one v_pk_mul_f32 by two v_mul_f32 makes
// the kernel clean; so does -packed-fp32-ops.
//
// Here: 512 workgroups of 256 lanes (two per CU, one wave of each per SIMD).  Workgroups of the first 256 are VICTIMS (loop over the
// three instructions above under a partial exec mask, compare with the same product computed by v_mul_f32), the others AGGRESSORS
// (a k-loop of the matrix instructions named on the command line).  Prints the number of wrong low / high results by 16-lane row.
//
//   hipcc --offload-arch=gfx950 -O3 -o pk_mul_opsel_probe pk_mul_opsel_probe.hip && ./pk_mul_opsel_probe [aggressor] [victim] [iters]
//     aggressor: 0 none (victims alone, one wave per SIMD)  1 f16 MFMA + ds_read_b128 (the transposed layer)  2 f16 MFMA only
//                3 ds_read_b128 only  4 f32 MFMA (16x16x4)  5 victims on both sides
//     victim:    0 the failing form   1 dst not overlapping src1   2 op_sel_hi:[1,0] form (pen in the LOW word)   3 s_nop 1 in front
//                4 without the exec change behind it   5 full exec mask
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int VICTIM> __device__ __forceinline__ void victim_op(float qx, float qy, float junk, float pen, float &rx, float &ry) {
  // fixed registers, as in the failing build
  if constexpr (VICTIM == 0 || VICTIM == 4 || VICTIM == 5)
    asm volatile("v_mov_b32 v84, %2\n\tv_mov_b32 v85, %3\n\tv_mov_b32 v80, %4\n\tv_mov_b32 v81, %5\n\ts_nop 4\n\t"
                 "v_div_fixup_f32 v84, v84, 1.0, v84\n\t"
                 "v_pk_mul_f32 v[80:81], v[84:85], v[80:81] op_sel:[0,1]\n\t"
                 "s_nop 4\n\tv_mov_b32 %0, v80\n\tv_mov_b32 %1, v81"
                 : "=v"(rx), "=v"(ry) : "v"(qx), "v"(qy), "v"(junk), "v"(pen) : "v80", "v81", "v84", "v85");
  else if constexpr (VICTIM == 1)
    asm volatile("v_mov_b32 v84, %2\n\tv_mov_b32 v85, %3\n\tv_mov_b32 v80, %4\n\tv_mov_b32 v81, %5\n\ts_nop 4\n\t"
                 "v_div_fixup_f32 v84, v84, 1.0, v84\n\t"
                 "v_pk_mul_f32 v[88:89], v[84:85], v[80:81] op_sel:[0,1]\n\t"
                 "s_nop 4\n\tv_mov_b32 %0, v88\n\tv_mov_b32 %1, v89"
                 : "=v"(rx), "=v"(ry) : "v"(qx), "v"(qy), "v"(junk), "v"(pen) : "v80", "v81", "v84", "v85", "v88", "v89");
  else if constexpr (VICTIM == 2)
    asm volatile("v_mov_b32 v84, %2\n\tv_mov_b32 v85, %3\n\tv_mov_b32 v81, %4\n\tv_mov_b32 v80, %5\n\ts_nop 4\n\t"
                 "v_div_fixup_f32 v84, v84, 1.0, v84\n\t"
                 "v_pk_mul_f32 v[80:81], v[84:85], v[80:81] op_sel_hi:[1,0]\n\t"
                 "s_nop 4\n\tv_mov_b32 %0, v80\n\tv_mov_b32 %1, v81"
                 : "=v"(rx), "=v"(ry) : "v"(qx), "v"(qy), "v"(junk), "v"(pen) : "v80", "v81", "v84", "v85");
  else
    asm volatile("v_mov_b32 v84, %2\n\tv_mov_b32 v85, %3\n\tv_mov_b32 v80, %4\n\tv_mov_b32 v81, %5\n\ts_nop 4\n\t"
                 "v_div_fixup_f32 v84, v84, 1.0, v84\n\ts_nop 1\n\t"
                 "v_pk_mul_f32 v[80:81], v[84:85], v[80:81] op_sel:[0,1]\n\t"
                 "s_nop 4\n\tv_mov_b32 %0, v80\n\tv_mov_b32 %1, v81"
                 : "=v"(rx), "=v"(ry) : "v"(qx), "v"(qy), "v"(junk), "v"(pen) : "v80", "v81", "v84", "v85");
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
  return x;
}

template <int AGG, int VICTIM> __global__ void __launch_bounds__(256) k_probe(unsigned *counts, float *sink, int iters, int victims_blocks) {
  __shared__ __attribute__((aligned(16))) _Float16 acts[64][200 + 8];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
#ifdef PROBE_BIG_REGS
  asm volatile("" ::: "v254");    // the failing kernel allocates 255 registers: the SIMD's second wave lives in the upper half of the file
#endif
  for (int i = tid; i < 64 * 208; i += 256) (&acts[0][0])[i] = (_Float16)(0.001f * (float)((i * 7) % 13));
  __syncthreads();
  const bool victim = (int)blockIdx.x < victims_blocks || AGG == 5;
  if (victim) {
    unsigned bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
      const uint32_t h = mix((uint32_t)(blockIdx.x * 256 + tid) * 2654435761u + (uint32_t)it);
      // values of the contact chain's range: quotients up to +-100, penetration terms 1e-9 .. 0.05
      const float qx = ((float)(int)(h & 0xffff) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      const float qy = ((float)(int)((h >> 16) & 0xffff) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      const float pen = 1e-9f + (float)(mix(h) & 0xffff) * (0.05f / 65536.0f);
      const float junk = qy * 3.0f;
      // partial exec mask that keeps the four copies of a row together (lanes l, l + 16, l + 32, l + 48), as the kernel's slow path
      const bool active = VICTIM == 5 || ((mix(h + (uint32_t)(lane & 15) * 977u - (uint32_t)lane * 0u) >> 3) & 1u) != 0 || (lane & 15) < 2;
      float rx = 0.0f, ry = 0.0f;
      if (active) {
        victim_op<VICTIM>(qx, qy, junk, pen, rx, ry);
      }
      if constexpr (VICTIM == 4) asm volatile("s_nop 8" ::: "memory");
      if (active) {
        float wx, wy;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(wx), "=&v"(wy) : "v"(qx), "v"(qy), "v"(pen));
        if (__float_as_uint(wx) != __float_as_uint(rx)) bad_lo += 1;
        if (__float_as_uint(wy) != __float_as_uint(ry)) bad_hi += 1;
      }
    }
    if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);
    if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);
    if (lane == 0) atomicAdd(&counts[8], 1u);
  } else {
    // the transposed second layer's k-loop: weights as the A operand (registers), activation rows reloaded from LDS as B
    const int col = lane & 15, hi = lane >> 4;
    f16x8 wgt[6];
    for (int s = 0; s < 6; ++s)
      for (int j = 0; j < 8; ++j) wgt[s][j] = (_Float16)(0.01f * (float)((lane + s + j) % 7));
    f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, accs[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    f32x4 facc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        f16x8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          if constexpr (AGG == 1 || AGG == 3) {
            ah[t] = *reinterpret_cast<const f16x8 *>(&acts[16 * t + col][32 * s + 8 * hi]);
            al[t] = *reinterpret_cast<const f16x8 *>(&acts[(16 * t + col + 7) & 63][32 * s + 8 * hi]);
          } else {
            ah[t] = wgt[(s + t) % 6];
            al[t] = wgt[(s + t + 1) % 6];
          }
        }
        if constexpr (AGG == 1 || AGG == 2) {
#pragma unroll
          for (int t = 0; t < 4; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[s], al[t], accs[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[s], ah[t], acc[t], 0, 0, 0);
#pragma unroll
          for (int t = 0; t < 4; ++t) accs[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[(s + 1) % 6], ah[t], accs[t], 0, 0, 0);
        } else if constexpr (AGG == 3) {
#pragma unroll
          for (int t = 0; t < 4; ++t) facc[t] += (float)ah[t][0] + (float)al[t][1];
        } else if constexpr (AGG == 4) {
#pragma unroll
          for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32((float)wgt[s][t], (float)wgt[s][t + 1], acc[t], 0, 0, 0);
        }
      }
    }
    float v = facc[0] + facc[1] + facc[2] + facc[3];
    for (int t = 0; t < 4; ++t) v += acc[t][0] + accs[t][1];
    if (v == 123.456f) sink[tid] = v;
    if (lane == 0) atomicAdd(&counts[9], 1u);
  }
}

// v2 (after the cuts of the real kernel still failed with only the matrix layer, a barrier and ONE contact chain left): every
// workgroup alternates, tick after tick, between the matrix phase (the transposed float16 k-loop, LDS reloads) and the victim
// phase (the packed multiply under a partial exec mask, checked against v_mul_f32), with workgroup barriers between them -- two
// workgroups per CU drift against each other as the real ones do.  AGG here: 1 = with the matrix phase, 0 = without.
template <int AGG, int VICTIM> __global__ void __launch_bounds__(256) k_probe_ticks(unsigned *counts, float *sink, int ticks, int per_tick, int mode) {
  __shared__ __attribute__((aligned(16))) _Float16 acts[64][200 + 8];
  __shared__ float pad[8192];     // ~57 KB per workgroup with acts: two workgroups per CU, not more than a few
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 208; i += 256) (&acts[0][0])[i] = (_Float16)(0.001f * (float)((i * 7) % 13));
  if (tid == 0 && ticks < 0) pad[0] = 1.0f;
  __syncthreads();
  const int col = lane & 15, hi = lane >> 4;
  f16x8 wgt[6];
  for (int s = 0; s < 6; ++s)
    for (int j = 0; j < 8; ++j) wgt[s][j] = (_Float16)(0.01f * (float)((lane + s + j) % 7));
  f32x4 acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}}, accs[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned bad_lo = 0, bad_hi = 0;
  // mode 0: every workgroup alternates between the two phases (the real kernel's structure)
  //      1: workgroups >= 256 run only matrix phases, the others only victim phases -- after ONE matrix phase at tick 0
  //      2: the same split, the victim workgroups never execute a matrix instruction
  //      3: mode 0 without the workgroup barriers
  const bool agg_only = (mode == 1 || mode == 2) && blockIdx.x >= 256, vic_only = (mode == 1 || mode == 2) && blockIdx.x < 256;
  for (int t = 0; t < ticks; ++t) {
    if (AGG != 0 && !(vic_only && (mode == 2 || t > 0))) {
#pragma unroll
      for (int s = 0; s < 6; ++s) {
        f16x8 ah[4], al[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          ah[q] = *reinterpret_cast<const f16x8 *>(&acts[16 * q + col][32 * s + 8 * hi]);
          al[q] = *reinterpret_cast<const f16x8 *>(&acts[(16 * q + col + 7) & 63][32 * s + 8 * hi]);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) accs[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[s], al[q], accs[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[s], ah[q], acc[q], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) accs[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wgt[(s + 1) % 6], ah[q], accs[q], 0, 0, 0);
      }
      // EVERY accumulator register is read by a vector instruction the compiler can see (it inserts the wait states a matrix
      // result needs), a little of the sum goes back into the activations, and 64 more wait states follow: no matrix instruction
      // of THIS wave is in flight when the hand-written victim instructions below touch their fixed registers
      float sum = 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) sum += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3] + accs[q][0] + accs[q][1] + accs[q][2] + accs[q][3];
      acts[(lane + t) & 63][(tid >> 6) * 8 + (t & 7)] = (_Float16)(sum * 1e-9f);
      asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    }
    if (mode != 3) __syncthreads();
    for (int it = 0; it < (agg_only ? 0 : per_tick); ++it) {
      const uint32_t h = mix((uint32_t)(blockIdx.x * 256 + tid) * 2654435761u + (uint32_t)(t * per_tick + it));
      const float qx = ((float)(int)(h & 0xffff) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      const float qy = ((float)(int)((h >> 16) & 0xffff) - 32768.0f) * (100.0f / 32768.0f) + 0.001f;
      const float pen = 1e-9f + (float)(mix(h) & 0xffff) * (0.05f / 65536.0f);
      const float junk = qy * 3.0f;
      const bool active = VICTIM == 5 || ((mix(h + (uint32_t)(lane & 15) * 977u) >> 3) & 1u) != 0 || (lane & 15) < 2;
      float rx = 0.0f, ry = 0.0f;
      if (active) victim_op<VICTIM>(qx, qy, junk, pen, rx, ry);
      if (active) {
        float wx, wy;
        asm volatile("v_mul_f32 %0, %2, %4\n\tv_mul_f32 %1, %3, %4" : "=&v"(wx), "=&v"(wy) : "v"(qx), "v"(qy), "v"(pen));
        if (__float_as_uint(wx) != __float_as_uint(rx)) bad_lo += 1;
        if (__float_as_uint(wy) != __float_as_uint(ry)) bad_hi += 1;
      }
    }
    if (mode != 3) __syncthreads();
  }
  if (bad_lo) atomicAdd(&counts[lane >> 4], bad_lo);
  if (bad_hi) atomicAdd(&counts[4 + (lane >> 4)], bad_hi);
  if (lane == 0) atomicAdd(&counts[8], 1u);
  float v = 0.0f;
  for (int q = 0; q < 4; ++q) v += acc[q][0] + accs[q][1];
  if (v == 123.456f) sink[tid] = v + pad[tid];
}

template <int AGG, int VICTIM> static void run(unsigned *counts, float *sink, int iters) {
  if (getenv("PROBE_TICKS")) {
    const int blocks = getenv("PROBE_BLOCKS") ? atoi(getenv("PROBE_BLOCKS")) : 512;
    const int threads = getenv("PROBE_THREADS") ? atoi(getenv("PROBE_THREADS")) : 256;
    const int mode = getenv("PROBE_MODE") ? atoi(getenv("PROBE_MODE")) : 0;
    hipLaunchKernelGGL((k_probe_ticks<(AGG != 0), VICTIM>), dim3(blocks), dim3(threads), 0, 0, counts, sink, iters / 8, 8, mode);
    return;
  }
  const int victims = AGG == 0 ? 256 : 256, blocks = AGG == 0 ? 256 : 512;
  hipLaunchKernelGGL((k_probe<AGG, VICTIM>), dim3(blocks), dim3(256), 0, 0, counts, sink, iters, victims);
}
template <int AGG> static void run_v(int v, unsigned *c, float *s, int iters) {
  switch (v) {
    case 0: run<AGG, 0>(c, s, iters); break;
    case 1: run<AGG, 1>(c, s, iters); break;
    case 2: run<AGG, 2>(c, s, iters); break;
    case 3: run<AGG, 3>(c, s, iters); break;
    case 4: run<AGG, 4>(c, s, iters); break;
    default: run<AGG, 5>(c, s, iters); break;
  }
}

int main(int argc, char **argv) {
  const int agg = argc > 1 ? atoi(argv[1]) : 1, vic = argc > 2 ? atoi(argv[2]) : 0, iters = argc > 3 ? atoi(argv[3]) : 20000;
  unsigned *counts;
  float *sink;
  CHECK(hipMalloc(&counts, 64));
  CHECK(hipMalloc(&sink, 1024));
  for (int rep = 0; rep < 3; ++rep) {
    CHECK(hipMemset(counts, 0, 64));
    switch (agg) {
      case 0: run_v<0>(vic, counts, sink, iters); break;
      case 1: run_v<1>(vic, counts, sink, iters); break;
      case 2: run_v<2>(vic, counts, sink, iters); break;
      case 3: run_v<3>(vic, counts, sink, iters); break;
      case 4: run_v<4>(vic, counts, sink, iters); break;
      default: run_v<5>(vic, counts, sink, iters); break;
    }
    CHECK(hipDeviceSynchronize());
    unsigned h[16];
    CHECK(hipMemcpy(h, counts, 64, hipMemcpyDeviceToHost));
    printf("aggressor %d victim %d iters %d: wrong LOW results by 16-lane row [%u %u %u %u]  wrong HIGH [%u %u %u %u]  (victim waves %u, aggressor waves %u)\n",
           agg, vic, iters, h[0], h[1], h[2], h[3], h[4], h[5], h[6], h[7], h[8], h[9]);
  }
  return 0;
}
