// Pieces shared by the particle actor (actor.hip, policy.hip) and the Checkers actor (actor_checkers.hip).
#pragma once
#include "common.h"
#include "philox.h"

namespace cm3 {

constexpr int kA = 5;  // l_action
constexpr uint32_t kPurposePolicy = 0x40000000u;

// The matrix kernels are built for TWO waves per SIMD.  With that as the declared minimum occupancy the register budget is 256 and
// the compiler's default selection keeps the matrix accumulators in architectural VGPRs (no v_accvgpr_read per value in the
// epilogues) -- round 4 forced the same code with the experimental -amdgpu-mfma-vgpr-form=1; round 5 showed that the attribute
// gives it by the standard path (tools/probes/policy_fault_variants.sh, variant agpr2w) and dropped the flag.  Consequence that
// stays: NO inline assembly may read a matrix instruction's result (the hazard recogniser does not see into asm statements).
#define CM3_MATRIX_KERNEL __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2)))

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// relu as ONE instruction: fmaxf(x, 0) first canonicalises x (a second v_max_f32 x, x) because the build honours signalling NaNs
// (v_med3_f32 against 0 and +inf is folded back into the same pair).  On the bit patterns it is a signed-integer maximum: a float
// with the sign bit set is a negative integer (-0.0 included, a NaN with the sign bit too; a positive NaN stays what it is), every
// other float is its own non-negative integer.  Plain C rather than inline assembly: the compiler's hazard recogniser does not look
// into an asm statement, and with the accumulators in architectural VGPRs (build.sh) the operand is the matrix instruction's own
// destination -- an opaque "v_max_f32" read it before the passes were through (caught by tests/test_gpu_actor.py on the first
// build with that flag).
__device__ __forceinline__ float relu_f32(float x) {
  const int xi = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, xi > 0 ? xi : 0);
}

// action ~ multinomial(probs) (alg_credit.py:120): inverse CDF in action order, one uniform per (seed, global env id, episode,
// step, agent) from the two-stage stream of philox.h -- stage 1 (actor_block_word: the agent's word of the env's Philox block with
// the policy purpose bit) depends on nothing loaded or computed, so a kernel draws it once, ahead of time (the fused policy rollout:
// once per LAUNCH, for all its ticks); stage 2 (actor_uniform_from) mixes the episode / step counters in, ~10 instructions.
__device__ __forceinline__ uint32_t actor_block_word(uint64_t seed, uint64_t genv, int agent) {
  u32x4 ctr;
  ctr.x = (uint32_t)genv;
  ctr.y = (uint32_t)(genv >> 32);
  ctr.z = 0u;
  ctr.w = kPurposePolicy | ((uint32_t)(agent >> 2) << 24);
  const u32x4 wd = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const int q = agent & 3;
  return q == 0 ? wd.x : (q == 1 ? wd.y : (q == 2 ? wd.z : wd.w));
}
__device__ __forceinline__ float actor_uniform_from(uint32_t block_word, uint32_t episode, int steps) {
  return (float)u01(action_word(block_word, episode, (uint32_t)steps));
}
__device__ __forceinline__ float actor_uniform(uint64_t seed, uint64_t genv, uint32_t episode, int steps, int agent) {
  return actor_uniform_from(actor_block_word(seed, genv, agent), episode, steps);
}

__device__ __forceinline__ int actor_pick(const float (&pr)[kA], float u) {
  // first action whose cumulative probability exceeds u, the last one otherwise.  The running sum never decreases (pr >= 0), so
  // the comparisons that fail form a prefix and the index is their count: no branches (the if-chain compiled to four exec-mask
  // regions on the path of every tick)
  int act = 0;
  float cdf = 0.0f;
#pragma unroll
  for (int a = 0; a < kA - 1; ++a) {
    cdf += pr[a];
    act += (u < cdf) ? 0 : 1;
  }
  return act;
}

__device__ __forceinline__ int actor_sample(const float (&pr)[kA], uint64_t seed, uint64_t genv, uint32_t episode,
                                            int steps, int agent) {
  return actor_pick(pr, actor_uniform(seed, genv, episode, steps, agent));
}

}  // namespace cm3
