// Pieces shared by the particle actor (actor.hip, policy.hip) and the Checkers actor (actor_checkers.hip).
#pragma once
#include "common.h"
#include "philox.h"

namespace cm3 {

constexpr int kA = 5;  // l_action
constexpr uint32_t kPurposePolicy = 0x40000000u;

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// action ~ multinomial(probs) (alg_credit.py:120): inverse CDF in action order, one uniform from the Philox stream
// keyed (seed, global env id, episode, step | agent)
// the uniform of (seed, global env id, episode, step | agent): depends on nothing the network computes, so a kernel can draw it
// while its loads are in flight
__device__ __forceinline__ float actor_uniform(uint64_t seed, uint64_t genv, uint32_t episode, int steps, int agent) {
  u32x4 ctr;
  ctr.x = (uint32_t)genv;
  ctr.y = (uint32_t)(genv >> 32);
  ctr.z = episode;
  ctr.w = kPurposePolicy | ((uint32_t)(agent >> 2) << 24) | ((uint32_t)steps & 0x00FFFFFFu);
  const u32x4 wd = philox4x32_10(ctr, (uint32_t)seed, (uint32_t)(seed >> 32));
  const int q = agent & 3;
  return (float)u01(q == 0 ? wd.x : (q == 1 ? wd.y : (q == 2 ? wd.z : wd.w)));
}

__device__ __forceinline__ int actor_pick(const float (&pr)[kA], float u) {
  int act = kA - 1;
  float cdf = 0.0f;
  bool chosen = false;
#pragma unroll
  for (int a = 0; a < kA - 1; ++a) {
    cdf += pr[a];
    if (!chosen && u < cdf) {
      act = a;
      chosen = true;
    }
  }
  return act;
}

__device__ __forceinline__ int actor_sample(const float (&pr)[kA], uint64_t seed, uint64_t genv, uint32_t episode,
                                            int steps, int agent) {
  return actor_pick(pr, actor_uniform(seed, genv, episode, steps, agent));
}

}  // namespace cm3
