// Philox4x32-10 counter-based RNG (Salmon et al., SC'11), the build's own random stream.
// The reference draws from two global MT19937 generators (multi-goal_spread.py:75-91,
// train_onpolicy.py:307); bit-reproducing those on a GPU is neither feasible nor asked for
//  -- parity runs inject states/actions, and this stream only has to
// reproduce the reference's DISTRIBUTIONS.  Keying: key = seed (64 bit), counter =
// (global env id lo, hi, episode, purpose | index) so results are independent of sharding and
// of launch geometry.  oracle/philox.py restates this bit-for-bit for the tests.
#pragma once
#include <stdint.h>

namespace cm3 {

struct u32x4 {
  uint32_t x, y, z, w;
};

__host__ __device__ inline uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

__host__ __device__ inline u32x4 philox4x32_10(u32x4 ctr, uint32_t k0, uint32_t k1) {
  const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    // one 32 x 32 -> 64 multiply per multiplier (v_mad_u64_u32 on the device) instead of a mul_hi / mul_lo pair:
    // the 40 quarter-rate multiplies of a draw were ~half of its 1200 cycles on the critical path of a step launch
    const uint64_t p0 = (uint64_t)M0 * (uint64_t)ctr.x, p1 = (uint64_t)M1 * (uint64_t)ctr.z;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    u32x4 n;
#if defined(__HIP_DEVICE_COMPILE__) && defined(__gfx950__) && __has_builtin(__builtin_amdgcn_bitop3_b32)
    // one three-input xor (v_bitop3_b32, truth table 0x96; gfx950 only) instead of two v_xor: 20 VALU instructions fewer per draw
    n.x = __builtin_amdgcn_bitop3_b32(hi1, ctr.y, k0, 0x96);
    n.z = __builtin_amdgcn_bitop3_b32(hi0, ctr.w, k1, 0x96);
#else
    n.x = hi1 ^ ctr.y ^ k0;
    n.z = hi0 ^ ctr.w ^ k1;
#endif
    n.y = lo1;
    n.w = lo0;
    ctr = n;
    k0 += W0;
    k1 += W1;
  }
  return ctr;
}

// purposes (top bits of counter word 3)
constexpr uint32_t kPurposeAction = 0x00000000u;  // | step (low 24 bits) ... | call<<24
constexpr uint32_t kPurposeReset = 0x80000000u;   // | call index

// uniform in (0,1), 32-bit resolution, never 0 or 1
__host__ __device__ inline double u01(uint32_t r) { return ((double)r + 0.5) * (1.0 / 4294967296.0); }

// uniform integer on {0..4}: multiply-shift of one 32-bit word
__host__ __device__ inline int rand5(uint32_t r) { return (int)mulhi32(r, 5u); }

// word q (0..3) of a block, selected with masks so that all four words stay live: with a q == 0 ? x : ... chain the compiler
// sinks the last Philox round into divergent per-word branches, which puts taken branches (~40 cycles each for a lone wave) on
// the critical path of a step launch
__host__ __device__ inline uint32_t pick_word(const u32x4 &w, int q) {
  const uint32_t m0 = q == 0 ? ~0u : 0u, m1 = q == 1 ? ~0u : 0u, m2 = q == 2 ? ~0u : 0u, m3 = q == 3 ? ~0u : 0u;
  return (w.x & m0) | (w.y & m1) | (w.z & m2) | (w.w & m3);
}

// ---- the action stream: TWO stages (round 4) --------------------------------------------------------------------------------
// The uniform actions of the reference's random-action branch (train_onpolicy.py:305-307) are drawn in-kernel, keyed (seed,
// global env id, episode, step, agent).  Until round 3 that was one Philox4x32-10 block over all of them -- and (episode, step)
// are env state, LOADED by the step launch: the ten rounds (~340 cycles of a 2860-cycle wave at BASELINE sizes) could only start
// once the launch's initial loads were back.  Now:
//   stage 1  action_block(seed, env, call): Philox4x32-10 over (env id, call) keyed by the seed -- four words, one per agent
//            4 call .. 4 call + 3, a per-env constant that depends on NOTHING the launch loads: the kernels compute it while
//            their state loads are in flight;
//   stage 2  action_word(a, episode, step) = fmix32((a ^ step) + episode * 0x9E3779B1): MurmurHash3's 32-bit finaliser (a
//            bijection with full avalanche) over the agent's stage-1 word offset by the episode / step counters -- SplitMix-style
//            counter hashing with a Philox-random 32-bit offset per (env, agent); ~10 instructions behind the loads.
// rand5 of that word is the action.  Same properties as before where they matter: a pure function of (seed, global env id,
// episode, step, agent) -- independent of sharding, launch geometry and launch mode -- restated bit for bit by
// oracle/philox.py, uniform and serially uncorrelated (tests/test_philox.py).  The reset stream keeps the full Philox block
// per draw (reset_words): it is off the per-tick path.
__host__ __device__ inline uint32_t fmix32(uint32_t h) {
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__host__ __device__ inline u32x4 action_block(uint64_t seed, uint64_t env, uint32_t call) {
  u32x4 c;
  c.x = (uint32_t)env;
  c.y = (uint32_t)(env >> 32);
  c.z = 0u;
  c.w = kPurposeAction | (call << 24);
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}
__host__ __device__ inline uint32_t action_word(uint32_t a, uint32_t episode, uint32_t step) {
  return fmix32((a ^ step) + episode * 0x9E3779B1u);
}
// (Until round 4 the Checkers step kernel kept a ONE-stage draw -- a full Philox block over (env, episode, step | call) -- because
// computing stage 1 in its instruction-bound prologue measured slower, profiles/r04_two_stage_action_stream.txt.  Round 5: stage 1 is
// a per-env constant, so the env object computes it ONCE (cm3_checkers_action_blocks) and a step launch LOADS it with its state;
// one definition of the stream for every kernel.)
// actions of agents 4c..4c+3 of (env, episode, step) come from call c
__host__ __device__ inline u32x4 action_words(uint64_t seed, uint64_t env, uint32_t episode, uint32_t step,
                                              uint32_t call) {
  u32x4 w = action_block(seed, env, call);
  w.x = action_word(w.x, episode, step);
  w.y = action_word(w.y, episode, step);
  w.z = action_word(w.z, episode, step);
  w.w = action_word(w.w, episode, step);
  return w;
}

__host__ __device__ inline u32x4 reset_words(uint64_t seed, uint64_t env, uint32_t episode, uint32_t call) {
  u32x4 c;
  c.x = (uint32_t)env;
  c.y = (uint32_t)(env >> 32);
  c.z = episode;
  c.w = kPurposeReset | call;
  return philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
}

}  // namespace cm3
