// A whole POLICY-DRIVEN episode in one launch: actor forward pass (matrix cores) + epsilon-mixed sampling + physics
// step + observation, tick after tick, with the network weights, the input tile and the env state resident in
// LDS / registers.
//
// Reference loop being replaced (alg/train_onpolicy.py:302-350, the branch taken for 49 950 of its 50 000 episodes):
//     actions = alg.run_actor(local_others, local_self, goals, epsilon, sess)      :313   (alg_credit.py:249-270)
//     next_... = env.step(actions)                                                 :323   (environment.py:81-123)
//     buf.add(transition) ; roll state forward                                     :338-347
// The unfused device path (ParticleRollout.collect(policy=actor)) alternates k_actor_particle and
// k_particle_step_pairs launches inside a hipGraph: 12.9 us per tick at 4096 envs x 4 agents, of which ~3.8k cycles
// re-stage the weights, ~3 us are launch boundaries and ~2 us re-load state/observations.  Here a workgroup owns
// 64 agent rows = 64/N whole envs for all T ticks:
//   * weights: first-layer tables + output layer in LDS, W2 B-operands in VGPRs -- staged ONCE per launch;
//   * per tick: actor_mlp (phase A / phase B on the matrix cores, 2 or 3 barriers) -> head: probabilities + action of row
//     16w + (l&15) in the 16 "part 0" lanes of wave w -> the same lanes advance their agent: own-side contact forces
//     against the other agents of the env (positions read from the LDS input tile), integration, reward / collision /
//     reached, per-env reductions with wave shuffles, optional same-tick re-initialisation, and the next observation is
//     written both to the trajectory (global) and back into the LDS tile that feeds the next tick's phase A.
// Arithmetic per agent is the reference's operation order, so trajectories are bit-identical to the unfused path
// (tests/test_gpu_actor.py::test_fused_policy_rollout_equals_launch_per_tick).
#define CM3_NO_ENTRY_POINTS 1
#define CM3_PARTICLE_F32 1
#include "particle.hip"
#include "actor.hip"
#include <stdlib.h>
#include <atomic>

namespace cm3 {

struct PolicyParams {
  ParticleParams p;  // trajectory pointers/strides of the fused rollout (state_in = slot 0, *_out = slot 1 ...)
  const float *obs_in;  // obs_others slot 0
  const float *packed;
  float *probs;         // optional [T][E][N][5]
  size_t st_probs;
  float eps;
  int stage;
};

// RT = 16-row tiles per workgroup: 4 (64 agent rows) where that already gives every CU several workgroups; 2 or 1 for smaller
// batches, so that two or more workgroups share a CU and one's matrix-core phases run under the other's physics / LDS staging
// (round 4: the per-env chain actor -> physics -> observation is serial, and at one wave per SIMD the tick was latency-bound:
// ~1 us of matrix time inside 5.0 us).  Waves w >= RT take part in the layers (every wave owns 16 columns of all row tiles) and
// in the barriers, but have no rows of their own in the head and the physics.
template <int N, int BF16, int RT> __global__ void CM3_MATRIX_KERNEL k_policy_rollout(const PolicyParams q) {
  using G = ActorGeom<N, BF16, RT>;
  using V4 = float4;
  using V2 = float2;
  constexpr int L = G::L, NO = N > 1 ? N - 1 : 1;
  const ParticleParams &p = q.p;
  CM3_ACTOR_LDS_RT(N, BF16, RT, lds);
  __shared__ __attribute__((aligned(16))) float4 ns[16 * RT];  // post-step (vx, vy, px, py) of every row, exchanged inside a wave
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t E = (size_t)p.E;
  const size_t rows = E * N;
  const size_t row_base = (size_t)blockIdx.x * (16 * RT);
  CM3_STAMP(13, false);   // kernel entry (probes only)

  // the row this lane owns in the head and in the physics: rl = 16w + (l&15); only part 0 (l < 16) of the waves w < RT writes
  // Which waves own rows: w < RT -- and, for two-tile workgroups, waves 2, 3 in every other group of 256 workgroups.  Wave w of a
  // workgroup runs on SIMD w of its CU, and two such workgroups share a CU once a batch has more than 256 of them (workgroup b and
  // b + 256, as the dispatcher fills the chip): the head and the physics of both would otherwise pile up on SIMDs 0 and 1 while
  // SIMDs 2 and 3 idle for 60 % of every tick (stamped build, profiles/r04_policy_row_tiles.txt).  Speed only: results do not
  // depend on the placement.
  const int wr = RT == 2 ? ((w + 2 * (int)((blockIdx.x >> 8) & 1u)) & 3) : w;
  const bool row_wave = wr < RT;
  const int rl = row_wave ? 16 * wr + (lane & 15) : (lane & 15);
  const bool part0 = (lane >> 4) == 0 && row_wave;
  size_t r = row_base + rl;
  const bool row_ok = r < rows;
  r = row_ok ? r : rows - 1;
  const size_t e = r / N;
  const int i = (int)(r - e * N);
  const int env_lane0 = (lane & 15) - i;  // lane (within the wave's part 0) of agent 0 of this env
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)e);
  const bool writer = row_ok && part0;
  const bool head_lane = writer && i == 0;

  // ---- once per launch: weights, live counters, the input tile ---------------------------------------------------------
  // every global request of the entry first (tables, the wave's W2 slice, counters, the rows' state / goals / observation), the
  // sampling key's Philox block while they are in flight, then the LDS stores: one memory round trip before the barrier
  // (the W2 slice first: it keeps its registers for the whole launch, and requested after the table copy the register allocator
  // parked the copy in those very registers -- a wait for the tables, a move, and only then the slice's requests)
  ActorB<N, BF16> b;
  actor_load_b<N, BF16>(q.packed, w, lane, b);
  ActorTabRegs<N> tab;
  actor_tables_fetch<N>(q.packed, tid, tab);
  const int2 meta = reinterpret_cast<const int2 *>(p.meta_in)[e];
  const bool auto_reset = (p.flags & CM3_FLAG_AUTO_RESET) != 0;
  uint32_t episode = (uint32_t)p.episode[e];
  V4 s0, o0[L / 4];   // (every lane requests its row -- lanes outside part 0 repeat one of part 0's -- rather than branch)
  V2 g0;
  {
    s0 = reinterpret_cast<const V4 *>(p.state_in)[(size_t)i * E + e];
    g0 = reinterpret_cast<const V2 *>(p.goals_in)[(size_t)i * E + e];
    const V4 *o4 = reinterpret_cast<const V4 *>(q.obs_in + r * L);
#pragma unroll
    for (int k = 0; k < L / 4; ++k) o0[k] = o4[k];
  }
  const uint32_t ublock = actor_block_word(p.seed, genv, i);   // stage 1 of the sampling uniforms: once per launch (actor_common.h)
  int steps = meta.x, collisions = meta.y;
  const uint32_t episode_in = episode;
  actor_tables_store<N, BF16, RT>(lds, tid, tab);
  if (part0) {
    lds.xs[rl][0] = s0.x; lds.xs[rl][1] = s0.y; lds.xs[rl][2] = s0.z; lds.xs[rl][3] = s0.w;
    lds.xs[rl][4] = g0.x; lds.xs[rl][5] = g0.y;
#pragma unroll
    for (int k = 0; k < L / 4; ++k) {
      lds.xs[rl][6 + 4 * k + 0] = o0[k].x; lds.xs[rl][6 + 4 * k + 1] = o0[k].y;
      lds.xs[rl][6 + 4 * k + 2] = o0[k].z; lds.xs[rl][6 + 4 * k + 3] = o0[k].w;
    }
  }
  __syncthreads();
  CM3_STAMP(14, true);    // tables, weights and the input tile are in
  ActorHeadB hb;
  actor_head_load(lds.wout, lane, hb);      // output-layer operands of this lane, once per launch
  ActorFirstB<N> f1;                        // ... and its first-layer operands (round 4: they were re-read from LDS every tick)
  actor_first_b<N, float>(&lds.ws_self[0][0], &lds.ws_oth[0][0], w, lane, q.stage > 1, f1);

  const float kDt = 0.1f, kKeep = 1.0f - 0.25f;
#pragma unroll 1
  for (int t = 0; t < p.n_ticks; ++t) {
    // ---- policy: forward pass, probabilities and the sampled action of row rl (alg_credit.py:113-122) -----------------
    CM3_STAMP(0, false);
    actor_mlp<N, BF16, RT>(lds, b, f1, w, lane, q.stage > 1);
#ifdef CM3_PROBE_P_NO_ROWS   // (probe builds only: the matrix phases and the barriers alone)
    if (false) {
#else
    if (row_wave) {   // (wave-uniform; no workgroup barrier inside)
#endif
    // The head, the physics and the stores of the rows are the part of the tick only this wave can do while its workgroup waits at
    // the closing barrier: it goes first on its SIMD, ahead of the matrix phases of the workgroup that shares the CU (measured,
    // same box, three alternating rounds: 4.31 -> 4.15 us per tick at 4 096 x 4, 45.1 -> 44.6 at 65 536 x 4; levels 1, 2, 3 are
    // the same, raising the priority for the WHOLE tick loses the gain; profiles/r04_policy_head.txt)
    __builtin_amdgcn_s_setprio(2);
    float pr[kA];
    const float u = actor_uniform_from(ublock, episode, steps);
#ifndef CM3_PROBE_P_NO_HEAD   // (probe builds only, tools/r6/policy_whatif.sh)
    actor_head_probs(lds.h2s, hb, wr, lane, q.eps, pr);
#else
    for (int a = 0; a < kA; ++a) pr[a] = 0.2f;
#endif
    // (the head leaves the row's probabilities in lanes 0..15 only; lanes 16..63 repeat the physics of lane l & 15 and get its
    // action so that they take the same branches -- nothing they compute is stored: see the exchange slot below)
    const int act = bcast_row0(actor_pick(pr, u));
    CM3_STAMP(7, false);

    // ---- physics of agent i (environment.py:81-123), lanes of part 0 ------------------------------------------------------
    V4 si;
    si.x = lds.xs[rl][0]; si.y = lds.xs[rl][1]; si.z = lds.xs[rl][2]; si.w = lds.xs[rl][3];
    V2 gl;
    gl.x = lds.xs[rl][4]; gl.y = lds.xs[rl][5];
    float ux = 0.0f, uy = 0.0f;
    if (act == 1) ux = -1.0f;
    if (act == 2) ux = +1.0f;
    if (act == 3) uy = -1.0f;
    if (act == 4) uy = +1.0f;
    float Fx = ux * 5.0f + 0.0f, Fy = uy * 5.0f + 0.0f;
#ifndef CM3_PROBE_P_NO_PHYS
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {  // the reference's accumulation order: other agents ascending (core.py:145-155)
      const int j = k < i ? k : k + 1;
      const int rj = rl - i + j;
      float f_x, f_y;
      contact_force<float>(si.z - lds.xs[rj][2], si.w - lds.xs[rj][3], f_x, f_y);
      Fx = f_x + Fx;
      Fy = f_y + Fy;
    }
#endif
    si.x = si.x * kKeep;
    si.y = si.y * kKeep;
    si.x = si.x + (Fx / 1.0f) * kDt;
    si.y = si.y + (Fy / 1.0f) * kDt;
    si.z = si.z + si.x * kDt;
    si.w = si.w + si.y * kDt;
    steps += 1;
    // (one 16-byte LDS store; the other agents' rows come back as 16-byte reads, once, for collisions AND observation.  Lanes 0..15
    // only: lanes 16..63 repeat their physics and used to store their copy into the same slot -- the same value by the code,
    // but this way nothing a copy computes is ever stored.  Round 5 found what made copies differ: a packed multiply of the contact
    // chain returning a wrong low half in lanes 48..63 under a co-resident matrix wave, profiles/r05_policy_fault.txt)
    if (part0) ns[rl] = si;
    wave_lds_sync();  // the other agents of this env live in the same wave
    V4 oth[NO];
#pragma unroll
    for (int k = 0; k < NO; ++k) {
      const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
      oth[k] = ns[rl - i + j];
    }

    CM3_STAMP(8, false);
    // ---- reward / reached / collisions (multi-goal_spread.py:114-143) ----------------------------------------------------------
    float rew;
    bool reached;
    {
      const float dx = si.z - gl.x, dy = si.w - gl.y;
      const float d2 = dx * dx + dy * dy;
      rew = 0.0f - sqrtf(d2);
      reached = d2 < Thresh<float>::kReach2;  // rew >= -0.05f (thresholds.h)
    }
    int hits = 0;
#pragma unroll
    for (int k = 0; k < N - 1; ++k) {
      if (is_collision<float>(oth[k].z - si.z, oth[k].w - si.w)) {  // is_collision(a = j, agent = i)
        rew = rew - 1.0f;
        hits += 1;
      }
    }
    float rews[N];
    int hit_sum = 0;
    bool all_reached = true;
    if constexpr (N == 4) {  // the env's four agents are one DPP quad: broadcasts and sums without the LDS crossbar
      rews[0] = dpp_f32<0x00>(rew);
      rews[1] = dpp_f32<0x55>(rew);
      rews[2] = dpp_f32<0xAA>(rew);
      rews[3] = dpp_f32<0xFF>(rew);
      hit_sum = hits + dpp_i32<kDppXor1>(hits);
      hit_sum += dpp_i32<kDppXor2>(hit_sum);
      int r4 = (int)reached;
      r4 &= dpp_i32<kDppXor1>(r4);
      r4 &= dpp_i32<kDppXor2>(r4);
      all_reached = r4 != 0;
    } else if constexpr (N == 8) {  // an env is half a DPP row: sums over its 8 lanes by xor 1, xor 2, half mirror (lane i <-> 7 - i)
      hit_sum = hits + dpp_i32<kDppXor1>(hits);
      hit_sum += dpp_i32<kDppXor2>(hit_sum);
      hit_sum += dpp_i32<kDppHalfMirror>(hit_sum);
      int r8 = (int)reached;
      r8 &= dpp_i32<kDppXor1>(r8);
      r8 &= dpp_i32<kDppXor2>(r8);
      r8 &= dpp_i32<kDppHalfMirror>(r8);
      all_reached = r8 != 0;
    } else {
#pragma unroll
      for (int a = 0; a < N; ++a) {  // per-env reductions over the N part-0 lanes of this env
        rews[a] = __shfl(rew, env_lane0 + a, 64);
        hit_sum += __shfl(hits, env_lane0 + a, 64);
        all_reached = all_reached && (__shfl((int)reached, env_lane0 + a, 64) != 0);
      }
    }
    collisions += hit_sum;
    float reward;
    if constexpr (N == 8) {
      // NumPy's pairwise tree for 8 values, ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)) (sum_agents), by the same three DPP steps: every
      // partial sum meets its partner with the operands possibly swapped, and float addition is commutative bit for bit
      float tsum = rew + dpp_f32<kDppXor1>(rew);
      tsum = tsum + dpp_f32<kDppXor2>(tsum);
      reward = tsum + dpp_f32<kDppHalfMirror>(tsum);
    } else {
      reward = sum_agents<float, N>(rews);
    }
    const bool done = (steps == p.max_steps) || all_reached;
    if (head_lane) {
      reinterpret_cast<float *>(tick_ptr(p.reward, p.st_reward, t))[e] = reward;
      tick_ptr(p.done, p.st_done, t)[e] = done ? 1 : 0;
      if (p.collisions_tick) tick_ptr(p.collisions_tick, p.st_coll, t)[e] = collisions;
    }

    CM3_STAMP(9, false);
    // ---- same-tick re-initialisation of finished episodes (CM3_FLAG_AUTO_RESET) ------------------------------------------------
    bool was_reset = false;
#ifdef CM3_PROBE_P_NO_RESET   // (probe builds only)
    if (false) {
#else
    if (auto_reset && done) {
#endif
      void *term_state = tick_ptr(p.term_state, p.st_term_state, t);
      void *term_obs = tick_ptr(p.term_obs_others, p.st_term_obs, t);
      if (writer && term_state) reinterpret_cast<V4 *>(term_state)[(size_t)i * E + e] = si;
      if (writer && term_obs) {
        V4 *o = reinterpret_cast<V4 *>(term_obs) + r * NO;
#pragma unroll
        for (int k = 0; k < NO; ++k) {
          const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
          (void)j;
          o[k] = sub4<float, V4>(oth[k], si);
        }
      }
      wave_lds_sync();  // every lane of the env has read the terminal states before they are overwritten
      episode += 1;
      const bool rnd = episode_is_random(p, genv, episode);
      static_assert(offsetof(PolicyParams, p) == 0, "preset_table(0): the ParticleParams lead the kernel's only argument");
      init_agent<float, N, true>(p, genv, episode, rnd, i, si, gl, preset_table(0));
      steps = 0;
      collisions = 0;
      was_reset = true;
      if (part0) ns[rl] = si;
      wave_lds_sync();
    }
    if (auto_reset && __any(done)) {  // fresh episodes in this wave: the observation is that of the reset states
#pragma unroll
      for (int k = 0; k < NO; ++k) {
        const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
        oth[k] = ns[rl - i + j];
      }
    }

    CM3_STAMP(10, false);
    // ---- trajectory stores + the LDS tile of the next tick ---------------------------------------------------------------------
    if (part0) {
#ifndef CM3_PROBE_P_NO_STORES
      if (row_ok) {   // every per-row store of the tick in ONE exec region (the action, its probabilities and the reward used to have
                      // regions of their own further up: ~6 scalar instructions and a branch each on the row waves' path)
        tick_ptr(p.actions, p.st_actions, t)[r] = act;
        if (q.probs) {
          float *pt = tick_ptr(q.probs, q.st_probs, t) + r * kA;
#pragma unroll
          for (int a = 0; a < kA; ++a) pt[a] = pr[a];
        }
        reinterpret_cast<float *>(tick_ptr(p.reward_n, p.st_reward_n, t))[r] = rew;
        reinterpret_cast<V4 *>(tick_ptr(p.state_out, p.st_state, t))[(size_t)i * E + e] = si;
        if (p.goals_out != p.goals_in || was_reset)
          reinterpret_cast<V2 *>(tick_ptr(p.goals_out, p.st_goals, t))[(size_t)i * E + e] = gl;
      }
#endif
      V4 *o = reinterpret_cast<V4 *>(tick_ptr(p.obs_others, p.st_obs, t)) + r * NO;
#pragma unroll
      for (int k = 0; k < NO; ++k) {  // observation (multi-goal_spread.py:145-154)
        const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
        (void)j;
        const V4 d = sub4<float, V4>(oth[k], si);
#ifndef CM3_PROBE_P_NO_STORES
        if (row_ok) o[k] = d;
#endif
        lds.xs[rl][6 + 4 * k + 0] = d.x; lds.xs[rl][6 + 4 * k + 1] = d.y;
        lds.xs[rl][6 + 4 * k + 2] = d.z; lds.xs[rl][6 + 4 * k + 3] = d.w;
      }
      lds.xs[rl][0] = si.x; lds.xs[rl][1] = si.y; lds.xs[rl][2] = si.z; lds.xs[rl][3] = si.w;
      lds.xs[rl][4] = gl.x; lds.xs[rl][5] = gl.y;
    }
    CM3_STAMP(11, false);
    __builtin_amdgcn_s_setprio(0);
    }  // row_wave
    __syncthreads();  // the tile (and the h1/h2 storage) is free for the next tick's phase A
    CM3_STAMP(12, false);
  }

  if (head_lane) {
    int2 m;
    m.x = steps;
    m.y = collisions;
    reinterpret_cast<int2 *>(p.meta_out)[e] = m;
    if (episode != episode_in) p.episode[e] = (int32_t)episode;
  }
  CM3_STAMP(15, true);    // last store acknowledged
}

template <int N, int RT> static int policy_launch_rt(const PolicyParams &q, int prec, hipStream_t s) {
  const size_t rows = (size_t)q.p.E * N;
  const unsigned blocks = (unsigned)((rows + 16 * RT - 1) / (16 * RT));
  note_variant("k_policy_rollout", 4, N, 4, q.p.n_ticks > 1, prec, 0, 0, 0, RT);
  if (prec == kPrecF16x3)
    hipLaunchKernelGGL((k_policy_rollout<N, kPrecF16x3, RT>), dim3(blocks), dim3(256), 0, s, q);
  else if (prec == kPrecBf16)
    hipLaunchKernelGGL((k_policy_rollout<N, kPrecBf16, RT>), dim3(blocks), dim3(256), 0, s, q);
  else
    hipLaunchKernelGGL((k_policy_rollout<N, kPrecF32, RT>), dim3(blocks), dim3(256), 0, s, q);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

// Row tiles per workgroup from the batch (measured on MI355X with the final kernel, f16x3, us per tick at RT = 4 / 2 / 1 --
// profiles/r04_policy_row_tiles.txt, re-swept after the accumulators moved to architectural VGPRs):  N = 4: 2048 envs 4.17 / 3.05 /
// 2.86, 4096 envs (C2) 4.32 / 3.97 / 5.12, 8192 envs 6.00 / 7.21 / 9.68, 65536 envs 43.3 / 54.0 / 73.8;  N = 8: 512 envs 5.38 /
// 3.94 / 3.23, 1024 envs 5.43 / 4.06 / 3.84, 2048 envs 5.63 / 5.37 / 6.99, 4096 envs 8.09 / 9.91 / 13.4, 8192 envs 15.2 / 19.4 /
// 26.4;  N = 2 at 8192 envs 3.85 / 3.62 / 4.72;  N = 1 at 16384 envs 3.58 / 3.36 / 4.43.  A lone workgroup per CU is latency-bound
// and the smaller one is faster (fewer matrix instructions per wave and tick); once a CU holds several, the 64-row workgroup does
// the most work per instruction issued.  So: 64-row workgroups when they already give every CU more than one, 32-row ones down to
// half a workgroup per CU, 16-row ones below -- for every N (N = 8 had a rule of its own while its 32-row build needed 261
// registers, one wave per SIMD; every build is under 256 now).  cm3_policy_force_row_tiles(1 | 2 | 4) overrides the choice (tests and
// measurements; its initial value is the environment's CM3_POLICY_RT, read ONCE -- not per launch).
constexpr size_t kPolicyCus = 256;
static std::atomic<int> g_policy_rt_override{-1};   // -1: not initialised, 0: the rule decides
static int policy_rt_override() {
  int v = g_policy_rt_override.load(std::memory_order_relaxed);
  if (v < 0) {
    const char *e = getenv("CM3_POLICY_RT");
    v = e ? atoi(e) : 0;
    if (v != 1 && v != 2 && v != 4) v = 0;
    int expected = -1;
    g_policy_rt_override.compare_exchange_strong(expected, v);
    v = g_policy_rt_override.load(std::memory_order_relaxed);
  }
  return v;
}
template <int N> static int policy_launch(const PolicyParams &q, int prec, hipStream_t s) {
  const size_t rows = (size_t)q.p.E * N, wg64 = (rows + 63) / 64;
  int rt = wg64 > kPolicyCus ? 4 : (wg64 > kPolicyCus / 2 ? 2 : 1);
  // One or two ticks per launch (the launch-per-tick mode): the entry -- 58 KB of tables and weight slice per workgroup through the
  // CU's L1 path -- is as long as the tick, so the larger workgroup wins one step earlier (N = 4, one tick per launch, RT = 4 / 2 / 1:
  // 1024 envs 8.34 / 7.50 / 6.75, 2048 envs 8.46 / 7.43 / 8.47, 4096 envs 8.72 / 9.71 / 14.2, 8192 envs 11.8 / 16.6 / 24.5 us;
  // tools/probes/policy_tick_tiles.sh, profiles/r04_policy_head.txt)
  if (q.p.n_ticks <= 2) rt = wg64 >= kPolicyCus ? 4 : (wg64 >= kPolicyCus / 2 ? 2 : 1);
  if (const int v = policy_rt_override()) rt = v;
  switch (rt) {
    case 1: return policy_launch_rt<N, 1>(q, prec, s);
    case 2: return policy_launch_rt<N, 2>(q, prec, s);
  }
  return policy_launch_rt<N, 4>(q, prec, s);
}

}  // namespace cm3

extern "C" int cm3_policy_force_row_tiles(int32_t row_tiles) {
  using namespace cm3;
  CM3_REQUIRE(row_tiles == 0 || row_tiles == 1 || row_tiles == 2 || row_tiles == 4, "row_tiles must be 0 (rule), 1, 2 or 4; got %d", row_tiles);
  g_policy_rt_override.store(row_tiles, std::memory_order_relaxed);
  return CM3_OK;
}

extern "C" int cm3_policy_rollout_f32(const cm3_particle_desc *d, const cm3_particle_traj *t,
                                      const cm3_actor_particle_desc *ad, const cm3_actor_particle_weights *wt,
                                      float *probs, size_t probs_stride, int32_t n_ticks, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(d && t && ad && wt, "null argument");
  CM3_REQUIRE(n_ticks >= 1, "n_ticks must be >= 1");
  CM3_REQUIRE(d->n_agents == 1 || d->n_agents == 2 || d->n_agents == 4 || d->n_agents == 8,
              "the fused policy rollout needs n_agents in {1, 2, 4, 8} (whole envs per 16-row wave tile); got %d",
              d->n_agents);
  CM3_REQUIRE(ad->n_agents == d->n_agents && ad->n_envs == d->n_envs, "actor / env descriptors disagree");
  CM3_REQUIRE(!(d->flags & CM3_FLAG_GEN_ACTIONS), "the policy draws the actions: CM3_FLAG_GEN_ACTIONS is meaningless here");
  int rc = actor_check_desc(ad);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(ad->seed == d->seed && ad->env_id_base == d->env_id_base,
              "actor and env must share seed and env_id_base (one Philox key; the streams differ by purpose bits)");
  CM3_REQUIRE(ad->epsilon >= 0.0f && ad->epsilon <= 1.0f, "epsilon must be in [0,1]");
  CM3_REQUIRE(wt->packed, "weights->packed is NULL: run cm3_actor_particle_pack once per weight update");
  CM3_REQUIRE(t->state && t->goals && t->obs_others && t->actions && t->reward_n && t->reward && t->done && t->meta &&
                  t->episode,
              "trajectory base pointers are required");
  auto at = [](void *base, size_t stride, int k) -> void * {
    return base ? (void *)((char *)base + stride * (size_t)k) : nullptr;
  };
  cm3_particle_bufs b;
  memset(&b, 0, sizeof(b));
  b.state_in = t->state;
  b.state_out = at(t->state, t->state_stride, 1);
  b.goals_in = t->goals;
  b.goals_out = at(t->goals, t->goals_stride, 1);
  b.meta_in = b.meta_out = t->meta;
  b.episode = t->episode;
  b.actions = t->actions;
  b.obs_others = at(t->obs_others, t->obs_others_stride, 1);
  b.reward_n = t->reward_n;
  b.reward = t->reward;
  b.done = t->done;
  b.term_state = t->term_state;
  b.term_obs_others = t->term_obs_others;
  b.collisions_tick = t->collisions;
  CM3_REQUIRE(d->env_offset == 0 && d->env_count == 0, "the fused policy rollout covers the whole batch");
  cm3_particle_desc dd = *d;
  dd.flags &= CM3_FLAG_AUTO_RESET;
  PolicyParams q;
  memset(&q, 0, sizeof(q));
  rc = fill_params(&dd, &b, kStep, nullptr, q.p);
  if (rc != CM3_OK) return rc;
  q.p.n_ticks = n_ticks;
  q.p.st_state = t->state_stride;
  q.p.st_goals = t->goals_stride;
  q.p.st_obs = t->obs_others_stride;
  q.p.st_actions = t->actions_stride;
  q.p.st_reward_n = t->reward_n_stride;
  q.p.st_reward = t->reward_stride;
  q.p.st_done = t->done_stride;
  q.p.st_term_state = t->term_state_stride;
  q.p.st_term_obs = t->term_obs_others_stride;
  q.p.st_coll = t->collisions_stride;
  q.obs_in = (const float *)t->obs_others;
  q.packed = (const float *)wt->packed;
  q.probs = probs;
  q.st_probs = probs_stride;
  q.eps = ad->epsilon;
  q.stage = ad->stage;
  hipStream_t s = (hipStream_t)stream;
  CM3_REQUIRE(ad->precision >= 0 && ad->precision <= 2, "precision must be 0, 1 or 2");
  const int bf16 = ad->precision;
  switch (d->n_agents) {
    case 1: return policy_launch<1>(q, bf16, s);
    case 2: return policy_launch<2>(q, bf16, s);
    case 4: return policy_launch<4>(q, bf16, s);
    case 8: return policy_launch<8>(q, bf16, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", d->n_agents);
}
