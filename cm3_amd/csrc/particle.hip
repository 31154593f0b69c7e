// Particle env (MultiAgentEnv + multi-goal_spread) for E environments per launch, gfx950.
//
// Reference being replaced (paths relative to /root/reference/env/multiagent-particle-envs/multiagent):
//   environment.py:81-123  step            environment.py:125-149 reset
//   environment.py:177-225 _set_action     core.py:117-196 World.step and its pieces
//   scenarios/multi-goal_spread.py:65-93 reset_world, :114-154 is_collision/reward/done/observation
//
// Mapping: ONE LANE PER ENVIRONMENT.  A lane keeps all N agents of its env in registers
// (4 reals each), loops the C(N,2) pairs locally and never talks to another lane for the
// dynamics, so every global load is one 16-byte vector per lane at unit stride over the env index
// (state is [N][E][4]: 1 KiB per wave-instruction, fully coalesced).  The one env-major (AoS)
// output that is wider than a vector per lane -- obs_others [E][N][L] -- is transposed through a
// wave-private LDS tile so that the wave writes whole contiguous rows (16 B per lane, unit stride)
// instead of 64 partial cache lines per store.  No MFMA: there is no contraction on this path.
//
// Arithmetic follows the reference's operation order exactly (compile with -ffp-contract=off):
// the double instantiation differs from NumPy only in the last ulps of exp/log1p.
#include "common.h"
#include "philox.h"
#include "thresholds.h"

// This file is compiled a THIRD time for float32 (build.sh: -DCM3_PARTICLE_ILP_TU with -mllvm -amdgpu-sched-strategy=max-ilp): only the
// two shared-env step kernels and their launchers, tagged with TU = 1 so that they are distinct symbols.  The max-ILP scheduling
// strategy shortens the executed path of these kernels where a SIMD holds one or two waves (C2 2.61 -> 2.55 us per tick, fused
// rollout +8 %) and costs occupancy where it holds many (N = 8 at 2^20 envs: 5.4 -> 5.0 TB/s), so the default build of a kernel is
// kept for launches of more than kIlpMaxWaves waves (profiles/r02_sched_strategy_max_ilp.txt).
#ifdef CM3_PARTICLE_ILP_TU
#define CM3_PARTICLE_TU 1
#else
#define CM3_PARTICLE_TU 0
#endif

namespace cm3 {

struct ParticleParams;
// launchers of the max-ILP translation unit (float32 only): waves = 1 or CM3_PAIR_WAVES / CM3_AGENT_WAVES per workgroup; n_agents 2..8
int particle_ilp_launch_pairs_f32(const ParticleParams &p, int n_agents, int waves_per_wg, hipStream_t stream);
int particle_ilp_launch_agents_f32(const ParticleParams &p, int n_agents, int waves_per_wg, hipStream_t stream);
[[maybe_unused]] constexpr size_t kIlpMaxWaves = 16384;

struct ParticleParams {
  int E;          // extent of the env axis of every array (row stride of the [N][E][..] arrays)
  int max_steps;
  uint32_t flags;
  int _pad;
  int E0, EN;     // this launch covers envs [E0, EN) of the arrays (desc->env_offset / env_count: a sub-batch of the arrays)
  int64_t env_id_base;
  uint64_t seed;
  double prob_random, initial_std;
  double ax[CM3_MAX_AGENTS], ay[CM3_MAX_AGENTS], lx[CM3_MAX_AGENTS], ly[CM3_MAX_AGENTS];
  const void *state_in;
  void *state_out;
  const void *goals_in;
  void *goals_out;
  const int32_t *meta_in;
  int32_t *meta_out;
  int32_t *episode;
  int32_t *actions;
  void *obs_others;
  void *reward_n;
  void *reward;
  uint8_t *done;
  void *term_state;
  void *term_obs_others;
  int32_t *collisions_tick;  // optional int32 [E] per tick: scenario.collisions after this tick (before any same-launch reset)
  void *state_copy, *goals_copy;  // optional: the post-step state / goals ALSO go here (cm3_particle_traj.state_live: the slot copy)
  const uint8_t *reset_mask;
  // tick loop inside one launch (CM3_FLAG_FUSED_TICKS): tick t uses <pointer> + t * <stride in bytes> for the
  // per-tick arrays (state_out / goals_out / obs_others / term_* point at slot 1 of their trajectories);
  // n_ticks == 1 with zero strides is the plain one-launch-per-tick step.
  int n_ticks;
  int _pad2;
  size_t st_state, st_goals, st_obs, st_actions, st_reward_n, st_reward, st_done, st_term_state, st_term_obs, st_coll;
  CM3_SPAN_FIELD  // (span build only, common.h: where this launch's per-wave time stamps go)
};

template <typename T> __device__ __forceinline__ T *tick_ptr(T *base, size_t stride, int t) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + stride * (size_t)t);
}
template <> __device__ __forceinline__ void *tick_ptr<void>(void *base, size_t stride, int t) {
  return base ? (void *)(reinterpret_cast<char *>(base) + stride * (size_t)t) : nullptr;
}

// ---- scalar math per real ----------------------------------------------------------------------
template <typename R> struct Math;
template <> struct Math<float> {
  static __device__ __forceinline__ float sqrt(float x) { return sqrtf(x); }
  static __device__ __forceinline__ float exp(float x) { return expf(x); }
  static __device__ __forceinline__ float log1p(float x) { return log1pf(x); }
  static __device__ __forceinline__ float abs(float x) { return fabsf(x); }
  static __device__ __forceinline__ float log(float x) { return logf(x); }
  static __device__ __forceinline__ float cos(float x) { return cosf(x); }
  static __device__ __forceinline__ float sin(float x) { return sinf(x); }
};
template <> struct Math<double> {
  static __device__ __forceinline__ double sqrt(double x) { return ::sqrt(x); }
  static __device__ __forceinline__ double exp(double x) { return ::exp(x); }
  static __device__ __forceinline__ double log1p(double x) { return ::log1p(x); }
  static __device__ __forceinline__ double abs(double x) { return fabs(x); }
  static __device__ __forceinline__ double log(double x) { return ::log(x); }
  static __device__ __forceinline__ double cos(double x) { return ::cos(x); }
  static __device__ __forceinline__ double sin(double x) { return ::sin(x); }
};

// np.logaddexp(0, x) (core.py:192): NumPy's npy_logaddexp specialised to a zero first argument --
// x == 0 -> ln 2; x < 0 -> 0 + log1p(exp(x)); x > 0 -> x + log1p(exp(-x)); NaN propagates.
template <typename R> __device__ __forceinline__ R logaddexp0(R x) {
  const R t = Math<R>::log1p(Math<R>::exp(-Math<R>::abs(x)));
  R r = (x > R(0) ? x : R(0)) + t;
  if (x == R(0)) r = R(0.693147180559945309417232121458176568);
  return r;
}

// Soft contact between two agents (core.py:180-196), own-side force for delta = p_self - p_other:
//   dist = sqrt(dx^2 + dy^2); pen = logaddexp(0, -(dist - 0.3)/1e-3) * 1e-3; f = 100 * delta / dist * pen.
// The reference evaluates this for every agent pair every tick.  Beyond kSkip the soft-plus underflows to
// exactly 0 in the working precision (float: exp(x) = 0 for x <= -110, i.e. dist >= 0.41; double: x <= -750,
// dist >= 1.05), pen == 0 and the force is (+-)0, which leaves every accumulator bit-unchanged -- so the
// transcendental chain is skipped there.  The reach test runs on the SQUARED distance against the exact threshold
// Thresh<R>::kSkip2 (thresholds.h: sqrt(d2) >= kSkip <=> d2 >= kSkip2), so the square root itself is only taken
// on the slow path.  NaN distances take the slow path (NaN propagates as in the reference).
template <typename R> struct Contact;
template <> struct Contact<float> {
#ifndef CM3_F32_LIBM_SOFTPLUS
  // hardware soft-plus (logaddexp0<float> below): at dist = 0.32f, x = -19.99998 and exp2(x * log2 e) = 2^-28.85; for every
  // larger dist the term 1 + t rounds to exactly 1 (needs t <= 2^-24: a margin of 4.85 in the exponent against the ~1 ulp of
  // v_exp_f32), log2(1) = 0, the soft-plus is 0 + 0 and the force +-0
  static constexpr float kSkip = 0.32f;
#else
  static constexpr float kSkip = 0.41f;  // libm: expf(x) = 0 for x <= -110
#endif
};
template <> struct Contact<double> {
  static constexpr double kSkip = 1.05;
};

// the slow path alone, for callers that know d2 = dx^2 + dy^2 is within reach (!(d2 >= kSkip2))
template <typename R> __device__ __forceinline__ void contact_force_near(R dx, R dy, R d2, R &f_x, R &f_y) {
  const R kMargin = R(1e-3), kForce = R(1e+2), kDistMin = R(0.15) + R(0.15);
  const R dist = Math<R>::sqrt(d2);
  const R pen = logaddexp0<R>(-(dist - kDistMin) / kMargin) * kMargin;
  f_x = kForce * dx / dist * pen;
  f_y = kForce * dy / dist * pen;
}

// float32: the soft-plus of the contact chain on the hardware transcendental units (v_exp_f32 / v_log_f32, base 2, ~1 ulp).
// libm's expf + log1pf were 136 of the ~580 instructions a lane of the C2 kernel executes per tick, and by
// tools/probes/issue_probe.hip a lone wave pays 4 cycles for every one of them.  What it costs in accuracy:
//   softplus(x) = max(x, 0) + log(1 + exp(-|x|)); the second term lies in (0, ln 2] and is now computed with an ABSOLUTE
//   error of ~2e-7 (for t = exp(-|x|) < 6e-8 the sum 1 + t rounds to 1 and the term is dropped: another 6e-8).  That error
//   reaches the force as 100 * 1e-3 * 3e-7 = 3e-8 and a velocity as 3e-9 per tick -- three orders of magnitude below the
//   rounding of the float32 distance itself (1 ulp of dist = 3e-8 is amplified by 1/1e-3 in x) and below the 1e-5 bound.
//   Measured (tools/f32_error_scan.py, 20 x 4096 crowded states per agent count, against the float64 oracle): the worst
//   errors are IDENTICAL to the libm build's (N = 4: state 1.63e-6, obs_others 2.33e-6; N = 8: 2.97e-6 / 4.30e-6).
// The reference's own np.logaddexp is libm-dependent in its last ulps, so there never was a bit pattern
// to match here; sqrt and the three divisions of the chain stay IEEE (replacing them as well measured +30 % error, see
// profiles/r02_f32_softplus_hw.txt).  x == 0 gives 0 + log2(2) * ln 2 = ln 2; NaN propagates; beyond dist = 0.3166 the force
// is exactly +-0 (libm: beyond 0.41), which is what kSkip2 relies on only as an upper bound.  The float64 instantiation is
// unchanged (libm, the 1e-11 parity path).  -DCM3_F32_LIBM_SOFTPLUS builds the libm chain for comparisons.
#ifndef CM3_F32_LIBM_SOFTPLUS
template <> __device__ __forceinline__ float logaddexp0<float>(float x) {
  const float t = __builtin_amdgcn_exp2f(-fabsf(x) * 1.44269504088896340736f);
  return (x > 0.0f ? x : 0.0f) + __builtin_amdgcn_logf(1.0f + t) * 0.693147180559945309417f;
}
#endif

template <typename R> __device__ __forceinline__ void contact_force(R dx, R dy, R &f_x, R &f_y) {
  const R d2 = dx * dx + dy * dy;
  f_x = R(0);
  f_y = R(0);
  if (!(d2 >= Thresh<R>::kSkip2)) contact_force_near<R>(dx, dy, d2, f_x, f_y);
}

// is_collision (multi-goal_spread.py:114-118): sqrt(dx^2 + dy^2) < 0.15 + 0.15, on the squared distance (thresholds.h)
template <typename R> __device__ __forceinline__ bool is_collision(R dx, R dy) {
  return dx * dx + dy * dy < Thresh<R>::kColl2;
}

template <typename R, typename V4> __device__ __forceinline__ V4 sub4(const V4 &a, const V4 &b) {
  V4 r;
  r.x = a.x - b.x;
  r.y = a.y - b.y;
  r.z = a.z - b.z;
  r.w = a.w - b.w;
  return r;
}

// How the observation rows are stored (no later tick reads them).  NT = non-temporal: chosen by particle_rollout when the
// rollout's observation slots are a STREAM (see obs_store_nt); a COMPILE-TIME parameter of the step kernels, so the plain
// instantiations are byte for byte the code without this feature.  (A first version selected among four flavours at run
// time: that alone cost 1-3 % on every kernel that carried it -- profiles/r02_base_vs_new_runtime_flavours.txt -- and the
// write-through flavours it also offered bought nothing, profiles/r02_store_policy_ab.txt.)
constexpr uint32_t kFlagObsStoreNt = 0x100000u;  // internal launch flag, set by particle_rollout only
constexpr size_t kWtMinObsBytes = (size_t)3 << 20;
#ifndef CM3_AGENTS2_MAX_ENVS
#define CM3_AGENTS2_MAX_ENVS 32768   // measured crossover, profiles/r03_two_lanes_per_agent.txt (macro: build variant for that measurement)
#endif
constexpr size_t kAgents2MaxEnvs = CM3_AGENTS2_MAX_ENVS;  // N = 8: two lanes per agent up to this many envs per launch
constexpr size_t kAgents2EarlyMaxEnvs = 16384;            // ... with its write-through stores ahead of the reward work up to here

typedef float cm3_f4 __attribute__((ext_vector_type(4)));
// Store policy of the observation rows, a COMPILE-TIME parameter of the step kernels (kSpPlain kernels are byte for byte the code
// without this feature):
//   kSpPlain  ordinary stores
//   kSpNt     non-temporal hint: the rollout's observation slots are a STREAM (>= 128 MB, obs_store_nt); C2 trajectory 3.97 -> 3.53 us
//   kSpWt     write-through + non-temporal ("sc1 nt"): round 3, for launches that write >= kWtMinObsBytes of observation rows.  A
//             kernel boundary writes back every line the launch left dirty in the XCD L2s (the gap between two launches grows from
//             1.2 us to 2.1 us behind C5's 10.6 MB: profiles/r03_kernel_span_first_record.txt); written through, the rows leave
//             during the launch instead.  Measured (profiles/r03_obs_store_write_through.txt, same box, in place): N = 8 at 8192
//             envs 5.01 -> 4.48 us, 16 384 envs 6.97 -> 5.87; N = 4 at 65 536 envs 6.33 -> 5.76, 262 144 envs 17.9 -> 15.9, 2^22
//             envs 337 -> 305; and a LOSS where a launch writes little (C2: 0.8 MB, 2.55 -> 2.93 us) or in 4-byte pieces (Checkers).
constexpr int kSpPlain = 0, kSpNt = 1, kSpWt = 2;
template <int SP> __device__ __forceinline__ void store_obs_vec(float4 *p, const float4 &v) {
  if constexpr (SP == kSpWt) {
    const cm3_f4 t = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1 nt\n\ts_nop 1" ::"v"(p), "v"(t) : "memory");
  } else if constexpr (SP == kSpNt) {
    const cm3_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<cm3_f4 *>(p));
  } else {
    *p = v;
  }
}
template <int SP> __device__ __forceinline__ void store_obs_vec(double4 *p, const double4 &v) { *p = v; }
// The per-agent 4-byte outputs (actions, reward_n) of the lane-per-agent mapping: a wave writes 256 contiguous bytes = whole
// lines of them per instruction, so on the non-temporal path they take the hint too (C5 trajectory 5.56 -> 5.48 us per tick, same
// box; the per-ENV outputs are partial lines per instruction and lose with it: profiles/r02_small_outputs_nt_store.txt)
template <int SP, typename T> __device__ __forceinline__ void store_small(T *p, T v) {
  if constexpr (SP == kSpNt && sizeof(T) == 4) __builtin_nontemporal_store(v, p); else *p = v;
}

// np.sum(reward_n) as NumPy reduces a contiguous float64 vector (environment.py:107): left to right for
// n < 8, eight interleaved accumulators folded as a fixed tree for n == 8 (oracle: np_list_sum).
template <typename R, int N> __device__ __forceinline__ R sum_agents(const R (&v)[N]) {
  if constexpr (N < 8) {
    R acc = v[0];
#pragma unroll
    for (int i = 1; i < N; ++i) acc = acc + v[i];
    return acc;
  } else {
    static_assert(N < 16, "one block of eight accumulators, then the remainder one by one (NumPy's pairwise sum below 128 values)");
    R acc = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
#pragma unroll
    for (int i = 8; i < N; ++i) acc = acc + v[i];
    return acc;
  }
}

// ---- small row loads / stores ([E][N] arrays, one row per lane) ---------------------------------
template <typename T, int N> __device__ __forceinline__ void load_row(const T *base, size_t e, T (&v)[N]) {
  const T *p = base + e * N;
  constexpr int bytes = N * (int)sizeof(T);
  if constexpr (bytes % 16 == 0) {
    const uint4 *q = reinterpret_cast<const uint4 *>(p);
    uint4 tmp[bytes / 16];
#pragma unroll
    for (int k = 0; k < bytes / 16; ++k) tmp[k] = q[k];
    __builtin_memcpy(v, tmp, bytes);
  } else if constexpr (bytes % 8 == 0) {
    const uint2 *q = reinterpret_cast<const uint2 *>(p);
    uint2 tmp[bytes / 8];
#pragma unroll
    for (int k = 0; k < bytes / 8; ++k) tmp[k] = q[k];
    __builtin_memcpy(v, tmp, bytes);
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = p[k];
  }
}

template <typename T, int N> __device__ __forceinline__ void store_row(T *base, size_t e, const T (&v)[N]) {
  T *p = base + e * N;
  constexpr int bytes = N * (int)sizeof(T);
  if constexpr (bytes % 16 == 0) {
    uint4 tmp[bytes / 16];
    __builtin_memcpy(tmp, v, bytes);
    uint4 *q = reinterpret_cast<uint4 *>(p);
#pragma unroll
    for (int k = 0; k < bytes / 16; ++k) q[k] = tmp[k];
  } else if constexpr (bytes % 8 == 0) {
    uint2 tmp[bytes / 8];
    __builtin_memcpy(tmp, v, bytes);
    uint2 *q = reinterpret_cast<uint2 *>(p);
#pragma unroll
    for (int k = 0; k < bytes / 8; ++k) q[k] = tmp[k];
  } else {
#pragma unroll
    for (int k = 0; k < N; ++k) p[k] = v[k];
  }
}

// ---- obs_others geometry + LDS-staged store -------------------------------------------------------
template <typename R, int N> struct ObsGeom {
  static constexpr int NO = N > 1 ? N - 1 : 1;  // "others" per agent (N == 1 stores self, mgs.py:148-151)
  static constexpr int REC = N * NO * 4;        // reals per env
  static constexpr int STRIDE = REC + 4;        // +4 reals: spreads consecutive rows over LDS banks
  static constexpr int kBudget = 16384;         // LDS bytes per wave
  static constexpr int rows() {
    int r = 64;
    while (r > 1 && r * STRIDE * (int)sizeof(R) > kBudget) r >>= 1;
    return r;
  }
  static constexpr int ROWS = rows();
  static constexpr int LDS_REALS = ROWS * STRIDE;
};

// Cross-lane moves inside a 16-lane DPP row: no LDS round trip (a waited-for ds_bpermute is ~66 clocks, a DPP move ~11).
//   kDppXor1 / kDppXor2: lane ^ 1 / lane ^ 2 inside each quad; kDppHalfMirror: lane i <-> 7 - i inside each 8 lanes;
//   kDppBcast + n: lane n of the row to every lane of the row (gfx90a+).
constexpr int kDppXor1 = 0xB1, kDppXor2 = 0x4E, kDppHalfMirror = 0x141, kDppBcast = 0x150;
template <int CTRL> __device__ __forceinline__ float dpp_f32(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL> __device__ __forceinline__ int dpp_i32(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }

__device__ __forceinline__ void wave_lds_sync() {
  // LDS operations of one wave execute in issue order; this only stops the compiler from moving
  // LDS accesses across the write -> read hand-off inside the wave.
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// s[i] = (vx, vy, px, py) of agent i.  Row i of obs_others = concat_{j != i, ascending}(s[j] - s[i]).
template <typename R, int N, int SP = kSpPlain>
__device__ __forceinline__ void store_obs_others_staged(const typename Vec<R>::v4 (&s)[N], R *lds, int lane,
                                                        size_t e0, int E, R *out) {
  using V4 = typename Vec<R>::v4;
  using G = ObsGeom<R, N>;
  constexpr int VPR = G::REC / 4;  // vectors per env record
  V4 *lds4 = reinterpret_cast<V4 *>(lds);
#pragma unroll 1
  for (int pass = 0; pass < 64 / G::ROWS; ++pass) {
    const int r = lane - pass * G::ROWS;
    if (r >= 0 && r < G::ROWS) {
#pragma unroll
      for (int i = 0; i < N; ++i) {
#pragma unroll
        for (int k = 0; k < G::NO; ++k) {
          const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
          lds4[(r * G::STRIDE) / 4 + i * G::NO + k] = sub4<R, V4>(s[j], s[i]);
        }
      }
    }
    wave_lds_sync();
    const size_t row0 = e0 + (size_t)pass * G::ROWS;
    long rows_here = (long)E - (long)row0;
    rows_here = rows_here < 0 ? 0 : (rows_here > G::ROWS ? G::ROWS : rows_here);
    const int nvec = (int)rows_here * VPR;
    V4 *out4 = reinterpret_cast<V4 *>(out + row0 * G::REC);
    for (int f = lane; f < nvec; f += 64) {
      const int row = f / VPR, q = f - row * VPR;
      store_obs_vec<SP>(out4 + f, lds4[(row * G::STRIDE) / 4 + q]);
    }
    wave_lds_sync();
  }
}

template <typename R, int N>
__device__ __forceinline__ void store_obs_others_direct(const typename Vec<R>::v4 (&s)[N], size_t e, R *out) {
  using V4 = typename Vec<R>::v4;
  using G = ObsGeom<R, N>;
  V4 *o = reinterpret_cast<V4 *>(out + e * G::REC);
#pragma unroll
  for (int i = 0; i < N; ++i)
#pragma unroll
    for (int k = 0; k < G::NO; ++k) {
      const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
      o[i * G::NO + k] = sub4<R, V4>(s[j], s[i]);
    }
}

// ---- episode initialisation (multi-goal_spread.py:65-93), double arithmetic for both reals ---------
// One Bernoulli(prob_random) per episode shared by agents AND landmarks (:75).
__device__ __forceinline__ bool episode_is_random(const ParticleParams &p, uint64_t genv, uint32_t episode, uint64_t seed) {
  return u01(reset_words(seed, genv, episode, 0).x) < p.prob_random;
}
__device__ __forceinline__ bool episode_is_random(const ParticleParams &p, uint64_t genv, uint32_t episode) {
  return episode_is_random(p, genv, episode, p.seed);
}
// The seed as the same-launch reset of a step kernel sees it: the same value behind an empty asm statement, taken INSIDE the reset
// branch.  The reset stream and the per-tick action stream share the Philox key, and left to itself the compiler computes the key
// schedule (18 scalar adds) once at kernel entry and -- short of scalar registers across the tick -- carries it to the reset branch
// in VGPRs: 18 v_mov_b32 on the path of EVERY tick of the C2 kernel (7 % of its ~240 vector instructions) for a branch an env takes
// once per episode (round 5, tools/isa_dump.py).  Behind the asm the schedule is a different value: it is computed where it is used.
__device__ __forceinline__ uint64_t reset_seed(const ParticleParams &p) {
  uint32_t lo = (uint32_t)p.seed, hi = (uint32_t)(p.seed >> 32);
#if defined(__HIP_DEVICE_COMPILE__)
  asm volatile("" : "+v"(lo), "+v"(hi));   // (a vector register: an "s" constraint is refused where the value already lives in one)
#endif
  return ((uint64_t)hi << 32) | lo;
}

// Agent i (and its landmark) of the fresh episode.  `i` may be a run-time value: the config arrays are read
// through a compare-select chain, never through a dynamically indexed private array.
// `presets` (optional): the ax | ay | lx | ly arrays of `p` as they lie in the kernel-argument segment (preset_table below).
// The kernels whose lanes hold ONE agent (run-time i) read their four values from there with one indexed load each: with
// the select chains all 4 x CM3_MAX_AGENTS doubles had to sit in SGPRs at once, and the register allocator paid for that
// on the main path (81 SGPR spill moves per launch in the lane-per-agent kernel at N = 8).
template <typename R, int N, bool TABLE = false>
__device__ __forceinline__ void init_agent(const ParticleParams &p, uint64_t genv, uint32_t episode, bool rnd, int i,
                                           typename Vec<R>::v4 &s, typename Vec<R>::v2 &g, const double *presets = nullptr,
                                           const uint64_t *seed_in = nullptr) {
  const uint64_t seed = seed_in ? *seed_in : p.seed;
  const u32x4 a = reset_words(seed, genv, episode, 1u + (uint32_t)i);
  double x, y;
  if (rnd) {  // :77-78
    x = 2.0 * u01(a.x) - 1.0;
    y = 2.0 * u01(a.y) - 1.0;
  } else {  // :80-83
    double cx = 0.0, cy = 0.0;
    if constexpr (TABLE) {
      cx = presets[i];
      cy = presets[CM3_MAX_AGENTS + i];
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        cx = (k == i) ? p.ax[k] : cx;
        cy = (k == i) ? p.ay[k] : cy;
      }
    }
    x = cx;
    y = cy;
    if (p.initial_std != 0.0) {  // Box-Muller pair; std == 0 gives preset + 0 exactly as the reference
      // evaluated in the working precision: the float instantiation keeps the (rare) reset path free of
      // double-precision transcendentals (685 v_add_f64 in the first version)
      const R u1 = (R)u01(a.z), u2 = (R)u01(a.w);
      const R rad = Math<R>::sqrt(R(-2.0) * Math<R>::log(u1));
      const R ang = R(6.283185307179586476925286766559) * u2;
      x = x + p.initial_std * (double)(rad * Math<R>::cos(ang));
      y = y + p.initial_std * (double)(rad * Math<R>::sin(ang));
    }
  }
  s.x = R(0);
  s.y = R(0);
  s.z = R(x);
  s.w = R(y);
  if (rnd) {  // :88-89
    const u32x4 l = reset_words(seed, genv, episode, 1u + (uint32_t)N + (uint32_t)i);
    g.x = R(2.0 * u01(l.x) - 1.0);
    g.y = R(2.0 * u01(l.y) - 1.0);
  } else {  // :91
    double lx = 0.0, ly = 0.0;
    if constexpr (TABLE) {
      lx = presets[2 * CM3_MAX_AGENTS + i];
      ly = presets[3 * CM3_MAX_AGENTS + i];
    } else {
#pragma unroll
      for (int k = 0; k < N; ++k) {
        lx = (k == i) ? p.lx[k] : lx;
        ly = (k == i) ? p.ly[k] : ly;
      }
    }
    g.x = R(lx);
    g.y = R(ly);
  }
}

template <typename R, int N, bool TABLE = false>
__device__ __forceinline__ void init_episode(const ParticleParams &p, uint64_t genv, uint32_t episode,
                                             typename Vec<R>::v4 (&s)[N], typename Vec<R>::v2 (&g)[N],
                                             const double *presets = nullptr) {
  const bool rnd = episode_is_random(p, genv, episode);
  if constexpr (TABLE) {  // step kernel: presets read where they are used (see preset_table), not as 4 x 8 kernel-argument SGPR pairs
#pragma unroll
    for (int i = 0; i < N; ++i) init_agent<R, N, true>(p, genv, episode, rnd, i, s[i], g[i], presets);
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) init_agent<R, N>(p, genv, episode, rnd, i, s[i], g[i]);
  }
}

// The explicit arguments of k_particle_step_pairs / k_particle_step_agents as the code object lays them out in the
// kernel-argument segment (each argument at its natural alignment, in declaration order): gives the offset of `p`.
struct SharedEnvKernArgs {
  const void *state_in, *goals_in;
  const int32_t *meta_in, *episode;
  int E;
  uint32_t flags;
  int E0, EN, max_steps;
  const int32_t *actions;
  ParticleParams p;
};
static_assert(offsetof(ParticleParams, ay) == offsetof(ParticleParams, ax) + CM3_MAX_AGENTS * sizeof(double) &&
                  offsetof(ParticleParams, lx) == offsetof(ParticleParams, ax) + 2 * CM3_MAX_AGENTS * sizeof(double) &&
                  offsetof(ParticleParams, ly) == offsetof(ParticleParams, ax) + 3 * CM3_MAX_AGENTS * sizeof(double),
              "init_agent reads ax | ay | lx | ly as one table");
// kernel_arg_offset: where the kernel's ParticleParams argument starts in its kernel-argument segment
__device__ __forceinline__ const double *preset_table(size_t kernel_arg_offset = offsetof(SharedEnvKernArgs, p)) {
#if defined(__HIP_DEVICE_COMPILE__)
  const char *args = (const char *)__builtin_amdgcn_kernarg_segment_ptr();  // constant address space -> generic
  return reinterpret_cast<const double *>(args + kernel_arg_offset + offsetof(ParticleParams, ax));
#else
  return nullptr;  // host pass of the single-source compile: never called
#endif
}

// ---- the step kernel --------------------------------------------------------------------------------
template <typename R, int N, int WAVES, bool FUSED, int SP = kSpPlain>
__global__ void __launch_bounds__(WAVES * 64) k_particle_step(const ParticleParams p) {
  using V4 = typename Vec<R>::v4;
  using V2 = typename Vec<R>::v2;
  using G = ObsGeom<R, N>;
  __shared__ __attribute__((aligned(32))) R lds_all[WAVES][G::LDS_REALS];

  CM3_SPAN_IN();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t e0 = (size_t)p.E0 + ((size_t)blockIdx.x * WAVES + wave) * 64;
  const size_t e = e0 + lane;
  const bool active = e < (size_t)p.EN;
  const size_t ec = active ? e : (size_t)p.EN - 1;  // clamped index for loads
  const size_t E = (size_t)p.E;
  CM3_STAMP(0, false);

  // ---- loads: one vector per lane per array row, unit stride over e --------------------------------
  V4 s[N];
  V2 g[N];
  const V4 *sin4 = reinterpret_cast<const V4 *>(p.state_in);
  const V2 *gin2 = reinterpret_cast<const V2 *>(p.goals_in);
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = sin4[(size_t)i * E + ec];
#pragma unroll
  for (int i = 0; i < N; ++i) g[i] = gin2[(size_t)i * E + ec];
  const int2 meta = reinterpret_cast<const int2 *>(p.meta_in)[ec];
  int steps = meta.x, collisions = meta.y;
  const bool auto_reset = (p.flags & CM3_FLAG_AUTO_RESET) != 0;

  const bool gen = (p.flags & CM3_FLAG_GEN_ACTIONS) != 0;
  uint32_t episode = 0;
  if (gen || (p.flags & CM3_FLAG_AUTO_RESET)) episode = (uint32_t)p.episode[ec];
  const uint32_t episode_in = episode;
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
  // stage 1 of the action stream (philox.h): the Philox blocks of this env depend on nothing the launch loads
  uint32_t ablock[4 * ((N + 3) / 4)];
  if (gen) {
#pragma unroll
    for (int c = 0; c < (N + 3) / 4; ++c) {
      const u32x4 w = action_block(p.seed, genv, (uint32_t)c);
      ablock[4 * c + 0] = w.x;
      ablock[4 * c + 1] = w.y;
      ablock[4 * c + 2] = w.z;
      ablock[4 * c + 3] = w.w;
    }
  }
  const R kDt = R(0.1), kKeep = R(1 - 0.25);
  if constexpr (N >= 2)  // (measured: N = 2..4 at 16 384 - 65 536 envs 1.5-4 % faster with it, N = 1 3 % slower)
    CM3_FETCH_EARLY(p.state_out, p.goals_out, p.actions, p.reward_n, p.reward, p.done, p.obs_others, p.meta_out, p.collisions_tick,
                    p.state_copy, p.max_steps);
  CM3_STAMP(1, true);

  // The state stays in registers across the ticks of this launch (n_ticks == 1: plain one-launch-per-tick step).
  // FUSED == false: exactly one tick, the loop and every per-tick pointer offset fold away at compile time
  const int n_ticks = FUSED ? p.n_ticks : 1;
#pragma unroll 1
  for (int t = 0; t < n_ticks; ++t) {
    int act[N];
    int32_t *actions_t = tick_ptr(p.actions, p.st_actions, t);
    if (gen) {  // train_onpolicy.py:305-307
#pragma unroll
      for (int i = 0; i < N; ++i) act[i] = rand5(action_word(ablock[i], episode, (uint32_t)steps));   // stage 2
      if (active) store_row<int32_t, N>(actions_t, e, act);
    } else {
      load_row<int32_t, N>(actions_t, ec, act);
    }

    // ---- _set_action (environment.py:193-214) + apply_action_force (core.py:134-140) ----------------
    R fx[N], fy[N];
#pragma unroll
    for (int i = 0; i < N; ++i) {
      R ux = R(0), uy = R(0);
      if (act[i] == 1) ux = R(-1);
      if (act[i] == 2) ux = R(+1);
      if (act[i] == 3) uy = R(-1);
      if (act[i] == 4) uy = R(+1);
      fx[i] = ux * R(5.0) + R(0.0);
      fy[i] = uy * R(5.0) + R(0.0);
    }

    // ---- apply_environment_force (core.py:143-155) / get_collision_force (:180-196) -----------------
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
      for (int b = a + 1; b < N; ++b) {
        R f_x, f_y;
        contact_force<R>(s[a].z - s[b].z, s[a].w - s[b].w, f_x, f_y);
        fx[a] = f_x + fx[a];
        fy[a] = f_y + fy[a];
        fx[b] = (-f_x) + fx[b];
        fy[b] = (-f_y) + fy[b];
      }
    }

    // ---- integrate_state (core.py:158-169): mass 1, max_speed None -----------------------------------
#pragma unroll
    for (int i = 0; i < N; ++i) {
      s[i].x = s[i].x * kKeep;
      s[i].y = s[i].y * kKeep;
      s[i].x = s[i].x + (fx[i] / R(1.0)) * kDt;
      s[i].y = s[i].y + (fy[i] / R(1.0)) * kDt;
      s[i].z = s[i].z + s[i].x * kDt;
      s[i].w = s[i].w + s[i].y * kDt;
    }
    steps += 1;  // environment.py:93
    CM3_STAMP(2, false);

    // ---- reward / reached (multi-goal_spread.py:121-143) ----------------------------------------------
    R rew[N];
    bool all_reached = true;
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const R dx = s[i].z - g[i].x, dy = s[i].w - g[i].y;
      const R d2 = dx * dx + dy * dy;
      rew[i] = R(0) - Math<R>::sqrt(d2);
      all_reached = all_reached && (d2 < Thresh<R>::kReach2);  // rew[i] >= -0.05 (thresholds.h)
    }
#pragma unroll
    for (int a = 0; a < N; ++a) {
#pragma unroll
      for (int b = a + 1; b < N; ++b) {
        // is_collision is symmetric, so the reference's two ordered visits (j,i) and (i,j) collapse into one
        if (is_collision<R>(s[b].z - s[a].z, s[b].w - s[a].w)) {
          rew[a] = rew[a] - R(1);
          rew[b] = rew[b] - R(1);
          collisions += 2;  // double counts by design (:135-137)
        }
      }
    }
    const R reward = sum_agents<R, N>(rew);                   // environment.py:107
    const bool done = (steps == p.max_steps) || all_reached;  // environment.py:118-121

    CM3_STAMP(3, false);
    if (active) {
      store_row<R, N>(reinterpret_cast<R *>(tick_ptr(p.reward_n, p.st_reward_n, t)), e, rew);
      reinterpret_cast<R *>(tick_ptr(p.reward, p.st_reward, t))[e] = reward;
      tick_ptr(p.done, p.st_done, t)[e] = done ? 1 : 0;
      if (p.collisions_tick) tick_ptr(p.collisions_tick, p.st_coll, t)[e] = collisions;
    }
    CM3_STAMP(8, false);

    // ---- same-launch re-initialisation of finished episodes -------------------------------------------
    bool was_reset = false;
    if (auto_reset && done) {
      if (active) {
        void *term_state = tick_ptr(p.term_state, p.st_term_state, t);
        void *term_obs = tick_ptr(p.term_obs_others, p.st_term_obs, t);
        if (term_state) {
          V4 *t4 = reinterpret_cast<V4 *>(term_state);
#pragma unroll
          for (int i = 0; i < N; ++i) t4[(size_t)i * E + e] = s[i];
        }
        if (term_obs) store_obs_others_direct<R, N>(s, e, reinterpret_cast<R *>(term_obs));
      }
      episode += 1;
      init_episode<R, N, true>(p, genv, episode, s, g, preset_table(0));  // (the kernel's only argument is `p`)
      steps = 0;
      collisions = 0;
      was_reset = true;
    }
    CM3_STAMP(9, false);

    if (active) {
      V4 *sout4 = reinterpret_cast<V4 *>(tick_ptr(p.state_out, p.st_state, t));
#pragma unroll
      for (int i = 0; i < N; ++i) sout4[(size_t)i * E + e] = s[i];
      if (p.goals_out != p.goals_in || was_reset) {
        V2 *gout2 = reinterpret_cast<V2 *>(tick_ptr(p.goals_out, p.st_goals, t));
#pragma unroll
        for (int i = 0; i < N; ++i) gout2[(size_t)i * E + e] = g[i];
      }
      if (p.state_copy) {  // live-state rollout: the trajectory slot gets a copy of the state (write-only stream)
#pragma unroll
        for (int i = 0; i < N; ++i) store_obs_vec<SP>(reinterpret_cast<V4 *>(p.state_copy) + ((size_t)i * E + e), s[i]);
      }
      if (was_reset && p.goals_copy) {  // goal slots are SPARSE wherever the goals live in place: written where an env restarts
#pragma unroll
        for (int i = 0; i < N; ++i) reinterpret_cast<V2 *>(p.goals_copy)[(size_t)i * E + e] = g[i];
      }
    }

    CM3_STAMP(4, false);
    // ---- observation (multi-goal_spread.py:145-154), env-major rows through the wave's LDS tile --------
    store_obs_others_staged<R, N, SP>(s, &lds_all[wave][0], lane, e0, p.EN,
                                      reinterpret_cast<R *>(tick_ptr(p.obs_others, p.st_obs, t)));
    CM3_STAMP(5, false);
  }

  // ---- live counters, once per launch ----------------------------------------------------------------------
  if (active) {
    int2 m;
    m.x = steps;
    m.y = collisions;
    reinterpret_cast<int2 *>(p.meta_out)[e] = m;
    if (episode != episode_in) p.episode[e] = (int32_t)episode;
  }
  CM3_STAMP(6, true);
  CM3_SPAN_OUT(p.span);
}

// ---- the step kernel, second mapping: ONE LANE PER ORDERED AGENT PAIR -------------------------------------
// For small and medium batches the lane-per-env kernel is latency-bound: a 4096-env tick is only 64 waves,
// each walking all C(N,2) contact evaluations, N(N-1)/2 collision tests and N reward distances serially.
// Here an env owns a group of G = pow2 >= N(N-1) consecutive lanes; lane (i,k) handles agent i and its k-th
// other agent j.  It evaluates ONE contact force (own side: delta = p_i - p_j, bit-identical to the
// reference's +-force of the unordered pair because IEEE subtraction/negation are sign-symmetric), the N-1
// contributions of agent i are gathered with wavefront shuffles and added in the reference's order (j
// ascending = pair order of core.py:145-147), the post-step positions of j come back by shuffle, and the
// collision tests / reached flags are reduced with wave ballots.  No LDS: lane (i,k) writes the obs_others
// vector (i,k), so a wave stores 64/G whole env records contiguously (16 B per lane, unit stride).



template <int N> struct PairGeom {
  static constexpr int NO = N - 1;
  static constexpr int SLOTS = N * NO;
  // lanes per agent: its N-1 pair lanes, padded to a power of two where that keeps the group size (N = 3, 4, 5) -- an agent is then
  // a pair / a quad of lanes and its forces come together with DPP quad permutes instead of an LDS shuffle round trip
  static constexpr int LA = (N >= 3 && N <= 5) ? (N == 3 ? 2 : 4) : NO;
  static constexpr int pow2ceil(int v) {
    int r = 1;
    while (r < v) r <<= 1;
    return r;
  }
  static constexpr int G = pow2ceil(N * LA);
  static constexpr int EPW = 64 / G;  // envs per wave
};

// (Round 1 gave the workgroup an extra "draw wave" that drew the NEXT launch's actions while the physics waves ran, handing the
// action row forward between launches: 3.43 -> 3.30 us per tick at C2 then.  Re-measured in round 2 on the same box
// (profiles/r02_draw_wave_on_off.txt) it had become a loss in every configuration that used it -- in place 0-6 %, trajectory mode
// 5-10 % -- and was removed: a fifth wave, two workgroup barriers and an extra store + reload of the action row cost more than the
// Philox chain they hid once the square roots had left the physics chain.)
template <typename R, int N, int WAVES, bool FUSED, int SP = kSpPlain, bool LIVE = false, int TU = CM3_PARTICLE_TU>
__global__ void __launch_bounds__(WAVES * 64)
    k_particle_step_pairs(const void *h_state_in, const void *h_goals_in, const int32_t *h_meta_in, const int32_t *h_episode,
                          const int h_E, const uint32_t h_flags, const int h_E0, const int h_EN, const int h_max_steps,
                          const int32_t *h_actions, const ParticleParams p) {
  // The leading arguments repeat the fields of `p` that the first loads need: scalar kernel arguments are preloaded into
  // SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count; the first 14 dwords in practice: everything up to and
  // including h_max_steps), so the addresses of the first loads do not wait for a kernarg fetch.
  // Index and address arithmetic is 32-bit: every array of one tick is below 4 GiB (checked by launch_pairs), so an element is
  // <uniform base pointer> + <32-bit byte offset of the lane>, which the hardware adds itself (global_load/store with a
  // scalar base).  By tools/probes/issue_probe.hip a lone wave issues one VALU instruction per 4 cycles whatever it is, so at
  // one wave per SIMD the launch time is the length of the executed path: the 64-bit multiplies, adds and compares of
  // size_t indexing were ~8 % of it.
  static_assert(N >= 2, "the pair mapping needs at least two agents");
  using V4 = typename Vec<R>::v4;
  using V2 = typename Vec<R>::v2;
  using PG = PairGeom<N>;
  constexpr int NO = PG::NO, SLOTS = PG::SLOTS, G = PG::G, EPW = PG::EPW, LA = PG::LA;
  static_assert(PairGeom<N>::pow2ceil(SLOTS) == G, "the padded lane layout must not change the group size");
  CM3_SPAN_IN();

  const int lane = threadIdx.x & 63, wave_all = threadIdx.x >> 6;
  const int wave = wave_all;
  const int gslot = lane & (G - 1), sub = lane / G, base = lane - gslot;
  const uint32_t E = (uint32_t)h_E, EN = (uint32_t)h_EN;  // array extent (row stride); this launch covers envs [h_E0, h_EN)
  const uint32_t e = (uint32_t)h_E0 + (cm3_xcd_block(h_flags) * WAVES + wave) * EPW + sub;
  const bool env_ok = e < EN;
  const uint32_t ec = env_ok ? e : EN - 1;
  // lane gslot of the group = pair lane k of agent i: gslot = i * LA + k, k < N - 1 (LA == N - 1 unless padded, see PairGeom)
  const int i_raw = gslot / LA, k_raw = gslot - i_raw * LA;
  const bool slot_ok = i_raw < N && k_raw < NO;
  const int i = slot_ok ? i_raw : 0, k = slot_ok ? k_raw : 0, j = k < i ? k : k + 1;
  const int vslot = i * NO + k;  // index of this lane's obs_others vector inside the env record
  const bool lead = slot_ok && k == 0;  // one lane per agent does the per-agent stores
  const bool head = gslot == 0;         // one lane per env does the per-env stores

  CM3_STAMP(0, false);
  // ---- loads (once per launch; the state then lives in registers across the ticks of this launch) -------------
  const uint32_t row_i = (uint32_t)i * E, row_j = (uint32_t)j * E;
  V4 si = *at32<const V4>(h_state_in, (row_i + ec) * (uint32_t)sizeof(V4));
  V4 sj = *at32<const V4>(h_state_in, (row_j + ec) * (uint32_t)sizeof(V4));
  V2 gl = *at32<const V2>(h_goals_in, (row_i + ec) * (uint32_t)sizeof(V2));
  const int2 meta = *at32<const int2>(h_meta_in, ec * 8u);
  int steps = meta.x, collisions = meta.y;
  const bool auto_reset = (h_flags & CM3_FLAG_AUTO_RESET) != 0;
  const bool gen = (h_flags & CM3_FLAG_GEN_ACTIONS) != 0;
  uint32_t episode = 0;
  if (gen || (h_flags & CM3_FLAG_AUTO_RESET)) episode = (uint32_t)*at32<const int32_t>(h_episode, ec * 4u);
  const uint32_t episode_in = episode;
  // (after the vector loads are in flight: the scalar fetches complete in their shadow)
  CM3_FETCH_EARLY(p.state_out, p.goals_out, p.goals_in, p.collisions_tick, p.reward_n, p.reward, p.done, p.obs_others, p.meta_out);
  if constexpr (LIVE) CM3_FETCH_EARLY(p.state_copy, p.goals_copy);
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
  // stage 1 of the action stream (philox.h): this agent's word of its env's Philox block -- ten rounds that depend on nothing the
  // launch loads, computed while the loads are in flight (until round 3 the whole draw waited for `steps` / `episode`)
  uint32_t aword = 0;
  if (gen) aword = pick_word(action_block(p.seed, genv, (uint32_t)(i >> 2)), i & 3);
  const R kDt = R(0.1), kKeep = R(1 - 0.25);

  CM3_STAMP(1, true);
  CM3_SPAN_MARK(0, true);   // loads back
  // FUSED == false: exactly one tick, the loop and every per-tick pointer offset fold away at compile time
  const int n_ticks = FUSED ? p.n_ticks : 1;
#pragma unroll 1
  for (int t = 0; t < n_ticks; ++t) {
    int32_t *actions_t = tick_ptr(p.actions, p.st_actions, t);
    int act = 0;
    R f_x, f_y;
    if (gen) {  // train_onpolicy.py:305-307
      act = rand5(action_word(aword, episode, (uint32_t)steps));  // stage 2 (stored with the other per-agent outputs at the end of the tick)
    } else {
      act = *at32<const int32_t>(actions_t, (ec * N + i) * 4u);
    }

    CM3_STAMP(2, false);
    CM3_SPAN_MARK(1, false);  // action drawn
    // ---- action force + own contact force --------------------------------------------------------------------
    R ux = R(0), uy = R(0);
    if (act == 1) ux = R(-1);
    if (act == 2) ux = R(+1);
    if (act == 3) uy = R(-1);
    if (act == 4) uy = R(+1);
    R Fx = ux * R(5.0) + R(0.0), Fy = uy * R(5.0) + R(0.0);
    contact_force<R>(si.z - sj.z, si.w - sj.w, f_x, f_y);
    CM3_STAMP(3, false);
    if constexpr (sizeof(R) == 4 && LA == 4 && NO >= 3) {  // the agent's lanes are a quad
      Fx = dpp_f32<0x00>(f_x) + Fx;  // quad_perm:[0,0,0,0]
      Fy = dpp_f32<0x00>(f_y) + Fy;
      Fx = dpp_f32<0x55>(f_x) + Fx;  // [1,1,1,1]
      Fy = dpp_f32<0x55>(f_y) + Fy;
      Fx = dpp_f32<0xAA>(f_x) + Fx;  // [2,2,2,2]
      Fy = dpp_f32<0xAA>(f_y) + Fy;
      if constexpr (NO == 4) {
        Fx = dpp_f32<0xFF>(f_x) + Fx;  // [3,3,3,3]
        Fy = dpp_f32<0xFF>(f_y) + Fy;
      }
    } else if constexpr (N == 2) {  // one pair lane per agent: its only contribution is its own
      Fx = f_x + Fx;
      Fy = f_y + Fy;
    } else if constexpr (sizeof(R) == 4 && LA == 2) {  // N = 3: two agents per quad
      Fx = dpp_f32<0xA0>(f_x) + Fx;  // quad_perm:[0,0,2,2]
      Fy = dpp_f32<0xA0>(f_y) + Fy;
      Fx = dpp_f32<0xF5>(f_x) + Fx;  // [1,1,3,3]
      Fy = dpp_f32<0xF5>(f_y) + Fy;
    } else {
#pragma unroll
      for (int kk = 0; kk < NO; ++kk) {  // contributions of agent i in the reference's order (j ascending)
        const int src = base + i * LA + kk;
        Fx = __shfl(f_x, src, 64) + Fx;
        Fy = __shfl(f_y, src, 64) + Fy;
      }
    }

    // ---- integrate agent i (every lane of agent i computes the same values) ---------------------------------
    si.x = si.x * kKeep;
    si.y = si.y * kKeep;
    si.x = si.x + (Fx / R(1.0)) * kDt;
    si.y = si.y + (Fy / R(1.0)) * kDt;
    si.z = si.z + si.x * kDt;
    si.w = si.w + si.y * kDt;
    steps += 1;
    if constexpr (N == 2 && sizeof(R) == 4) {  // the other agent is the neighbouring lane
      sj.x = dpp_f32<kDppXor1>(si.x);
      sj.y = dpp_f32<kDppXor1>(si.y);
      sj.z = dpp_f32<kDppXor1>(si.z);
      sj.w = dpp_f32<kDppXor1>(si.w);
    } else {
      const int src = base + j * LA;  // lead lane of agent j
      sj.x = __shfl(si.x, src, 64);
      sj.y = __shfl(si.y, src, 64);
      sj.z = __shfl(si.z, src, 64);
      sj.w = __shfl(si.w, src, 64);
    }

    CM3_STAMP(4, true);
    CM3_SPAN_MARK(2, false);  // forces, integration, post-step positions exchanged
    // ---- reward / reached / collisions (multi-goal_spread.py:114-143) ----------------------------------------
    R rew;
    bool reached;
    {
      const R dx = si.z - gl.x, dy = si.w - gl.y;
      const R d2 = dx * dx + dy * dy;
      rew = R(0) - Math<R>::sqrt(d2);
      reached = d2 < Thresh<R>::kReach2;  // rew >= -0.05 (thresholds.h)
    }
    const bool hit = slot_ok && is_collision<R>(sj.z - si.z, sj.w - si.w);  // is_collision(a = j, agent = i)
    const unsigned long long hits = __ballot(hit);
    const unsigned long long grp = (G == 64) ? hits : ((hits >> base) & ((1ull << (G & 63)) - 1ull));
    const int c_i = __popcll((grp >> (i * LA)) & ((1ull << NO) - 1ull));
#pragma unroll
    for (int c = 0; c < NO; ++c)
      if (c < c_i) rew = rew - R(1);
    collisions += __popcll(grp);  // every ordered visit counts (:135-137)
    const unsigned long long rb = __ballot(reached && lead);
    const unsigned long long rgrp = (G == 64) ? rb : ((rb >> base) & ((1ull << (G & 63)) - 1ull));
    const bool all_reached = __popcll(rgrp) == N;
    R rews[N];
    if constexpr (N == 2 && sizeof(R) == 4) {  // G == 2: agents 0 / 1 are the even / odd lane of each pair
      rews[0] = dpp_f32<0xA0>(rew);  // quad_perm:[0,0,2,2]
      rews[1] = dpp_f32<0xF5>(rew);  // [1,1,3,3]
    } else if constexpr (N == 4 && sizeof(R) == 4) {  // G == 16: an env is one DPP row, agent a's lead lane is lane a * LA of it
      rews[0] = dpp_f32<kDppBcast + 0 * LA>(rew);
      rews[1] = dpp_f32<kDppBcast + 1 * LA>(rew);
      rews[2] = dpp_f32<kDppBcast + 2 * LA>(rew);
      rews[3] = dpp_f32<kDppBcast + 3 * LA>(rew);
    } else {
#pragma unroll
      for (int a = 0; a < N; ++a) rews[a] = __shfl(rew, base + a * LA, 64);
    }
    const R reward = sum_agents<R, N>(rews);
    const bool done = (steps == h_max_steps) || all_reached;

    const int collisions_tick = collisions;  // scenario.collisions after this tick, before any same-launch reset

    CM3_STAMP(5, true);
    CM3_SPAN_MARK(3, false);  // rewards / done stored
    // ---- same-launch re-initialisation -------------------------------------------------------------------------
    bool was_reset = false;
    if (auto_reset && done) {
      if (env_ok) {
        void *term_state = tick_ptr(p.term_state, p.st_term_state, t);
        void *term_obs = tick_ptr(p.term_obs_others, p.st_term_obs, t);
        if (term_state && lead) *at32<V4>(term_state, (row_i + e) * (uint32_t)sizeof(V4)) = si;
        if (term_obs && slot_ok) *at32<V4>(term_obs, (e * SLOTS + vslot) * (uint32_t)sizeof(V4)) = sub4<R, V4>(sj, si);
      }
      episode += 1;
      const uint64_t rseed = reset_seed(p);
      const bool rnd = episode_is_random(p, genv, episode, rseed);
      V2 gj;
      init_agent<R, N, true>(p, genv, episode, rnd, i, si, gl, preset_table(), &rseed);
      init_agent<R, N, true>(p, genv, episode, rnd, j, sj, gj, preset_table(), &rseed);
      steps = 0;
      collisions = 0;
      was_reset = true;
    }

    CM3_STAMP(6, false);
    // ---- per-tick stores ------------------------------------------------------------------------------------------
    // Stores grouped by predicate -- per agent, per pair, per env: an exec-masked region (s_and_saveexec ... s_or exec) costs a lone
    // wave about twice its instruction count (tools/probes/salu_valu_probe.hip), and the first version had one region per output,
    // eight per tick.  (Storing from ALL lanes instead -- duplicates writing the same value -- was measured too: +8 % at C2.)
    if (env_ok) {
      if (lead) {
        if (gen) *at32<int32_t>(actions_t, (e * N + i) * 4u) = act;
        *at32<R>(tick_ptr(p.reward_n, p.st_reward_n, t), (e * N + i) * (uint32_t)sizeof(R)) = rew;
        *at32<V4>(tick_ptr(p.state_out, p.st_state, t), (row_i + e) * (uint32_t)sizeof(V4)) = si;
        if (p.goals_out != p.goals_in || was_reset)
          *at32<V2>(tick_ptr(p.goals_out, p.st_goals, t), (row_i + e) * (uint32_t)sizeof(V2)) = gl;
        if constexpr (LIVE)   // live-state rollout (a compile-time variant: the others carry none of it): the slot gets a copy
          store_obs_vec<SP>(at32<V4>(p.state_copy, (row_i + e) * (uint32_t)sizeof(V4)), si);
        if constexpr (!FUSED) {   // goal slots are SPARSE wherever the goals live in place (live state, or goals_live alone)
          if (was_reset && p.goals_copy) *at32<V2>(p.goals_copy, (row_i + e) * (uint32_t)sizeof(V2)) = gl;
        }
      }
      // observation (multi-goal_spread.py:145-154): vector (i,k) of env e; lanes of a wave cover whole records
      if (slot_ok)
        store_obs_vec<SP>(at32<V4>(tick_ptr(p.obs_others, p.st_obs, t), (e * SLOTS + vslot) * (uint32_t)sizeof(V4)), sub4<R, V4>(sj, si));
      if (head) {
        *at32<R>(tick_ptr(p.reward, p.st_reward, t), e * (uint32_t)sizeof(R)) = reward;
        *at32<uint8_t>(tick_ptr(p.done, p.st_done, t), e) = done ? 1 : 0;
        if (p.collisions_tick) *at32<int32_t>(tick_ptr(p.collisions_tick, p.st_coll, t), e * 4u) = collisions_tick;
        if constexpr (!FUSED) {  // the live counters, once per launch
          int2 m;
          m.x = steps;
          m.y = collisions;
          *at32<int2>(p.meta_out, e * 8u) = m;
          if (episode != episode_in) *at32<int32_t>(p.episode, e * 4u) = (int32_t)episode;
        }
      }
    }
  }

  CM3_SPAN_MARK(4, false);  // state + observation stores issued
  CM3_STAMP(7, false);
  // ---- live counters, once per launch -------------------------------------------------------------------------------
  if constexpr (FUSED) {
    if (env_ok && head) {
      int2 m;
      m.x = steps;
      m.y = collisions;
      *at32<int2>(p.meta_out, e * 8u) = m;
      if (episode != episode_in) *at32<int32_t>(p.episode, e * 4u) = (int32_t)episode;
    }
  }
  CM3_STAMP(8, true);
  CM3_SPAN_OUT(p.span);
}

// ---- the step kernel, third mapping: ONE LANE PER AGENT -----------------------------------------------------
// With many agents the pair mapping spends its time on per-lane overhead that every one of the N(N-1) lanes of an env
// repeats (action draw, integration, reward, bookkeeping): at N = 8 it is VALU-bound (531 VALU instructions on each of
// 8192 waves at 8192 envs = 7 us of issue time) and over-fetches 2.6x (a wave = one env touches a separate cache line
// per agent row).  Here an env owns G = pow2 >= N consecutive lanes, lane i = agent i: it walks its N-1 contact forces
// itself (other agents' positions by wavefront shuffle, contributions added in the reference's order), integrates once,
// counts its own collisions, and a wave covers 64/G consecutive envs -- agent-row loads and stores are whole cache lines.
// obs_others rows (N-1 vectors per lane) are transposed through a wave-private LDS tile laid out exactly like the
// 64/G env records in memory, so the wave stores them as contiguous 16-byte-per-lane rows.
// Bit-identical to the other two mappings (same expressions, same summation orders).
template <int N> struct AgentGeom {
  static constexpr int NO = N - 1;
  static constexpr int G = PairGeom<N>::pow2ceil(N);
  static constexpr int EPW = 64 / G;        // envs per wave
  static constexpr int VPE = N * NO;        // obs vectors per env record
};

template <typename R, int N, int WAVES, bool FUSED, int SP = kSpPlain, bool LIVE = false, int TU = CM3_PARTICLE_TU>
__global__ void __launch_bounds__(WAVES * 64)
    k_particle_step_agents(const void *h_state_in, const void *h_goals_in, const int32_t *h_meta_in, const int32_t *h_episode,
                           const int h_E, const uint32_t h_flags, const int h_E0, const int h_EN, const int h_max_steps,
                           const int32_t *h_actions, const ParticleParams p) {
  // Leading scalar arguments as in k_particle_step_pairs: they repeat the fields of `p` that the first loads need and are
  // preloaded into SGPRs at wave launch (-mllvm -amdgpu-kernarg-preload-count), so the first addresses do not wait for a
  // kernarg fetch (worth 7.7 % on the pair kernel at C2, profiles/r02_remaining_round1_tunings_rechecked.txt).  32-bit index
  // and address arithmetic as in the pair kernel (every per-tick array below 4 GiB, checked by launch_agents).
  static_assert(N >= 2, "the agent mapping needs at least two agents");
  using V4 = typename Vec<R>::v4;
  using V2 = typename Vec<R>::v2;
  using AG = AgentGeom<N>;
  constexpr int NO = AG::NO, G = AG::G, EPW = AG::EPW, VPE = AG::VPE;
  __shared__ __attribute__((aligned(32))) R lds_all[WAVES][EPW * VPE * 4];
  CM3_SPAN_IN();

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane & (G - 1), sub = lane / G, base = lane - gi;
  const uint32_t E = (uint32_t)h_E, EN = (uint32_t)h_EN;  // array extent (row stride); this launch covers envs [E0, EN)
  const uint32_t e0 = (uint32_t)h_E0 + (cm3_xcd_block(h_flags) * WAVES + wave) * EPW;
  const uint32_t e = e0 + sub;
  const bool env_ok = e < EN;
  const uint32_t ec = env_ok ? e : EN - 1;
  const bool agent_ok = gi < N;
  const int i = agent_ok ? gi : 0;
  const bool mine = env_ok && agent_ok;  // this lane owns agent i of env e
  const bool head = env_ok && gi == 0;   // one lane per env does the per-env stores
  V4 *lds4 = reinterpret_cast<V4 *>(&lds_all[wave][0]);

  CM3_STAMP(0, false);
  // ---- loads (once per launch; the state then lives in registers across the ticks of this launch) -------------
  const uint32_t row_i = (uint32_t)i * E;
  V4 si = *at32<const V4>(h_state_in, (row_i + ec) * (uint32_t)sizeof(V4));
  V2 gl = *at32<const V2>(h_goals_in, (row_i + ec) * (uint32_t)sizeof(V2));
  const int2 meta = *at32<const int2>(h_meta_in, ec * 8u);
  int steps = meta.x, collisions = meta.y;
  const bool auto_reset = (h_flags & CM3_FLAG_AUTO_RESET) != 0;
  const bool gen = (h_flags & CM3_FLAG_GEN_ACTIONS) != 0;
  uint32_t episode = 0;
  if (gen || (h_flags & CM3_FLAG_AUTO_RESET)) episode = (uint32_t)*at32<const int32_t>(h_episode, ec * 4u);
  const uint32_t episode_in = episode;
  // (after the vector loads are in flight: the scalar fetches complete in their shadow)
  CM3_FETCH_EARLY(p.state_out, p.goals_out, p.goals_in, p.collisions_tick, p.reward_n, p.reward, p.done, p.obs_others, p.meta_out);
  if constexpr (LIVE) CM3_FETCH_EARLY(p.state_copy, p.goals_copy);
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
  // stage 1 of the action stream (philox.h): this agent's word of its env's Philox block, computed while the loads are in flight
  uint32_t aword = 0;
  if (gen) aword = pick_word(action_block(p.seed, genv, (uint32_t)(i >> 2)), i & 3);
  const R kDt = R(0.1), kKeep = R(1 - 0.25);

  // rows of the wave's obs tile that exist (envs past the launch's range are not stored)
  long envs_here = (long)EN - (long)e0;
  envs_here = envs_here < 0 ? 0 : (envs_here > EPW ? EPW : envs_here);
  const int nvec = (int)envs_here * VPE;

  // obs rows of this lane (vector k of agent i = s_j - s_i, j the k-th other agent) -> LDS tile -> contiguous rows at dst
  // Straight-line code (no branch per vector, no loop with a run-time trip count): the k-th other agent of agent i is k or k + 1,
  // picked with selects; the tile goes out in ceil(EPW * VPE / 64) unrolled steps, all LDS reads issued before the first wait.
  // (The first version -- `if (j != i) tile[..] = ..` and `for (f = lane; f < nvec; f += 64)` -- compiled to a branch per
  // vector and seven read / wait / store / taken-branch rounds at N = 8.)
  auto fill_tile = [&](const V4 (&oj)[N]) {
    if (agent_ok) {
#pragma unroll
      for (int k = 0; k < NO; ++k) {
        const bool below = k < i;  // other agent k (below i) or k + 1
        V4 o;
        o.x = below ? oj[k].x : oj[k + 1].x;
        o.y = below ? oj[k].y : oj[k + 1].y;
        o.z = below ? oj[k].z : oj[k + 1].z;
        o.w = below ? oj[k].w : oj[k + 1].w;
        lds4[sub * VPE + i * NO + k] = sub4<R, V4>(o, si);
      }
    }
    wave_lds_sync();
  };
  auto store_obs = [&](const V4 (&oj)[N], void *dst) {
    fill_tile(oj);
    constexpr int STEPS = (EPW * VPE + 63) / 64;
    V4 row[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; ++q) row[q] = lds4[(q * 64 + lane) < EPW * VPE ? q * 64 + lane : 0];
#pragma unroll
    for (int q = 0; q < STEPS; ++q) {
      const int f = q * 64 + lane;
      if (f < nvec) store_obs_vec<SP>(at32<V4>(dst, (e0 * VPE + f) * (uint32_t)sizeof(V4)), row[q]);
    }
    wave_lds_sync();
  };

  CM3_STAMP(1, true);
  CM3_SPAN_MARK(0, true);   // loads back
  const int n_ticks = FUSED ? p.n_ticks : 1;
#pragma unroll 1
  for (int t = 0; t < n_ticks; ++t) {
    int32_t *actions_t = tick_ptr(p.actions, p.st_actions, t);
    int act;
    if (gen) {  // train_onpolicy.py:305-307
      act = rand5(action_word(aword, episode, (uint32_t)steps));   // stage 2 of the action stream
      if (mine) store_small<SP>(at32<int32_t>(actions_t, (e * N + i) * 4u), act);
    } else {
      act = *at32<const int32_t>(actions_t, (ec * N + i) * 4u);
    }

    CM3_STAMP(2, false);
    CM3_SPAN_MARK(1, false);  // action drawn
    // ---- action force + contact forces of agent i, other agents in ascending order (core.py:143-155) ------------
    R ux = R(0), uy = R(0);
    if (act == 1) ux = R(-1);
    if (act == 2) ux = R(+1);
    if (act == 3) uy = R(-1);
    if (act == 4) uy = R(+1);
    R Fx = ux * R(5.0) + R(0.0), Fy = uy * R(5.0) + R(0.0);
    // Beyond Contact<R>::kSkip the force is exactly (+-)0 and leaves Fx / Fy bit-unchanged (see contact_force), so only
    // the agents within reach are visited, in ascending order; the wave iterates max-over-lanes(#neighbours in reach)
    // times (typically 0-2) instead of running the transcendental chain N-1 times because SOME lane needs it.
    // The scan runs on squared distances against the exact threshold (thresholds.h): no square root outside the loop.
    R dxs[N], dys[N], d2s[N];
    unsigned near_mask = 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      dxs[j] = si.z - __shfl(si.z, base + j, 64);
      dys[j] = si.w - __shfl(si.w, base + j, 64);
      d2s[j] = dxs[j] * dxs[j] + dys[j] * dys[j];
      near_mask |= (unsigned)((j != i) & !(d2s[j] >= Thresh<R>::kSkip2)) << j;  // no branch
    }
    while (__any(near_mask != 0u)) {
      const int jn = near_mask ? (__ffs((int)near_mask) - 1) : 0;
      R dx = dxs[0], dy = dys[0], d2 = d2s[0];
#pragma unroll
      for (int j = 1; j < N; ++j) {
        dx = (jn == j) ? dxs[j] : dx;
        dy = (jn == j) ? dys[j] : dy;
        d2 = (jn == j) ? d2s[j] : d2;
      }
      if (near_mask) {
        R f_x, f_y;
        contact_force_near<R>(dx, dy, d2, f_x, f_y);  // within reach: the one square root of this neighbour
        Fx = f_x + Fx;
        Fy = f_y + Fy;
      }
      near_mask &= near_mask - 1u;
    }

    CM3_STAMP(3, false);
    CM3_SPAN_MARK(2, false);  // neighbour scan + contact forces
    // ---- integrate agent i (core.py:158-169) ---------------------------------------------------------------------
    si.x = si.x * kKeep;
    si.y = si.y * kKeep;
    si.x = si.x + (Fx / R(1.0)) * kDt;
    si.y = si.y + (Fy / R(1.0)) * kDt;
    si.z = si.z + si.x * kDt;
    si.w = si.w + si.y * kDt;
    steps += 1;
    V4 oj[N];  // post-step state of every agent of this env
#pragma unroll
    for (int j = 0; j < N; ++j) {
      oj[j].x = __shfl(si.x, base + j, 64);
      oj[j].y = __shfl(si.y, base + j, 64);
      oj[j].z = __shfl(si.z, base + j, 64);
      oj[j].w = __shfl(si.w, base + j, 64);
    }

    // state / goals stores of this tick (issued last: moving the write-through stores of this kernel ahead of the reward work, as
    // the two-lanes kernel does for small batches, LOST 2-6 % at the batch sizes this kernel serves: profiles/r03_early_wt_stores.txt)
    auto emit_state = [&](bool fresh_goals) {
      if (mine) {
        *at32<V4>(tick_ptr(p.state_out, p.st_state, t), (row_i + e) * (uint32_t)sizeof(V4)) = si;
        if (p.goals_out != p.goals_in || fresh_goals)
          *at32<V2>(tick_ptr(p.goals_out, p.st_goals, t), (row_i + e) * (uint32_t)sizeof(V2)) = gl;
        if constexpr (LIVE)   // live-state rollout (a compile-time variant: the others carry none of it): the slot gets a copy
          store_obs_vec<SP>(at32<V4>(p.state_copy, (row_i + e) * (uint32_t)sizeof(V4)), si);
        if constexpr (!FUSED) {   // goal slots are SPARSE wherever the goals live in place (live state, or goals_live alone)
          if (fresh_goals && p.goals_copy) *at32<V2>(p.goals_copy, (row_i + e) * (uint32_t)sizeof(V2)) = gl;
        }
      }
    };
    CM3_STAMP(4, true);
    CM3_SPAN_MARK(3, false);  // integrated, post-step states exchanged
    // ---- reward / reached / collisions (multi-goal_spread.py:114-143) ----------------------------------------
    R rew;
    bool reached;
    {
      const R dx = si.z - gl.x, dy = si.w - gl.y;
      const R d2 = dx * dx + dy * dy;
      rew = R(0) - Math<R>::sqrt(d2);
      reached = d2 < Thresh<R>::kReach2;  // rew >= -0.05 (thresholds.h)
    }
    int c_i = 0;
#pragma unroll
    for (int j = 0; j < N; ++j)  // is_collision(a = j, agent = i)
      c_i += (int)((j != i) & is_collision<R>(oj[j].z - si.z, oj[j].w - si.w));  // no branch
    c_i = agent_ok ? c_i : 0;
#pragma unroll
    for (int c = 0; c < NO; ++c)
      if (c < c_i) rew = rew - R(1);
    int c_env = c_i;  // every ordered visit counts (:135-137)
    if constexpr (G == 8) {
      c_env += dpp_i32<kDppXor1>(c_env);
      c_env += dpp_i32<kDppXor2>(c_env);
      c_env += dpp_i32<kDppHalfMirror>(c_env);  // the quads are uniform by now: lane 7 - i holds the other quad's sum
    } else if constexpr (G == 4) {
      c_env += dpp_i32<kDppXor1>(c_env);
      c_env += dpp_i32<kDppXor2>(c_env);
    } else {
#pragma unroll
      for (int off = G / 2; off > 0; off >>= 1) c_env += __shfl_xor(c_env, off, 64);
    }
    collisions += c_env;
    const unsigned long long rb = __ballot(reached && agent_ok);
    const unsigned long long rgrp = (G == 64) ? rb : ((rb >> base) & ((1ull << (G & 63)) - 1ull));
    const bool all_reached = __popcll(rgrp) == N;
    R reward;
    if constexpr (N == 8 && sizeof(R) == 4) {
      // NumPy's pairwise tree for 8 values (sum_agents): ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)), as three DPP adds -- IEEE addition
      // commutes, so every lane of the env ends with exactly that value
      R tsum = rew + dpp_f32<kDppXor1>(rew);
      tsum = tsum + dpp_f32<kDppXor2>(tsum);
      const R other = dpp_f32<kDppHalfMirror>(tsum);
      reward = gi < 4 ? tsum + other : other + tsum;
    } else {
      R rews[N];
#pragma unroll
      for (int a = 0; a < N; ++a) rews[a] = __shfl(rew, base + a, 64);
      reward = sum_agents<R, N>(rews);
    }
    const bool done = (steps == h_max_steps) || all_reached;

    if (mine) store_small<SP>(at32<R>(tick_ptr(p.reward_n, p.st_reward_n, t), (e * N + i) * (uint32_t)sizeof(R)), rew);
    if (head) {
      *at32<R>(tick_ptr(p.reward, p.st_reward, t), e * (uint32_t)sizeof(R)) = reward;
      *at32<uint8_t>(tick_ptr(p.done, p.st_done, t), e) = done ? 1 : 0;
      if (p.collisions_tick) *at32<int32_t>(tick_ptr(p.collisions_tick, p.st_coll, t), e * 4u) = collisions;
    }

    CM3_STAMP(5, false);
    CM3_SPAN_MARK(4, false);  // rewards / done stored
    // ---- same-launch re-initialisation -------------------------------------------------------------------------
    bool was_reset = false;
    if (auto_reset) {
      void *term_state = tick_ptr(p.term_state, p.st_term_state, t);
      void *term_obs = tick_ptr(p.term_obs_others, p.st_term_obs, t);
      if (__any(done)) {  // wave-uniform: the tile store needs every lane
        if (term_state && done && mine) *at32<V4>(term_state, (row_i + e) * (uint32_t)sizeof(V4)) = si;
        if (term_obs) {
          // terminal observations of the finished envs; rows of unfinished envs in the tile are not written out
          fill_tile(oj);
          const unsigned long long done_bits = __ballot(done);
          for (int f = lane; f < nvec; f += 64) {
            const int row = f / VPE;
            if ((done_bits >> (row * G)) & 1ull) *at32<V4>(term_obs, (e0 * VPE + f) * (uint32_t)sizeof(V4)) = lds4[f];
          }
          wave_lds_sync();
        }
        if (done) {
          episode += 1;
          const uint64_t rseed = reset_seed(p);
          const bool rnd = episode_is_random(p, genv, episode, rseed);
          init_agent<R, N, true>(p, genv, episode, rnd, i, si, gl, preset_table(), &rseed);
          steps = 0;
          collisions = 0;
          was_reset = true;
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {  // fresh episodes: the observation is that of the reset state
          oj[j].x = __shfl(si.x, base + j, 64);
          oj[j].y = __shfl(si.y, base + j, 64);
          oj[j].z = __shfl(si.z, base + j, 64);
          oj[j].w = __shfl(si.w, base + j, 64);
        }
      }
    }

    CM3_STAMP(6, false);
    CM3_SPAN_MARK(5, false);  // reset handled
    // ---- per-tick stores ------------------------------------------------------------------------------------------
    emit_state(was_reset);
    store_obs(oj, tick_ptr(p.obs_others, p.st_obs, t));  // observation (multi-goal_spread.py:145-154)
  }

  CM3_SPAN_MARK(6, false);  // state + observation stores issued
  CM3_STAMP(7, false);
  // ---- live counters, once per launch -------------------------------------------------------------------------------
  if (head) {
    int2 m;
    m.x = steps;
    m.y = collisions;
    *at32<int2>(p.meta_out, e * 8u) = m;
    if (episode != episode_in) *at32<int32_t>(p.episode, e * 4u) = (int32_t)episode;
  }
  CM3_STAMP(8, true);
  CM3_SPAN_OUT(p.span);
}

// ---- lane-per-agent mapping with TWO LANES PER AGENT (N = 8, float32, per-tick launches; round 3) --------------------------------
// At C5 the one-lane-per-agent kernel above runs one wave per SIMD for ~7500 cycles, most of them in work that is proportional to
// the number of OTHER agents a lane walks: the neighbour scan + contact forces (1770 cycles), the exchange of the post-step states
// (530), the collision tests (1000 with the reward) and the observation tile (1300).  Here an env owns 16 lanes = one DPP row; lane
// (i, h) = agent i, half h handles the other agents j in [4h, 4h + 4): half the scan, half the contact passes, half the state
// exchange, half of the agent's observation row -- and twice the waves, i.e. two per SIMD, where a lone wave leaves three of four
// issue slots empty (tools/probes/issue_probe.hip).  What both lanes of an agent do redundantly (action draw, integration, reward
// distance, reset) is lockstep work.  Bit-identical to the other mappings:
//   * contact forces are summed in the reference's order (other agents ascending, core.py:145-155) by two chained passes -- every
//     lane adds its four contributions (exactly +0 for agents beyond reach or itself, which leaves a sum bit-unchanged: a partial
//     sum is never -0) to a start value; pass 1 starts from the action force, pass 2 from the EVEN lane's pass-1 result (DPP), so
//     the odd lane ends with f7 + (f6 + (f5 + (f4 + (f3 + (f2 + (f1 + (f0 + F_action))))))) and hands it back to the even lane;
//   * collision counts add up over the pair, the env's count and NumPy's 8-value pairwise reward tree are DPP row reductions.
template <int WAVES, int SP = kSpPlain, bool LIVE = false, bool EARLY = false, int TU = CM3_PARTICLE_TU>
__global__ void __launch_bounds__(WAVES * 64)
    k_particle_step_agents2(const void *h_state_in, const void *h_goals_in, const int32_t *h_meta_in, const int32_t *h_episode,
                            const int h_E, const uint32_t h_flags, const int h_E0, const int h_EN, const int h_max_steps,
                            const int32_t *h_actions, const ParticleParams p) {
  using R = float;
  using V4 = float4;
  using V2 = float2;
  constexpr int N = 8, NO = 7, G = 16, EPW = 4, VPE = N * NO, NH = 4;  // NH: other agents per half
  __shared__ __attribute__((aligned(32))) R lds_all[WAVES][EPW * VPE * 4];
  CM3_SPAN_IN();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gi = lane & (G - 1), sub = lane / G, base = lane - gi;
  const int i = gi >> 1, h = gi & 1;
  const uint32_t E = (uint32_t)h_E, EN = (uint32_t)h_EN;
  const uint32_t e0 = (uint32_t)h_E0 + (cm3_xcd_block(h_flags) * WAVES + wave) * EPW;
  const uint32_t e = e0 + sub;
  const bool env_ok = e < EN;
  const uint32_t ec = env_ok ? e : EN - 1;
  const bool mine = env_ok && h == 0;    // the even lane of an agent's pair does the per-agent stores
  const bool head = env_ok && gi == 0;   // one lane per env does the per-env stores
  V4 *lds4 = reinterpret_cast<V4 *>(&lds_all[wave][0]);

  const uint32_t row_i = (uint32_t)i * E;
  V4 si = *at32<const V4>(h_state_in, (row_i + ec) * (uint32_t)sizeof(V4));
  V2 gl = *at32<const V2>(h_goals_in, (row_i + ec) * (uint32_t)sizeof(V2));
  const int2 meta = *at32<const int2>(h_meta_in, ec * 8u);
  int steps = meta.x, collisions = meta.y;
  const bool auto_reset = (h_flags & CM3_FLAG_AUTO_RESET) != 0;
  const bool gen = (h_flags & CM3_FLAG_GEN_ACTIONS) != 0;
  uint32_t episode = 0;
  if (gen || (h_flags & CM3_FLAG_AUTO_RESET)) episode = (uint32_t)*at32<const int32_t>(h_episode, ec * 4u);
  const uint32_t episode_in = episode;
  CM3_FETCH_EARLY(p.state_out, p.goals_out, p.goals_in, p.collisions_tick, p.reward_n, p.reward, p.done, p.obs_others, p.meta_out);
  if constexpr (LIVE) CM3_FETCH_EARLY(p.state_copy, p.goals_copy);
  const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
  // stage 1 of the action stream (philox.h): this agent's word of its env's Philox block, computed while the loads are in flight
  uint32_t aword = 0;
  if (gen) aword = pick_word(action_block(p.seed, genv, (uint32_t)(i >> 2)), i & 3);
  const R kDt = R(0.1), kKeep = R(1 - 0.25);
  long envs_here = (long)EN - (long)e0;
  envs_here = envs_here < 0 ? 0 : (envs_here > EPW ? EPW : envs_here);
  const int nvec = (int)envs_here * VPE;
  const int src0 = base + 8 * h;  // lane of agent 4h (its even lane); agent 4h + q sits two lanes further per q

  // this half's share of agent i's observation row -> the wave's tile.  The slot of agent i itself (when it falls into this half)
  // repeats the next one: the same value written twice to the same place, no predicate.
  auto fill_tile = [&](const V4 (&oj)[NH]) {
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      const int j = 4 * h + q;
      const bool self = j == i;
      const int jj = self ? 4 * h + ((q + 1) & 3) : j;
      V4 o;
      o.x = self ? oj[(q + 1) & 3].x : oj[q].x;
      o.y = self ? oj[(q + 1) & 3].y : oj[q].y;
      o.z = self ? oj[(q + 1) & 3].z : oj[q].z;
      o.w = self ? oj[(q + 1) & 3].w : oj[q].w;
      const int k = jj < i ? jj : jj - 1;
      lds4[sub * VPE + i * NO + k] = sub4<R, V4>(o, si);
    }
    wave_lds_sync();
  };
  auto gather = [&](V4 (&oj)[NH]) {  // post-step (or fresh) state of this half's four other agents
#pragma unroll
    for (int q = 0; q < NH; ++q) {
      oj[q].x = __shfl(si.x, src0 + 2 * q, 64);
      oj[q].y = __shfl(si.y, src0 + 2 * q, 64);
      oj[q].z = __shfl(si.z, src0 + 2 * q, 64);
      oj[q].w = __shfl(si.w, src0 + 2 * q, 64);
    }
  };
  CM3_SPAN_MARK(0, true);   // loads back

  int32_t *actions_t = p.actions;
  int act;
  if (gen) {  // train_onpolicy.py:305-307
    act = rand5(action_word(aword, episode, (uint32_t)steps));   // stage 2 of the action stream
    if (mine) store_small<SP>(at32<int32_t>(actions_t, (e * N + i) * 4u), act);
  } else {
    act = *at32<const int32_t>(actions_t, (ec * N + i) * 4u);
  }
  CM3_SPAN_MARK(1, false);  // action drawn
  R ux = R(0), uy = R(0);
  if (act == 1) ux = R(-1);
  if (act == 2) ux = R(+1);
  if (act == 3) uy = R(-1);
  if (act == 4) uy = R(+1);
  const R Fax = ux * R(5.0) + R(0.0), Fay = uy * R(5.0) + R(0.0);
  // ---- this half's neighbours within reach (exact test on the squared distance, thresholds.h) and their contact forces ------------
  R dxs[NH], dys[NH], d2s[NH], fx[NH], fy[NH];
  unsigned near_mask = 0;
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    dxs[q] = si.z - __shfl(si.z, src0 + 2 * q, 64);
    dys[q] = si.w - __shfl(si.w, src0 + 2 * q, 64);
    d2s[q] = dxs[q] * dxs[q] + dys[q] * dys[q];
    near_mask |= (unsigned)((4 * h + q != i) & !(d2s[q] >= Thresh<R>::kSkip2)) << q;
    fx[q] = R(0);
    fy[q] = R(0);
  }
  while (__any(near_mask != 0u)) {
    const int qn = near_mask ? (__ffs((int)near_mask) - 1) : 0;
    R dx = dxs[0], dy = dys[0], d2 = d2s[0];
#pragma unroll
    for (int q = 1; q < NH; ++q) {
      dx = (qn == q) ? dxs[q] : dx;
      dy = (qn == q) ? dys[q] : dy;
      d2 = (qn == q) ? d2s[q] : d2;
    }
    if (near_mask) {
      R f_x, f_y;
      contact_force_near<R>(dx, dy, d2, f_x, f_y);
#pragma unroll
      for (int q = 0; q < NH; ++q) {
        fx[q] = (qn == q) ? f_x : fx[q];
        fy[q] = (qn == q) ? f_y : fy[q];
      }
    }
    near_mask &= near_mask - 1u;
  }
  // two chained passes in the reference's order (see the head of the kernel)
  R Px = Fax, Py = Fay;
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    Px = fx[q] + Px;
    Py = fy[q] + Py;
  }
  Px = dpp_f32<0xA0>(Px);  // quad_perm:[0,0,2,2]: the even lane's sum over agents 0..3
  Py = dpp_f32<0xA0>(Py);
#pragma unroll
  for (int q = 0; q < NH; ++q) {
    Px = fx[q] + Px;
    Py = fy[q] + Py;
  }
  // quad_perm:[1,1,3,3]: the odd lane holds the complete sum.  The moves run with ALL lanes enabled, before the select: a DPP
  // move inside the h == 0 branch would read lanes that are switched off there (and get 0)
  const R Qx = dpp_f32<0xF5>(Px), Qy = dpp_f32<0xF5>(Py);
  const R Fx = h ? Px : Qx;
  const R Fy = h ? Py : Qy;
  CM3_SPAN_MARK(2, false);  // neighbour scan + contact forces
  // ---- integrate agent i (core.py:158-169) ---------------------------------------------------------------------------------------
  si.x = si.x * kKeep;
  si.y = si.y * kKeep;
  si.x = si.x + (Fx / R(1.0)) * kDt;
  si.y = si.y + (Fy / R(1.0)) * kDt;
  si.z = si.z + si.x * kDt;
  si.w = si.w + si.y * kDt;
  steps += 1;
  V4 oj[NH];
  gather(oj);
  // observation (multi-goal_spread.py:145-154): the tile goes out as contiguous 16-byte-per-lane rows.  EARLY (write-through
  // launches of up to kAgents2EarlyMaxEnvs envs): NOW, before the rewards are worked out -- write-through stores are acknowledged
  // only by memory (~800 cycles) and the wave cannot end before they are; issued here their round trip overlaps the ~1000 cycles
  // of reward / reset work, issued last the wave sat waiting for it (marked build: 839 cycles from 'stores issued' to 'drained').
  // Envs that reset in this tick get their rows written once more below, with the observation of the reset state (same lanes, same
  // addresses, program order).  With more waves per SIMD the wait is hidden anyway and the early stores only get in the way
  // (32768 envs: +2 %), hence the bound (profiles/r03_early_wt_stores.txt).
  auto emit_obs = [&](const V4 (&o)[NH]) {
    fill_tile(o);
    constexpr int STEPS = (EPW * VPE + 63) / 64;
    V4 row[STEPS];
#pragma unroll
    for (int q = 0; q < STEPS; ++q) row[q] = lds4[(q * 64 + lane) < EPW * VPE ? q * 64 + lane : 0];
#pragma unroll
    for (int q = 0; q < STEPS; ++q) {
      const int f = q * 64 + lane;
      if (f < nvec) store_obs_vec<SP>(at32<V4>(p.obs_others, (e0 * VPE + f) * (uint32_t)sizeof(V4)), row[q]);
    }
    wave_lds_sync();   // the tile may be refilled (terminal capture, a second emit)
  };
  if constexpr (EARLY) emit_obs(oj);
  // ... and the agent's state / goals (the slot copy of a live rollout is a write-through store as well), for the same reason
  auto emit_state = [&](bool fresh_goals) {
    if (mine) {
      *at32<V4>(p.state_out, (row_i + e) * (uint32_t)sizeof(V4)) = si;
      if (p.goals_out != p.goals_in || fresh_goals) *at32<V2>(p.goals_out, (row_i + e) * (uint32_t)sizeof(V2)) = gl;
      if constexpr (LIVE) store_obs_vec<SP>(at32<V4>(p.state_copy, (row_i + e) * (uint32_t)sizeof(V4)), si);
      if (fresh_goals && p.goals_copy) *at32<V2>(p.goals_copy, (row_i + e) * (uint32_t)sizeof(V2)) = gl;   // sparse goal slots
    }
  };
  if constexpr (EARLY) emit_state(false);
  CM3_SPAN_MARK(3, false);  // integrated, post-step states exchanged (EARLY: state + observation stores issued)
  // ---- reward / reached / collisions (multi-goal_spread.py:114-143) -----------------------------------------------------------------
  R rew;
  bool reached;
  {
    const R dx = si.z - gl.x, dy = si.w - gl.y;
    const R d2 = dx * dx + dy * dy;
    rew = R(0) - Math<R>::sqrt(d2);
    reached = d2 < Thresh<R>::kReach2;
  }
  int c_half = 0;
#pragma unroll
  for (int q = 0; q < NH; ++q) c_half += (int)((4 * h + q != i) & is_collision<R>(oj[q].z - si.z, oj[q].w - si.w));
  const int c_i = c_half + dpp_i32<kDppXor1>(c_half);
#pragma unroll
  for (int c = 0; c < NO; ++c)
    if (c < c_i) rew = rew - R(1);
  int c_env = c_half;  // every ordered visit counts (:135-137): the sum over the env's 16 lanes
  c_env += dpp_i32<kDppXor1>(c_env);
  c_env += dpp_i32<kDppXor2>(c_env);
  c_env += dpp_i32<kDppHalfMirror>(c_env);
  c_env += dpp_i32<0x140>(c_env);  // row_mirror: lane i <-> 15 - i (the two halves of the row are uniform by now)
  collisions += c_env;
  const unsigned long long rb = __ballot(reached && h == 0);
  const bool all_reached = __popcll((rb >> base) & 0xffffull) == N;
  // NumPy's pairwise tree for 8 values: ((v0+v1)+(v2+v3)) + ((v4+v5)+(v6+v7)); agents i, i^1 are two lanes apart
  R tsum = rew + dpp_f32<kDppXor2>(rew);
  tsum = tsum + dpp_f32<kDppHalfMirror>(tsum);
  const R other = dpp_f32<0x140>(tsum);
  const R reward = gi < 8 ? tsum + other : other + tsum;
  const bool done = (steps == h_max_steps) || all_reached;
  if (mine) store_small<SP>(at32<R>(p.reward_n, (e * N + i) * (uint32_t)sizeof(R)), rew);
  if (head) {
    *at32<R>(p.reward, e * (uint32_t)sizeof(R)) = reward;
    *at32<uint8_t>(p.done, e) = done ? 1 : 0;
    if (p.collisions_tick) *at32<int32_t>(p.collisions_tick, e * 4u) = collisions;
  }
  CM3_SPAN_MARK(4, false);  // rewards / done stored
  // ---- same-launch re-initialisation ---------------------------------------------------------------------------------------------
  bool was_reset = false;
  if (auto_reset) {
    if (__any(done)) {  // wave-uniform: the tile store needs every lane
      if (p.term_state && done && mine) *at32<V4>(p.term_state, (row_i + e) * (uint32_t)sizeof(V4)) = si;
      if (p.term_obs_others) {
        fill_tile(oj);
        const unsigned long long done_bits = __ballot(done);
        for (int f = lane; f < nvec; f += 64) {
          const int row = f / VPE;
          if ((done_bits >> (row * G)) & 1ull) *at32<V4>(p.term_obs_others, (e0 * VPE + f) * (uint32_t)sizeof(V4)) = lds4[f];
        }
        wave_lds_sync();
      }
      if (done) {
        episode += 1;
        const uint64_t rseed = reset_seed(p);
        const bool rnd = episode_is_random(p, genv, episode, rseed);
        init_agent<R, N, true>(p, genv, episode, rnd, i, si, gl, preset_table(), &rseed);
        steps = 0;
        collisions = 0;
        was_reset = true;
      }
      gather(oj);  // fresh episodes: the observation is that of the reset state
      if constexpr (EARLY) {
        emit_obs(oj);
        if (was_reset) emit_state(true);
      }
    }
  }
  CM3_SPAN_MARK(5, false);  // reset handled
  if constexpr (!EARLY) {  // the usual place: last
    emit_state(was_reset);
    emit_obs(oj);
  }
  CM3_SPAN_MARK(6, false);  // (the state and the observation went out right after the step, see emit_obs / emit_state)
  if (head) {
    int2 m;
    m.x = steps;
    m.y = collisions;
    *at32<int2>(p.meta_out, e * 8u) = m;
    if (episode != episode_in) *at32<int32_t>(p.episode, e * 4u) = (int32_t)episode;
  }
  CM3_SPAN_OUT(p.span);
}

// ---- reset kernel (environment.py:125-149) ------------------------------------------------------------
template <typename R, int N, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_particle_reset(const ParticleParams p) {
  using V4 = typename Vec<R>::v4;
  using V2 = typename Vec<R>::v2;
  using G = ObsGeom<R, N>;
  __shared__ __attribute__((aligned(32))) R lds_all[WAVES][G::LDS_REALS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t e0 = (size_t)p.E0 + ((size_t)blockIdx.x * WAVES + wave) * 64;
  const size_t e = e0 + lane;
  const bool active = e < (size_t)p.EN;
  const size_t ec = active ? e : (size_t)p.EN - 1;
  const size_t E = (size_t)p.E;
  const bool sel = p.reset_mask ? (p.reset_mask[ec] != 0) : true;
  V4 s[N];
  V2 g[N];
  V4 *sout4 = reinterpret_cast<V4 *>(p.state_out);
  V2 *gout2 = reinterpret_cast<V2 *>(p.goals_out);
  if (sel) {
    const uint32_t episode = (uint32_t)p.episode[ec] + 1u;
    init_episode<R, N>(p, (uint64_t)(p.env_id_base + (int64_t)ec), episode, s, g);
    if (active) {
      p.episode[e] = (int32_t)episode;
#pragma unroll
      for (int i = 0; i < N; ++i) sout4[(size_t)i * E + e] = s[i];
#pragma unroll
      for (int i = 0; i < N; ++i) gout2[(size_t)i * E + e] = g[i];
      int2 m;
      m.x = 0;
      m.y = 0;
      reinterpret_cast<int2 *>(p.meta_out)[e] = m;
    }
  } else {
    // unselected envs keep their state; it is re-read so the observation rows of the whole tile are rewritten
#pragma unroll
    for (int i = 0; i < N; ++i) s[i] = sout4[(size_t)i * E + ec];
  }
  store_obs_others_staged<R, N>(s, &lds_all[wave][0], lane, e0, p.EN, reinterpret_cast<R *>(p.obs_others));
}

// ---- observe kernel (multi-goal_spread.py:145-154 after a state injection) -----------------------------
template <typename R, int N, int WAVES>
__global__ void __launch_bounds__(WAVES * 64) k_particle_observe(const ParticleParams p) {
  using V4 = typename Vec<R>::v4;
  using G = ObsGeom<R, N>;
  __shared__ __attribute__((aligned(32))) R lds_all[WAVES][G::LDS_REALS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t e0 = (size_t)p.E0 + ((size_t)blockIdx.x * WAVES + wave) * 64;
  const size_t e = e0 + lane;
  const size_t ec = e < (size_t)p.EN ? e : (size_t)p.EN - 1;
  V4 s[N];
  const V4 *sin4 = reinterpret_cast<const V4 *>(p.state_in);
#pragma unroll
  for (int i = 0; i < N; ++i) s[i] = sin4[(size_t)i * p.E + ec];
  store_obs_others_staged<R, N>(s, &lds_all[wave][0], lane, e0, p.EN, reinterpret_cast<R *>(p.obs_others));
}

// ---- host side -------------------------------------------------------------------------------------------
enum ParticleOp { kStep = 0, kReset = 1, kObserve = 2 };

static int fill_params(const cm3_particle_desc *d, const cm3_particle_bufs *b, ParticleOp op, const uint8_t *mask,
                       ParticleParams &p) {
  CM3_REQUIRE(d && b, "null desc/bufs");
  CM3_REQUIRE(d->n_envs > 0, "n_envs must be positive (got %d)", d->n_envs);
  CM3_REQUIRE(d->n_agents >= 1 && d->n_agents <= CM3_MAX_AGENTS, "n_agents must be in 1..%d (got %d)",
              CM3_MAX_AGENTS, d->n_agents);
  CM3_REQUIRE(d->max_steps >= 1, "max_steps must be >= 1");
  CM3_REQUIRE((d->flags & ~(CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS | CM3_FLAG_KERNEL_LANE_PER_ENV |
                            CM3_FLAG_KERNEL_LANE_PER_PAIR | CM3_FLAG_KERNEL_LANE_PER_AGENT | CM3_FLAG_FUSED_TICKS)) == 0,
              "unknown flag bits 0x%x", d->flags);
  {
    const uint32_t k = d->flags & (CM3_FLAG_KERNEL_LANE_PER_ENV | CM3_FLAG_KERNEL_LANE_PER_PAIR | CM3_FLAG_KERNEL_LANE_PER_AGENT);
    CM3_REQUIRE((k & (k - 1)) == 0, "more than one kernel-mapping flag set");
  }
  CM3_REQUIRE(!((d->flags & (CM3_FLAG_KERNEL_LANE_PER_PAIR | CM3_FLAG_KERNEL_LANE_PER_AGENT)) && d->n_agents < 2),
              "the lane-per-pair / lane-per-agent kernels need n_agents >= 2");
  CM3_REQUIRE(b->obs_others, "obs_others is required");
  if (op == kStep) {
    CM3_REQUIRE(b->state_in && b->state_out && b->goals_in && b->goals_out && b->meta_in && b->meta_out,
                "step: state/goals/meta pointers are required");
    CM3_REQUIRE(b->actions && b->reward_n && b->reward && b->done, "step: actions/reward_n/reward/done are required");
    if (d->flags & (CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS))
      CM3_REQUIRE(b->episode, "episode counter is required with AUTO_RESET / GEN_ACTIONS");
  } else if (op == kReset) {
    CM3_REQUIRE(b->state_out && b->goals_out && b->meta_out && b->episode, "reset: state_out/goals_out/meta_out/episode");
  } else {
    CM3_REQUIRE(b->state_in, "observe: state_in is required");
  }
  CM3_REQUIRE(d->env_offset >= 0 && d->env_count >= 0 && (int64_t)d->env_offset + d->env_count <= d->n_envs &&
                  (d->env_count > 0 || d->env_offset == 0),
              "env_offset / env_count must select a range inside [0, n_envs) (got %d / %d of %d)", d->env_offset,
              d->env_count, d->n_envs);
  memset(&p, 0, sizeof(p));
  p.n_ticks = 1;
  p.E = d->n_envs;
  p.E0 = d->env_offset;
  p.EN = d->env_count > 0 ? d->env_offset + d->env_count : d->n_envs;
  p.max_steps = d->max_steps;
  p.flags = d->flags & ~CM3_FLAG_FUSED_TICKS;
  p.env_id_base = d->env_id_base;
  p.seed = d->seed;
  p.prob_random = d->prob_random;
  p.initial_std = d->initial_std;
  for (int i = 0; i < CM3_MAX_AGENTS; ++i) {
    p.ax[i] = d->agents_x[i];
    p.ay[i] = d->agents_y[i];
    p.lx[i] = d->landmarks_x[i];
    p.ly[i] = d->landmarks_y[i];
  }
  p.state_in = b->state_in;
  p.state_out = b->state_out;
  p.goals_in = b->goals_in;
  p.goals_out = b->goals_out;
  p.meta_in = b->meta_in;
  p.meta_out = b->meta_out;
  p.episode = b->episode;
  p.actions = b->actions;
  p.obs_others = b->obs_others;
  p.reward_n = b->reward_n;
  p.reward = b->reward;
  p.done = b->done;
  p.term_state = b->term_state;
  p.term_obs_others = b->term_obs_others;
  p.collisions_tick = b->collisions_tick;
  p.state_copy = p.goals_copy = nullptr;
  p.reset_mask = mask;
  if (op == kStep) CM3_SPAN_SET(p);
  return CM3_OK;
}

template <typename R, int N, int WAVES>
static int launch_one(const ParticleParams &p, ParticleOp op, hipStream_t stream) {
  const unsigned per_block = WAVES * 64;
  const unsigned blocks = (unsigned)(((size_t)(p.EN - p.E0) + per_block - 1) / per_block);
  switch (op) {
    case kStep: {
      constexpr int kNt = sizeof(R) == 4 ? kSpNt : kSpPlain, kWt = sizeof(R) == 4 ? kSpWt : kSpPlain;
      const bool nt = sizeof(R) == 4 && (p.flags & kFlagObsStoreNt);   // streaming-size trajectory (obs_store_nt)
      // a launch that writes >= kWtMinObsBytes of observation rows writes them through (kSpWt; per-tick launches only)
      const bool wt = sizeof(R) == 4 && (size_t)(p.EN - p.E0) * ObsGeom<R, N>::REC * sizeof(R) >= kWtMinObsBytes && !p.state_copy;
      note_variant("k_particle_step", (int)sizeof(R), N, WAVES, p.n_ticks > 1,
                   p.n_ticks > 1 ? (nt ? kNt : kSpPlain) : (wt ? kWt : (nt ? kNt : kSpPlain)), 0, CM3_PARTICLE_TU);
      if (p.n_ticks > 1) {
        if (nt) hipLaunchKernelGGL((k_particle_step<R, N, WAVES, true, kNt>), dim3(blocks), dim3(per_block), 0, stream, p);
        else hipLaunchKernelGGL((k_particle_step<R, N, WAVES, true>), dim3(blocks), dim3(per_block), 0, stream, p);
      } else {
        if (wt) hipLaunchKernelGGL((k_particle_step<R, N, WAVES, false, kWt>), dim3(blocks), dim3(per_block), 0, stream, p);
        else if (nt) hipLaunchKernelGGL((k_particle_step<R, N, WAVES, false, kNt>), dim3(blocks), dim3(per_block), 0, stream, p);
        else hipLaunchKernelGGL((k_particle_step<R, N, WAVES, false>), dim3(blocks), dim3(per_block), 0, stream, p);
      }
      break;
    }
    case kReset:
      note_variant("k_particle_reset", (int)sizeof(R), N, WAVES, 0, kSpPlain, 0, CM3_PARTICLE_TU);
      hipLaunchKernelGGL((k_particle_reset<R, N, WAVES>), dim3(blocks), dim3(per_block), 0, stream, p);
      break;
    case kObserve:
      note_variant("k_particle_observe", (int)sizeof(R), N, WAVES, 0, kSpPlain, 0, CM3_PARTICLE_TU);
      hipLaunchKernelGGL((k_particle_observe<R, N, WAVES>), dim3(blocks), dim3(per_block), 0, stream, p);
      break;
  }
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

template <typename R, int N, int WAVES> static int launch_pairs(const ParticleParams &p, hipStream_t stream) {
  if constexpr (N >= 2 && N <= 8) {   // (N (N - 1) pair lanes must fit a wave)
    const size_t envs_per_block = (size_t)WAVES * PairGeom<N>::EPW;
    const unsigned raw_blocks = (unsigned)(((size_t)(p.EN - p.E0) + envs_per_block - 1) / envs_per_block);
    const unsigned blocks = cm3_xcd_grid(raw_blocks);          // XCD-aware block order (common.h)
    const uint32_t xf = cm3_xcd_flags(raw_blocks);
    constexpr int kF32 = sizeof(R) == 4 ? kSpNt : kSpPlain;
    const bool nt = sizeof(R) == 4 && (p.flags & kFlagObsStoreNt);   // streaming-size trajectory (obs_store_nt)
#define CM3_LAUNCH_PAIRS(...)                                                                                               \
  hipLaunchKernelGGL((k_particle_step_pairs<R, N, WAVES, __VA_ARGS__>), dim3(blocks), dim3(WAVES * 64), 0, stream,          \
                     p.state_in, p.goals_in, p.meta_in, (const int32_t *)p.episode, p.E, p.flags | xf, p.E0, p.EN, p.max_steps,   \
                     (const int32_t *)p.actions, p)
    // the kernel indexes with 32-bit byte offsets: its largest per-tick array (obs_others) must stay below 4 GiB
    if ((size_t)p.E * PairGeom<N>::SLOTS * 4 * sizeof(R) >= ((size_t)1 << 32))
      return fail(CM3_ERR_INVALID, "the lane-per-pair kernel addresses at most 4 GiB per array: %d envs x %d agents is too large "
                  "(use the default kernel choice)", p.E, N);
    const bool live = p.state_copy != nullptr;   // cm3_particle_traj.state_live (per-tick launches only)
    note_variant("k_particle_step_pairs", (int)sizeof(R), N, WAVES, p.n_ticks > 1, nt ? kF32 : kSpPlain, p.n_ticks == 1 && live,
                 CM3_PARTICLE_TU);
    if (p.n_ticks > 1) {
      if (nt) CM3_LAUNCH_PAIRS(true, kF32);
      else CM3_LAUNCH_PAIRS(true, kSpPlain);
    } else if (live) {
      if (nt) CM3_LAUNCH_PAIRS(false, kF32, true);
      else CM3_LAUNCH_PAIRS(false, kSpPlain, true);
    } else {
      if (nt) CM3_LAUNCH_PAIRS(false, kF32);
      else CM3_LAUNCH_PAIRS(false, kSpPlain);
    }
#undef CM3_LAUNCH_PAIRS
    CM3_HIP_CHECK(hipGetLastError());
    return CM3_OK;
  } else {
    return fail(CM3_ERR_INVALID, "the lane-per-pair kernel needs n_agents in 2..8");
  }
}

// Largest batch for which the lane-per-pair mapping is preferred when no per-N entry says otherwise (see launch_n).
constexpr size_t kPairsMaxEnvs = (size_t)1 << 14;

template <typename R, int N, int WAVES> static int launch_agents(const ParticleParams &p, hipStream_t stream) {
  if constexpr (N >= 2) {
    const size_t envs_per_block = (size_t)WAVES * AgentGeom<N>::EPW;
    const unsigned raw_blocks = (unsigned)(((size_t)(p.EN - p.E0) + envs_per_block - 1) / envs_per_block);
    const unsigned blocks = cm3_xcd_grid(raw_blocks);          // XCD-aware block order (common.h)
    const uint32_t xf = cm3_xcd_flags(raw_blocks);
    const bool nt = sizeof(R) == 4 && (p.flags & kFlagObsStoreNt);   // streaming-size trajectory (obs_store_nt)
#define CM3_LAUNCH_AGENTS(...)                                                                                              \
  hipLaunchKernelGGL((k_particle_step_agents<R, N, WAVES, __VA_ARGS__>), dim3(blocks), dim3(WAVES * 64), 0, stream,         \
                     p.state_in, p.goals_in, p.meta_in, (const int32_t *)p.episode, p.E, p.flags | xf, p.E0, p.EN, p.max_steps,   \
                     (const int32_t *)p.actions, p)
    // 32-bit byte offsets inside the kernel: the largest per-tick array (obs_others) must stay below 4 GiB
    if ((size_t)p.E * AgentGeom<N>::VPE * 4 * sizeof(R) >= ((size_t)1 << 32))
      return fail(CM3_ERR_INVALID, "the lane-per-agent kernel addresses at most 4 GiB per array: %d envs x %d agents is too large", p.E, N);
    const bool live = p.state_copy != nullptr;   // cm3_particle_traj.state_live (per-tick launches only)
    constexpr int kNt = sizeof(R) == 4 ? kSpNt : kSpPlain, kWt = sizeof(R) == 4 ? kSpWt : kSpPlain;
    // a launch that writes >= kWtMinObsBytes of observation rows writes them through (kSpWt; per-tick launches, no slot copies)
    const bool wt = sizeof(R) == 4 && (size_t)(p.EN - p.E0) * AgentGeom<N>::VPE * 4 * sizeof(R) >= kWtMinObsBytes;
    if constexpr (N == 8 && sizeof(R) == 4) {
      // two lanes per agent (k_particle_step_agents2) while a SIMD holds few waves: per-tick launches up to kAgents2MaxEnvs
      if (p.n_ticks == 1 && (size_t)(p.EN - p.E0) <= kAgents2MaxEnvs) {
        const unsigned raw2 = (unsigned)(((size_t)(p.EN - p.E0) + (size_t)WAVES * 4 - 1) / ((size_t)WAVES * 4));
        const unsigned blocks2 = cm3_xcd_grid(raw2);
        const uint32_t xf2 = cm3_xcd_flags(raw2);
#define CM3_LAUNCH_AGENTS2(...)                                                                                             \
  hipLaunchKernelGGL((k_particle_step_agents2<WAVES, __VA_ARGS__>), dim3(blocks2), dim3(WAVES * 64), 0, stream, p.state_in,  \
                     p.goals_in, p.meta_in, (const int32_t *)p.episode, p.E, p.flags | xf2, p.E0, p.EN, p.max_steps,         \
                     (const int32_t *)p.actions, p)
        const bool early = wt && (size_t)(p.EN - p.E0) <= kAgents2EarlyMaxEnvs;
        note_variant("k_particle_step_agents2", 4, 8, WAVES, 0, wt ? kWt : (nt ? kNt : kSpPlain), live, CM3_PARTICLE_TU, early);
        if (live) {
          if (early) CM3_LAUNCH_AGENTS2(kWt, true, true);
          else if (wt) CM3_LAUNCH_AGENTS2(kWt, true);
          else if (nt) CM3_LAUNCH_AGENTS2(kNt, true);
          else CM3_LAUNCH_AGENTS2(kSpPlain, true);
        } else {
          if (early) CM3_LAUNCH_AGENTS2(kWt, false, true);
          else if (wt) CM3_LAUNCH_AGENTS2(kWt);
          else if (nt) CM3_LAUNCH_AGENTS2(kNt);
          else CM3_LAUNCH_AGENTS2(kSpPlain);
        }
#undef CM3_LAUNCH_AGENTS2
        CM3_HIP_CHECK(hipGetLastError());
        return CM3_OK;
      }
    }
    note_variant("k_particle_step_agents", (int)sizeof(R), N, WAVES, p.n_ticks > 1,
                 p.n_ticks > 1 ? (nt ? kNt : kSpPlain) : (wt ? kWt : (nt ? kNt : kSpPlain)), p.n_ticks == 1 && live, CM3_PARTICLE_TU);
    if (p.n_ticks > 1) {
      if (nt) CM3_LAUNCH_AGENTS(true, kNt);
      else CM3_LAUNCH_AGENTS(true, kSpPlain);
    } else if (live) {
      if (wt) CM3_LAUNCH_AGENTS(false, kWt, true);
      else if (nt) CM3_LAUNCH_AGENTS(false, kNt, true);
      else CM3_LAUNCH_AGENTS(false, kSpPlain, true);
    } else {
      if (wt) CM3_LAUNCH_AGENTS(false, kWt);
      else if (nt) CM3_LAUNCH_AGENTS(false, kNt);
      else CM3_LAUNCH_AGENTS(false, kSpPlain);
    }
#undef CM3_LAUNCH_AGENTS
    CM3_HIP_CHECK(hipGetLastError());
    return CM3_OK;
  } else {
    return fail(CM3_ERR_INVALID, "the lane-per-agent kernel needs n_agents >= 2");
  }
}

#ifndef CM3_PAIR_WAVES
#define CM3_PAIR_WAVES 4   // waves per workgroup of the shared-env mappings at >= 256 waves (macros: build variants for comparisons)
#endif
#ifndef CM3_AGENT_WAVES
#define CM3_AGENT_WAVES 4
#endif
template <typename R, int N> static int launch_n(const ParticleParams &p, ParticleOp op, hipStream_t stream) {
  if (op == kStep) {
    // Which mapping for (N, E): measured on MI355X in round 2, after the exact squared-distance thresholds took the square roots
    // out of every mapping (in place, us per tick, pair / agent / env; profiles/r02_n2345_mapping_sweep_*.txt, r02_n678_mid_sweep.txt,
    // r02_n8_mapping_sweep.txt, r02_n567_mapping_sweep.txt).  Round 1's rule (pair for N >= 3 up to 16384 envs, agent for N >= 6 between
    // 6144 and 2^17) had been tuned on N = 4 and N = 8 only and before that change; it left `auto` up to 41 % behind the best kernel:
    //   N = 3   24576: 4.43 / 5.15 / 4.97   32768: 5.14 / 5.31 / 5.14   65536: 7.7 / 7.6 / 5.8     -> pair to 32768, then env
    //   N = 4   16384: 5.06 / 5.13 / 6.65   24576: 6.40 / 6.04 / 7.29   32768: 7.85 / 6.42 / 7.67   65536: 12.6 / 9.1 / 8.4
    //           -> pair to 16384, agent to 49152, then env
    //   N = 5    8192: 5.14 / 5.33 / 9.73   16384: 7.64 / 6.42 / 10.0   32768: 12.8 / 9.2 / 11.5    65536: 21 / 14.2 / 12.3
    //           -> pair to 12288, agent to 49152, then env
    //   N = 6    8192: 5.7 / 6.0 / 13.2     16384: 8.8 / 7.5 / 13.5     65536: 23.6 / 16.3 / 16.7   2^17: - / 26.6 / 23.5
    //           -> pair to 12288, agent to 65536, then env
    //   N = 7    6144: 7.1 / 6.4 / 18.5     65536: 41 / 17.9 / 22.6     2^17 .. 2^20: agent = env within 1 %   -> agent 6144 .. 2^17
    //   N = 8    4096: 6.2 / 6.7 / 23.8      6144: 7.9 / 7.0 / 23.7     2^18: - / 68.9 / 71.1       2^20: - / 286 / 319
    //           -> pair below 6144, agent from 6144 up (no upper bound)
    // Crossovers that fall between two measured sizes (12288, 49152) are interpolated, not measured.
    constexpr size_t kInf = ~(size_t)0;
    // Re-measured after the executed paths of the pair / agent kernels were shortened (hardware soft-plus, 32-bit addressing, no
    // spills; tools/mapping_sweep.py, profiles/r02_mapping_sweep_after_path_shortening.txt): the table moved a little --
    //   N = 2: pair wins by 10 % up to 6144 envs (2.31 / - / 2.56 at 2048), level with env above  -> pair to 6144
    //   N = 3: pair to 24576 (32768: 4.41 / 4.43 / 4.20)
    //   N = 4: pair to 12288 (3.80 / 3.88 / 4.70), agent to 40960 (16384: 4.41 / 4.06 / 4.85; 49152: 9.0 / 6.40 / 6.30), then env
    //   N = 5, 6: agent from 8192 (N = 5: 4.31 / 4.23 / 6.41); N = 5 up to 40960, N = 6 up to 65536, then env
    //   N = 7, 8: agent from 6144 up, no upper bound (N = 7 at 2^20: - / 218 / 231)
    // ... and once more on the final build (max-ILP pair / agent kernels; profiles/r02_mapping_sweep_final_build.txt):
    //   N = 2: pair up to 32768 envs (16384: 2.57 / - / 2.68; 32768: 2.89 / - / 3.01); N = 6: agent from 6144 (4.27 / 4.11);
    //   N = 7, 8: agent from 4096 (N = 8: 5.12 / 4.57, N = 7: 4.73 / 4.34; at 2048 pair: 3.83 / 4.29)
    // round 4, after the two-stage action draw shortened every mapping's path: re-swept (profiles/r04_mapping_sweep.txt, merge8
    // positions): `auto` within ~1 % of the best mapping at 88 of 91 (N, E) points.  The two candidates for a move -- N = 2 pair up
    // to 65536, N = 4 agent up to 65536 -- were tried and taken back: on the antipodal config of the bench N = 4 at 65536 envs ran
    // 6.11 us with the agent mapping against 5.70 with lane-per-env; the crossovers depend on how crowded a config is.
    constexpr size_t kPairMax = N > 8 ? 0 : (N == 2 ? 32768 : (N == 3 ? 24576 : (N == 4 ? 12288 : kPairsMaxEnvs)));
    // round 3: N = 8 with two lanes per agent (k_particle_step_agents2; profiles/r03_two_lanes_per_agent.txt) moved its crossover
    // to 2048 envs; the XCD-aware block order (common.h) then sped the pair mapping up most at exactly these sizes
    // (profiles/r03_xcd_block_order.txt; pair / agent, in place): N = 8: 2048 3.42 / 3.74, 4096 4.47 / 3.84 -> agent from 4096;
    // N = 7: 4096 4.12 / 4.31, 6144 5.03 / 4.29 -> agent from 6144; N = 6: 6144 3.86 / 4.14, 8192 4.37 / 4.11 -> agent from 8192
    // N = 5: 8192 3.86 / 4.02, 12288 4.74 / 4.42 -> agent from 10240 (profiles/r03_mapping_sweep_xcd.txt)
    // N = 9, 10 (round 4; no pair mapping: N (N - 1) lanes do not fit a wave): lane per agent from 1024 envs (256 waves), as N = 8 above
    constexpr size_t kAgentLo = N == 4 ? 12289 : (N == 5 ? 10240 : (N == 6 ? 8192 : (N == 7 ? 6144 : (N == 8 ? 4096 : (N > 8 ? 1024 : kInf)))));
    // round 3, large batches after the write-through observation stores (profiles/r03_mapping_sweep_large.txt; env / agent):
    //   N = 6: 2^17 17.7 / 17.4, 2^19 66.3 / 62.3, 2^20 125.5 / 121.9, 2^21 290 / 326   -> agent up to 1.5 M envs (was 65536)
    //   N = 7: 2^19 89 / 80, 2^20 164-170 / 153-217 (the agent mapping is bimodal there: it depends on where the allocator
    //          put the buffers), 2^21 392-412 / 437-466                                -> agent up to 768 K envs (was unbounded)
    //   N = 8: 2^19 110.5 / 99.9, 2^20 222-233 / 278-281 (6.1 vs 4.9 TB/s), 2^21 489-556 / 477-505 -> agent up to 768 K envs
    constexpr size_t kAgentHi = N == 4 || N == 5 ? 40960 : (N == 6 ? 1572864 : (N >= 7 ? 786432 : 0));
    bool pairs = N >= 2 && (size_t)p.E <= kPairMax;
    bool agents = N >= 4 && (size_t)p.E >= kAgentLo && (size_t)p.E <= kAgentHi;
    // both shared-env mappings index with 32-bit byte offsets (obs_others below 4 GiB per tick); beyond that only a forced choice
    // reaches them (and is refused by their launchers)
    if ((size_t)p.E * N * (N - 1) * 4 * sizeof(R) >= ((size_t)1 << 32)) pairs = agents = false;
    if (p.flags & CM3_FLAG_KERNEL_LANE_PER_ENV) pairs = agents = false;
    if (p.flags & CM3_FLAG_KERNEL_LANE_PER_PAIR) { pairs = true; agents = false; }
    if (N > 8 && pairs) return fail(CM3_ERR_INVALID, "the lane-per-pair kernel needs n_agents in 2..8");
    if (p.flags & CM3_FLAG_KERNEL_LANE_PER_AGENT) agents = true;
    if (agents) {
      const size_t waves = ((size_t)p.E + AgentGeom<(N >= 2 ? N : 2)>::EPW - 1) / AgentGeom<(N >= 2 ? N : 2)>::EPW;
#ifndef CM3_PARTICLE_ILP_TU
      if constexpr (sizeof(R) == 4 && N >= 2)
        if (waves <= kIlpMaxWaves) return particle_ilp_launch_agents_f32(p, N, waves < 256 ? 1 : CM3_AGENT_WAVES, stream);
#endif
      if (waves < 256) return launch_agents<R, N, 1>(p, stream);
      return launch_agents<R, N, CM3_AGENT_WAVES>(p, stream);
    }
    if (pairs) {
      // 4 waves per workgroup (one per SIMD of a CU) measured faster than 1 or 2 from 1024 waves up
      // (tools/probes/step_timeline.hip: 4.38 vs 4.82 us at E=4096, 6.36 vs 7.32 us at E=16384, stamped build);
      // below 256 waves single-wave workgroups spread the work over more CUs.
      constexpr int NP = (N >= 2 && N <= 8) ? N : 2;
      const size_t waves = ((size_t)p.E + PairGeom<NP>::EPW - 1) / PairGeom<NP>::EPW;
#ifndef CM3_PARTICLE_ILP_TU
      if constexpr (sizeof(R) == 4 && N >= 2)
        if (waves <= kIlpMaxWaves) return particle_ilp_launch_pairs_f32(p, N, waves < 256 ? 1 : CM3_PAIR_WAVES, stream);
#endif
      if (waves < 256) return launch_pairs<R, N, 1>(p, stream);
      return launch_pairs<R, N, CM3_PAIR_WAVES>(p, stream);
    }
  }
  // lane-per-env.  Small batches: one wave per workgroup; large: 4 waves per workgroup (one per SIMD).
  // measured crossover (N=4): 11.27 vs 11.43 us at 2^17 envs, 20.9 vs 19.7 us at 2^18
  if ((size_t)p.E <= (size_t)128 * 1024) return launch_one<R, N, 1>(p, op, stream);
  return launch_one<R, N, 4>(p, op, stream);
}

template <typename R> static int launch(const ParticleParams &p, int n_agents, ParticleOp op, hipStream_t stream) {
  switch (n_agents) {
    case 1: return launch_n<R, 1>(p, op, stream);
    case 2: return launch_n<R, 2>(p, op, stream);
    case 3: return launch_n<R, 3>(p, op, stream);
    case 4: return launch_n<R, 4>(p, op, stream);
    case 5: return launch_n<R, 5>(p, op, stream);
    case 6: return launch_n<R, 6>(p, op, stream);
    case 7: return launch_n<R, 7>(p, op, stream);
    case 8: return launch_n<R, 8>(p, op, stream);
    case 9: return launch_n<R, 9>(p, op, stream);
    case 10: return launch_n<R, 10>(p, op, stream);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", n_agents);
}

template <typename R>
static int particle_call(const cm3_particle_desc *d, const cm3_particle_bufs *b, ParticleOp op, const uint8_t *mask,
                         void *stream) {
  ParticleParams p;
  int rc = fill_params(d, b, op, mask, p);
  if (rc != CM3_OK) return rc;
  return launch<R>(p, d->n_agents, op, (hipStream_t)stream);
}

// Non-temporal observation stores for a rollout whose observation slots are a STREAM -- more bytes than the cache hierarchy
// keeps (half of the 256 MB Infinity Cache) -- and plain ones when they are small enough to stay cached and be re-read (returns
// / normalisation / sampling right after a short rollout).  float32 only.  Measured support: see DESIGN.md section 0.
static uint32_t obs_store_nt(size_t obs_stride, int n_ticks) {
  return obs_stride * (size_t)n_ticks >= ((size_t)128 << 20) ? kFlagObsStoreNt : 0u;
}

template <typename R>
static int particle_rollout(const cm3_particle_desc *d, const cm3_particle_traj *t, int32_t n_ticks, void *stream) {
  CM3_REQUIRE(d && t, "null desc/traj");
  CM3_REQUIRE(n_ticks >= 1, "n_ticks must be >= 1");
  CM3_REQUIRE(t->state && t->goals && t->obs_others && t->actions && t->reward_n && t->reward && t->done && t->meta,
              "rollout: trajectory base pointers are required");
  CM3_REQUIRE(!t->state_live || t->goals_live, "rollout: state_live needs goals_live (goals_live alone = sparse goal slots)");
  CM3_REQUIRE(!t->goals_live || (t->goals_stride != 0 && (!t->state_live || t->state_stride != 0)),
              "rollout: live buffers are for slot trajectories (state_stride / goals_stride must be non-zero)");
  auto at = [](void *base, size_t stride, int k) -> void * {
    return base ? (void *)((char *)base + stride * (size_t)k) : nullptr;
  };
  if (d->flags & CM3_FLAG_FUSED_TICKS) {
    // ONE launch runs all n_ticks ticks with the state in registers (no per-tick launch, no state re-load).
    // Only possible when no host/policy step is needed between ticks: actions are drawn in-kernel, or all
    // n_ticks action slots were filled beforehand.
    CM3_REQUIRE((d->flags & CM3_FLAG_GEN_ACTIONS) || n_ticks == 1 || t->actions_stride != 0,
                "fused rollout needs in-kernel actions or one pre-filled action slot per tick");
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
    ParticleParams p;
    int rc = fill_params(d, &b, kStep, nullptr, p);
    if (rc != CM3_OK) return rc;
    p.n_ticks = n_ticks;
    p.st_state = t->state_stride;
    p.st_goals = t->goals_stride;
    p.st_obs = t->obs_others_stride;
    p.st_actions = t->actions_stride;
    p.st_reward_n = t->reward_n_stride;
    p.st_reward = t->reward_stride;
    p.st_done = t->done_stride;
    p.st_term_state = t->term_state_stride;
    p.st_term_obs = t->term_obs_others_stride;
    p.st_coll = t->collisions_stride;
    p.flags |= obs_store_nt(t->obs_others_stride, n_ticks);
    return launch<R>(p, d->n_agents, kStep, (hipStream_t)stream);
  }
  for (int k = 0; k < n_ticks; ++k) {
    cm3_particle_bufs b;
    memset(&b, 0, sizeof(b));
    const bool live = t->state_live != nullptr;
    const bool sparse_goals = !live && t->goals_live != nullptr;   // goals live in place, slot k + 1 only where an env restarts
    b.state_in = live ? t->state_live : at(t->state, t->state_stride, k);
    b.state_out = live ? t->state_live : at(t->state, t->state_stride, k + 1);
    b.goals_in = (live || sparse_goals) ? t->goals_live : at(t->goals, t->goals_stride, k);
    b.goals_out = (live || sparse_goals) ? t->goals_live : at(t->goals, t->goals_stride, k + 1);
    b.meta_in = t->meta;
    b.meta_out = t->meta;
    b.episode = t->episode;
    b.actions = (int32_t *)at(t->actions, t->actions_stride, k);
    b.obs_others = at(t->obs_others, t->obs_others_stride, k + 1);
    b.reward_n = at(t->reward_n, t->reward_n_stride, k);
    b.reward = at(t->reward, t->reward_stride, k);
    b.done = (uint8_t *)at(t->done, t->done_stride, k);
    b.term_state = at(t->term_state, t->term_state_stride, k);
    b.term_obs_others = at(t->term_obs_others, t->term_obs_others_stride, k);
    b.collisions_tick = (int32_t *)at(t->collisions, t->collisions_stride, k);
    ParticleParams p;
    int rc = fill_params(d, &b, kStep, nullptr, p);
    if (rc != CM3_OK) return rc;
    p.flags |= obs_store_nt(t->obs_others_stride, n_ticks);
    if (live) {
      p.state_copy = at(t->state, t->state_stride, k + 1);
      p.goals_copy = at(t->goals, t->goals_stride, k + 1);
    } else if (sparse_goals) {
      p.goals_copy = at(t->goals, t->goals_stride, k + 1);
    }
    rc = launch<R>(p, d->n_agents, kStep, (hipStream_t)stream);
    if (rc != CM3_OK) return rc;
  }
  return CM3_OK;
}


}  // namespace cm3

#ifdef CM3_PARTICLE_ILP_TU
namespace cm3 {
template <int N> static int ilp_pairs_n(const ParticleParams &p, int w, hipStream_t s) {
  return w == 1 ? launch_pairs<float, N, 1>(p, s) : launch_pairs<float, N, CM3_PAIR_WAVES>(p, s);
}
template <int N> static int ilp_agents_n(const ParticleParams &p, int w, hipStream_t s) {
  return w == 1 ? launch_agents<float, N, 1>(p, s) : launch_agents<float, N, CM3_AGENT_WAVES>(p, s);
}
int particle_ilp_launch_pairs_f32(const ParticleParams &p, int n_agents, int w, hipStream_t s) {
  switch (n_agents) {
    case 2: return ilp_pairs_n<2>(p, w, s);
    case 3: return ilp_pairs_n<3>(p, w, s);
    case 4: return ilp_pairs_n<4>(p, w, s);
    case 5: return ilp_pairs_n<5>(p, w, s);
    case 6: return ilp_pairs_n<6>(p, w, s);
    case 7: return ilp_pairs_n<7>(p, w, s);
    case 8: return ilp_pairs_n<8>(p, w, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", n_agents);
}
int particle_ilp_launch_agents_f32(const ParticleParams &p, int n_agents, int w, hipStream_t s) {
  switch (n_agents) {
    case 2: return ilp_agents_n<2>(p, w, s);
    case 3: return ilp_agents_n<3>(p, w, s);
    case 4: return ilp_agents_n<4>(p, w, s);
    case 5: return ilp_agents_n<5>(p, w, s);
    case 6: return ilp_agents_n<6>(p, w, s);
    case 7: return ilp_agents_n<7>(p, w, s);
    case 8: return ilp_agents_n<8>(p, w, s);
    case 9: return ilp_agents_n<9>(p, w, s);
    case 10: return ilp_agents_n<10>(p, w, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", n_agents);
}
}  // namespace cm3
#define CM3_NO_ENTRY_POINTS 1
#endif

// The library compiles this file twice (build.sh): -DCM3_PARTICLE_F32 and -DCM3_PARTICLE_F64 each instantiate one
// real type, which halves the wall-clock of the build.  Without either macro both are instantiated.
#if !defined(CM3_PARTICLE_F32) && !defined(CM3_PARTICLE_F64)
#define CM3_PARTICLE_F32 1
#define CM3_PARTICLE_F64 1
#endif

#ifndef CM3_NO_ENTRY_POINTS
extern "C" {
#ifdef CM3_PARTICLE_F32
int cm3_particle_step_f32(const cm3_particle_desc *d, const cm3_particle_bufs *b, void *s) {
  return cm3::particle_call<float>(d, b, cm3::kStep, nullptr, s);
}
int cm3_particle_reset_f32(const cm3_particle_desc *d, const cm3_particle_bufs *b, const uint8_t *m, void *s) {
  return cm3::particle_call<float>(d, b, cm3::kReset, m, s);
}
int cm3_particle_observe_f32(const cm3_particle_desc *d, const cm3_particle_bufs *b, void *s) {
  return cm3::particle_call<float>(d, b, cm3::kObserve, nullptr, s);
}
int cm3_particle_rollout_f32(const cm3_particle_desc *d, const cm3_particle_traj *t, int32_t n, void *s) {
  return cm3::particle_rollout<float>(d, t, n, s);
}
#endif
#ifdef CM3_PARTICLE_F64
int cm3_particle_step_f64(const cm3_particle_desc *d, const cm3_particle_bufs *b, void *s) {
  return cm3::particle_call<double>(d, b, cm3::kStep, nullptr, s);
}
int cm3_particle_reset_f64(const cm3_particle_desc *d, const cm3_particle_bufs *b, const uint8_t *m, void *s) {
  return cm3::particle_call<double>(d, b, cm3::kReset, m, s);
}
int cm3_particle_observe_f64(const cm3_particle_desc *d, const cm3_particle_bufs *b, void *s) {
  return cm3::particle_call<double>(d, b, cm3::kObserve, nullptr, s);
}
int cm3_particle_rollout_f64(const cm3_particle_desc *d, const cm3_particle_traj *t, int32_t n, void *s) {
  return cm3::particle_rollout<double>(d, t, n, s);
}
#endif
}
#endif  // CM3_NO_ENTRY_POINTS
