/*
 * cm3_amd.h -- C ABI of libcm3_hip.so: the MI355X (gfx950) vectorised rollout engine for the
 * CM3 hot path (cooperative-navigation particle env, Checkers env, trajectory collection).
 *
 * The reference (011235813/cm3) is pure Python and has no FFI of its own; the interface the
 * hot path sits behind is the Python env protocol (reset()/step()) used by
 * alg/train_onpolicy.py:119,126,282,294,321,323.  Each entry point below names the reference
 * function(s) it replaces for E environments at once.  INTEGRATION.md shows the ctypes stub a
 * reference maintainer would add.
 *
 * Conventions
 *   - Plain pointers and sizes only; no torch / HIP types in signatures (`stream` is a
 *     hipStream_t passed as void*, NULL = the default stream).
 *   - Every pointer is a DEVICE pointer owned by the caller unless stated otherwise.  The
 *     library allocates nothing persistent, keeps no pointer after return, and only ENQUEUES
 *     work on `stream` (asynchronous).
 *   - Return value: 0 (CM3_OK) or a negative CM3_ERR_* code; cm3_last_error() returns a
 *     thread-local message for the last failure on the calling thread.
 *   - E = environments in this launch (this rank's shard), N = agents per environment.
 *
 * Device layouts (particle; `real` = float for *_f32, double for *_f64)
 *   state       real [N][E][4]   (vx, vy, px, py) of agent i in env e at [i][e]: one 16/32-byte
 *                                vector per lane, unit-stride over e (coalesced).  The
 *                                reference's global_state[N,4] (environment.py:113-116) and
 *                                obs_self (multi-goal_spread.py:154) for env e are the rows
 *                                state[:, e, :].
 *   goals       real [N][E][2]   landmark positions (train_onpolicy.py:283-285)
 *   meta        int32 [E][2]     {steps, collisions}   (environment.py:93; multi-goal_spread.py:93,137).  As in the
 *                                reference, an env that is stepped after its episode ended keeps counting both
 *                                (MultiAgentEnv.step never looks at a previous `done`); the COLLECTOR stops using it
 *                                (train_onpolicy.py:302) -- see cm3_particle_traj.collisions for the per-episode value.
 *   episode     int32 [E]        episodes started so far by env e (RNG key; touched on reset only)
 *   actions     int32 [E][N]     discrete actions 0..4 (environment.py:197-200)
 *   obs_others  real [E][N][L]   L = 4*max(N-1,1)  (multi-goal_spread.py:145-154)
 *   reward_n    real [E][N]      (multi-goal_spread.py:121-138)
 *   reward      real [E]         np.sum(reward_n) (environment.py:107)
 *   done        uint8 [E]        (environment.py:118-121)
 */
#ifndef CM3_AMD_H
#define CM3_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CM3_ABI_VERSION 7   /* 7 (round 6): REMOVED cm3_particle_rollout_chains_f32 / _f64 (sub-batch chains on several streams: a tested,
                               measured regression since round 2 -- profiles/r02_chains_diag.txt; tools/chains_diag.py reproduces it
                               with desc->env_offset / env_count).  ADDED cm3_td_target_f64, cm3_policy_rollout_checkers; cm3_actor_checkers_packed_bytes grew by the others-branch table; the
                               precision-2 Checkers actor adds branch_others W_others_h2 to h2's accumulators BEFORE branch_self
                               W_self_h2 (probabilities move in the last bits).
                               6: cm3_policy_force_row_tiles, cm3_rows_scatter / cm3_rows_gather / cm3_rows_tile / cm3_transitions_gather_f32 (round 5).
                               5: the in-kernel ACTION STREAM of the particle kernels and the actors' sampling uniforms became two stages
                               (Philox4x32-10 block per (seed, global env id, call) + fmix32((word ^ step) + episode * 0x9E3779B1);
                               csrc/philox.h, oracle/philox.py; since ABI 6 the Checkers kernels too) -- a given (seed, env, episode, step, agent) draws a DIFFERENT action
                               than under ABI 4, whose counter layout masked `step & 0x00FFFFFF` into one Philox call (Checkers kept
                               that one-stage draw until ABI 6).  Also 5: cm3_last_kernel_variant; cm3_returns_moments_* takes a zero-initialised-once scratch (see there);
                               cm3_returns_normalize_*, cm3_copy_shift, cm3_source_id, actor precision 2 (all added under 4) */
#define CM3_MAX_AGENTS 10   /* the reference's make_world takes up to ten agents (its colour table, multi-goal_spread.py:7-16);
                              8 until ABI 5.  Checkers, the lane-per-pair mapping and the fused policy rollout stay at <= 8 */

#define CM3_OK 0
#define CM3_ERR_INVALID (-1)     /* bad argument / unsupported configuration */
#define CM3_ERR_HIP (-2)         /* a HIP runtime call failed */
#define CM3_ERR_STATE (-3)       /* call sequence error (graph capture etc.) */

/* cm3_*_desc.flags */
#define CM3_FLAG_AUTO_RESET 1u   /* an env whose episode ends is re-initialised inside the same launch:
                                    its reward/done outputs are the terminal ones, its state/obs outputs are
                                    those of the fresh episode (train_onpolicy.py:282 folded into :321) and
                                    the true terminal next-state goes to term_* when those are non-NULL */
#define CM3_FLAG_GEN_ACTIONS 2u  /* actions are drawn in-kernel, uniform on {0..4} (train_onpolicy.py:305-307),
                                    Philox4x32-10 keyed (seed, global env id, episode, step), and WRITTEN to
                                    `actions` */

#define CM3_FLAG_FUSED_TICKS 0x400u  /* cm3_particle_rollout_* only: run all n_ticks ticks in ONE launch with the
                                       state in registers (identical results).  Needs CM3_FLAG_GEN_ACTIONS or one
                                       pre-filled action slot per tick: nothing can act between the ticks */
#define CM3_FLAG_KERNEL_LANE_PER_ENV 0x100u  /* particle step: force the one-lane-per-env mapping   */
#define CM3_FLAG_KERNEL_LANE_PER_PAIR 0x200u /* particle step: force the one-lane-per-agent-pair mapping
                                                (default: chosen from n_envs; both give identical results) */
#define CM3_FLAG_KERNEL_LANE_PER_AGENT 0x800u /* particle step: force the one-lane-per-agent mapping (n_agents >= 2) */
/* The two shared-env mappings index with 32-bit byte offsets: every per-tick array (the largest is obs_others,
   n_envs * N * (N-1) * 4 reals) must stay below 4 GiB.  The default choice respects that; a FORCED mapping beyond it returns
   CM3_ERR_INVALID.  (N = 8, float32: 4.79 M envs.) */

int cm3_abi_version(void);
/* First 16 hex digits of the SHA-256 over the library's sources (the .hip and .h files of csrc/ in byte order of their names,
 * then this header), baked in by csrc/build.sh.  The Python binding compares it with the sources next to it and refuses a stale build. */
const char *cm3_source_id(void);
const char *cm3_last_error(void);
/* Debug / test query (ABI 5): the kernel instantiation that the most recent cm3_particle_* / cm3_checkers_* call ON THE CALLING
 * THREAD launched last, e.g. "k_particle_step_pairs<f32,N=4,waves=4,fused=0,sp=nt,live=1,early=0,g=0,tu=ilp>" -- the launchers
 * choose among size-gated builds of one template (mapping, waves per workgroup, observation store policy plain / nt / wt, the
 * live-state variant, the max-ILP translation unit); tests/test_gpu_dispatch_sizes.py asserts which build it checked.
 * Thread-local storage, valid until the next call on the thread.  No reference counterpart. */
const char *cm3_last_kernel_variant(void);
/* Number of visible HIP devices (0 when none); fills name (<= len bytes) of device `dev` if name != NULL. */
int cm3_device_count(void);
int cm3_device_name(int dev, char *name, int len);

/* ------------------------------------------------------------------------------------------
 * Particle env: MultiAgentEnv + multi-goal_spread scenario
 * ---------------------------------------------------------------------------------------- */
typedef struct cm3_particle_desc {
  int32_t n_envs;      /* E */
  int32_t n_agents;    /* N, 1..CM3_MAX_AGENTS */
  int32_t max_steps;   /* config.json:61 */
  uint32_t flags;      /* CM3_FLAG_* */
  int32_t env_offset;  /* launch only envs [env_offset, env_offset + env_count) of the E-extent arrays (a sub-batch; the */
  int32_t env_count;   /* arrays and RNG keys are untouched by the split).  env_count == 0 (with offset 0) = all E envs */
  int64_t env_id_base; /* global id of local env 0: RNG is keyed by GLOBAL env id, so results do not
                          depend on how envs are sharded over GPUs */
  uint64_t seed;
  double prob_random;  /* multi-goal_spread.py:75 */
  double initial_std;  /* multi-goal_spread.py:80-81 */
  double agents_x[CM3_MAX_AGENTS], agents_y[CM3_MAX_AGENTS];       /* config_particle_*.json */
  double landmarks_x[CM3_MAX_AGENTS], landmarks_y[CM3_MAX_AGENTS];
} cm3_particle_desc;

typedef struct cm3_particle_bufs {
  const void *state_in; /* real [N][E][4] pre-step state */
  void *state_out;      /* real [N][E][4] post-step state; may alias state_in */
  const void *goals_in; /* real [N][E][2] */
  void *goals_out;      /* real [N][E][2]; may alias goals_in (then only re-initialised envs are written) */
  const int32_t *meta_in; /* int32 [E][2] */
  int32_t *meta_out;      /* may alias meta_in */
  int32_t *episode;     /* int32 [E], in/out */
  int32_t *actions;     /* int32 [E][N]; input, or output under CM3_FLAG_GEN_ACTIONS */
  void *obs_others;     /* real [E][N][L] out */
  void *reward_n;       /* real [E][N] out */
  void *reward;         /* real [E] out */
  uint8_t *done;        /* uint8 [E] out */
  void *term_state;      /* optional real [N][E][4]: written only for envs re-initialised by AUTO_RESET */
  void *term_obs_others; /* optional real [E][N][L]: same */
  int32_t *collisions_tick; /* optional int32 [E] out, written for EVERY env: scenario.collisions (multi-goal_spread.py:137)
                               after this tick and before any same-launch re-initialisation.  At the tick that ends an
                               episode it is that episode's count, whose `!= 0` is the reference's is_bad flag of the dual
                               replay buffer (train_onpolicy.py:356): AUTO_RESET zeroes the live counter in the same launch,
                               and without it the live counter keeps counting if the env is stepped further */
} cm3_particle_bufs;

/* One tick for E envs in ONE kernel launch: replaces MultiAgentEnv.step (environment.py:81-123) =
 * _set_action (:177-225) + World.step (core.py:117-131) + Scenario.observation/reward/done
 * (multi-goal_spread.py:121-154). */
int cm3_particle_step_f32(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, void *stream);
int cm3_particle_step_f64(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, void *stream);

/* (Re)initialise envs: replaces MultiAgentEnv.reset (environment.py:125-149) +
 * Scenario.reset_world (multi-goal_spread.py:65-93).  Uses state_out, goals_out, meta_out, episode,
 * obs_others of `bufs`; `mask` (uint8 [E], optional) selects the envs to reset (NULL = all).
 * The build's own counter-based RNG reproduces the reference's DISTRIBUTIONS (one Bernoulli(prob_random)
 * per episode shared by agents and landmarks; U(-1,1)^2, or preset + N(0, initial_std) on agents only). */
int cm3_particle_reset_f32(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, const uint8_t *mask,
                           void *stream);
int cm3_particle_reset_f64(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, const uint8_t *mask,
                           void *stream);

/* Observation only (no dynamics): obs_others from state_in.  Replaces Scenario.observation
 * (multi-goal_spread.py:145-154) after a state injection. */
int cm3_particle_observe_f32(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, void *stream);
int cm3_particle_observe_f64(const cm3_particle_desc *desc, const cm3_particle_bufs *bufs, void *stream);

/* Trajectory collection (train_onpolicy.py:302-350): n_ticks launches of the step kernel over a
 * time-major trajectory.  Slot t of each array is base + t*stride (strides in BYTES).  state/obs_others/
 * goals have n_ticks+1 slots (slot t = before tick t, slot t+1 = after); the per-tick outputs have
 * n_ticks slots.  meta/episode are live (in place).  goals_stride == 0 keeps one live goals array.
 * term_* are optional (n_ticks slots).  All strides 0 = every tick overwrites the same live buffers.
 * With CM3_FLAG_GEN_ACTIONS every launch draws its own actions and writes them to its action slot. */
typedef struct cm3_particle_traj {
  void *state;        size_t state_stride;
  void *goals;        size_t goals_stride;
  void *obs_others;   size_t obs_others_stride;
  int32_t *actions;   size_t actions_stride;
  void *reward_n;     size_t reward_n_stride;
  void *reward;       size_t reward_stride;
  uint8_t *done;      size_t done_stride;
  int32_t *meta;      /* int32 [E][2], live */
  int32_t *episode;   /* int32 [E], live */
  void *term_state;      size_t term_state_stride;
  void *term_obs_others; size_t term_obs_others_stride;
  int32_t *collisions;  size_t collisions_stride; /* optional, n_ticks slots of int32 [E]: cm3_particle_bufs.collisions_tick */
  /* optional (ABI 4), both or neither; per-tick launches only (ignored with CM3_FLAG_FUSED_TICKS, where the state never leaves
     the registers): LIVE state real [N][E][4] / goals real [N][E][2].  When given, every tick reads and overwrites the live
     buffers IN PLACE (they hold the current state on entry and the final state on return) and writes its post-step state ALSO to
     slot k + 1 of `state` (slot 0 is neither read nor written: the caller keeps the initial state there if it wants it).  The
     GOALS slots are sparse from ABI 5 on (see goals_live below): slot k + 1 of `goals` is written only for the envs that restart
     at tick k -- landmarks do not move otherwise.  Values are identical either way; what changes is where a tick LOADS from -- lines its predecessor read and
     overwrote stay resident, lines of a fresh slot that was only ever written do not: measured 0.19 us of 2.83 us per tick at
     4096 envs x 4 agents (tools/trajectory_gap.py). */
  void *state_live;
  void *goals_live;
  /* goals_live WITHOUT state_live (ABI 5): SPARSE goal slots.  Landmarks move only when an episode starts (multi-goal_spread.py:
     88-91), yet a slot trajectory wrote all of them every tick: 8 N of the ~450 bytes a tick moves per env at streaming sizes, where
     the launch runs at the chip's copy rate (profiles/r04_trajectory_stream.txt).  Here the goals live in goals_live (read and, on a
     restart, updated in place) and slot k + 1 of `goals` is written ONLY for the envs re-initialised at tick k: the goals in effect
     at slot t are those of the last written slot <= t (slot 0 is the caller's).  Per-tick launches only. */
} cm3_particle_traj;

int cm3_particle_rollout_f32(const cm3_particle_desc *desc, const cm3_particle_traj *traj, int32_t n_ticks,
                             void *stream);
int cm3_particle_rollout_f64(const cm3_particle_desc *desc, const cm3_particle_traj *traj, int32_t n_ticks,
                             void *stream);


/* ------------------------------------------------------------------------------------------
 * Checkers env (env/checkers.py).  Compact live state, reference-shaped outputs.
 *   mask     uint64 [E]      bit (k*n_columns + j) set <=> reward cell (row k, col j) collected
 *   agents   uint32 [N][E]   r | c<<8 | n_green<<16 | n_orange<<24   (expanded-grid coordinates)
 *   steps    int32  [E]
 *   goals    uint8  [E][N]   index of the colour the agent wants (0 green, 1 orange): argmax of the
 *                            reference's one-hot goals (checkers.py:234)
 *   actions  int32  [E][N]
 * Outputs (values identical to the reference's float64 arrays; integer-valued ones are stored as
 * integers):
 *   grid        int8   [E][n_rows][n_columns+1][2]        get_valid_grid (checkers.py:66-76); env records may be
 *                                                         padded, see grid_stride / obs_self_t_stride
 *   vec         int32  [E][N][4]                          (r, c, n_green, n_orange) (:79-94)
 *   obs_others  double [E][N][2*max(N-1,1)]               (:128-154, normalize :112-125)
 *   obs_self_t  int8   [E][N][2*n_obs+1][2*n_obs+1][3]    get_obs (:97-109)
 *   obs_self_v  double [E][N][4]
 *   local_rewards double [E][N]; reward double [E]; done uint8 [E]   (:228-262)
 * ---------------------------------------------------------------------------------------- */
typedef struct cm3_checkers_desc {
  int32_t n_envs;
  int32_t n_agents;   /* 1..CM3_MAX_AGENTS */
  int32_t n_rows;     /* odd  (checkers.py:16) */
  int32_t n_columns;  /* even (checkers.py:17); n_rows*n_columns <= 64 */
  int32_t n_obs;
  int32_t max_steps;
  uint32_t flags;     /* CM3_FLAG_AUTO_RESET, CM3_FLAG_GEN_ACTIONS */
  int32_t grid_stride;       /* bytes between consecutive envs' grid records; 0 = packed (n_rows*(n_columns+1)*2) */
  int32_t obs_self_t_stride; /* bytes between consecutive envs' obs_self_t records; 0 = packed (N*K*K*3).
                                Records padded to a multiple of 4 bytes (56 / 152 for the reference geometry with
                                N = 2) enable the multi-lane fast kernel, which writes each record up to its payload
                                rounded up to 4 bytes (the rounding bytes as 0; bytes of a LARGER stride beyond that are
                                left untouched) and indexes with 32-bit byte offsets: n_envs times the widest per-env record
                                of any per-tick array -- max(obs_self_t_stride, grid_stride, 32 N, 16 N max(N-1, 1)) bytes
                                (obs_others is the widest from N = 6 on: 896 B at N = 8) -- must stay below 4 GiB, else
                                CM3_ERR_INVALID */
  int32_t _pad;
  int64_t env_id_base;
  uint64_t seed;
  int32_t agents_r[CM3_MAX_AGENTS]; /* before expansion, as in config_checkers_*.json */
  int32_t agents_c[CM3_MAX_AGENTS];
} cm3_checkers_desc;

typedef struct cm3_checkers_bufs {
  uint64_t *mask;      /* in/out */
  uint32_t *agents;    /* in/out */
  int32_t *steps;      /* in/out */
  int32_t *episode;    /* in/out (RNG key for GEN_ACTIONS) */
  uint8_t *goals;      /* in; rewritten only for N == 1 under AUTO_RESET (train_onpolicy.py:288-291) */
  int32_t *actions;    /* in, or out under GEN_ACTIONS */
  int8_t *grid;
  int32_t *vec;
  double *obs_others;
  int8_t *obs_self_t;
  double *obs_self_v;
  double *local_rewards;
  double *reward;
  uint8_t *done;
  /* optional, all five or none: written ONLY for envs re-initialised by CM3_FLAG_AUTO_RESET in this launch -- their TRUE
     post-step observation, i.e. the next_* columns of the terminal transition, which the reference stores before it resets
     (train_onpolicy.py:336-347; checkers.py:246-262).  Same layouts / record strides as the five arrays above. */
  int8_t *term_grid;
  int32_t *term_vec;
  double *term_obs_others;
  int8_t *term_obs_self_t;
  double *term_obs_self_v;
  uint8_t *goals_next; /* optional uint8 [E][N]: the goals in effect AFTER this tick -- the goals column of the NEXT
                          transition (they change only when a single-agent env restarts, train_onpolicy.py:288-291) */
  const uint32_t *action_block; /* optional (ABI 6), uint32 [E][4], 16-byte aligned, filled ONCE per (seed, env_id_base) by
                          cm3_checkers_action_blocks: stage 1 of the in-kernel action stream (CM3_FLAG_GEN_ACTIONS), a constant per env.
                          With it a step launch loads the block with its state instead of running ten Philox rounds behind the
                          loads; NULL: the kernel computes the same block itself (identical actions). */
} cm3_checkers_bufs;

/* Trajectory collection for Checkers (train_onpolicy.py:302-350, 16-column transitions): n_ticks step launches over
 * a time-major trajectory, or ONE launch with CM3_FLAG_FUSED_TICKS (fast kernel only).  Observation arrays have
 * n_ticks+1 slots (slot t = before tick t); per-tick outputs n_ticks slots; strides in BYTES between slots.
 * With CM3_FLAG_AUTO_RESET (continuous collection) slot t+1 of an env whose episode ended at tick t holds the FRESH
 * episode's observation (what the policy acts on next) and the terminal next_* go to the term_* slots of tick t. */
typedef struct cm3_checkers_traj {
  uint64_t *mask; uint32_t *agents; int32_t *steps; int32_t *episode; uint8_t *goals; /* live, in place */
  int32_t *actions;      size_t actions_stride;
  int8_t *grid;          size_t grid_slot_stride;
  int32_t *vec;          size_t vec_stride;
  double *obs_others;    size_t obs_others_stride;
  int8_t *obs_self_t;    size_t obs_self_t_slot_stride;
  double *obs_self_v;    size_t obs_self_v_stride;
  double *local_rewards; size_t local_rewards_stride;
  double *reward;        size_t reward_stride;
  uint8_t *done;         size_t done_stride;
  /* optional terminal capture (see cm3_checkers_bufs), n_ticks slots each */
  int8_t *term_grid;        size_t term_grid_slot_stride;
  int32_t *term_vec;        size_t term_vec_stride;
  double *term_obs_others;  size_t term_obs_others_stride;
  int8_t *term_obs_self_t;  size_t term_obs_self_t_slot_stride;
  double *term_obs_self_v;  size_t term_obs_self_v_stride;
  uint8_t *goals_slots;     size_t goals_slots_stride; /* optional, n_ticks+1 slots of uint8 [E][N]; slot 0 is the caller's,
                                                          tick t writes slot t+1 (cm3_checkers_bufs.goals_next) */
  const uint32_t *action_block;  /* optional (ABI 6): see cm3_checkers_bufs.action_block */
} cm3_checkers_traj;

int cm3_checkers_rollout(const cm3_checkers_desc *desc, const cm3_checkers_traj *traj, int32_t n_ticks, void *stream);

/* Stage 1 of the in-kernel action stream for every env of `desc` (seed, env_id_base, n_envs): out = uint32 [E][4], the Philox block
 * action_block(seed, global env id, 0) of csrc/philox.h.  Once per env object (ABI 6; since round 5 the Checkers kernels draw with
 * the particle kernels' two-stage stream: action = rand5(fmix32((word ^ step) + episode * 0x9E3779B1))). */
int cm3_checkers_action_blocks(const cm3_checkers_desc *desc, uint32_t *out, void *stream);
/* SCOPE of that table: it holds Philox call 0 = the words of agents 0..3; with 5..8 agents the second block (agents
 * 4..7) is still computed in every step launch, so the round-5 gain applies to n_agents <= 4.  The table is a function of
 * (desc->seed, desc->env_id_base, n_envs): REBUILD it whenever one of them changes -- a step launch cannot check, a stale table
 * silently draws another action stream (cm3_amd.checkers.VecCheckersEnv builds it once per env object, whose seed and base are fixed). */
/* Replaces Checkers.step (checkers.py:228-262): agents act sequentially in index order inside one lane. */
int cm3_checkers_step(const cm3_checkers_desc *desc, const cm3_checkers_bufs *bufs, void *stream);
/* Replaces Checkers.reset (checkers.py:265-291) for the envs selected by mask (NULL = all). */
int cm3_checkers_reset(const cm3_checkers_desc *desc, const cm3_checkers_bufs *bufs, const uint8_t *mask,
                       void *stream);

/* ------------------------------------------------------------------------------------------
 * On-device particle actor: networks.actor_particle (networks.py:517-538) + the epsilon-mixed categorical
 * sampling of alg_credit.py:119-120 / run_actor :249-270, for all E*N agent rows in one launch.
 * Weights are float32, row-major [in][out] exactly as the TF variables are shaped:
 *   w_self [6][64] + b_self [64]            "actor_branch_self" (input = concat(v_obs[4], v_goal[2]))
 *   w_self_h2 [64][64]                      "W_branch_self_h2"
 *   w_others [L][128] + b_others [128]      "stage-2/actor_others"    (stage > 1 only; L = 4*max(N-1,1))
 *   w_others_h2 [128][64]                   "stage-2/W_others_h2"
 *   b_h2 [64]                               "b"
 *   w_out [64][5] + b_out [5]               "actor_out"
 * Inputs are the env's own buffers (obs_others [E][N][L], state [N][E][4], goals [N][E][2], meta, episode);
 * outputs actions int32 [E][N] (what cm3_particle_step_* consumes) and optionally the mixed probabilities
 * float [E][N][5].  Sampling inverts the CDF in action order with one uniform per agent-step from the Philox
 * stream keyed (seed, global env id, episode, step).
 * ---------------------------------------------------------------------------------------- */
typedef struct cm3_actor_particle_desc {
  int32_t n_envs, n_agents;
  int32_t stage;       /* 1: self branch only; 2: + others branch */
  int32_t n_h1_self, n_h1_others, n_h2, n_actions; /* 64, 128, 64, 5 */
  float epsilon;
  int32_t precision;   /* 0: float32 throughout (exact-f32 MFMA: a k-ordered fmaf chain).  1: second layer on the bf16
                          matrix cores with float32 accumulation (activations and W2 rounded to bf16; probabilities move
                          by up to ~1e-2).  2: SPLIT float16 -- activations and W2 as hi + lo float16 pairs, three f16 MFMA
                          passes (hi hi + hi lo + lo hi), float32 accumulation (22 of 24 significand bits per factor):
                          probabilities within the 2e-5 of the float64 oracle that precision 0 is held to, ~5x fewer
                          matrix-core cycles; first-layer activations must stay below 65504.  With 16..32 others inputs (5..9
                          agents) actor_others runs in the same split float16 (inputs and weights split alike; inputs below
                          65504 too: they are relative positions / velocities) */
  int32_t _pad;
  int64_t env_id_base;
  uint64_t seed;
} cm3_actor_particle_desc;

typedef struct cm3_actor_particle_weights {
  const float *w_self, *b_self, *w_self_h2, *w_others, *b_others, *w_others_h2, *b_h2, *w_out, *b_out;
  const void *packed; /* kernel-layout copy written by cm3_actor_particle_pack (cm3_actor_particle_packed_bytes()
                         bytes); the forward launch reads ONLY this buffer, re-pack after every weight update */
} cm3_actor_particle_weights;

typedef struct cm3_actor_particle_bufs {
  const void *obs_others;
  const void *state;
  const void *goals;
  const int32_t *meta;
  const int32_t *episode;
  int32_t *actions;
  float *probs; /* optional */
  const float *epsilon_dev; /* optional device float: read at launch INSTEAD of desc->epsilon, so a captured hipGraph
                               follows the epsilon annealing of train_onpolicy.py:369 without being re-captured */
} cm3_actor_particle_bufs;

size_t cm3_actor_particle_packed_bytes(int32_t n_agents);
/* Re-arranges the TensorFlow-shaped weights into the forward kernel's layout (unit-major first-layer tables, per-lane
 * MFMA B operands in f32 and bf16) -- one small launch per weight update; only desc->n_agents / stage are read. */
int cm3_actor_particle_pack(const cm3_actor_particle_desc *desc, const cm3_actor_particle_weights *weights,
                            void *packed, void *stream);
int cm3_actor_particle_f32(const cm3_actor_particle_desc *desc, const cm3_actor_particle_weights *weights,
                           const cm3_actor_particle_bufs *bufs, void *stream);

/* A whole policy-driven episode in ONE launch: for every tick, actor forward pass + sampling (as cm3_actor_particle_f32)
 * followed by the env step (as cm3_particle_step_f32), with the network weights, the observation tile and the env
 * state resident in LDS / registers.  Same trajectory layout and flags as cm3_particle_rollout_f32 (AUTO_RESET honoured;
 * GEN_ACTIONS rejected: the policy draws the actions, written to traj->actions).  probs (optional) receives the mixed
 * probabilities per tick, float [n_ticks][E][N][5] with probs_stride bytes between ticks.  n_agents in {1,2,4,8};
 * actor_desc and desc must agree on n_envs / n_agents / seed / env_id_base.  Bit-identical to alternating
 * cm3_actor_particle_f32 and cm3_particle_step_f32 launches. */
int cm3_policy_rollout_f32(const cm3_particle_desc *desc, const cm3_particle_traj *traj,
                           const cm3_actor_particle_desc *actor_desc, const cm3_actor_particle_weights *weights,
                           float *probs, size_t probs_stride, int32_t n_ticks, void *stream);
/* Test / measurement knob (ABI 6): 16-row tiles per workgroup of cm3_policy_rollout_f32 -- 1, 2 or 4 forces that build of the
 * kernel for every later launch of the process, 0 gives the choice back to the library's rule (by batch size).  Results do not
 * depend on it (tests/test_gpu_actor.py forces each build on one batch).  Initial value: the environment's CM3_POLICY_RT, read
 * once at the first launch. */
int cm3_policy_force_row_tiles(int32_t row_tiles);

/* ------------------------------------------------------------------------------------------
 * On-device Checkers actor: networks.convnet_1 + networks.actor_checkers (networks.py:67-75, :549-578) + the
 * epsilon-mixed categorical sampling of alg_credit_checkers.py:107-113 / run_actor :229-253, for all E*N agent rows in
 * one launch; every layer on the matrix cores (exact-f32 MFMA).  Weights are float32, shaped as the TF variables:
 *   conv_w [3][3][3][6] + conv_b [6]        "conv/Conv/weights", "conv/Conv/biases"  (3x3, stride 1, SAME, relu; NHWC)
 *   lin_w [150][32] + lin_b [32]            "conv_linear"
 *   self_w [43][256] + self_b [256]         "branch_self"  (input = concat(conv_linear 32, v_obs_self 4, a_prev 5, goal 2))
 *   w_self_h2 [256][256]                    "W_self_h2"
 *   others_w [2(N-1)][256] + others_b       "stage-2/branch_others"   (stage > 1 only)
 *   w_others_h2 [256][256]                  "stage-2/W_others_h2"
 *   b_h2 [256]                              "b"
 *   out_w [256][5] + out_b [5]              "actor_out"
 * Inputs are the env's own output buffers: obs_self_t int8 (env records of obs_self_t_stride bytes, agent i at byte
 * 75 i), obs_self_v double [E][N][4], obs_others double [E][N][2 max(N-1,1)] (cast to float32 like a TF feed), goals uint8
 * [E][N] (0 green / 1 orange -> one-hot), actions_prev int32 [E][N] (NULL = zeros, train_onpolicy.py:295), steps, episode.
 * Outputs: actions int32 [E][N] (what cm3_checkers_step consumes), optional mixed probabilities float [E][N][5].
 * ---------------------------------------------------------------------------------------- */
typedef struct cm3_actor_checkers_desc {
  int32_t n_envs, n_agents;
  int32_t stage;  /* 1: self branch only; 2: + others branch */
  int32_t n_obs;  /* 2 (5x5 window) */
  int32_t conv_f, n_conv_linear, n_h1, n_h2, n_actions; /* 6, 32, 256, 256, 5 */
  float epsilon;
  int32_t precision;         /* 0: float32 throughout (parity path).  1: the two 256x256 layers on the bf16 matrix cores with
                                float32 accumulation (first-layer activations and those weights rounded to bf16; probabilities
                                move by up to ~1e-2).  2: EVERY layer in split float16 -- activations and weights as float16
                                hi + lo, hi*hi + hi*lo + lo*hi on v_mfma_f32_16x16x32_f16, float32 accumulation: a parity
                                path (held to the 2e-5 of precision 0) in about half the time of precision 0 */
  int32_t obs_self_t_stride; /* bytes between env records of obs_self_t (cm3_checkers_desc.obs_self_t_stride) */
  int64_t env_id_base;
  uint64_t seed;
} cm3_actor_checkers_desc;

typedef struct cm3_actor_checkers_weights {
  const float *conv_w, *conv_b, *lin_w, *lin_b, *self_w, *self_b, *w_self_h2, *others_w, *others_b, *w_others_h2, *b_h2,
      *out_w, *out_b;
  const void *packed; /* written by cm3_actor_checkers_pack (cm3_actor_checkers_packed_bytes() bytes); the forward launch
                         reads ONLY this buffer */
} cm3_actor_checkers_weights;

typedef struct cm3_actor_checkers_bufs {
  const int8_t *obs_self_t;
  const double *obs_self_v;
  const double *obs_others;
  const uint8_t *goals;
  const int32_t *actions_prev; /* optional */
  const int32_t *steps;
  const int32_t *episode;
  int32_t *actions;
  float *probs; /* optional */
  const uint8_t *prev_done;  /* optional uint8 [E]: envs whose previous tick ended an episode (continuous collection with
                                CM3_FLAG_AUTO_RESET) feed actions_prev = zeros, as a fresh episode does (train_onpolicy.py:295) */
  const float *epsilon_dev;  /* optional device float read at launch instead of desc->epsilon (see cm3_actor_particle_bufs) */
} cm3_actor_checkers_bufs;

size_t cm3_actor_checkers_packed_bytes(void);
int cm3_actor_checkers_pack(const cm3_actor_checkers_desc *desc, const cm3_actor_checkers_weights *weights, void *packed,
                            void *stream);
int cm3_actor_checkers_f32(const cm3_actor_checkers_desc *desc, const cm3_actor_checkers_weights *weights,
                           const cm3_actor_checkers_bufs *bufs, void *stream);

/* A whole POLICY-DRIVEN Checkers rollout in ONE launch (ABI 7; csrc/policy_checkers.hip): per tick the actor above (precision 2),
 * epsilon-mixed sampling and Checkers.step with the actions just drawn -- train_onpolicy.py:309-347 / train_offpolicy.py:309-368 without
 * leaving the kernel; env state in registers, the network's inputs written straight into LDS, the others branch of the two-agent
 * network read from a table that cm3_actor_checkers_pack builds with the forward pass's own code.  `traj` as for
 * cm3_checkers_rollout (slot 0 of the observation arrays = the caller's current observation is NOT read: the live state is);
 * actions_prev0: optional int32 [E][N], actions_prev of tick 0 (NULL = zeros, train_onpolicy.py:295); later ticks use the actions just
 * taken, zeros after a tick that ended an episode under CM3_FLAG_AUTO_RESET; actions_prev_next: optional int32 [E][N] output, the
 * actions_prev of the NEXT rollout's first tick (a different buffer); probs: optional float [n_ticks][E][N][5] with
 * probs_stride BYTES between ticks; epsilon_dev: optional device float read at launch instead of actor->epsilon.
 * One or two agents (config_checkers_stage1 / stage2; actor->stage = n_agents), reference geometry, 4-byte padded records.
 * Every output equals, bit for bit, what n_ticks x (cm3_actor_checkers_f32, cm3_checkers_step) write. */
int cm3_policy_rollout_checkers(const cm3_checkers_desc *desc, const cm3_checkers_traj *traj, const cm3_actor_checkers_desc *actor,
                                const cm3_actor_checkers_weights *weights, const int32_t *actions_prev0, int32_t *actions_prev_next, float *probs,
                                size_t probs_stride, const float *epsilon_dev, const cm3_checkers_bufs *final_obs,
                                int32_t n_ticks, void *stream);

/* ------------------------------------------------------------------------------------------
 * Advantage normalisation (build-defined; the reference's advantage, alg_credit.py:334-357, is not normalised).
 * Discounted return-to-go over a time-major trajectory, G[t] = x[t] + gamma * (1 - done[t]) * G[t+1], G[T] = 0:
 *   x, out  real [T][E][C]   (C = N for reward_n, 1 for the team reward); out may alias x
 *   done    uint8 [T][E];  valid uint8 [T][E] optional (invalid entries: out = 0, excluded from the moments)
 *   scratch >= cm3_returns_scratch_bytes() bytes, ZERO-INITIALISED ONCE by the caller (per-block partials + an arrival
 *   counter; the last block to arrive folds the partials in block order and resets the counter for the
 *   next call on the same stream -- no memset launch per call);
 *   moments double[3] = (sum, sum of squares, count) of this rank's valid returns, computed deterministically.
 *   The host all-gathers the three numbers over the ranks (RCCL); cm3_normalize_* sums the n_parts triples in rank order
 *   and applies x = (x - (real)mean) / (real)(std + eps) with the GLOBAL moments (x real [n_elem], valid indexed by
 *   element / C; the same expression as the host helper cm3_amd.shard.normalize_advantages, bit for bit);
 *   stats (optional) receives (mean, std, count); apply = 0 only computes stats.
 * ---------------------------------------------------------------------------------------- */
size_t cm3_returns_scratch_bytes(void);
/* NOTE (scratch layout): cm3_returns_moments_* keeps its arrival ticket BEHIND 3 * 512 partial slots of `scratch`;
 * cm3_returns_normalize_segments_* with n_segments >= 2 uses that range for partials.  Give the two entry-point families SEPARATE
 * scratch buffers (cm3_returns_scratch_bytes / cm3_returns_segments_scratch_bytes): a shared one leaves a garbage ticket and the
 * moments silently stale. */
int cm3_returns_moments_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream);
int cm3_returns_moments_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                            double *moments, int32_t T, int32_t E, int32_t C, double gamma, void *stream);
int cm3_normalize_f32(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, double *stats, size_t n_elem,
                      int32_t C, double eps, int32_t apply, void *stream);
int cm3_normalize_f64(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, double *stats, size_t n_elem,
                      int32_t C, double eps, int32_t apply, void *stream);
/* ONE rank's whole advantage step (the all-gather is the identity) as TWO launches: cm3_returns_moments_* + cm3_normalize_* with
 * n_parts = 1 on its own moments, bit for bit.  Launch A computes the returns and per-block partial moments (short trajectories,
 * T <= 40: the whole column in one memory round trip), launch B folds the partials in block order in every block -- the fold of
 * cm3_returns_moments_*, so `moments` receives the same bits -- and normalises; no atomics, no arrival counter.
 * `shift` (optional) adds the slot bookkeeping of the rollout whose rewards these are to launch A, element by element:
 * first_dst[r] <- mid[r], then mid[r] <- last_src[r] (r < n <= 4; 16-byte aligned pointers and sizes).  With mid = the env's live
 * buffers, first_dst = slot 0 and last_src = slot T of a trajectory whose first tick READ the live buffers, this records the
 * initial state into slot 0 and leaves the final state in the live buffers (train_onpolicy.py:340-343 "state = next_state" across
 * rollouts) without launches of its own: a 33-tick rollout is a chain of dependent launches of ~2.6 us each, and its tail was four. */
typedef struct cm3_copy_shift {
  int32_t n;
  int32_t _pad;
  void *first_dst[4];
  void *mid[4];
  const void *last_src[4];
  size_t bytes[4];
} cm3_copy_shift;
int cm3_returns_normalize_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                              double *moments, double *stats, int32_t T, int32_t E, int32_t C, double gamma, double eps,
                              int32_t apply, const cm3_copy_shift *shift, void *stream);
int cm3_returns_normalize_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                              double *moments, double *stats, int32_t T, int32_t E, int32_t C, double gamma, double eps,
                              int32_t apply, const cm3_copy_shift *shift, void *stream);
/* The same two launches for n_segments consecutive rollouts of ONE collection phase (ABI 5; train_onpolicy.py:359: episodes_per_train
 * rollouts between two training steps): x / out are time-major over n_segments * T ticks, done / valid likewise; segment k = ticks
 * [k T, (k + 1) T) has its own returns (G = 0 beyond its last tick), moments[k][3] and stats[k][3] and is normalised with its own
 * statistics -- bit for bit what n_segments calls of cm3_returns_normalize_* on the slices give.  A phase of K rollouts is then ONE
 * chain of K T step launches + 2, and one hipGraph replay, instead of K chains of T + 2 and K replay boundaries.
 * scratch >= cm3_returns_segments_scratch_bytes(n_segments); `shift` is applied once (by segment 0's blocks).
 * apply = 0 leaves the raw returns in `out` and only fills moments / stats: the multi-rank form, followed by ONE all-gather of every
 * rank's moments[n_segments][3] and cm3_normalize_segments_* with parts = [n_parts ranks][n_segments][3]. */
size_t cm3_returns_segments_scratch_bytes(int32_t n_segments);
int cm3_returns_normalize_segments_f32(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                                       double *moments, double *stats, int32_t T, int32_t n_segments, int32_t E, int32_t C,
                                       double gamma, double eps, int32_t apply, const cm3_copy_shift *shift, void *stream);
int cm3_returns_normalize_segments_f64(const void *x, const uint8_t *done, const uint8_t *valid, void *out, void *scratch,
                                       double *moments, double *stats, int32_t T, int32_t n_segments, int32_t E, int32_t C,
                                       double gamma, double eps, int32_t apply, const cm3_copy_shift *shift, void *stream);
/* x: n_segments slabs of n_elem elements; parts [n_parts][n_segments][3]; stats (optional) [n_segments][3]. */
int cm3_normalize_segments_f32(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, int32_t n_segments, double *stats,
                               size_t n_elem, int32_t C, double eps, int32_t apply, void *stream);
int cm3_normalize_segments_f64(void *x, const uint8_t *valid, const double *parts, int32_t n_parts, int32_t n_segments, double *stats,
                               size_t n_elem, int32_t C, double eps, int32_t apply, void *stream);
/* Up to 8 device-to-device copies (16-byte aligned pointers and sizes) in ONE launch: trajectory slot <-> live env buffers
 * of the collection loop (train_onpolicy.py:340-343 "state = next_state" across rollouts). */
int cm3_copy_list(int32_t n, void *const *dst, const void *const *src, const size_t *bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Transition export and replay rings as single launches (ABI 6)
 * ---------------------------------------------------------------------------------------- */
/* The nine distinct columns of the reference's 11-field particle transition (alg/train_onpolicy.py:338; v_global == v_local ==
 * state, alg_credit.py:458-470), each contiguous over the B gathered transitions. */
typedef struct cm3_transition_cols {
  void *state;            /* float [B][N][4]            global_state / local_self */
  void *obs_others;       /* float [B][N][4 max(N-1,1)] */
  int32_t *actions;       /* int32 [B][N] */
  void *reward;           /* float [B] */
  void *reward_n;         /* float [B][N] */
  void *next_state;       /* float [B][N][4]   slot t + 1, or the captured terminal state where the env restarted in that launch */
  void *next_obs_others;  /* float [B][N][L]   likewise */
  uint8_t *done;          /* uint8 [B] */
  void *goals;            /* float [B][N][2] */
  int64_t ring_start;     /* ring_size > 0: the columns are REPLAY RINGS of ring_size rows and transition b is written to row */
  int64_t ring_size;      /* (ring_start + b) mod ring_size -- export and replay_buffer.add in one launch; 0: row b */
} cm3_transition_cols;
/* INDEX CONTRACT of cm3_rows_scatter / cm3_rows_gather / cm3_transitions_gather_f32: row indices (dst_row, src_row, tt, ee) are
 * NOT bounds-checked by the launches -- an index outside its array is an out-of-bounds device access.  The host classes that call
 * them (cm3_amd/replay.py RingIndex, rollout.py) derive every index from sizes they own; a caller of the C ABI must do the same.
 * n == 0 (or zero rows) returns CM3_OK without touching any pointer. */
/* Gathers transitions (tt[b], ee[b]), b < n, out of a time-major trajectory (the cm3_particle_traj the collector wrote: state
 * [T+1][N][E][4], obs_others [T+1][E][N][L], actions / reward_n [T][E][N], reward / done [T][E], optional term_* [T][...]) in ONE
 * launch.  goals: traj->goals with goals_stride bytes per slot (stride 0: one live array); goal_slot (optional, int32 with
 * goal_slot_stride bytes per tick, [T+1][E]) names the slot that holds the goals in effect at (t, e) -- sparse goal slots; NULL:
 * slot t itself.  tt == ee == NULL: ALL transitions of ticks [0, n / E) in time-major order (b = t E + e).  desc supplies n_envs /
 * n_agents only.  Replaces alg/replay_buffer.py's list-of-arrays + np.stack(batch[:, k]). */
int cm3_transitions_gather_f32(const cm3_particle_desc *desc, const cm3_particle_traj *traj, const int32_t *goal_slot,
                               size_t goal_slot_stride, const int64_t *tt, const int64_t *ee, int64_t n,
                               const cm3_transition_cols *out, void *stream);

/* Up to 16 columns of row-major records (row_bytes[k] bytes per row of column k), moved together. */
typedef struct cm3_row_cols {
  int32_t n_cols;
  int32_t reserved;
  void *dst[16];
  const void *src[16];
  uint32_t row_bytes[16];
} cm3_row_cols;
/* dst[k][row(b)] = src[k][b] for b < n_rows, every column in ONE launch.  row(b) = dst_row[b] when dst_row is given (a negative
 * entry skips the row: the two rings of replay_buffer_dual.py:12-38 take the same batch with complementary index arrays), else the
 * ring position (ring_start + b) mod ring_size (replay_buffer.py:11-16: overwrite the oldest; n_rows <= ring_size). */
int cm3_rows_scatter(const cm3_row_cols *cols, int64_t n_rows, const int64_t *dst_row, int64_t ring_start, int64_t ring_size,
                     void *stream);
/* dst[k][b] = src[k][src_row[b]] (replay_buffer.py:28-37 sample_batch: the sampled transitions as contiguous columns). */
int cm3_rows_gather(const cm3_row_cols *cols, int64_t n_rows, const int64_t *src_row, void *stream);

/* Tiling of per-agent rows into the feeds of the reference's train_step (ABI 6): process_actions / process_global_state
 * (alg/alg_credit.py:406-443, :528-557), the n x n credit repeats (:614-658) and the n x n x l_action counterfactual tiling
 * (:730-751).  Every output column is "destination row r <- source row f(r)":
 *   others_n == 0:  f(r) = ((r / div[0]) % mod[0]) * mul[0] + ((r / div[1]) % mod[1]) * mul[1]      (mod 0 = no remainder taken)
 *   others_n == N:  f(r) = (r / others_divq) * N + j,  j = k + (k >= n),  k = r % (N - 1),  n = (r / others_divn) % N
 *                   (the reference's `x[:, np.arange(N) != n]` selections, whatever repeats sit around them)
 * and what is written per element depends on `kind`.  Up to 16 columns in ONE launch. */
#define CM3_TILE_COPY 0         /* elements of elem_bytes (1, 4, 8) copied */
#define CM3_TILE_F32_TO_F64 1   /* float -> double (the reference's float64 np.zeros targets) */
#define CM3_TILE_ONEHOT_I64 2   /* source rows are ONE int32 each; destination rows elems_per_row int64: 1 at the value's index */
#define CM3_TILE_ONEHOT_F64 3   /* ... as double */
#define CM3_TILE_EYE_F64 4      /* np.tile(np.eye(elems_per_row)): row r has its 1.0 at r % elems_per_row (src unused) */
#define CM3_TILE_NOT_I64 5      /* int64 1 - (byte != 0):  -(done.astype(int) - 1), alg_credit.py:590 */
typedef struct cm3_tile_col {
  void *dst;
  const void *src;
  int64_t n_rows;           /* destination rows */
  uint32_t elems_per_row;   /* destination (= source, except the one-hot / eye kinds) elements per row */
  uint32_t kind;            /* CM3_TILE_* */
  uint32_t elem_bytes;      /* CM3_TILE_COPY only */
  uint32_t others_n, others_divq, others_divn;
  uint32_t div[2], mod[2], mul[2];
} cm3_tile_col;
int cm3_rows_tile(const cm3_tile_col *cols, int32_t n_cols, void *stream);
/* out[i] = (double)reward[i] + (gamma * q[i]) * (double)multiplier[i]  (ABI 7): the TD targets of the reference's train_step
 * (alg_credit.py:594, :640, :684: reward + gamma * Q_target * done_multiplier), NumPy's evaluation order, no contraction.
 * reward float32 (reward_is_f64 = 0) or float64 [n], q float64 [n], multiplier int64 [n] (0 / 1), out float64 [n]. */
int cm3_td_target_f64(const void *reward, int32_t reward_is_f64, const double *q, const int64_t *multiplier, double gamma, double *out,
                      int64_t n, void *stream);

/* ------------------------------------------------------------------------------------------
 * Measurement and launch plumbing
 * ---------------------------------------------------------------------------------------- */
/* Launch-structure floor probe: `blocks` x `threads` lanes read read_bytes (16-byte vectors, all loads first), then write
 * write_bytes whose value depends on what was read -- the load -> store skeleton of a step launch with no arithmetic. */
int cm3_traffic_floor_bench(const void *src, size_t read_bytes, void *dst, size_t write_bytes, int32_t blocks,
                            int32_t threads, void *stream);
/* Streaming 16-byte-per-lane read of `bytes` bytes (multiple of 16); writes one checksum word per
 * workgroup to sink (>= 4*cm3_hbm_bench_sink_words() bytes).  The measured read-bandwidth roofline. */
int cm3_hbm_read_bench(const void *buf, size_t bytes, void *sink, void *stream);
int cm3_hbm_bench_sink_words(void);
/* The same probe with an explicit configuration (independent 16-byte loads in flight per lane: 1/2/4/8/16; workgroups of
 * 256 lanes per CU: 1..32; nt != 0: non-temporal loads) -- tools/hbm_probe_sweep.py picks cm3_hbm_read_bench's. */
int cm3_hbm_read_bench_cfg(const void *buf, size_t bytes, void *sink, int32_t unroll, int32_t wg_per_cu, int32_t nt,
                           void *stream);
/* Streaming 16-byte-per-lane copy of `bytes` bytes (the pattern behind the guide's 6.29 TB/s "float4 copy" figure);
 * the bandwidth is 2 * bytes / time.  unroll 1/2/4/8. */
int cm3_hbm_copy_bench(void *dst, const void *src, size_t bytes, void *stream);
int cm3_hbm_copy_bench_cfg(void *dst, const void *src, size_t bytes, int32_t unroll, int32_t wg_per_cu, int32_t nt,
                           void *stream);

/* hipGraph capture of whatever is enqueued on `stream` between begin and end (launch-bound inner
 * loops: 33 ticks per replay). */
int cm3_graph_begin(void *stream);
int cm3_graph_end(void *stream, void **graph_exec);
int cm3_graph_launch(void *graph_exec, void *stream);
int cm3_graph_destroy(void *graph_exec);

/* HIP events on the caller's stream (bench.py times the kernel on the stream it is launched on). */
int cm3_event_create(void **event);
int cm3_event_record(void *event, void *stream);
int cm3_event_synchronize(void *event);
int cm3_event_elapsed_ms(void *start, void *stop, float *ms);
int cm3_event_destroy(void *event);
int cm3_stream_synchronize(void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CM3_AMD_H */
