// Checkers env (reference: /root/reference/env/checkers.py) for E environments per launch, gfx950.
//
// Live state per env is compact -- a 64-bit "collected" mask over the n_rows x n_columns reward cells,
// one packed word per agent (r, c, #green, #orange) and a step counter -- instead of the reference's
// dense float64 world[total_rows][total_columns][3] (checkers.py:267).  Everything the reference reads
// from the dense world is a pure function of that state:
//   channel 0/1 (green/orange): cell (k, j) of the reward band is green iff (k + j) is even
//                               (populate_world :54-63); value -1 uncollected, +1 collected (:204-222)
//   channel 2 (invalid):        +1 on the static wall pattern (:43-46), -1 where an agent stands
//                               (:49-51, :168-183), 0 elsewhere.
// (Agents can never share a cell: agent_act refuses occupied targets; distinct start cells are enforced
// on the host.)
//
// Mapping: one lane per env; agents act SEQUENTIALLY in index order inside the lane, because agent i+1
// must see agent i's move and pick-up (checkers.py:233-237).  Integer-valued outputs are written as
// integers (int8 / int32), the normalised ones with the reference's own float64 expressions, so every
// output is value-identical to the reference (bit-exact parity).  The two byte-granular env-major
// outputs (grid, obs_self_t) are assembled in a wave-private LDS tile and written out as contiguous
// 16-byte vectors.
#include "common.h"
#include "philox.h"

namespace cm3 {

struct CheckersParams {
  int E, R, C, O, TR, TC, K, max_steps;
  uint32_t flags;
  int max_collectible;
  int64_t env_id_base;
  uint64_t seed;
  uint64_t green_mask, orange_mask;
  int start_r[CM3_MAX_AGENTS], start_c[CM3_MAX_AGENTS];  // expanded coordinates
  uint64_t *mask;
  uint32_t *agents;
  int32_t *steps;
  int32_t *episode;
  uint8_t *goals;
  const uint32_t *ablock;   // optional uint32 [E][4]: stage 1 of the action stream per env (cm3_checkers_action_blocks); NULL: computed here
  int32_t *actions;
  int8_t *grid;
  int32_t *vec;
  double *obs_others;
  int8_t *obs_self_t;
  double *obs_self_v;
  double *local_rewards;
  double *reward;
  uint8_t *done;
  // optional: the TRUE post-step observation of envs re-initialised by CM3_FLAG_AUTO_RESET in this launch (the next_* columns
  // of the terminal transition, stored by the reference before it resets: train_onpolicy.py:336-347)
  int8_t *term_grid;
  int32_t *term_vec;
  double *term_obs_others;
  int8_t *term_obs_self_t;
  double *term_obs_self_v;
  uint8_t *goals_next;  // optional uint8 [E][N]: the goals in effect AFTER this tick (they change only for N == 1 restarts)
  const uint8_t *reset_mask;
  int grid_rec, obst_rec;        // payload bytes per env of grid / obs_self_t
  int grid_stride, obst_stride;  // bytes between consecutive env records (>= payload)
  // the observation of a FRESH episode (fast kernel, N <= 2; filled by the host, ck_fresh_fill): see CkFresh
  alignas(16) uint32_t fresh[104];
  // tick loop inside one launch (CM3_FLAG_FUSED_TICKS, fast kernel only): tick t uses <pointer> + t * <stride in
  // bytes>; the observation pointers then address slot 1 of their trajectories.  n_ticks == 1 with zero strides
  // is the plain one-launch-per-tick step.
  int n_ticks;
  int _pad2;
  size_t st_actions, st_grid, st_vec, st_obs_others, st_obs_self_t, st_obs_self_v, st_local, st_reward, st_done;
  size_t st_term_grid, st_term_vec, st_term_obs_others, st_term_obs_self_t, st_term_obs_self_v, st_goals_next;
  CM3_SPAN_FIELD  // (span build only, common.h)
};

// the five observation arrays one emit writes (the trajectory slot of a tick, or the terminal-capture slot)
struct CkOut {
  int8_t *grid;
  int32_t *vec;
  double *obs_others;
  int8_t *obs_self_t;
  double *obs_self_v;
};

template <typename T> __device__ __forceinline__ T *ck_tick_ptr(T *base, size_t stride, int t) {
  return reinterpret_cast<T *>(reinterpret_cast<char *>(base) + stride * (size_t)t);
}

// internal flag bits set by ck_rollout only (never part of the ABI)
constexpr uint32_t kCkObsStoreNt = 0x100000u;  // internal: non-temporal observation stores (set by ck_rollout only)

constexpr int kCkLdsBytes = 40960;  // per-wave staging tile (64 rows x up to 640 bytes)

__device__ __forceinline__ bool ck_wall(const CheckersParams &p, int r, int c) {
  return c < p.O || r < p.O || r >= p.O + p.R || c >= p.O + p.C + 1;
}

template <int N> struct CkState {
  uint64_t mask;
  int r[N], c[N], ng[N], no[N];
};

template <int N> __device__ __forceinline__ int ck_ch2(const CheckersParams &p, const CkState<N> &s, int r, int c) {
  if (ck_wall(p, r, c)) return 1;
#pragma unroll
  for (int j = 0; j < N; ++j)
    if (s.r[j] == r && s.c[j] == c) return -1;
  return 0;
}

// value of channels 0 and 1 at expanded cell (r, c)
template <int N>
__device__ __forceinline__ void ck_ch01(const CheckersParams &p, const CkState<N> &s, int r, int c, int &g, int &o) {
  const int k = r - p.O, j = c - p.O;
  g = 0;
  o = 0;
  if (k >= 0 && k < p.R && j >= 0 && j < p.C) {
    const int v = ((s.mask >> (k * p.C + j)) & 1ull) ? 1 : -1;
    if (((k + j) & 1) == 0) g = v; else o = v;
  }
}

__device__ __forceinline__ void ck_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// contiguous copy of the wave's `nbytes` staged bytes to global memory (dst 16-byte aligned)
__device__ __forceinline__ void ck_copy_out(const int8_t *lds, int8_t *dst, int nbytes, int lane) {
  const int nvec = nbytes >> 4;
  const uint4 *src4 = reinterpret_cast<const uint4 *>(lds);
  uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
  for (int f = lane; f < nvec; f += 64) dst4[f] = src4[f];
  for (int b = (nvec << 4) + lane; b < nbytes; b += 64) dst[b] = lds[b];
}

template <int N>
__device__ __forceinline__ void ck_emit_vectors(const CheckersParams &p, const CkState<N> &s, size_t e, const CkOut &out);

// Terminal capture of the generic kernel: one lane writes its env's whole observation straight to global memory (rare:
// only envs whose episode ended this tick under CM3_FLAG_AUTO_RESET).
template <int N>
__device__ __forceinline__ void ck_emit_direct(const CheckersParams &p, const CkState<N> &s, size_t e, const CkOut &out) {
  int8_t *grow = out.grid + e * (size_t)p.grid_stride;
  for (int k = 0; k < p.R; ++k)
    for (int j = 0; j <= p.C; ++j) {
      int g, o;
      ck_ch01<N>(p, s, k + p.O, j + p.O, g, o);
      const int off = (k * (p.C + 1) + j) * 2;
      grow[off] = (int8_t)g;
      grow[off + 1] = (int8_t)o;
    }
  int8_t *orow = out.obs_self_t + e * (size_t)p.obst_stride;
#pragma unroll
  for (int i = 0; i < N; ++i)
    for (int dr = 0; dr < p.K; ++dr)
      for (int dc = 0; dc < p.K; ++dc) {
        const int rr = s.r[i] - p.O + dr, cc = s.c[i] - p.O + dc;
        int g, o;
        ck_ch01<N>(p, s, rr, cc, g, o);
        int inv = ck_ch2<N>(p, s, rr, cc);
        if (dr == p.O && dc == p.O) inv = 0;
        int8_t *q = orow + ((i * p.K + dr) * p.K + dc) * 3;
        q[0] = (int8_t)g;
        q[1] = (int8_t)o;
        q[2] = (int8_t)inv;
      }
  ck_emit_vectors<N>(p, s, e, out);
}

template <int N>
__device__ __forceinline__ void ck_emit(const CheckersParams &p, const CkState<N> &s, int8_t *lds, int lane, size_t e0,
                                        size_t e, bool active) {
  long rows_here = (long)p.E - (long)e0;
  rows_here = rows_here < 0 ? 0 : (rows_here > 64 ? 64 : rows_here);

  // ---- grid [R][C+1][2] int8 (get_valid_grid :66-76) -------------------------------------------------
  {
    int8_t *row = lds + lane * p.grid_rec;
    for (int k = 0; k < p.R; ++k)
      for (int j = 0; j <= p.C; ++j) {
        int g, o;
        ck_ch01<N>(p, s, k + p.O, j + p.O, g, o);
        const int off = (k * (p.C + 1) + j) * 2;
        *reinterpret_cast<int16_t *>(row + off) = (int16_t)((g & 0xff) | ((o & 0xff) << 8));
      }
    ck_wave_sync();
    ck_copy_out(lds, p.grid + e0 * (size_t)p.grid_rec, (int)rows_here * p.grid_rec, lane);
    ck_wave_sync();
  }
  // ---- obs_self_t [N][K][K][3] int8 (get_obs :97-109) ------------------------------------------------
  {
    int8_t *row = lds + lane * p.obst_rec;
#pragma unroll
    for (int i = 0; i < N; ++i)
      for (int dr = 0; dr < p.K; ++dr)
        for (int dc = 0; dc < p.K; ++dc) {
          const int rr = s.r[i] - p.O + dr, cc = s.c[i] - p.O + dc;
          int g, o;
          ck_ch01<N>(p, s, rr, cc, g, o);
          int inv = ck_ch2<N>(p, s, rr, cc);
          if (dr == p.O && dc == p.O) inv = 0;  // the agent's own cell is valid (:105-107)
          int8_t *q = row + ((i * p.K + dr) * p.K + dc) * 3;
          q[0] = (int8_t)g;
          q[1] = (int8_t)o;
          q[2] = (int8_t)inv;
        }
    ck_wave_sync();
    ck_copy_out(lds, p.obs_self_t + e0 * (size_t)p.obst_rec, (int)rows_here * p.obst_rec, lane);
    ck_wave_sync();
  }
  if (!active) return;
  CkOut o;
  o.vec = p.vec;
  o.obs_others = p.obs_others;
  o.obs_self_v = p.obs_self_v;
  ck_emit_vectors<N>(p, s, e, o);
}

// ---- vec (:79-94), obs_self_v / obs_others (normalize :112-125, :128-154) of one env --------------------------
template <int N>
__device__ __forceinline__ void ck_emit_vectors(const CheckersParams &p, const CkState<N> &s, size_t e, const CkOut &out) {
  constexpr int NO = N > 1 ? N - 1 : 1;
  double nr[N], nc[N];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    nr[i] = ((double)s.r[i] - (double)p.TR / 2.0) / (double)p.TR;
    nc[i] = ((double)s.c[i] - (double)p.TC / 2.0) / (double)p.TC;
  }
  const double half = (double)p.max_collectible / 2.0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    int4 v;
    v.x = s.r[i];
    v.y = s.c[i];
    v.z = s.ng[i];
    v.w = s.no[i];
    reinterpret_cast<int4 *>(out.vec)[e * N + i] = v;
    double4 sv;
    sv.x = nr[i];
    sv.y = nc[i];
    sv.z = (double)s.ng[i] / half;
    sv.w = (double)s.no[i] / half;
    reinterpret_cast<double4 *>(out.obs_self_v)[e * N + i] = sv;
    double2 *oo = reinterpret_cast<double2 *>(out.obs_others) + (e * N + i) * NO;
#pragma unroll
    for (int k = 0; k < NO; ++k) {
      const int j = (N > 1) ? (k < i ? k : k + 1) : 0;
      double2 t;
      t.x = nr[j];
      t.y = nc[j];
      oo[k] = t;
    }
  }
}

// the pointers / scalars the first loads of a step launch need; the fast kernel receives them as LEADING scalar kernel arguments
// (preloaded into SGPRs at wave launch, -mllvm -amdgpu-kernarg-preload-count) so these loads do not wait for a kernarg fetch
struct CkHead {
  const uint64_t *mask;
  const uint32_t *agents;
  const int32_t *steps;
  const int32_t *episode;
  const uint8_t *goals;
  const uint32_t *ablock;
  int E;
  uint32_t flags;
  int64_t env_id_base;     // (only read when ablock is NULL and the actions are drawn in-kernel)
  uint64_t seed;
};
__device__ __forceinline__ CkHead ck_head(const CheckersParams &p) {
  CkHead h;
  h.mask = p.mask; h.agents = p.agents; h.steps = p.steps; h.episode = p.episode; h.goals = p.goals; h.ablock = p.ablock; h.E = p.E;
  h.flags = p.flags; h.env_id_base = p.env_id_base; h.seed = p.seed;
  return h;
}

template <int N> __device__ __forceinline__ void ck_load(const CkHead &p, size_t ec, CkState<N> &s) {
  s.mask = p.mask[ec];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const uint32_t w = p.agents[(size_t)i * p.E + ec];
    s.r[i] = (int)(w & 0xff);
    s.c[i] = (int)((w >> 8) & 0xff);
    s.ng[i] = (int)((w >> 16) & 0xff);
    s.no[i] = (int)((w >> 24) & 0xff);
  }
}

template <int N> __device__ __forceinline__ void ck_store(const CheckersParams &p, size_t e, const CkState<N> &s) {
  p.mask[e] = s.mask;
#pragma unroll
  for (int i = 0; i < N; ++i)
    p.agents[(size_t)i * p.E + e] =
        (uint32_t)s.r[i] | ((uint32_t)s.c[i] << 8) | ((uint32_t)s.ng[i] << 16) | ((uint32_t)s.no[i] << 24);
}

// reset (:265-291): empty mask, agents on their start cells; N == 1 starts on row 0 or 2 by goal (:271-276)
template <int N> __device__ __forceinline__ void ck_init(const CheckersParams &p, const uint8_t (&goal)[N], CkState<N> &s) {
  s.mask = 0;
#pragma unroll
  for (int i = 0; i < N; ++i) {
    s.r[i] = p.start_r[i];
    s.c[i] = p.start_c[i];
    s.ng[i] = 0;
    s.no[i] = 0;
  }
  if (N == 1) s.r[0] = (goal[0] == 0 ? 0 : 2) + p.O;
}

// Live per-env scalars that travel with the compact state across the ticks of a launch.
template <int N> struct CkLive {
  int steps;
  uint32_t episode, episode_in;
  uint8_t goal[N];
  uint32_t aword[N];   // stage 1 of the in-kernel action stream: this env's Philox words, one per agent (philox.h)
};

template <int N>
__device__ __forceinline__ void ck_load_env(const CkHead &p, size_t ec, CkState<N> &s, CkLive<N> &lv) {
  ck_load<N>(p, ec, s);
  lv.steps = p.steps[ec];
#pragma unroll
  for (int i = 0; i < N; ++i) lv.goal[i] = p.goals[ec * N + i];
  lv.episode = 0;
  if (p.flags & (CM3_FLAG_GEN_ACTIONS | CM3_FLAG_AUTO_RESET)) lv.episode = (uint32_t)p.episode[ec];
  lv.episode_in = lv.episode;
#pragma unroll
  for (int i = 0; i < N; ++i) lv.aword[i] = 0u;
  if (p.flags & CM3_FLAG_GEN_ACTIONS) {
    // Round 5: the two-stage action stream of the particle kernels here too.  Stage 1 (a Philox block per env: a constant of
    // (seed, global env id)) comes from the env's table -- ONE more load in the round trip of the state loads -- and only the
    // ten-instruction counter mix of stage 2 follows them; the one-stage draw put ten Philox rounds (~550 cycles of a 4500-cycle
    // wave at C3) behind the loads, and computing stage 1 in the prologue instead measured SLOWER in round 4 (the prologue is
    // instruction-bound: profiles/r04_two_stage_action_stream.txt).  Without a table (ablock == NULL) the block is computed here.
    auto put = [&](int idx, uint32_t v) {
      if (idx < N) lv.aword[idx < N ? idx : 0] = v;
    };
    const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
    if constexpr (N <= 2) {
      uint2 w;
      if (p.ablock) {
        w = reinterpret_cast<const uint2 *>(p.ablock)[2 * ec];
      } else {
        const u32x4 b = action_block(p.seed, genv, 0u);
        w.x = b.x;
        w.y = b.y;
      }
      put(0, w.x);
      put(1, w.y);
    } else {
#pragma unroll
      for (int c = 0; c < (N + 3) / 4; ++c) {   // (agents 4 .. 7 have a block of their own: the table holds the first, the second is computed)
        const u32x4 w = (c == 0 && p.ablock) ? *reinterpret_cast<const u32x4 *>(p.ablock + 4 * ec) : action_block(p.seed, genv, (uint32_t)c);
        put(4 * c + 0, w.x);
        put(4 * c + 1, w.y);
        put(4 * c + 2, w.z);
        put(4 * c + 3, w.w);
      }
    }
  }
}

template <int N>
__device__ __forceinline__ void ck_store_env(const CheckersParams &p, size_t e, const CkState<N> &s, const CkLive<N> &lv) {
  ck_store<N>(p, e, s);
  p.steps[e] = lv.steps;
  if (lv.episode != lv.episode_in) p.episode[e] = (int32_t)lv.episode;
}

// One tick of one env (checkers.py:228-262), executed by every lane that maps to env `ec`; `writer` selects the
// single lane that performs the per-env stores of tick t.  Leaves the POST-STEP state in `s` and returns whether the
// episode ended under CM3_FLAG_AUTO_RESET: the caller then captures the terminal observation (term_*) and calls
// ck_restart_env.
// FAST: the reference geometry (3 x 8 band, n_obs 2) as compile-time constants instead of kernel arguments.
// GIVEN: the actions are the caller's registers (the whole-episode policy kernel, policy_checkers.hip, which has just sampled and
// stored them) instead of the tick's action slot.
template <int N, bool FAST = false, bool GIVEN = false>
__device__ __forceinline__ bool ck_tick_env(const CheckersParams &p, int t, size_t e, size_t ec, bool writer, CkState<N> &s,
                                            CkLive<N> &lv, const char *fast_tab = nullptr
#ifdef CM3_SPAN_MARKS
                                            , unsigned long long *_span_mk = nullptr
#endif
                                            , const int *given = nullptr) {
  const int gO = FAST ? 2 : p.O, gR = FAST ? 3 : p.R, gC = FAST ? 8 : p.C;
  const int g_collectible = FAST ? 24 : p.max_collectible;
  const bool active = writer;
  const int g_max_steps = p.max_steps;
  int steps = lv.steps;
  uint32_t episode = lv.episode;
  uint8_t goal[N];
#pragma unroll
  for (int i = 0; i < N; ++i) goal[i] = lv.goal[i];
  int act[N];
  int32_t *actions_t = ck_tick_ptr(p.actions, p.st_actions, t);
  if constexpr (GIVEN) {
#pragma unroll
    for (int i = 0; i < N; ++i) act[i] = given[i];
  } else if (p.flags & CM3_FLAG_GEN_ACTIONS) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      act[i] = rand5(action_word(lv.aword[i], episode, (uint32_t)steps));   // stage 2 (philox.h); stage 1 came with the state loads
      if (active) {
        if constexpr (FAST) *at32<int32_t>(actions_t, ((uint32_t)e * N + i) * 4u) = act[i];
        else actions_t[e * N + i] = act[i];
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      if constexpr (FAST) act[i] = *at32<const int32_t>(actions_t, ((uint32_t)ec * N + i) * 4u);
      else act[i] = actions_t[ec * N + i];
    }
  }
  CM3_STAMP(2, false);
  CM3_SPAN_MARK(3, false);  // actions drawn

  // ---- agents act in index order (step :233-237) ---------------------------------------------------------
  double local[N];
  if constexpr (FAST) {
    // Straight-line version for the reference geometry (round 3).  The if / else form below compiles to ~10 exec-mask regions per
    // agent; by the marked two-stamp build the draw + act phase took 1935 of the wave's 4700 cycles for ~235 instructions.  Same
    // semantics, every decision a select:
    //   move (agent_act :157-187): target = cell + delta(action) (delta 0 for "stay" AND for an out-of-range action); a move happens iff
    //   the target is inside the band and no agent stands on it -- the agent's own cell is occupied by itself, so delta 0 never moves;
    //   penalty -0.1 iff action != 0 and no move.  reward (get_reward :190-225) as below.  local = penalty + reward comes from a
    //   six-entry table of those very float64 sums (kCkBoardTab.norm[34..39], in the wave's LDS copy).
    uint32_t m = (uint32_t)s.mask;
    uint32_t rcp[N];
#pragma unroll
    for (int j = 0; j < N; ++j) rcp[j] = (uint32_t)s.r[j] | ((uint32_t)s.c[j] << 8);
#pragma unroll
    for (int i = 0; i < N; ++i) {
      const uint32_t a = (uint32_t)act[i];
      const uint32_t sh = (a < 7u ? a : 7u) * 8u;
      const int dr = __builtin_amdgcn_sbfe((int)(uint32_t)(0x0000000001ff00ull >> sh), 0, 8);   // action 1: -1, 2: +1
      const int dc = __builtin_amdgcn_sbfe((int)(uint32_t)(0x00000001ff000000ull >> sh), 0, 8);   // action 3: -1, 4: +1
      const int tr = s.r[i] + dr, tc = s.c[i] + dc;
      const uint32_t trc = (uint32_t)tr | ((uint32_t)tc << 8);
      bool blocked = ((unsigned)(tr - 2) >= 3u) | ((unsigned)(tc - 2) >= 9u);
#pragma unroll
      for (int j = 0; j < N; ++j) blocked = blocked | (rcp[j] == trc);
      s.r[i] = blocked ? s.r[i] : tr;
      s.c[i] = blocked ? s.c[i] : tc;
      rcp[i] = blocked ? rcp[i] : trc;
      const uint32_t pen = (uint32_t)((a != 0u) & blocked);
      const int k = s.r[i] - 2, j = s.c[i] - 2;
      const uint32_t bit = 1u << ((k * 8 + j) & 31);
      const bool fresh = ((unsigned)k < 3u) & ((unsigned)j < 8u) & ((m & bit) == 0u);
      m |= fresh ? bit : 0u;
      const uint32_t colour = (uint32_t)(k + j) & 1u;
      s.ng[i] += (int)(fresh & (colour == 0u));
      s.no[i] += (int)(fresh & (colour != 0u));
      const uint32_t kind = fresh ? (colour == (uint32_t)goal[i] ? 1u : 2u) : 0u;
      local[i] = *reinterpret_cast<const double *>(fast_tab + 448u + 8u * (34u + 3u * pen + kind));
    }
    s.mask = (uint64_t)m;
  } else {
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const int a = act[i];
    double penalty = 0.0;
    if (a != 0) {  // agent_act :157-187
      bool moved = false;
      if (a >= 1 && a <= 4) {
        const int tr = s.r[i] + (a == 1 ? -1 : (a == 2 ? 1 : 0));
        const int tc = s.c[i] + (a == 3 ? -1 : (a == 4 ? 1 : 0));
        bool free_cell;
        if constexpr (FAST) {
          const bool wall = tc < 2 || tr < 2 || tr >= 5 || tc >= 11;
          bool agent = false;
          const int trc = tr | (tc << 8);
#pragma unroll
          for (int j = 0; j < N; ++j) agent = agent | ((s.r[j] | (s.c[j] << 8)) == trc);
          free_cell = !wall && !agent;
        } else {
          free_cell = ck_ch2<N>(p, s, tr, tc) == 0;
        }
        if (free_cell) {
          s.r[i] = tr;
          s.c[i] = tc;
          moved = true;
        }
      }
      if (!moved) penalty = -0.1;
    }
    double rew = 0.0;  // get_reward :190-225
    const int k = s.r[i] - gO, j = s.c[i] - gO;
    if (k >= 0 && k < gR && j >= 0 && j < gC) {
      bool fresh;
      if constexpr (FAST) {  // 24 reward cells: the collected mask fits 32 bits
        const uint32_t bit = 1u << (k * gC + j), m = (uint32_t)s.mask;
        fresh = !(m & bit);
        if (fresh) s.mask = (uint64_t)(m | bit);
      } else {
        const uint64_t bit = 1ull << (k * gC + j);
        fresh = !(s.mask & bit);
        if (fresh) s.mask |= bit;
      }
      if (fresh) {
        const int colour = (k + j) & 1;
        if (colour == 0) s.ng[i] += 1; else s.no[i] += 1;
        rew = (colour == (int)goal[i]) ? 1.0 : -0.5;
      }
    }
    local[i] = penalty + rew;
  }
  }
  double total = local[0];  // np.sum(local_rewards) :243 (left to right for n < 8)
  if constexpr (N < 8) {
#pragma unroll
    for (int i = 1; i < N; ++i) total = total + local[i];
  } else {
    total = ((local[0] + local[1]) + (local[2] + local[3])) + ((local[4] + local[5]) + (local[6] + local[7]));
  }
  steps += 1;
  CM3_STAMP(3, false);
  CM3_SPAN_MARK(4, false);  // agents have acted
  bool done;  // :246-260
  if (steps == g_max_steps) {
    done = true;
  } else if (N == 1) {
    const uint64_t want = goal[0] == 0 ? p.green_mask : p.orange_mask;
    done = (s.mask & want) == want;
  } else {
    done = (FAST ? (int)__popc((uint32_t)s.mask) : (int)__popcll(s.mask)) == g_collectible;
  }
  if (active) {
    double *local_t = ck_tick_ptr(p.local_rewards, p.st_local, t);
    if constexpr (FAST) {
#pragma unroll
      for (int i = 0; i < N; ++i) *at32<double>(local_t, ((uint32_t)e * N + i) * 8u) = local[i];
      *at32<double>(ck_tick_ptr(p.reward, p.st_reward, t), (uint32_t)e * 8u) = total;
      *at32<uint8_t>(ck_tick_ptr(p.done, p.st_done, t), (uint32_t)e) = done ? 1 : 0;
    } else {
#pragma unroll
      for (int i = 0; i < N; ++i) local_t[e * N + i] = local[i];
      ck_tick_ptr(p.reward, p.st_reward, t)[e] = total;
      ck_tick_ptr(p.done, p.st_done, t)[e] = done ? 1 : 0;
    }
  }
  lv.steps = steps;
  return (p.flags & CM3_FLAG_AUTO_RESET) && done;
}

// Same-launch re-initialisation of an env whose episode ended (train_onpolicy.py:282 folded into :321).
template <int N>
__device__ __forceinline__ void ck_restart_env(const CheckersParams &p, size_t e, size_t ec, bool writer, CkState<N> &s,
                                               CkLive<N> &lv) {
  lv.episode += 1;
  if (N == 1) {  // train_onpolicy.py:288-291: a fresh random goal per episode
    const uint64_t genv = (uint64_t)(p.env_id_base + (int64_t)ec);
    lv.goal[0] = (uint8_t)(reset_words(p.seed, genv, lv.episode, 0).x & 1u);
    if (writer) p.goals[e] = lv.goal[0];
  }
  ck_init<N>(p, lv.goal, s);
  lv.steps = 0;
}

// plain one-tick step of the generic kernel: load, tick 0, terminal capture + restart, store
template <int N>
__device__ __forceinline__ void ck_step_env(const CheckersParams &p, size_t e, size_t ec, bool writer, CkState<N> &s) {
  CkLive<N> lv;
  ck_load_env<N>(ck_head(p), ec, s, lv);
  if (ck_tick_env<N>(p, 0, e, ec, writer, s, lv)) {
    if (writer && p.term_grid) {
      CkOut o;
      o.grid = p.term_grid;
      o.vec = p.term_vec;
      o.obs_others = p.term_obs_others;
      o.obs_self_t = p.term_obs_self_t;
      o.obs_self_v = p.term_obs_self_v;
      ck_emit_direct<N>(p, s, e, o);
    }
    ck_restart_env<N>(p, e, ec, writer, s, lv);
  }
  if (writer) {
    ck_store_env<N>(p, e, s, lv);
    if (p.goals_next) {
#pragma unroll
      for (int i = 0; i < N; ++i) p.goals_next[e * N + i] = lv.goal[i];
    }
  }
}


// ---- fast path: the reference geometry (3 x 8 band, n_obs 2), G lanes per env ------------------------------
// The generic kernel above is latency-bound at the BASELINE batch (8192 envs = 128 waves, each lane filling
// ~200 bytes one at a time).  Here an env owns G = 8 consecutive lanes: every lane redundantly runs the (short)
// sequential state update -- lanes of a wave execute in lockstep, so that costs nothing and needs no shuffle --
// and then produces a 1/G share of the env's outputs: dwords g, g + G, ... of the (4-byte padded) grid record and of
// the padded obs_self_t record, and one of the small vector outputs.  No LDS; a wave stores 64/G whole env records
// contiguously.
// Lanes per env G.  In-place stepping (re-used buffers), measured at C3 (8192 envs, us per tick / fused env-steps/s) in round 1:
// 2: 8.3 / 1.3e9, 4: 6.7 / 1.7e9, 8: 5.1 / 2.5e9, 16: 5.7 / 2.4e9, 32: 7.6 / 1.5e9; round 2 (profiles/
// r02_checkers_lanes_per_env_sweep.txt, three alternating rounds): in place 4: 6.42, 8: 5.05, 16: 5.48 -- but for a streaming-size
// trajectory (every tick its own slot, non-temporal stores) 4: 7.95, 8: 6.22, 16: 5.87.  So G = 8 everywhere except on the
// non-temporal path, which used G = 16 (the round-1 tuning was specific to in-place stepping) -- until the emit was re-cut along
// lane lines at the end of round 2: since then 8 lanes win there too (streaming trajectory 8: 4.22, 16: 4.30; in place 8: 3.87,
// 16: 4.16, 32: 5.4), so both paths use 8.
#ifndef CM3_CK_G
#define CM3_CK_G 8
#endif
#ifndef CM3_CK_G_STREAM
#define CM3_CK_G_STREAM 8   // (16 until the emit was re-cut along lane lines; re-measured after that: 8 lanes 4.22 us, 16 lanes 4.30)
#endif
constexpr int kCkG = CM3_CK_G, kCkGStream = CM3_CK_G_STREAM;  // (macros: build variants for tools/ab_builds.sh style comparisons)
template <int N, int G_ = kCkG> struct CkFast {
  static constexpr int R = 3, C = 8, O = 2, K = 5, TR = 7, TC = 13;
  static constexpr int GRID_REC = R * (C + 1) * 2;  // 54
  static constexpr int OBST_REC = N * K * K * 3;    // 75 N
  static constexpr int G = G_;
  static constexpr int EPW = 64 / G;
};

// One window / grid cell as packed bytes, branch-free.  (A first version decoded and evaluated every output BYTE with
// nested ifs: 1022 VALU instructions and ~50 exec-mask branches per wave.)
//   value of channel 0 (green) / 1 (orange) at expanded cell (rr, cc): 0 outside the 3 x 8 band or on the other colour,
//   else +1 collected / -1 still there (checkers.py:54-63, :203-224); the band cell (kk, jj) is green iff kk + jj is even.
template <int N> __device__ __forceinline__ uint32_t ckf_colour16(uint32_t m32, int kk, int jj) {
  using F = CkFast<N>;
  const bool inband = (unsigned)kk < (unsigned)F::R && (unsigned)jj < (unsigned)F::C;
  const uint32_t bit = (m32 >> ((kk * F::C + jj) & 31)) & 1u;
  const uint32_t sgn = bit ? 0x01u : 0xffu;          // +1 / -1 as a byte
  const uint32_t par = (uint32_t)(kk + jj) & 1u;     // 0 green -> byte 0, 1 orange -> byte 1
  return inband ? (sgn << (8u * par)) : 0u;
}

// the three channel bytes (ch0 | ch1 << 8 | ch2 << 16) of global window cell cg = 25 i + 5 dr + dc of agent i's 5 x 5
// window (get_obs, checkers.py:97-109); cells past the last agent's record (padding) are 0
template <int N> __device__ __forceinline__ uint32_t ckf_cell3(const CkState<N> &s, uint32_t m32, int cg) {
  using F = CkFast<N>;
  constexpr int KK = F::K * F::K;
  const int i = cg / KK, cell = cg - KK * i;
  const int dr = cell / F::K, dc = cell - F::K * dr;
  int ar = s.r[0], ac = s.c[0];
#pragma unroll
  for (int a = 1; a < N; ++a) {
    ar = (i == a) ? s.r[a] : ar;
    ac = (i == a) ? s.c[a] : ac;
  }
  const int rr = ar - F::O + dr, cc = ac - F::O + dc;
  const int kk = rr - F::O, jj = cc - F::O;
  const uint32_t v01 = ckf_colour16<N>(m32, kk, jj);
  // channel 2: walls +1 (2-wide border and right of the start column), agents -1, own cell 0 (:43-51, :105-107)
  const bool wall = (unsigned)kk >= (unsigned)F::R || (unsigned)jj >= (unsigned)(F::C + 1);
  bool agent = false;
  const int rc = rr | (cc << 8);  // one compare per agent on the packed (row, column)
#pragma unroll
  for (int a = 0; a < N; ++a) agent = agent | ((s.r[a] | (s.c[a] << 8)) == rc);
  const uint32_t v2 = (cell == KK / 2) ? 0u : (wall ? 0x01u : (agent ? 0xffu : 0u));
  return cg < N * KK ? (v01 | (v2 << 16)) : 0u;
}

// dword d of the grid record: cells 2d and 2d + 1 of get_valid_grid (world[2:5, 2:11, 0:2], checkers.py:66-76)
template <int N> __device__ __forceinline__ uint32_t ckf_grid_dword(uint32_t m32, int d) {
  using F = CkFast<N>;
  uint32_t w = 0;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int cl = 2 * d + h;
    const int k = cl / (F::C + 1), j = cl - k * (F::C + 1);
    const uint32_t v = cl < F::R * (F::C + 1) ? ckf_colour16<N>(m32, k, j) : 0u;
    w |= v << (16 * h);
  }
  return w;
}

// dword d of the obs_self_t record: bytes 4d .. 4d+3 = parts of window cells ca = 4d / 3 and ca + 1
template <int N> __device__ __forceinline__ uint32_t ckf_obst_dword(const CkState<N> &s, uint32_t m32, int d) {
  const int b0 = 4 * d, ca = b0 / 3, sh = 8 * (b0 - 3 * ca);
  const uint64_t t = (uint64_t)ckf_cell3<N>(s, m32, ca) | ((uint64_t)ckf_cell3<N>(s, m32, ca + 1) << 24);
  return (uint32_t)(t >> sh);
}

// the observation slot written by tick t (slot t + 1 of the trajectory) / the terminal-capture slot of tick t
__device__ __forceinline__ CkOut ck_out_tick(const CheckersParams &p, int t) {
  CkOut o;
  o.grid = ck_tick_ptr(p.grid, p.st_grid, t);
  o.vec = ck_tick_ptr(p.vec, p.st_vec, t);
  o.obs_others = ck_tick_ptr(p.obs_others, p.st_obs_others, t);
  o.obs_self_t = ck_tick_ptr(p.obs_self_t, p.st_obs_self_t, t);
  o.obs_self_v = ck_tick_ptr(p.obs_self_v, p.st_obs_self_v, t);
  return o;
}
__device__ __forceinline__ CkOut ck_out_term(const CheckersParams &p, int t) {
  CkOut o;
  o.grid = ck_tick_ptr(p.term_grid, p.st_term_grid, t);
  o.vec = ck_tick_ptr(p.term_vec, p.st_term_vec, t);
  o.obs_others = ck_tick_ptr(p.term_obs_others, p.st_term_obs_others, t);
  o.obs_self_t = ck_tick_ptr(p.term_obs_self_t, p.st_term_obs_self_t, t);
  o.obs_self_v = ck_tick_ptr(p.term_obs_self_v, p.st_term_obs_self_v, t);
  return o;
}

// Observation stores of the fast kernel.  NT = non-temporal: chosen by ck_rollout when the rollout's observation slots are a
// stream (>= 128 MB, like the particle kernels: nothing re-reads them soon); a compile-time parameter, so the plain
// instantiation is the code without this feature.
typedef uint32_t ck_u4 __attribute__((ext_vector_type(4)));
typedef double ck_d2 __attribute__((ext_vector_type(2)));
template <bool NT> __device__ __forceinline__ void ck_st(uint32_t *p, uint32_t v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}
template <bool NT> __device__ __forceinline__ void ck_st(int4 *p, const int4 &v) {
  if constexpr (NT) {
    const ck_u4 t = {(uint32_t)v.x, (uint32_t)v.y, (uint32_t)v.z, (uint32_t)v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<ck_u4 *>(p));
  } else {
    *p = v;
  }
}
template <bool NT> __device__ __forceinline__ void ck_st(double2 *p, const double2 &v) {
  if constexpr (NT) {
    const ck_d2 t = {v.x, v.y};
    __builtin_nontemporal_store(t, reinterpret_cast<ck_d2 *>(p));
  } else {
    *p = v;
  }
}
template <bool NT> __device__ __forceinline__ void ck_st(double4 *p, const double4 &v) {
  if constexpr (NT) {
    ck_st<true>(reinterpret_cast<double2 *>(p), make_double2(v.x, v.y));
    ck_st<true>(reinterpret_cast<double2 *>(p) + 1, make_double2(v.z, v.w));
  } else {
    *p = v;
  }
}

template <bool NT> __device__ __forceinline__ void ck_st(double *p, double v) {
  if constexpr (NT) __builtin_nontemporal_store(v, p); else *p = v;
}

// Observation of one env, written by its G lanes.  Everything is sized at compile time (the fast kernel's geometry is fixed), so
// there are no loops with run-time trip counts and no run-time selects of WHAT a lane does beyond its lane index:
//   grid        dword g of the 14-dword record (two cells each)
//   obs_self_t  lane q takes window cells 4q .. 4q+3 = bytes 12q .. 12q+11 = three whole dwords (the first version took dwords
//               g, g + G, ..: two cell evaluations per dword, 6 per lane at N = 2 against 4 here)
//   vec         lane i < N writes agent i's int4
//   obs_self_v / obs_others: the 4N + 2N(N-1) normalised doubles of the env, ONE per lane: every one of them is an IEEE
//               float64 division (normalize, checkers.py:112-125), ~12 double-rate instructions; lanes 0, 1 used to walk six of
//               them each while the other lanes of the wave waited.  Value v sits at double v of the env's obs_self_v record
//               (v < 4N) or double v - 4N of its obs_others record, so a wave's lanes store consecutive doubles.
// Record bytes past the payload rounded up to 4 (caller-chosen strides larger than that) are not written.
template <int N, bool NT = false, int G = kCkG>
__device__ __forceinline__ void ckf_emit(const CheckersParams &p, const CkState<N> &s, int g, uint32_t e, bool env_ok,
                                         const CkOut &out) {
  using F = CkFast<N, G>;
  constexpr int NO = N > 1 ? N - 1 : 1;
  if (!env_ok) return;
  const uint32_t m32 = (uint32_t)s.mask;  // 24 collected bits
  // ---- grid ---------------------------------------------------------------------------------------------------------------
  constexpr int GD = (F::GRID_REC + 3) / 4;
  const uint32_t grow = e * (uint32_t)p.grid_stride;
#pragma unroll
  for (int d0 = 0; d0 < GD; d0 += G) {
    const int d = d0 + g;
    if (d < GD) ck_st<NT>(at32<uint32_t>(out.grid, grow + 4u * d), ckf_grid_dword<N>(m32, d));
  }
  CM3_STAMP(5, false);
  // ---- obs_self_t ---------------------------------------------------------------------------------------------------------
  constexpr int OD = (F::OBST_REC + 3) / 4, NQ = (OD + 2) / 3;
  const uint32_t orow = e * (uint32_t)p.obst_stride;
#pragma unroll
  for (int q0 = 0; q0 < NQ; q0 += G) {
    const int q = q0 + g;
    if (q < NQ) {
      const uint32_t c0 = ckf_cell3<N>(s, m32, 4 * q), c1 = ckf_cell3<N>(s, m32, 4 * q + 1);
      const uint32_t c2 = ckf_cell3<N>(s, m32, 4 * q + 2), c3 = ckf_cell3<N>(s, m32, 4 * q + 3);
      uint32_t *dst = at32<uint32_t>(out.obs_self_t, orow + 12u * q);
      ck_st<NT>(dst, c0 | (c1 << 24));
      if (3 * q + 1 < OD) ck_st<NT>(dst + 1, (c1 >> 8) | (c2 << 16));
      if (3 * q + 2 < OD) ck_st<NT>(dst + 2, (c2 >> 16) | (c3 << 8));
    }
  }
  CM3_STAMP(6, false);
  // ---- vec ----------------------------------------------------------------------------------------------------------------
  if (g < N) {
    int4 v;
    v.x = s.r[0]; v.y = s.c[0]; v.z = s.ng[0]; v.w = s.no[0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
      v.x = (g == a) ? s.r[a] : v.x;
      v.y = (g == a) ? s.c[a] : v.y;
      v.z = (g == a) ? s.ng[a] : v.z;
      v.w = (g == a) ? s.no[a] : v.w;
    }
    ck_st<NT>(at32<int4>(out.vec, (e * N + g) * 16u), v);
  }
  CM3_STAMP(7, false);
  // ---- obs_self_v, obs_others ---------------------------------------------------------------------------------------------
  constexpr int NSV = 4 * N, NOO = 2 * N * NO, NV = NSV + NOO;
  const double half = (double)(F::R * F::C) / 2.0;
#pragma unroll
  for (int v0 = 0; v0 < NV; v0 += G) {
    const int v = v0 + g;
    if (v < NV) {
      const bool others = v >= NSV;
      const int w = others ? v - NSV : v;
      // agent whose coordinate / count is normalised, and which of its four numbers
      int a, comp;
      if (others) {
        const int i = w / (2 * NO), k = (w - i * 2 * NO) >> 1;
        a = (N > 1) ? (k < i ? k : k + 1) : 0;  // k-th other agent of agent i (N == 1: itself)
        comp = w & 1;
      } else {
        a = w >> 2;
        comp = w & 3;
      }
      int ra = s.r[0], ca = s.c[0], ga = s.ng[0], oa = s.no[0];
#pragma unroll
      for (int x = 1; x < N; ++x) {
        ra = (a == x) ? s.r[x] : ra;
        ca = (a == x) ? s.c[x] : ca;
        ga = (a == x) ? s.ng[x] : ga;
        oa = (a == x) ? s.no[x] : oa;
      }
      const int num_i = comp == 0 ? ra : (comp == 1 ? ca : (comp == 2 ? ga : oa));
      const double off = comp == 0 ? (double)F::TR / 2.0 : (comp == 1 ? (double)F::TC / 2.0 : 0.0);
      const double den = comp == 0 ? (double)F::TR : (comp == 1 ? (double)F::TC : half);
      // counts: (double)n / half; coordinates: ((double)r - TR / 2.0) / TR -- x - 0.0 == x exactly, so one expression serves
      const double val = ((double)num_i - off) / den;
      if (others) ck_st<NT>(at32<double>(out.obs_others, (e * (uint32_t)NOO + (uint32_t)w) * 8u), val);
      else ck_st<NT>(at32<double>(out.obs_self_v, (e * (uint32_t)NSV + (uint32_t)w) * 8u), val);
    }
  }
}

// ---- table-driven observation emit of the step kernel (round 3) ------------------------------------------------------------
// The emit above evaluates every window cell from scratch (~45 instructions per cell) and needs ~185 lane-geometry instructions
// (which cell, which agent, which row) that the compiler placed AFTER the tick: at one wave per SIMD every one of them is ~4.5
// cycles of the launch.  What those instructions compute does not depend on the state:
//   * the STATIC content of an expanded board cell (rr, cc) -- wall byte, colour channel, which bit of the collected mask it
//     shows -- is a constant of the geometry: one dword per cell of the 7 x 16 padded board (CkBoardTab::board);
//   * the twelve-or-so normalised float64 values of an env are (r - 3.5) / 7, (c - 6.5) / 13 or n / 12 with r in 0..6, c in
//     0..12, n in 0..12: 33 possible doubles (CkBoardTab::norm; constexpr IEEE divisions = the device's IEEE divisions);
//   * which cells / grid dword / value a LANE produces is a function of its lane index: per-N plan tables (CkPlanTab<N>).
// All three are compile-time constants in device memory.  A wave copies the 768-byte board / norm table into a private LDS region
// and loads its lanes' plan entries while the state loads are in flight (nothing of it waits on the critical path); after the
// tick a window cell is  e = board[agent base + plan offset];  cell = e ^ (e & 0xfefe & -collected(e >> 24)) | agent byte
// (~13 instructions), four cells become three dwords with three byte permutes, a normalised value is one LDS read.
// Bit-exact by construction (same values), checked by the same tests as the emit above (which the reset kernel keeps using).
struct CkBoardTab {
  uint32_t board[7 * 16];  // entry of board cell (rr, cc) at [rr * 16 + cc]: ch0 | ch1 << 8 | ch2 << 16 | bit index << 24
  double norm[40];         // [0..6] row, [7..19] column, [20..32] count; [34..39] local reward = penalty + reward (ck_tick_env)
  constexpr CkBoardTab() : board{}, norm{} {
    for (int rr = 0; rr < 7; ++rr)
      for (int cc = 0; cc < 16; ++cc) {
        const int kk = rr - 2, jj = cc - 2;
        const bool band_row = kk >= 0 && kk < 3;
        const bool inband = band_row && jj >= 0 && jj < 8, valid = band_row && jj >= 0 && jj < 9;
        uint32_t e = 0;
        if (inband) e |= 0xffu << (8 * ((kk + jj) & 1));  // -1: still there (the collected bit flips it to +1)
        if (!valid) e |= 1u << 16;                        // wall
        e |= (uint32_t)(inband ? kk * 8 + jj : 31) << 24;  // bit 31 of the 24-bit mask is never set
        board[rr * 16 + cc] = e;
      }
    for (int r = 0; r < 7; ++r) norm[r] = ((double)r - 7.0 / 2.0) / 7.0;
    for (int c = 0; c < 13; ++c) norm[7 + c] = ((double)c - 13.0 / 2.0) / 13.0;
    for (int n = 0; n < 13; ++n) norm[20 + n] = ((double)n - 0.0) / (24.0 / 2.0);
    const double penalty[2] = {0.0, -0.1}, reward[3] = {0.0, 1.0, -0.5};  // checkers.py:186, :213-222 (matching / other colour)
    for (int pp = 0; pp < 2; ++pp)
      for (int kk = 0; kk < 3; ++kk) norm[34 + 3 * pp + kk] = penalty[pp] + reward[kk];
  }
};
static_assert(sizeof(CkBoardTab) == 768, "the kernel stages 48 x 16 bytes");
__device__ const CkBoardTab kCkBoardTab = CkBoardTab();

template <int N> struct CkPlanTab {
  static constexpr int NO = N > 1 ? N - 1 : 1;
  static constexpr int KK = 25, OD = (75 * N + 3) / 4, NQ = (OD + 2) / 3, NCELL = 4 * NQ;
  static constexpr int NSV = 4 * N, NOO = 2 * N * NO, NV = NSV + NOO, GD = 14;
  static constexpr int NGV = (GD > NV ? GD : NV);
  // window cell cg, two dwords: { board byte offset from the agent's cell (low 16 bits, signed) | packed (row, column) offset << 16,
  //                              agent | agent byte mask (0xff0000 or 0) | invalid << 31 }
  uint32_t cell[NCELL][2];
  // lane slot j = grid dword j AND normalised value j: { grid static dword, bit index of grid cell 2j | of cell 2j + 1 << 8,
  //   norm index of number 0 | agent << 8 | bit offset of the number in the agent word << 16, byte offset | others << 31 }
  uint32_t gv[NGV + 1][4];
  constexpr CkPlanTab() : cell{}, gv{} {
    for (int cg = 0; cg < NCELL; ++cg) {
      const bool ok = cg < KK * N;
      const int i = ok ? cg / KK : 0, c = ok ? cg - KK * i : 12;
      const int dr = c / 5 - 2, dc = c % 5 - 2;
      cell[cg][0] = ((uint32_t)((dr * 16 + dc) * 4) & 0xffffu) | ((uint32_t)(dr + dc * 256) << 16);
      cell[cg][1] = (uint32_t)i | ((ok && c != 12) ? 0xff0000u : 0u) | (ok ? 0u : 0x80000000u);  // own (centre) cell: valid (checkers.py:105-107)
    }
    for (int j = 0; j <= NGV; ++j) {
      uint32_t w = 0, b[2] = {31, 31};
      for (int h = 0; h < 2; ++h) {
        const int cl = 2 * j + h;
        if (j < GD && cl < 27) {
          const int k = cl / 9, jj = cl - 9 * k;
          if (jj < 8) {
            w |= 0xffu << (16 * h + 8 * ((k + jj) & 1));
            b[h] = (uint32_t)(k * 8 + jj);
          }
        }
      }
      gv[j][0] = w;
      gv[j][1] = b[0] | (b[1] << 8);
      const int v = j < NV ? j : NV - 1;
      const bool others = v >= NSV;
      const int wv = others ? v - NSV : v;
      int a = 0, comp = 0;
      if (others) {
        const int i = wv / (2 * NO), k = (wv - i * 2 * NO) >> 1;
        a = (N > 1) ? (k < i ? k : k + 1) : 0;
        comp = wv & 1;
      } else {
        a = wv >> 2;
        comp = wv & 3;
      }
      gv[j][2] = (comp == 0 ? 0u : (comp == 1 ? 7u : 20u)) | ((uint32_t)a << 8) | ((uint32_t)(8 * comp) << 16);
      gv[j][3] = (uint32_t)(wv * 8) | (others ? 0x80000000u : 0u);
    }
  }
};
template <int N> __device__ const CkPlanTab<N> kCkPlanTab = CkPlanTab<N>();

// What a lane produces (registers; loaded once per launch while the state loads are in flight and decoded there).  Every wave of a
// launch reads these tables through its CU's 64 B / clock vector cache, so they are kept small: 7 sixteen-byte loads per lane at
// N = 2 (a first version with one dword per field took 13 -- 14 KB per wave -- and the wave ended up waiting for them).
template <int N, int G> struct CkLanePlan {
  using T = CkPlanTab<N>;
  static constexpr int NSLOT = (T::NQ + G - 1) / G, NGV = (T::NGV + G - 1) / G;
  uint32_t boff[NSLOT][4], rcd[NSLOT][4], amask[NSLOT][4], agent[NSLOT][4];
  bool invalid[NSLOT][4];
  uint32_t gstatic[NGV], gb0[NGV], gb1[NGV];
  uint32_t vidx[NGV], vagent[NGV], vshift[NGV], voff[NGV];
};

template <int N, int G> struct CkLanePlanRaw {
  using P = CkLanePlan<N, G>;
  uint4 c01[P::NSLOT], c23[P::NSLOT], gv[P::NGV];
};
// the lane's plan entries as loaded (requests only; nothing waits here) ...
template <int N, int G> __device__ __forceinline__ void ckf_plan_fetch(int g, CkLanePlanRaw<N, G> &raw) {
  using T = CkPlanTab<N>;
  using P = CkLanePlan<N, G>;
  const CkPlanTab<N> *tab = &kCkPlanTab<N>;
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it) {
    const int q = it * G + g, qc = q < T::NQ ? q : T::NQ - 1;
    raw.c01[it] = *reinterpret_cast<const uint4 *>(&tab->cell[4 * qc][0]);
    raw.c23[it] = *reinterpret_cast<const uint4 *>(&tab->cell[4 * qc + 2][0]);
  }
#pragma unroll
  for (int it = 0; it < P::NGV; ++it) {
    const int j = it * G + g;
    raw.gv[it] = *reinterpret_cast<const uint4 *>(&tab->gv[j < T::NGV ? j : T::NGV][0]);
  }
}
// ... and decoded into the fields the emit uses
template <int N, int G> __device__ __forceinline__ void ckf_plan_decode(const CkLanePlanRaw<N, G> &raw, CkLanePlan<N, G> &pl) {
  using P = CkLanePlan<N, G>;
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it) {
    const uint32_t w0[4] = {raw.c01[it].x, raw.c01[it].z, raw.c23[it].x, raw.c23[it].z};
    const uint32_t w1[4] = {raw.c01[it].y, raw.c01[it].w, raw.c23[it].y, raw.c23[it].w};
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      pl.boff[it][x] = (uint32_t)__builtin_amdgcn_sbfe((int)w0[x], 0, 16);
      pl.rcd[it][x] = (uint32_t)((int)w0[x] >> 16);
      pl.amask[it][x] = w1[x] & 0xff0000u;
      pl.agent[it][x] = w1[x] & 0xffu;
      pl.invalid[it][x] = (int)w1[x] < 0;
    }
  }
#pragma unroll
  for (int it = 0; it < P::NGV; ++it) {
    pl.gstatic[it] = raw.gv[it].x;
    pl.gb0[it] = raw.gv[it].y & 0xffu;
    pl.gb1[it] = raw.gv[it].y >> 8;
    pl.vidx[it] = raw.gv[it].z & 0xffu;
    pl.vagent[it] = (raw.gv[it].z >> 8) & 0xffu;
    pl.vshift[it] = raw.gv[it].z >> 16;
    pl.voff[it] = raw.gv[it].w;
  }
}

// the wave's private copy of kCkBoardTab in LDS (48 x 16 bytes, written by lanes 0..47) and the lane's plan: ALL loads are
// requested before the first of them is waited for (the LDS write placed right behind its load made the wave wait for the table
// before it had even requested its plan: a second memory round trip on the critical path)
template <int N, int G> __device__ __forceinline__ void ckf_plan_load(int g, int lane, uint4 *lds_tab, CkLanePlan<N, G> &pl,
                                                                      uint4 *lds_fresh = nullptr, const uint4 *fresh_src = nullptr) {
  const uint4 board_vec = reinterpret_cast<const uint4 *>(&kCkBoardTab)[lane < 48 ? lane : 47];
  uint4 fresh_vec = make_uint4(0u, 0u, 0u, 0u);
  if (fresh_src) fresh_vec = fresh_src[lane < 26 ? lane : 25];   // (the fresh-episode record, CkFresh: 26 vectors at most)
  CkLanePlanRaw<N, G> raw;
  ckf_plan_fetch<N, G>(g, raw);
  if (lane < 48) lds_tab[lane] = board_vec;
  if (lds_fresh && lane < 26) lds_fresh[lane] = fresh_vec;
  ckf_plan_decode<N, G>(raw, pl);
}

// ---- the observation of a fresh episode as a constant record (round 4) ---------------------------------------------------------
// A Checkers env restarts from a state that is a constant of (geometry, start cells) -- and, for N = 1, of the goal bit, which picks
// the start row (checkers.py:265-291 / ck_init).  With auto-reset a wave that held ONE finished env used to run the whole emit twice
// (terminal observation, then the observation of the reset state), and a launch lasts as long as its slowest wave: 3.35 -> 3.52 us
// per tick at C3, 4.02 with rare resets (profiles/r03_checkers_table_emit.txt, end).  Now the host works the fresh observation out
// once per call (ck_fresh_fill: the expressions of the emit on the start state), it travels in the kernel-argument segment, every
// wave stages it into LDS in the shadow of its state loads, and a finished env's lanes COPY it to the slot (11 dword stores per
// lane at N = 2); the one emit of such a wave writes the terminal slot for finished envs and the regular slot for the others.
// Layout (dwords): obs_self_t payload | grid (14) | vec | obs_self_v | obs_others, padded to 16 bytes; N = 1: two variants (goal 0, 1).
template <int N> struct CkFresh {
  static constexpr bool kOn = N <= 2;
  static constexpr int NO = N > 1 ? N - 1 : 1;
  static constexpr int OD = (75 * N + 3) / 4, GD = 14, VD = 4 * N, SD = 8 * N, XD = 4 * N * NO;
  static constexpr int oObst = 0, oGrid = OD, oVec = oGrid + GD, oSelf = oVec + VD, oOth = oSelf + SD, FD = oOth + XD;
  static constexpr int FDP = (FD + 3) / 4 * 4, NVAR = N == 1 ? 2 : 1;
  static constexpr int VECS = FDP * NVAR / 4;   // 16-byte vectors a wave stages
  static_assert(!kOn || FDP * NVAR <= 104, "CheckersParams::fresh");
};

constexpr CkBoardTab kCkBoardTabHost = CkBoardTab();

// host: the fresh observation(s) of this call's configuration into p.fresh (fast geometry: 3 x 8 band, n_obs 2)
static void ck_fresh_fill(CheckersParams &p, int n) {
  memset(p.fresh, 0, sizeof(p.fresh));
  if (n > 2) return;
  const int NO = n > 1 ? n - 1 : 1;
  const int OD = (75 * n + 3) / 4, oGrid = OD, oVec = oGrid + 14, oSelf = oVec + 4 * n, oOth = oSelf + 8 * n, FD = oOth + 4 * n * NO;
  const int FDP = (FD + 3) / 4 * 4;
  for (int var = 0; var < (n == 1 ? 2 : 1); ++var) {
    uint32_t *rec = p.fresh + var * FDP;
    int r[2], c[2];
    for (int i = 0; i < n; ++i) {
      r[i] = p.start_r[i];
      c[i] = p.start_c[i];
    }
    if (n == 1) r[0] = (var == 0 ? 0 : 2) + 2;     // ck_init: the goal picks the start row
    uint8_t *ob = reinterpret_cast<uint8_t *>(rec);
    for (int i = 0; i < n; ++i)                      // get_obs (checkers.py:97-109): 5 x 5 x 3 window, nothing collected yet
      for (int dr = 0; dr < 5; ++dr)
        for (int dc = 0; dc < 5; ++dc) {
          const int rr = r[i] - 2 + dr, cc = c[i] - 2 + dc;
          const uint32_t en = kCkBoardTabHost.board[rr * 16 + cc];
          bool agent = false;
          for (int a = 0; a < n; ++a) agent = agent || (r[a] == rr && c[a] == cc);
          uint8_t *cell = ob + ((i * 25 + dr * 5 + dc) * 3);
          cell[0] = (uint8_t)(en & 0xffu);
          cell[1] = (uint8_t)((en >> 8) & 0xffu);
          cell[2] = (uint8_t)(((en >> 16) & 0xffu) | ((agent && !(dr == 2 && dc == 2)) ? 0xffu : 0u));
        }
    uint8_t *gr = reinterpret_cast<uint8_t *>(rec + oGrid);   // get_valid_grid (checkers.py:66-76): 3 x 9 cells x 2 channels
    for (int k = 0; k < 3; ++k)
      for (int j = 0; j < 8; ++j) gr[(k * 9 + j) * 2 + ((k + j) & 1)] = 0xffu;
    for (int i = 0; i < n; ++i) {
      rec[oVec + 4 * i + 0] = (uint32_t)r[i];
      rec[oVec + 4 * i + 1] = (uint32_t)c[i];
      double v[4] = {kCkBoardTabHost.norm[r[i]], kCkBoardTabHost.norm[7 + c[i]], kCkBoardTabHost.norm[20], kCkBoardTabHost.norm[20]};
      memcpy(rec + oSelf + 8 * i, v, 32);
      for (int k = 0; k < NO; ++k) {
        const int a = n > 1 ? (k < i ? k : k + 1) : 0;
        double o[2] = {kCkBoardTabHost.norm[r[a]], kCkBoardTabHost.norm[7 + c[a]]};
        memcpy(rec + oOth + 4 * (i * NO + k), o, 16);
      }
    }
  }
}

// the explicit arguments of k_checkers_step_fast as they lie in the kernel-argument segment: gives the offset of `p`
struct CkFastKernArgs {
  const uint64_t *mask;
  const uint32_t *agents;
  const int32_t *steps;
  const int32_t *episode;
  const uint8_t *goals;
  const uint32_t *ablock;
  int E;
  uint32_t flags;
  CheckersParams p;
};
__device__ __forceinline__ const uint4 *ck_fresh_kernarg() {
#if defined(__HIP_DEVICE_COMPILE__)
  const char *args = (const char *)__builtin_amdgcn_kernarg_segment_ptr();
  return reinterpret_cast<const uint4 *>(args + offsetof(CkFastKernArgs, p) + offsetof(CheckersParams, fresh));
#else
  return nullptr;
#endif
}

// a finished env's G lanes copy the fresh record (variant var of the wave's LDS copy) to the slot `out`
template <int N, bool NT, int G>
__device__ __forceinline__ void ckf_fresh_copy(const CheckersParams &p, const uint4 *lds_fresh, int var, int g, uint32_t e, const CkOut &out) {
  using FR = CkFresh<N>;
  const uint32_t *rec = reinterpret_cast<const uint32_t *>(lds_fresh) + var * FR::FDP;
  auto run = [&](void *base, uint32_t row, int off, int nd) {
#pragma unroll
    for (int d0 = 0; d0 < nd; d0 += G) {
      const int d = d0 + g;
      if (d < nd) ck_st<NT>(at32<uint32_t>(base, row + 4u * d), rec[off + d]);
    }
  };
  run(out.obs_self_t, e * (uint32_t)p.obst_stride, FR::oObst, FR::OD);
  run(out.grid, e * (uint32_t)p.grid_stride, FR::oGrid, FR::GD);
  run(out.vec, e * (uint32_t)(FR::VD * 4), FR::oVec, FR::VD);
  run(out.obs_self_v, e * (uint32_t)(FR::SD * 4), FR::oSelf, FR::SD);
  run(out.obs_others, e * (uint32_t)(FR::XD * 4), FR::oOth, FR::XD);
}

// SEL: a lane's five observation arrays are `alt` instead of `out` where use_alt is set (the one emit of a wave that holds finished
// envs: their terminal slot) -- the addresses then are per-lane 64-bit values instead of a uniform base plus a 32-bit lane offset,
// which is why the common case (no finished env in the wave) keeps the plain instantiation.
// SINK (the whole-episode policy kernel, policy_checkers.hip): the window cells and the normalised values this lane produces are
// ALSO handed to sink.cell(agent, window cell, three channel bytes) / sink.value(others, index, value) -- that kernel's next network
// inputs, written to LDS from the registers that hold them anyway (lanes writing a terminal slot, use_alt, hand over nothing).
struct CkNoSink {
  static constexpr bool kOn = false;
  __device__ __forceinline__ void cell(uint32_t, int, uint32_t) const {}
  __device__ __forceinline__ void value(bool, uint32_t, double) const {}
};
template <int N, bool NT = false, int G = kCkG, bool SEL = false, class SINK = CkNoSink>
__device__ __forceinline__ void ckf_emit_tab(const CheckersParams &p, const CkState<N> &s, const CkLanePlan<N, G> &pl,
                                             const uint4 *lds_tab, int g, uint32_t e, bool env_ok, const CkOut &out_in,
                                             const CkOut *alt = nullptr, bool use_alt = false, const SINK &sink = SINK()) {
  using T = CkPlanTab<N>;
  using P = CkLanePlan<N, G>;
  if (!env_ok) return;
  CkOut out = out_in;
  if constexpr (SEL) {
    out.grid = use_alt ? alt->grid : out.grid;
    out.vec = use_alt ? alt->vec : out.vec;
    out.obs_others = use_alt ? alt->obs_others : out.obs_others;
    out.obs_self_t = use_alt ? alt->obs_self_t : out.obs_self_t;
    out.obs_self_v = use_alt ? alt->obs_self_v : out.obs_self_v;
  }
  const char *tab = reinterpret_cast<const char *>(lds_tab);
  const uint32_t m32 = (uint32_t)s.mask;  // 24 collected bits
  uint32_t rc[N], base[N], word[N];
#pragma unroll
  for (int a = 0; a < N; ++a) {
    rc[a] = (uint32_t)s.r[a] | ((uint32_t)s.c[a] << 8);
    base[a] = (uint32_t)s.r[a] * 64u + (uint32_t)s.c[a] * 4u;
    word[a] = rc[a] | ((uint32_t)s.ng[a] << 16) | ((uint32_t)s.no[a] << 24);
  }
  // ---- every table read of this lane first (cell entries, normalised values), ONE wait, then arithmetic and stores: a waited-for
  //      LDS read is ~100+ cycles for a lone wave, and reads issued behind exec-masked store blocks each got their own wait ---------
  uint32_t ent[P::NSLOT][4], rcx[P::NSLOT][4];
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it) {
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      const uint32_t ia = pl.agent[it][x];
      uint32_t bb = base[0], r0 = rc[0];
#pragma unroll
      for (int a = 1; a < N; ++a) {
        bb = (ia == (uint32_t)a) ? base[a] : bb;
        r0 = (ia == (uint32_t)a) ? rc[a] : r0;
      }
      ent[it][x] = *reinterpret_cast<const uint32_t *>(tab + (bb + pl.boff[it][x]));
      rcx[it][x] = r0 + pl.rcd[it][x];
    }
  }
  double nval[P::NGV];
#pragma unroll
  for (int it = 0; it < P::NGV; ++it) {
    nval[it] = 0.0;
    if (it * G < T::NV) {  // (compile time)
      uint32_t wa = word[0];
#pragma unroll
      for (int a = 1; a < N; ++a) wa = (pl.vagent[it] == (uint32_t)a) ? word[a] : wa;
      const uint32_t num = (wa >> pl.vshift[it]) & 0xffu;
      nval[it] = *reinterpret_cast<const double *>(tab + 448u + 8u * (pl.vidx[it] + num));
    }
  }
  // (the compiler otherwise sinks the reads of the later slots into the exec-masked blocks that consume them)
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it)
#pragma unroll
    for (int x = 0; x < 4; ++x) asm volatile("" : "+v"(ent[it][x]));
  // ---- obs_self_t: lane q takes window cells 4q .. 4q + 3 = three whole dwords ------------------------------------------------
  const uint32_t orow = e * (uint32_t)p.obst_stride;
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it) {
    const int q = it * G + g;
    uint32_t c[4];
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      bool agent = false;
#pragma unroll
      for (int a = 0; a < N; ++a) agent = agent | (rc[a] == rcx[it][x]);
      const uint32_t en = ent[it][x];
      const uint32_t got = (uint32_t)__builtin_amdgcn_sbfe((int)m32, en >> 24, 1);  // 0 / ~0: the cell's collected bit
      uint32_t v = (en ^ (en & got & 0xfefeu)) | (agent ? pl.amask[it][x] : 0u);
      if ((it + 1) * G * 4 > T::KK * N) v = pl.invalid[it][x] ? 0u : v;  // (compile time: only the slots that can hold padding cells)
      c[x] = v;
      if constexpr (SINK::kOn) {
        if (q < T::NQ && !pl.invalid[it][x] && !(SEL && use_alt)) sink.cell(pl.agent[it][x], 4 * q + x - T::KK * (int)pl.agent[it][x], v);
      }
    }
    if (q < T::NQ) {
      // bytes: c0[0..2] c1[0..2] c2[0..2] c3[0..2] -> three dwords (v_perm_b32: selector byte k picks byte k of {hi, lo})
      const uint32_t d0 = __builtin_amdgcn_perm(c[1], c[0], 0x04020100u);
      const uint32_t d1 = __builtin_amdgcn_perm(c[2], c[1], 0x05040201u);
      const uint32_t d2 = __builtin_amdgcn_perm(c[3], c[2], 0x06050402u);
      uint32_t *dst = at32<uint32_t>(out.obs_self_t, orow + 12u * q);
      ck_st<NT>(dst, d0);
      if (3 * q + 1 < T::OD) ck_st<NT>(dst + 1, d1);
      if (3 * q + 2 < T::OD) ck_st<NT>(dst + 2, d2);
    }
  }
  // ---- grid: dword d = cells 2d, 2d + 1 of get_valid_grid (checkers.py:66-76) --------------------------------------------------
  const uint32_t grow = e * (uint32_t)p.grid_stride;
#pragma unroll
  for (int it = 0; it < P::NGV; ++it) {
    const int d = it * G + g;
    if (it * G < T::GD) {  // (compile time)
      const uint32_t gs = pl.gstatic[it];
      const uint32_t g0 = (uint32_t)__builtin_amdgcn_sbfe((int)m32, pl.gb0[it], 1), g1 = (uint32_t)__builtin_amdgcn_sbfe((int)m32, pl.gb1[it], 1);
      const uint32_t w = gs ^ (gs & g0 & 0x0000fefeu) ^ (gs & g1 & 0xfefe0000u);
      if (d < T::GD) ck_st<NT>(at32<uint32_t>(out.grid, grow + 4u * d), w);
    }
  }
  // ---- vec ------------------------------------------------------------------------------------------------------------------
  if (g < N) {
    uint32_t wa = word[0];
#pragma unroll
    for (int a = 1; a < N; ++a) wa = (g == a) ? word[a] : wa;
    int4 v;
    v.x = (int)(wa & 0xffu); v.y = (int)((wa >> 8) & 0xffu); v.z = (int)((wa >> 16) & 0xffu); v.w = (int)(wa >> 24);
    ck_st<NT>(at32<int4>(out.vec, (e * N + g) * 16u), v);
  }
  // ---- obs_self_v, obs_others ------------------------------------------------------------------------------------------------
#pragma unroll
  for (int it = 0; it < P::NGV; ++it) {
    const int v = it * G + g;
    if (it * G < T::NV) {  // (compile time)
      if (v < T::NV) {
        const uint32_t off = pl.voff[it] & 0x7fffffffu;
        if ((int)pl.voff[it] < 0) ck_st<NT>(at32<double>(out.obs_others, e * (uint32_t)(T::NOO * 8) + off), nval[it]);
        else ck_st<NT>(at32<double>(out.obs_self_v, e * (uint32_t)(T::NSV * 8) + off), nval[it]);
        if constexpr (SINK::kOn) {
          if (!(SEL && use_alt)) sink.value((int)pl.voff[it] < 0, off >> 3, nval[it]);
        }
      }
    }
  }
}

// (Round 1 added a fifth "draw wave" per workgroup that drew the next launch's actions, as in the particle pair mapping: 5.08 ->
// 4.96 us per tick at C3 then.  Re-measured in round 2 (profiles/r02_draw_wave_on_off.txt) it had become a loss -- stage 2: 4.99 ->
// 4.88 us, stage 1: 4.19 -> 3.98 us without it -- and was removed together with the particle one.)
template <int N, bool FUSED, bool NT = false, int G = kCkG>
__global__ void __launch_bounds__(256)
    k_checkers_step_fast(const uint64_t *h_mask, const uint32_t *h_agents, const int32_t *h_steps, const int32_t *h_episode,
                         const uint8_t *h_goals, const uint32_t *h_ablock, const int h_E, const uint32_t h_flags, const CheckersParams p) {
  using F = CkFast<N, G>;
  CM3_SPAN_IN();
  CkHead hd;   // leading, preloaded kernel arguments (see CkHead)
  hd.mask = h_mask; hd.agents = h_agents; hd.steps = h_steps; hd.episode = h_episode; hd.goals = h_goals; hd.ablock = h_ablock; hd.E = h_E;
  hd.flags = h_flags; hd.env_id_base = p.env_id_base; hd.seed = p.seed;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane & (F::G - 1), sub = lane / F::G;
  // 32-bit env index and byte offsets (every per-tick array below 4 GiB, checked by ck_launch): addresses are
  // <uniform base> + <lane offset>, as in the particle kernels
  const uint32_t e = (cm3_xcd_block(h_flags) * 4 + wave) * F::EPW + sub;
  const bool env_ok = e < (uint32_t)h_E;
  const uint32_t ec = env_ok ? e : (uint32_t)h_E - 1;
  const bool writer = env_ok && g == 0;
  CkState<N> s;
  CkLive<N> lv;
  __shared__ __attribute__((aligned(16))) uint4 lds_tab_all[4][48];
  __shared__ __attribute__((aligned(16))) uint4 lds_fresh_all[4][CkFresh<N>::kOn ? 26 : 1];
  uint4 *lds_tab = &lds_tab_all[wave][0];
  uint4 *lds_fresh = &lds_fresh_all[wave][0];
  constexpr bool kFresh = CkFresh<N>::kOn;
  CM3_STAMP(0, false);
  ck_load_env<N>(hd, ec, s, lv);
  // while those loads are in flight: the wave's copy of the board / norm table and this lane's plan (see ckf_emit_tab)
  CkLanePlan<N, G> pl;
  if constexpr (kFresh) ckf_plan_load<N, G>(g, lane, lds_tab, pl, lds_fresh, (h_flags & CM3_FLAG_AUTO_RESET) ? ck_fresh_kernarg() : nullptr);
  else ckf_plan_load<N, G>(g, lane, lds_tab, pl);
  ck_wave_sync();
  CM3_SPAN_MARK(0, true);   // loads back
  // the kernel arguments the tick needs, requested while the state loads are in flight (fetched at their first use they made
  // the wave wait for a scalar load six times along its critical path)
  CM3_FETCH_EARLY(p.actions, p.local_rewards, p.reward, p.done, p.grid, p.vec, p.obs_others, p.obs_self_t, p.obs_self_v,
                  p.goals_next, p.term_grid, p.max_steps, p.grid_stride, p.obst_stride, p.seed, p.env_id_base);
  CM3_STAMP(1, true);
  // FUSED == false: exactly one tick; the loop and the per-tick offsets fold away
  const int n_ticks = FUSED ? p.n_ticks : 1;
#pragma unroll 1
  for (int t = 0; t < n_ticks; ++t) {
    const bool ended = ck_tick_env<N, true>(p, t, e, ec, writer, s, lv, reinterpret_cast<const char *>(lds_tab)
#ifdef CM3_SPAN_MARKS
                                                , _span_mk
#endif
                                                );
    CM3_STAMP(4, false);
    CM3_SPAN_MARK(1, false);  // draw + agents act done
    if constexpr (kFresh) {
      if (!__any(ended)) {
        ckf_emit_tab<N, NT, G>(p, s, pl, lds_tab, g, e, env_ok, ck_out_tick(p, t));
      } else {
        // a wave with finished envs: ONE emit of the post-step state -- into the terminal slot for the finished envs (the next_*
        // columns of their last transition, train_onpolicy.py:336-347; nowhere without terminal capture), into the regular slot
        // for the others -- then the finished envs restart and their regular slot gets the fresh-episode record by copy (CkFresh)
        const CkOut o_tick = ck_out_tick(p, t), o_term = ck_out_term(p, t);
        ckf_emit_tab<N, NT, G, true>(p, s, pl, lds_tab, g, e, env_ok && (!ended || p.term_grid != nullptr), o_tick, &o_term, ended);
        if (ended) {
          ck_restart_env<N>(p, e, ec, writer, s, lv);
          if (env_ok) ckf_fresh_copy<N, NT, G>(p, lds_fresh, N == 1 ? (int)lv.goal[0] : 0, g, e, o_tick);
        }
      }
    } else {
      if (ended) {  // AUTO_RESET: terminal observation (train_onpolicy.py:336-347), then the fresh episode
        if (p.term_grid) ckf_emit_tab<N, NT, G>(p, s, pl, lds_tab, g, e, env_ok, ck_out_term(p, t));
        ck_restart_env<N>(p, e, ec, writer, s, lv);
      }
      ckf_emit_tab<N, NT, G>(p, s, pl, lds_tab, g, e, env_ok, ck_out_tick(p, t));
    }
    CM3_STAMP(8, false);
    CM3_SPAN_MARK(2, false);  // observation stores issued
    if (p.goals_next && writer) {
      uint8_t *gn = ck_tick_ptr(p.goals_next, p.st_goals_next, t);
#pragma unroll
      for (int i = 0; i < N; ++i) *at32<uint8_t>(gn, e * N + i) = lv.goal[i];
    }
  }
  if (writer) ck_store_env<N>(p, e, s, lv);
  CM3_STAMP(9, true);
  CM3_SPAN_OUT(p.span);
}

template <int N> __global__ void __launch_bounds__(256) k_checkers_reset_fast(const CheckersParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int g = lane & (CkFast<N>::G - 1), sub = lane / CkFast<N>::G;
  const size_t e = ((size_t)blockIdx.x * 4 + wave) * CkFast<N>::EPW + sub;
  const bool env_ok = e < (size_t)p.E;
  const size_t ec = env_ok ? e : (size_t)p.E - 1;
  CkState<N> s;
  const bool sel = p.reset_mask ? (p.reset_mask[ec] != 0) : true;
  if (sel) {
    uint8_t goal[N];
#pragma unroll
    for (int i = 0; i < N; ++i) goal[i] = p.goals[ec * N + i];
    ck_init<N>(p, goal, s);
    if (env_ok && g == 0) {
      ck_store<N>(p, e, s);
      p.steps[e] = 0;
      if (p.episode) p.episode[e] = p.episode[e] + 1;
    }
  } else {
    ck_load<N>(ck_head(p), ec, s);
  }
  ckf_emit<N>(p, s, g, e, env_ok, ck_out_tick(p, 0));
}

template <int N> __global__ void __launch_bounds__(64) k_checkers_step(const CheckersParams p) {
  __shared__ __attribute__((aligned(16))) int8_t lds[kCkLdsBytes];
  const int lane = threadIdx.x;
  const size_t e0 = (size_t)blockIdx.x * 64;
  const size_t e = e0 + lane;
  const bool active = e < (size_t)p.E;
  const size_t ec = active ? e : (size_t)p.E - 1;

  CkState<N> s;
  ck_step_env<N>(p, e, ec, active, s);
  ck_emit<N>(p, s, lds, lane, e0, e, active);
}

template <int N> __global__ void __launch_bounds__(64) k_checkers_reset(const CheckersParams p) {
  __shared__ __attribute__((aligned(16))) int8_t lds[kCkLdsBytes];
  const int lane = threadIdx.x;
  const size_t e0 = (size_t)blockIdx.x * 64;
  const size_t e = e0 + lane;
  const bool active = e < (size_t)p.E;
  const size_t ec = active ? e : (size_t)p.E - 1;
  CkState<N> s;
  const bool sel = p.reset_mask ? (p.reset_mask[ec] != 0) : true;
  if (sel) {
    uint8_t goal[N];
#pragma unroll
    for (int i = 0; i < N; ++i) goal[i] = p.goals[ec * N + i];
    ck_init<N>(p, goal, s);
    if (active) {
      ck_store<N>(p, e, s);
      p.steps[e] = 0;
      if (p.episode) p.episode[e] = p.episode[e] + 1;
    }
  } else {
    ck_load<N>(ck_head(p), ec, s);
  }
  ck_emit<N>(p, s, lds, lane, e0, e, active);
}

// the multi-lane kernel handles the reference geometry with dword-aligned env records
static bool ck_fast_ok(const CheckersParams &p) {
  return p.R == 3 && p.C == 8 && p.O == 2 && (p.grid_stride % 4) == 0 && (p.obst_stride % 4) == 0;
}

static int ck_fill(const cm3_checkers_desc *d, const cm3_checkers_bufs *b, const uint8_t *mask, bool step,
                   CheckersParams &p) {
  CM3_REQUIRE(d && b, "null desc/bufs");
  CM3_REQUIRE(d->n_envs > 0, "n_envs must be positive");
  CM3_REQUIRE(d->n_agents >= 1 && d->n_agents <= 8, "Checkers: n_agents must be in 1..8");
  CM3_REQUIRE(d->n_rows >= 1 && d->n_rows % 2 == 1, "n_rows must be odd (checkers.py:16)");
  CM3_REQUIRE(d->n_columns >= 2 && d->n_columns % 2 == 0, "n_columns must be even (checkers.py:17)");
  CM3_REQUIRE(d->n_rows * d->n_columns <= 64, "n_rows*n_columns must be <= 64 (collected-mask width)");
  CM3_REQUIRE(d->n_obs >= 0 && d->n_obs <= 8, "n_obs out of range");
  CM3_REQUIRE(d->max_steps >= 1, "max_steps must be >= 1");
  CM3_REQUIRE((d->flags & ~(CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS | CM3_FLAG_FUSED_TICKS)) == 0, "unknown flag bits");
  memset(&p, 0, sizeof(p));
  p.n_ticks = 1;
  p.E = d->n_envs;
  p.R = d->n_rows;
  p.C = d->n_columns;
  p.O = d->n_obs;
  p.TR = p.R + 2 * p.O;
  p.TC = p.C + 2 * p.O + 1;
  p.K = 2 * p.O + 1;
  CM3_REQUIRE(p.TR <= 255 && p.TC <= 255, "grid too large for packed agent words");
  p.max_steps = d->max_steps;
  p.flags = d->flags & ~CM3_FLAG_FUSED_TICKS;
  p.max_collectible = p.R * p.C;
  p.env_id_base = d->env_id_base;
  p.seed = d->seed;
  for (int k = 0; k < p.R; ++k)
    for (int j = 0; j < p.C; ++j) {
      const uint64_t bit = 1ull << (k * p.C + j);
      if (((k + j) & 1) == 0) p.green_mask |= bit; else p.orange_mask |= bit;
    }
  for (int i = 0; i < d->n_agents; ++i) {
    p.start_r[i] = d->agents_r[i] + p.O;
    p.start_c[i] = d->agents_c[i] + p.O;
    CM3_REQUIRE(p.start_r[i] >= p.O && p.start_r[i] < p.O + p.R && p.start_c[i] >= p.O && p.start_c[i] <= p.O + p.C,
                "agent %d starts outside the playable band", i);
    for (int j = 0; j < i; ++j)
      CM3_REQUIRE(!(p.start_r[i] == p.start_r[j] && p.start_c[i] == p.start_c[j]),
                  "agents %d and %d start on the same cell (unsupported: channel 2 is derived from agent cells)", j, i);
  }
  p.grid_rec = p.R * (p.C + 1) * 2;
  p.obst_rec = d->n_agents * p.K * p.K * 3;
  p.grid_stride = d->grid_stride ? d->grid_stride : p.grid_rec;
  p.obst_stride = d->obs_self_t_stride ? d->obs_self_t_stride : p.obst_rec;
  CM3_REQUIRE(p.grid_stride >= p.grid_rec && p.obst_stride >= p.obst_rec, "record strides smaller than the records");
  if (!ck_fast_ok(p)) {
    CM3_REQUIRE(p.grid_stride == p.grid_rec && p.obst_stride == p.obst_rec,
                "padded records are only supported by the fast kernel (3x8 band, n_obs 2, strides multiple of 4)");
    CM3_REQUIRE(64 * p.grid_rec <= kCkLdsBytes && 64 * p.obst_rec <= kCkLdsBytes,
                "observation record too large for the staging tile");
  }
  CM3_REQUIRE(b->mask && b->agents && b->steps && b->goals, "state pointers (mask/agents/steps/goals) are required");
  CM3_REQUIRE(b->grid && b->vec && b->obs_others && b->obs_self_t && b->obs_self_v, "observation outputs are required");
  if (step) {
    CM3_REQUIRE(b->actions && b->local_rewards && b->reward && b->done, "step outputs/actions are required");
    if (d->flags) CM3_REQUIRE(b->episode, "episode counter is required with AUTO_RESET / GEN_ACTIONS");
  }
  p.mask = b->mask;
  p.agents = b->agents;
  p.steps = b->steps;
  p.episode = b->episode;
  p.goals = const_cast<uint8_t *>(b->goals);
  p.ablock = b->action_block;
  p.actions = b->actions;
  p.grid = b->grid;
  p.vec = b->vec;
  p.obs_others = b->obs_others;
  p.obs_self_t = b->obs_self_t;
  p.obs_self_v = b->obs_self_v;
  p.local_rewards = b->local_rewards;
  p.reward = b->reward;
  p.done = b->done;
  {
    const int have = (b->term_grid != nullptr) + (b->term_vec != nullptr) + (b->term_obs_others != nullptr) +
                     (b->term_obs_self_t != nullptr) + (b->term_obs_self_v != nullptr);
    CM3_REQUIRE(have == 0 || have == 5, "terminal capture needs all five term_* arrays (or none)");
  }
  p.term_grid = b->term_grid;
  p.term_vec = b->term_vec;
  p.term_obs_others = b->term_obs_others;
  p.term_obs_self_t = b->term_obs_self_t;
  p.term_obs_self_v = b->term_obs_self_v;
  p.goals_next = b->goals_next;
  p.reset_mask = mask;
  if (step && ck_fast_ok(p) && (p.flags & CM3_FLAG_AUTO_RESET)) ck_fresh_fill(p, d->n_agents);
  if (step) CM3_SPAN_SET(p);
  return CM3_OK;
}

template <int N> static int ck_launch(const CheckersParams &p, bool step, hipStream_t stream) {
  if (ck_fast_ok(p)) {
    const bool nt = (p.flags & kCkObsStoreNt) != 0;   // streaming-size trajectory (ck_rollout): non-temporal stores, kCkGStream lanes per env
    const unsigned epb = 4u * (nt ? CkFast<N, kCkGStream>::EPW : CkFast<N>::EPW);  // 4 waves x EPW envs per workgroup
    const unsigned raw_blocks = (unsigned)(((size_t)p.E + epb - 1) / epb);
    const unsigned fblocks = cm3_xcd_grid(raw_blocks);         // XCD-aware block order (common.h)
    const uint32_t xf = cm3_xcd_flags(raw_blocks);
    if (step) {  // the step kernel indexes with 32-bit byte offsets: the widest per-env record of any per-tick array bounds E
      // obs_self_t (stride), grid (stride), obs_self_v 32 N, obs_others 16 N max(N-1, 1), vec 16 N, local_rewards 8 N
      size_t widest = (size_t)p.obst_stride;
      const size_t others = (size_t)16 * N * (N > 1 ? N - 1 : 1);
      if ((size_t)p.grid_stride > widest) widest = (size_t)p.grid_stride;
      if ((size_t)N * 32 > widest) widest = (size_t)N * 32;
      if (others > widest) widest = others;
      if ((size_t)p.E * widest >= ((size_t)1 << 32))
        return fail(CM3_ERR_INVALID, "the Checkers step kernel addresses at most 4 GiB per array: %d envs x %d agents is too large", p.E, N);
    }
#define CM3_LAUNCH_CKF(...)                                                                                                  \
  hipLaunchKernelGGL((k_checkers_step_fast<N, __VA_ARGS__>), dim3(fblocks), dim3(256), 0, stream, (const uint64_t *)p.mask,    \
                     (const uint32_t *)p.agents, (const int32_t *)p.steps, (const int32_t *)p.episode, (const uint8_t *)p.goals, \
                     p.ablock, p.E, p.flags | xf, p)
    note_variant(step ? "k_checkers_step_fast" : "k_checkers_reset_fast", 0, N, 4, step && p.n_ticks > 1, step && nt ? 1 : 0, 0, 0, 0,
                 step ? (nt ? kCkGStream : kCkG) : 0);
    if (step && p.n_ticks > 1) {
      if (nt) CM3_LAUNCH_CKF(true, true, kCkGStream);
      else CM3_LAUNCH_CKF(true);
    } else if (step) {
      if (nt) CM3_LAUNCH_CKF(false, true, kCkGStream);
      else CM3_LAUNCH_CKF(false);
    } else
      hipLaunchKernelGGL((k_checkers_reset_fast<N>), dim3(fblocks), dim3(256), 0, stream, p);
#undef CM3_LAUNCH_CKF
    CM3_HIP_CHECK(hipGetLastError());
    return CM3_OK;
  }
  const unsigned blocks = (unsigned)(((size_t)p.E + 63) / 64);
  note_variant(step ? "k_checkers_step" : "k_checkers_reset", 0, N, 1, 0, 0, 0, 0);
  if (step)
    hipLaunchKernelGGL((k_checkers_step<N>), dim3(blocks), dim3(64), 0, stream, p);
  else
    hipLaunchKernelGGL((k_checkers_reset<N>), dim3(blocks), dim3(64), 0, stream, p);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

static int ck_dispatch(const CheckersParams &p, int n_agents, bool step, hipStream_t s) {
  switch (n_agents) {
    case 1: return ck_launch<1>(p, step, s);
    case 2: return ck_launch<2>(p, step, s);
    case 3: return ck_launch<3>(p, step, s);
    case 4: return ck_launch<4>(p, step, s);
    case 5: return ck_launch<5>(p, step, s);
    case 6: return ck_launch<6>(p, step, s);
    case 7: return ck_launch<7>(p, step, s);
    case 8: return ck_launch<8>(p, step, s);
  }
  return fail(CM3_ERR_INVALID, "n_agents %d unsupported", n_agents);
}

static int ck_call(const cm3_checkers_desc *d, const cm3_checkers_bufs *b, const uint8_t *mask, bool step, void *stream) {
  CheckersParams p;
  int rc = ck_fill(d, b, mask, step, p);
  if (rc != CM3_OK) return rc;
  return ck_dispatch(p, d->n_agents, step, (hipStream_t)stream);
}

// tick k of a trajectory as one step call's buffers: observation slot k + 1, per-tick outputs slot k
static void ck_traj_bufs(const cm3_checkers_traj *t, int k, cm3_checkers_bufs &b) {
  auto at = [](void *base, size_t stride, int k) -> void * {
    return base ? (void *)((char *)base + stride * (size_t)k) : nullptr;
  };
  memset(&b, 0, sizeof(b));
  b.mask = t->mask;
  b.agents = t->agents;
  b.steps = t->steps;
  b.episode = t->episode;
  b.goals = t->goals;
  b.action_block = t->action_block;
  b.actions = (int32_t *)at(t->actions, t->actions_stride, k);
  b.grid = (int8_t *)at(t->grid, t->grid_slot_stride, k + 1);
  b.vec = (int32_t *)at(t->vec, t->vec_stride, k + 1);
  b.obs_others = (double *)at(t->obs_others, t->obs_others_stride, k + 1);
  b.obs_self_t = (int8_t *)at(t->obs_self_t, t->obs_self_t_slot_stride, k + 1);
  b.obs_self_v = (double *)at(t->obs_self_v, t->obs_self_v_stride, k + 1);
  b.local_rewards = (double *)at(t->local_rewards, t->local_rewards_stride, k);
  b.reward = (double *)at(t->reward, t->reward_stride, k);
  b.done = (uint8_t *)at(t->done, t->done_stride, k);
  b.term_grid = (int8_t *)at(t->term_grid, t->term_grid_slot_stride, k);
  b.term_vec = (int32_t *)at(t->term_vec, t->term_vec_stride, k);
  b.term_obs_others = (double *)at(t->term_obs_others, t->term_obs_others_stride, k);
  b.term_obs_self_t = (int8_t *)at(t->term_obs_self_t, t->term_obs_self_t_slot_stride, k);
  b.term_obs_self_v = (double *)at(t->term_obs_self_v, t->term_obs_self_v_stride, k);
  b.goals_next = (uint8_t *)at(t->goals_slots, t->goals_slots_stride, k + 1);
}

// the tick loop of ONE launch over that trajectory (p filled from tick 0's buffers): tick k uses <pointer> + k * <stride>
static void ck_traj_strides(const cm3_checkers_traj *t, int32_t n_ticks, CheckersParams &p) {
  p.n_ticks = n_ticks;
  p.st_actions = t->actions_stride;
  p.st_grid = t->grid_slot_stride;
  p.st_vec = t->vec_stride;
  p.st_obs_others = t->obs_others_stride;
  p.st_obs_self_t = t->obs_self_t_slot_stride;
  p.st_obs_self_v = t->obs_self_v_stride;
  p.st_local = t->local_rewards_stride;
  p.st_reward = t->reward_stride;
  p.st_done = t->done_stride;
  p.st_term_grid = t->term_grid_slot_stride;
  p.st_term_vec = t->term_vec_stride;
  p.st_term_obs_others = t->term_obs_others_stride;
  p.st_term_obs_self_t = t->term_obs_self_t_slot_stride;
  p.st_term_obs_self_v = t->term_obs_self_v_stride;
  p.st_goals_next = t->goals_slots_stride;
}

static int ck_rollout(const cm3_checkers_desc *d, const cm3_checkers_traj *t, int32_t n_ticks, void *stream) {
  CM3_REQUIRE(d && t, "null desc/traj");
  CM3_REQUIRE(n_ticks >= 1, "n_ticks must be >= 1");
  // the rollout's observation slots as a stream (>= 128 MB, beyond what the cache hierarchy keeps): non-temporal stores
  const size_t obs_bytes = (t->grid_slot_stride + t->vec_stride + t->obs_others_stride + t->obs_self_t_slot_stride +
                            t->obs_self_v_stride) * (size_t)n_ticks;
  const uint32_t nt_flag = obs_bytes >= ((size_t)128 << 20) ? kCkObsStoreNt : 0u;
  cm3_checkers_bufs b;
  if (d->flags & CM3_FLAG_FUSED_TICKS) {
    CM3_REQUIRE((d->flags & CM3_FLAG_GEN_ACTIONS) || n_ticks == 1 || t->actions_stride != 0,
                "fused rollout needs in-kernel actions or one pre-filled action slot per tick");
    ck_traj_bufs(t, 0, b);
    CheckersParams p;
    int rc = ck_fill(d, &b, nullptr, true, p);
    if (rc != CM3_OK) return rc;
    CM3_REQUIRE(ck_fast_ok(p), "fused Checkers rollouts need the fast kernel (3x8 band, n_obs 2, 4-byte padded records)");
    p.flags |= nt_flag;
    ck_traj_strides(t, n_ticks, p);
    return ck_dispatch(p, d->n_agents, true, (hipStream_t)stream);
  }
  for (int k = 0; k < n_ticks; ++k) {
    ck_traj_bufs(t, k, b);
    CheckersParams p;
    int rc = ck_fill(d, &b, nullptr, true, p);
    if (rc != CM3_OK) return rc;
    p.flags |= nt_flag;
    rc = ck_dispatch(p, d->n_agents, true, (hipStream_t)stream);
    if (rc != CM3_OK) return rc;
  }
  return CM3_OK;
}

}  // namespace cm3

#ifndef CM3_NO_ENTRY_POINTS
extern "C" {
int cm3_checkers_rollout(const cm3_checkers_desc *d, const cm3_checkers_traj *t, int32_t n_ticks, void *stream) {
  return cm3::ck_rollout(d, t, n_ticks, stream);
}
namespace cm3 {
__global__ void __launch_bounds__(256) k_checkers_action_blocks(uint32_t *out, int E, int64_t env_id_base, uint64_t seed) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= E) return;
  const u32x4 w = action_block(seed, (uint64_t)(env_id_base + (int64_t)e), 0u);
  reinterpret_cast<u32x4 *>(out)[e] = w;
}
}  // namespace cm3

int cm3_checkers_action_blocks(const cm3_checkers_desc *d, uint32_t *out, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(d && out, "null argument");
  CM3_REQUIRE(d->n_envs > 0, "n_envs must be positive");
  CM3_REQUIRE(((uintptr_t)out % 16) == 0, "the table must be 16-byte aligned");
  hipLaunchKernelGGL(k_checkers_action_blocks, dim3((unsigned)((d->n_envs + 255) / 256)), dim3(256), 0, (hipStream_t)stream, out, d->n_envs,
                     d->env_id_base, d->seed);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

int cm3_checkers_step(const cm3_checkers_desc *d, const cm3_checkers_bufs *b, void *stream) {
  return cm3::ck_call(d, b, nullptr, true, stream);
}
int cm3_checkers_reset(const cm3_checkers_desc *d, const cm3_checkers_bufs *b, const uint8_t *mask, void *stream) {
  return cm3::ck_call(d, b, mask, false, stream);
}
}
#endif  // CM3_NO_ENTRY_POINTS
