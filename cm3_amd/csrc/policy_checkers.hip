// A whole POLICY-DRIVEN Checkers episode in one launch: actor forward pass (matrix cores, split float16) + epsilon-mixed sampling +
// the sequential env step + observation, tick after tick, with the env state in registers and the network's inputs in LDS.
//
// Reference loop being replaced (alg/train_onpolicy.py:302-350 / train_offpolicy.py:309-368, the branch taken after the 50 pretrain
// episodes):
//     actions = alg.run_actor(actions_prev, obs_others, obs_self_t, obs_self_v, goals, epsilon, sess)   :309-313
//               (alg_credit_checkers.py:229-253; networks.convnet_1 + actor_checkers, networks.py:67-75, :549-578)
//     next_... = env.step(actions)                                                                      :321   (checkers.py:228-262)
//     buf.add(transition) ; roll state forward, actions_prev = actions                                  :336-347
// The unfused device path (CheckersRollout.collect(policy=actor)) alternates k_ck_actor_x3 and k_checkers_step_fast launches inside a
// hipGraph: 29.6 us per tick at 8192 envs x 2 agents -- 23.4 us of actor launch, of which ~8 k of 45 k cycles stage the observation the
// step launch has just written to memory and ~11 k run the others branch, plus two launch boundaries and the step launch per tick.
// Here a 512-thread workgroup owns 64 agent rows = 64 / N whole envs for all T ticks:
//   * the env state (collected mask, packed agent words, counters, goals) lives in the registers of the env's 8 N lanes (the mapping of
//     k_checkers_step_fast, twice the lanes for two agents: every lane runs the short sequential update, each produces its share of
//     the env's outputs);
//   * after a tick those lanes write the NEXT tick's network inputs straight into LDS (window bytes as float16, the normalised
//     vector, the one-hots of the actions just taken and of the goal) next to the trajectory slot in memory;
//   * the others branch (two agents: v_obs_others = the normalised cell of the ONE other agent, 91 possible values) is a table
//     lookup: h2's accumulators start from the row cm3_actor_checkers_pack built with the forward pass's own function
//     (k_ck_actor_others_table), bit for bit what the unfused actor computes; the tick's 64 rows come straight into LDS
//     (global_load_lds_dwordx4, CkpTableHooks);
//   * conv -> conv_linear -> branch_self -> h2 -> actor_out as in the stand-alone kernel (ck_x3_self_chain), weights streamed from L2.
// Trajectories (every array of CheckersRollout, actions and probabilities included) are bit-identical to the alternating launches
// (tests/test_gpu_actor_checkers.py::test_checkers_policy_rollout_equals_launch_per_tick).  Precision 2 (split float16) only; N = 1
// (stage 1) or N = 2 (stage 2), the reference's Checkers configurations, reference geometry (3 x 8 band, n_obs 2).
#ifndef CM3_POLICY_CHECKERS_BODY_ONLY   // (tools/probes/ck_policy_timeline.hip includes the two files WITH their entry points first)
#define CM3_NO_ENTRY_POINTS 1
#include "checkers.hip"
#include "actor_checkers.hip"
#endif

namespace cm3 {

struct CkPolicyParams {
  CheckersParams ck;      // FIRST (the fresh-episode record is read from the kernel-argument segment): trajectory pointers / strides of
                          // a CM3_FLAG_FUSED_TICKS step (observation pointers = slot 1, per-tick outputs = slot 0)
  const float *packed;
  const int32_t *prev0;   // optional int32 [E][N]: actions_prev of tick 0 (NULL = zeros, train_onpolicy.py:295)
  int32_t *prev0_next;    // optional int32 [E][N], out: actions_prev of the NEXT rollout's first tick (the last actions, zeros where the
                          // last tick ended an episode under auto-reset)
  float *probs;           // optional float [T][E][N][5]
  size_t st_probs;
  const float *eps_dev;
  float eps;
  int stage;
  // slot 0 of the observation arrays (+ of the goal slots): written at entry from the live state, i.e. what the env's current
  // observation holds -- the caller need not copy it there
  CkOut slot0;
  uint8_t *goals_slot0;
  // optional: the env's own current-observation buffers and last-actions buffer, written after the last tick (= slot n_ticks /
  // the last action slot): the caller need not copy them back
  CkOut final_obs;
  int32_t *final_actions;
};

// Lanes per env: 8 N -- the 512 lanes of the workgroup over its 64 / N envs.  (The step kernel of the random-action path uses 8: every
// lane repeats the short sequential update and produces 1 / G of the outputs.  Here the env phase sits on every tick's critical path
// between two matrix passes and all eight waves would otherwise wait for four of them: with 16 lanes per two-agent env a lane's share
// of the emit halves.)
template <int N> struct CkpGeom { static constexpr int G = 8 * N; };

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is also a release fence for GLOBAL memory: behind the tick's
// trajectory stores it made every wave wait for their acknowledgement (~1 k cycles, twice per tick).  Nothing in this kernel reads
// back what it stored to global memory, so the two barriers that follow stores wait for the LDS only.
__device__ __forceinline__ void ckp_barrier_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");   // LDS traffic only (no wait for global stores in flight)
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// the 5 x 5 x 3 windows of an env's agents -> float16 rows of X0 (what ck_x3_stage_inputs makes of the obs_self_t bytes): the cell
// expressions of ckf_emit_tab, lane q of the env takes window cells 4q .. 4q + 3
template <int N>
__device__ __forceinline__ void ckp_windows_to_x0(const CkState<N> &s, const CkLanePlan<N, CkpGeom<N>::G> &pl, const uint4 *lds_tab, int g,
                                                  _Float16 *X0, int row0) {
  using T = CkPlanTab<N>;
  using P = CkLanePlan<N, CkpGeom<N>::G>;
  constexpr int kCkpG = CkpGeom<N>::G;
  const char *tab = reinterpret_cast<const char *>(lds_tab);
  const uint32_t m32 = (uint32_t)s.mask;
  uint32_t rc[N], base[N];
#pragma unroll
  for (int a = 0; a < N; ++a) {
    rc[a] = (uint32_t)s.r[a] | ((uint32_t)s.c[a] << 8);
    base[a] = (uint32_t)s.r[a] * 64u + (uint32_t)s.c[a] * 4u;
  }
  uint32_t ent[P::NSLOT][4], rcx[P::NSLOT][4];
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it)
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
  uint16_t *X0w = reinterpret_cast<uint16_t *>(X0);
#pragma unroll
  for (int it = 0; it < P::NSLOT; ++it) {
    const int q = it * kCkpG + g;
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      bool agent = false;
#pragma unroll
      for (int a = 0; a < N; ++a) agent = agent | (rc[a] == rcx[it][x]);
      const uint32_t en = ent[it][x];
      const uint32_t got = (uint32_t)__builtin_amdgcn_sbfe((int)m32, en >> 24, 1);
      const uint32_t v = (en ^ (en & got & 0xfefeu)) | (agent ? pl.amask[it][x] : 0u);
      // channel byte b in {0xff, 0, 1} -> float16 {-1, 0, +1}: (b & 1) ? 0x3c00 : 0, sign from bit 7
      const int i = (int)pl.agent[it][x];
      const int cw = 4 * q + x - T::KK * i;
      const int at = (row0 + i) * ck_actor::kLhX0 + 3 * cw;
      if (q < T::NQ && !pl.invalid[it][x]) {
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
          const uint32_t b = v >> (8 * ch);
          X0w[at + ch] = (uint16_t)(((b & 1u) ? 0x3c00u : 0u) | ((b & 0x80u) << 8));
        }
      }
    }
  }
}

// the k padding of an X0 row (window bytes 75 .. 95 of the 96 the conv contracts over): X0 lives inside the H storage, which branch_self's
// output overwrites every tick -- and 0 x (whatever it left) must be 0
__device__ __forceinline__ void ckp_zero_x0_pad(_Float16 *X0, int row) {
  using namespace ck_actor;
  static_assert(kObs == 75 && kKConvX == 96 && (kLhX0 * 2) % 8 == 0, "one 2-byte and five 8-byte stores per row");
  _Float16 *r = X0 + row * kLhX0;
  r[kObs] = (_Float16)0.0f;
#pragma unroll
  for (int k = 0; k < 5; ++k) *reinterpret_cast<uint2 *>(r + kObs + 1 + 4 * k) = make_uint2(0u, 0u);
}

// lane g < N of an env: the tail of agent g's X2 row -- v_obs_self (the env's normalised values, cast to float32 like a TF feed and
// split), the one-hots of a_prev and of the goal -- and, for two agents, the table row of the others branch: the OTHER agent's cell
template <int N>
__device__ __forceinline__ void ckp_row_inputs(const CkState<N> &s, const uint8_t (&goal)[N], const int (&aprev)[N], const uint4 *lds_tab,
                                               int g, const CkX3Planes &L, int32_t *sCell, int row0) {
  using namespace ck_actor;
  if (g < N) {
    ckp_zero_x0_pad(L.X0, row0 + g);
    int r = s.r[0], c = s.c[0], ng = s.ng[0], no = s.no[0], gl = goal[0], ap = aprev[0];
    int orr = s.r[N > 1 ? 1 : 0], occ = s.c[N > 1 ? 1 : 0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
      const bool me = g == a;
      r = me ? s.r[a] : r; c = me ? s.c[a] : c; ng = me ? s.ng[a] : ng; no = me ? s.no[a] : no;
      gl = me ? (int)goal[a] : gl; ap = me ? aprev[a] : ap;
      orr = me ? s.r[0] : orr; occ = me ? s.c[0] : occ;     // (N == 2: the other agent of agent 1 is agent 0)
    }
    const double *norm = reinterpret_cast<const double *>(reinterpret_cast<const char *>(lds_tab) + 448);
    const double v[4] = {norm[r], norm[7 + c], norm[20 + ng], norm[20 + no]};
    const int at0 = (row0 + g) * kLhX2 + kLin;
#pragma unroll
    for (int k = 0; k < 4; ++k) put_split(L.X2h, L.X2l, at0 + k, (float)v[k]);
#pragma unroll
    for (int k = 0; k < kA; ++k) L.X2h[at0 + 4 + k] = ap == k ? (_Float16)1.0f : (_Float16)0.0f;
    L.X2h[at0 + 9] = gl == 0 ? (_Float16)1.0f : (_Float16)0.0f;
    L.X2h[at0 + 10] = gl == 0 ? (_Float16)0.0f : (_Float16)1.0f;
    sCell[row0 + g] = orr * 13 + occ;
  }
}

// ... and the per-tick form: the emit hands its window cells and normalised values over (CkpSink, from the registers that hold them
// anyway -- recomputing them as above cost 3.8 k cycles of every tick), lane g < N adds the one-hots and the others-branch cell
struct CkpSink {
  static constexpr bool kOn = true;
  uint16_t *X0w;
  _Float16 *X2h, *X2l;
  int row0;
  __device__ __forceinline__ void cell(uint32_t agent, int cw, uint32_t v) const {
    // three channel bytes in {0xff, 0, 1} -> the HIGH bytes of their float16 values {0xbc, 0, 0x3c}, all three at once
    const uint32_t hb = (v & 0x808080u) | ((v & 0x010101u) * 0x3cu);
    const int at = (row0 + (int)agent) * ck_actor::kLhX0 + 3 * cw;
    X0w[at] = (uint16_t)((hb & 0xffu) << 8);
    X0w[at + 1] = (uint16_t)(hb & 0xff00u);
    X0w[at + 2] = (uint16_t)((hb >> 8) & 0xff00u);
  }
  __device__ __forceinline__ void value(bool others, uint32_t idx, double val) const {
    if (!others) put_split(X2h, X2l, (row0 + (int)(idx >> 2)) * ck_actor::kLhX2 + ck_actor::kLin + (int)(idx & 3u), (float)val);
  }
};
template <int N>
__device__ __forceinline__ void ckp_row_flags(const CkState<N> &s, const uint8_t (&goal)[N], const int (&aprev)[N], int g,
                                              const CkX3Planes &L, int32_t *sCell, int row0) {
  using namespace ck_actor;
  if (g < N) {
    ckp_zero_x0_pad(L.X0, row0 + g);
    int gl = goal[0], ap = aprev[0];
    int orr = s.r[N > 1 ? 1 : 0], occ = s.c[N > 1 ? 1 : 0];
#pragma unroll
    for (int a = 1; a < N; ++a) {
      const bool me = g == a;
      gl = me ? (int)goal[a] : gl; ap = me ? aprev[a] : ap;
      orr = me ? s.r[0] : orr; occ = me ? s.c[0] : occ;
    }
    const int at0 = (row0 + g) * kLhX2 + kLin;
#pragma unroll
    for (int k = 0; k < kA; ++k) L.X2h[at0 + 4 + k] = ap == k ? (_Float16)1.0f : (_Float16)0.0f;
    L.X2h[at0 + 9] = gl == 0 ? (_Float16)1.0f : (_Float16)0.0f;
    L.X2h[at0 + 10] = gl == 0 ? (_Float16)0.0f : (_Float16)1.0f;
    sCell[row0 + g] = orr * 13 + occ;
  }
}

// h2's accumulators start from the others-branch table (two agents; stage 1: zeros): row = the OTHER agent's cell.  A lane's
// accumulators cover 16 agent rows per tile -- read straight from the table that is 16 scattered 64-byte pieces per load, 128
// cache-line requests per wave and tick, and cost 1.0 - 1.5 us of a 16.5 us tick.  Instead wave w fetches the eight WHOLE table rows
// of agent rows [8w, 8w + 8), one coalesced 1 KB request each, STRAIGHT INTO LDS (global_load_lds_dwordx4: wave-uniform LDS base + 16
// bytes per lane = a row of sT) in the middle of the conv phase -- behind the conv's own matrix instructions: the compiler waits for EVERY
// vector-memory operation in flight at the next use of a loaded register once an LDS-direct load is among them, and at the top of the tick that
// made the conv wait for the rows too -- and every wave picks its units out of LDS before the h2 pass.  (Until late
// round 6 the rows went through 32 registers per lane -- requested behind the conv because X0 shared sT's storage, parked behind
// conv_linear -- and one of them through scratch memory.)  The rows are ordered for the readers by the conv's closing barrier: the
// compiler drains the vector-memory counter ahead of every barrier while an LDS-direct load is in flight.
constexpr int kCkpTabLd = 256 + CM3_CK_LD_PAD / 2;   // (floats: 8 dwords mod 64 -- the same lane groups read it with 16-byte loads, see kLdHb)
struct CkpTableHooks {
  const float *tab;     // NULL: stage 1, no others branch
  const float *pk;
  const int32_t *sCell;
  float *sT;
  int w, lane;
  __device__ __forceinline__ void mid_conv() {
#if !defined(CM3_PROBE_NO_TABLE) && defined(__HIP_DEVICE_COMPILE__)
    if (tab) {
      typedef const __attribute__((address_space(1))) void *GlobalPtr;
      typedef __attribute__((address_space(3))) void *LdsPtr;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float *src = tab + (size_t)sCell[8 * w + j] * ck_actor::kH2 + 4 * lane;
        __builtin_amdgcn_global_load_lds((GlobalPtr)src, (LdsPtr)(sT + (8 * w + j) * kCkpTabLd), 16, 0, 0);
      }
    }
#endif
  }
  __device__ __forceinline__ void after_conv() {}
  // Behind conv_linear's barrier, ahead of branch_self's: this wave's rows have landed (vmcnt(0), written out: today the compiler's own
  // wait at conv_linear's first use of a loaded weight already drains the counter, but nothing obliges it to) -- with the barrier that
  // follows, that is what orders LDS-direct data for the other waves' reads in before_h2()
  __device__ __forceinline__ void after_lin() {
#if !defined(CM3_PROBE_NO_TABLE) && defined(__HIP_DEVICE_COMPILE__)
    if (tab) __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), the other counters untouched
#endif
  }
  __device__ __forceinline__ void before_h2(f32x4 (&acc2)[4][kCkBCT]) {
#ifndef CM3_PROBE_NO_TABLE
    if (tab) {
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int c = 0; c < kCkBCT; ++c) {
          const float4 v = *reinterpret_cast<const float4 *>(sT + (16 * tt + (lane & 15)) * kCkpTabLd + 16 * (kCkBCT * w + c) + 4 * (lane >> 4));
          acc2[tt][c] = f32x4{v.x, v.y, v.z, v.w};
        }
      return;
    }
#endif
    ck_x3_h2_bias(pk, w, lane, acc2);   // stage 1: no others branch, the accumulators start from h2's bias
  }
};

template <int N> __global__ void __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2))) k_ck_policy_rollout(const CkPolicyParams q_arg) {
  using namespace ck_actor;
  constexpr int G = CkpGeom<N>::G, EPW = 64 / N, ENV_WAVES = EPW * G / 64;
  static_assert(ENV_WAVES == 8, "every wave takes part in the env phase");
  static_assert(N == 1 || N == 2, "the whole-episode Checkers kernel covers the reference's configurations: one or two agents");
  // The ~45 pointers / strides of the arguments are read from the kernel-argument segment where they are used, through a pointer the
  // compiler cannot see through inside the tick loop: kept in scalar registers for the whole launch, ~100 of them were spilled to
  // vector-register lanes and came back one v_readlane at a time (98 in the conv phase, 111 in the env phase of every tick).
#if defined(__HIP_DEVICE_COMPILE__)
  typedef const __attribute__((address_space(4))) CkPolicyParams *CkpArgs;
  const CkpArgs qa = (CkpArgs)__builtin_amdgcn_kernarg_segment_ptr();
#else
  typedef const CkPolicyParams *CkpArgs;
  const CkpArgs qa = &q_arg;
#endif
  const CkPolicyParams &q = *(const CkPolicyParams *)qa;
  const CheckersParams &p = q.ck;
  __shared__ __attribute__((aligned(16))) _Float16 sH[kCkX3HBytes / 2];
  __shared__ __attribute__((aligned(16))) float sT[64 * kCkpTabLd];    // the tick's 64 table rows
  // X0 lives in the H storage BEHIND the C1 planes: written by the env phase (nothing else touches the H storage between the h2 pass's
  // partial logits, its first 16 KB, and the conv), read by the conv (whose epilogue writes C1 below it), dead when branch_self writes H
  static_assert(2 * 64 * kLhC1 * 2 + kCkX3X0Bytes <= kCkX3HBytes && 8 * 64 * 8 * 4 <= 2 * 64 * kLhC1 * 2, "X0 behind C1 inside the H storage");
  _Float16 *sX0 = sH + 2 * 64 * kLhC1;
  __shared__ __attribute__((aligned(16))) _Float16 sX2[kCkX3X2Bytes / 2];
  __shared__ float sLG[64][8];
  __shared__ __attribute__((aligned(16))) uint4 lds_tab[48];           // board / norm table (CkBoardTab), one copy per workgroup
  __shared__ __attribute__((aligned(16))) uint4 lds_fresh[26];         // the fresh-episode observation record (CkFresh)
  __shared__ __attribute__((aligned(16))) uint4 sPlan[G][sizeof(CkLanePlanRaw<N, G>) / 16];   // the emit plan of lane g of ANY env
  __shared__ int32_t sAct[64], sCell[64];
  __shared__ int2 sMeta[EPW];                                          // {episode, steps} of every env, for the sampling uniforms
  CkX3Planes L;
  L.Hh = sH; L.Hl = sH + 64 * kLdHb;
  L.X0 = sX0; L.C1h = sH; L.C1l = sH + 64 * kLhC1;
  L.X2h = sX2; L.X2l = sX2 + 64 * kLhX2; L.XOh = nullptr; L.XOl = nullptr;
  L.LG = sLG;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float *pk = q.packed;
  const size_t rows = (size_t)p.E * N;
  const size_t row_base = (size_t)blockIdx.x * 64;
  const bool stage2 = q.stage > 1;

  CkConvB b_conv;

  // ---- the env side: lane (el, g) of the env waves ------------------------------------------------------------------------------
  const bool env_wave = w < ENV_WAVES;                 // (wave-uniform)
  const int el = tid / G, g = tid & (G - 1), row0 = el * N;
  const uint32_t e = (uint32_t)blockIdx.x * EPW + (uint32_t)el;
  const bool env_ok = env_wave && e < (uint32_t)p.E;
  const uint32_t ec = e < (uint32_t)p.E ? e : (uint32_t)p.E - 1;
  const bool writer = env_ok && g == 0;
  CkState<N> s;
  CkLive<N> lv;
  int aprev[N];
  const bool auto_reset = (p.flags & CM3_FLAG_AUTO_RESET) != 0;
  if (env_wave) {
    CkHead hd = ck_head(p);
    hd.flags = (hd.flags & ~CM3_FLAG_GEN_ACTIONS) | CM3_FLAG_AUTO_RESET;   // (the load reads the episode counter under either flag: the
                                                                           // sampling uniforms are keyed by it with or without auto-reset)
    ck_load_env<N>(hd, ec, s, lv);
    static_assert(offsetof(CkPolicyParams, ck) == 0, "the fresh record is addressed inside the kernel-argument segment");
    const uint4 *fresh_src = nullptr;
#if defined(__HIP_DEVICE_COMPILE__)
    if (auto_reset)
      fresh_src = reinterpret_cast<const uint4 *>((const char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(CheckersParams, fresh));
#endif
    {   // the board / norm table and the fresh record into LDS (every env wave writes the same 48 + 26 vectors)
      const uint4 board_vec = reinterpret_cast<const uint4 *>(&kCkBoardTab)[lane < 48 ? lane : 47];
      uint4 fresh_vec = make_uint4(0u, 0u, 0u, 0u);
      if (fresh_src) fresh_vec = fresh_src[lane < 26 ? lane : 25];
      if (lane < 48) lds_tab[lane] = board_vec;
      if (lane < 26) lds_fresh[lane] = fresh_vec;
    }
#pragma unroll
    for (int i = 0; i < N; ++i) aprev[i] = q.prev0 ? q.prev0[(size_t)ec * N + i] : 0;
  }
  // the head side: lane l < 16 of wave w < 4 finishes agent row 16w + l
  const int row_l = 16 * (w & 3) + (lane & 15);
  const size_t row_h = row_base + row_l;
  const int el_h = row_l / N, i_h = row_l - el_h * N;
  const size_t e_h = row_h / N;
  const bool head_lane = w < 4 && lane < 16;
  uint32_t ublock = 0u;
  if (head_lane) ublock = actor_block_word(p.seed, (uint64_t)(p.env_id_base + (int64_t)(e_h < (size_t)p.E ? e_h : (size_t)p.E - 1)), i_h);
  // zero fill, once: the tail of X2 beyond the concat (43 .. 63) and the lo plane of its one-hots
  for (int idx = tid; idx < 64 * 32; idx += 512) {
    const int r = idx >> 5, k = kLin + (idx & 31);
    L.X2h[r * kLhX2 + k] = (_Float16)0.0f;
    L.X2l[r * kLhX2 + k] = (_Float16)0.0f;
  }
  __syncthreads();   // tables and zero fills are in
  if (env_wave) {
    // (the lane's emit plan -- which cells, which values it produces -- is NOT kept across the matrix phases: ~54 registers next to
    // the gemm's.  It depends on g alone: the first env's lanes park the 6 sixteen-byte entries in LDS, every tick re-reads them)
    CkLanePlanRaw<N, G> raw;
    CkLanePlan<N, G> pl;
    ckf_plan_fetch<N, G>(g, raw);
    if (tid < G) *reinterpret_cast<CkLanePlanRaw<N, G> *>(&sPlan[g][0]) = raw;
    ckf_plan_decode<N, G>(raw, pl);
    ckp_windows_to_x0<N>(s, pl, lds_tab, g, sX0, row0);
    ckp_row_inputs<N>(s, lv.goal, aprev, lds_tab, g, L, sCell, row0);
    if (g == 0) sMeta[el] = make_int2((int)lv.episode, lv.steps);
    ckf_emit_tab<N, false, G>(p, s, pl, lds_tab, g, e, env_ok, q.slot0);     // slot 0 = the observation the first tick acts on
    if (q.goals_slot0 && writer) {
#pragma unroll
      for (int i = 0; i < N; ++i) q.goals_slot0[(size_t)e * N + i] = lv.goal[i];
    }
  }
  __syncthreads();

  const int n_ticks = p.n_ticks;
  const float *const pk_loop = pk;
  // epsilon once per launch (inside the loop `q.eps_dev ? *q.eps_dev : q.eps` became a select between a device pointer and the
  // kernel-argument segment: one flat_load per tick, which also counts on the LDS counter)
  const float eps_now = q.eps_dev ? *q.eps_dev : q.eps;
#pragma unroll 1
  for (int t = 0; t < n_ticks; ++t) {
    CkpArgs qt = qa;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+s"(qt));
#endif
    const CkPolicyParams &q = *(const CkPolicyParams *)qt;   // (shadow the launch-wide references inside the loop)
    const CheckersParams &p = q.ck;
    const float *pk = pk_loop;
    // ---- policy ---------------------------------------------------------------------------------------------------------------
    // (a per-tick copy of the weight pointer the compiler cannot see through: with the loop-invariant one it hoisted every tile
    // address of every layer out of the tick loop and spilled ~100 vector registers to scratch memory.  The copy is typed as a
    // GLOBAL-memory pointer: through a plain one the compiler no longer knew the address space and every weight load became a
    // flat_load, which counts on the LDS counter too -- each wait for an LDS read then waited for the weights in flight)
    typedef const __attribute__((address_space(1))) float *CkpGlobalF;
    CkpGlobalF pkg = (CkpGlobalF)pk;
    asm volatile("" : "+s"(pkg));
    const float *pkt = (const float *)pkg;
    // (the conv's first weights are requested here, not ahead of the env work: held across it or across the closing barrier they cost
    // the two-agent build 14 spilled registers)
    ck_x3_load_conv(pkt, w, lane, b_conv);
    f32x4 acc2[4][kCkBCT];
    CkpTableHooks hooks;
    hooks.tab = stage2 ? pkt + kPOthTab : nullptr; hooks.pk = pkt; hooks.sCell = sCell; hooks.sT = sT; hooks.w = w; hooks.lane = lane;
    ck_x3_self_chain<CkpTableHooks &>(L, pkt, w, lane, b_conv, acc2, hooks);
    if (w < 4) {
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      if (lane < 16) {
        float o[kA], pr[kA];
#pragma unroll
        for (int a = 0; a < kA; ++a) o[a] = sLG[row_l][a];
        ck_actor_probs(o, eps_now, pr);
        const int2 m = sMeta[el_h];
        const int act = actor_pick(pr, actor_uniform_from(ublock, (uint32_t)m.x, m.y));
        sAct[row_l] = act;
        if (row_h < rows) {
          ck_tick_ptr(p.actions, p.st_actions, t)[row_h] = act;
          if (q.probs) {
            float *pt = ck_tick_ptr(q.probs, q.st_probs, t) + row_h * kA;
#pragma unroll
            for (int a = 0; a < kA; ++a) pt[a] = pr[a];
          }
        }
      }
    }
    CM3_STAMP(13, false);
#ifndef CM3_PROBE_NO_TAIL_BARRIERS
    ckp_barrier_lds();   // actions are in LDS; every wave is done with this tick's inputs and activations
#endif
    CM3_STAMP(14, false);
    // ---- env step of the workgroup's envs (checkers.py:228-262): k_checkers_step_fast's tick with the actions just sampled --------
#ifndef CM3_PROBE_SKIP_ENV   // (probe builds only, tools/r6/ck_whatif.sh: what each part of the tick costs the UNSTAMPED kernel)
    if (env_wave) {
#ifdef CM3_SPAN_MARKS
      unsigned long long span_marks_unused[8];   // (the marks build instruments the STEP kernels; ck_tick_env's marks land here)
#endif
      const CkLanePlanRaw<N, G> raw = *reinterpret_cast<const CkLanePlanRaw<N, G> *>(&sPlan[g][0]);
      int act[N];
#pragma unroll
      for (int i = 0; i < N; ++i) act[i] = sAct[row0 + i];
      const bool ended = ck_tick_env<N, true, true>(p, t, e, ec, writer, s, lv, reinterpret_cast<const char *>(lds_tab)
#ifdef CM3_SPAN_MARKS
                                                        , span_marks_unused
#endif
                                                        , act);
      CM3_STAMP(9, false);
      CkLanePlan<N, G> pl;
      ckf_plan_decode<N, G>(raw, pl);
      CkpSink sink;
      sink.X0w = reinterpret_cast<uint16_t *>(sX0); sink.X2h = L.X2h; sink.X2l = L.X2l; sink.row0 = row0;
      if (!__any(ended)) {
        ckf_emit_tab<N, false, G, false, CkpSink>(p, s, pl, lds_tab, g, e, env_ok, ck_out_tick(p, t), nullptr, false, sink);
      } else {
        // a wave with finished envs (k_checkers_step_fast): ONE emit -- terminal slot for the finished envs, regular slot for the
        // others -- then the finished envs restart and their regular slot gets the fresh-episode record
        const CkOut o_tick = ck_out_tick(p, t), o_term = ck_out_term(p, t);
        // (env_ok decides about the global stores only; the LDS inputs of a clamped lane's rows are never used)
        ckf_emit_tab<N, false, G, true, CkpSink>(p, s, pl, lds_tab, g, e, env_ok && (!ended || p.term_grid != nullptr), o_tick, &o_term, ended, sink);
        if (ended) {
          ck_restart_env<N>(p, e, ec, writer, s, lv);
          if (env_ok) ckf_fresh_copy<N, false, G>(p, lds_fresh, N == 1 ? (int)lv.goal[0] : 0, g, e, o_tick);
        }
      }
      if (p.goals_next && writer) {
        uint8_t *gn = ck_tick_ptr(p.goals_next, p.st_goals_next, t);
#pragma unroll
        for (int i = 0; i < N; ++i) *at32<uint8_t>(gn, e * N + i) = lv.goal[i];
      }
      CM3_STAMP(12, false);
      // the next tick's inputs: a fresh episode starts from actions_prev = zeros (train_onpolicy.py:295), otherwise the actions just taken
#pragma unroll
      for (int i = 0; i < N; ++i) aprev[i] = ended ? 0 : act[i];
      if (ended) {   // the fresh episode's observation was COPIED to the slot (CkFresh): its LDS form is worked out from the state
        ckp_windows_to_x0<N>(s, pl, lds_tab, g, sX0, row0);
        ckp_row_inputs<N>(s, lv.goal, aprev, lds_tab, g, L, sCell, row0);
      } else {
        ckp_row_flags<N>(s, lv.goal, aprev, g, L, sCell, row0);
      }
      if (g == 0) sMeta[el] = make_int2((int)lv.episode, lv.steps);
      CM3_STAMP(15, false);
    }
#endif
    ckp_barrier_lds();
    CM3_STAMP(8, false);
  }
  if (writer) {
    ck_store_env<N>(p, e, s, lv);
    if (q.prev0_next) {
#pragma unroll
      for (int i = 0; i < N; ++i) q.prev0_next[(size_t)e * N + i] = aprev[i];
    }
  }
  if (q.final_obs.grid && env_wave) {   // the env object's current observation <- the state the rollout leaves
    const CkLanePlanRaw<N, G> raw = *reinterpret_cast<const CkLanePlanRaw<N, G> *>(&sPlan[g][0]);
    CkLanePlan<N, G> pl;
    ckf_plan_decode<N, G>(raw, pl);
    ckf_emit_tab<N, false, G>(p, s, pl, lds_tab, g, e, env_ok, q.final_obs);
    if (q.final_actions && writer) {
#pragma unroll
      for (int i = 0; i < N; ++i) q.final_actions[(size_t)e * N + i] = sAct[row0 + i];
    }
  }
}

template <int N> static int ckp_launch(const CkPolicyParams &q, hipStream_t s) {
  const size_t rows = (size_t)q.ck.E * N;
  // the step's stores use 32-bit byte offsets (ck_launch): the widest per-env record of any per-tick array bounds E
  size_t widest = (size_t)q.ck.obst_stride;
  if ((size_t)q.ck.grid_stride > widest) widest = (size_t)q.ck.grid_stride;
  if ((size_t)N * 32 > widest) widest = (size_t)N * 32;
  if ((size_t)q.ck.E * widest >= ((size_t)1 << 32))
    return fail(CM3_ERR_INVALID, "the Checkers step addresses at most 4 GiB per array: %d envs x %d agents is too large", q.ck.E, N);
  note_variant("k_ck_policy_rollout", 0, N, 8, q.ck.n_ticks > 1, 2, 0, 0, 0, CkpGeom<N>::G);
  hipLaunchKernelGGL((k_ck_policy_rollout<N>), dim3((unsigned)((rows + 63) / 64)), dim3(512), 0, s, q);
  CM3_HIP_CHECK(hipGetLastError());
  return CM3_OK;
}

}  // namespace cm3

extern "C" int cm3_policy_rollout_checkers(const cm3_checkers_desc *d, const cm3_checkers_traj *t, const cm3_actor_checkers_desc *ad,
                                           const cm3_actor_checkers_weights *wt, const int32_t *actions_prev0, int32_t *actions_prev_next, float *probs,
                                           size_t probs_stride, const float *epsilon_dev, const cm3_checkers_bufs *final_obs,
                                           int32_t n_ticks, void *stream) {
  using namespace cm3;
  CM3_REQUIRE(d && t && ad && wt, "null argument");
  CM3_REQUIRE(n_ticks >= 1, "n_ticks must be >= 1");
  CM3_REQUIRE(d->n_agents == 1 || d->n_agents == 2,
              "the whole-episode Checkers policy rollout covers one or two agents (config_checkers_stage1 / stage2); got %d -- use "
              "alternating cm3_actor_checkers_f32 / cm3_checkers_step launches", d->n_agents);
  int rc = ck_actor_check(ad);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(ad->n_agents == d->n_agents && ad->n_envs == d->n_envs, "actor / env descriptors disagree");
  CM3_REQUIRE(ad->precision == 2, "the whole-episode Checkers policy rollout runs the split-float16 actor (precision 2)");
  CM3_REQUIRE((ad->stage > 1) == (d->n_agents > 1), "stage 1 has one agent, stage 2 two (the others branch is a table over the ONE other agent's cell)");
  CM3_REQUIRE(ad->seed == d->seed && ad->env_id_base == d->env_id_base, "actor and env must share seed and env_id_base");
  CM3_REQUIRE(ad->epsilon >= 0.0f && ad->epsilon <= 1.0f, "epsilon must be in [0,1]");
  CM3_REQUIRE(wt->packed, "weights->packed is NULL: run cm3_actor_checkers_pack once per weight update");
  CM3_REQUIRE(!(d->flags & ~CM3_FLAG_AUTO_RESET), "only CM3_FLAG_AUTO_RESET applies: the policy draws the actions, the ticks are fused by definition");
  CM3_REQUIRE(t->actions && (n_ticks == 1 || t->actions_stride != 0), "one action slot per tick is required");
  CM3_REQUIRE(t->episode, "the episode counter is required (it keys the sampling uniforms)");
  cm3_checkers_bufs b;
  ck_traj_bufs(t, 0, b);
  CkPolicyParams q;
  memset(&q, 0, sizeof(q));
  rc = ck_fill(d, &b, nullptr, true, q.ck);
  if (rc != CM3_OK) return rc;
  CM3_REQUIRE(ck_fast_ok(q.ck), "the whole-episode Checkers policy rollout needs the reference geometry (3x8 band, n_obs 2, 4-byte padded records)");
  CM3_REQUIRE(ad->obs_self_t_stride == q.ck.obst_stride, "actor / env obs_self_t strides disagree");
  ck_traj_strides(t, n_ticks, q.ck);
  q.packed = (const float *)wt->packed;
  q.prev0 = actions_prev0;
  q.prev0_next = actions_prev_next;
  q.probs = probs;
  q.st_probs = probs_stride;
  q.eps = ad->epsilon;
  q.eps_dev = epsilon_dev;
  q.stage = ad->stage;
  q.slot0.grid = t->grid; q.slot0.vec = t->vec; q.slot0.obs_others = t->obs_others; q.slot0.obs_self_t = t->obs_self_t;
  q.slot0.obs_self_v = t->obs_self_v;
  q.goals_slot0 = t->goals_slots;
  if (final_obs) {
    CM3_REQUIRE(final_obs->grid && final_obs->vec && final_obs->obs_others && final_obs->obs_self_t && final_obs->obs_self_v,
                "final_obs needs the five observation arrays (actions optional)");
    q.final_obs.grid = final_obs->grid; q.final_obs.vec = final_obs->vec; q.final_obs.obs_others = final_obs->obs_others;
    q.final_obs.obs_self_t = final_obs->obs_self_t; q.final_obs.obs_self_v = final_obs->obs_self_v;
    q.final_actions = final_obs->actions;
  }
  return d->n_agents == 1 ? ckp_launch<1>(q, (hipStream_t)stream) : ckp_launch<2>(q, (hipStream_t)stream);
}
