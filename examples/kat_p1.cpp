// A consumer of libcm3_hip.so that uses nothing but the C ABI of include/cm3_amd.h and the HIP runtime for its buffers
// (no Python, no PyTorch): replays known-answer vector KAT-P1 -- values recorded from the reference's own env in float64
// (two agents head-on, MultiAgentEnv.step of environment.py:81-123) -- through
// cm3_particle_reset_f64 / cm3_particle_step_f64 and compares every number (within 1e-12), then KAT-C1 (Checkers, exact)
// through cm3_checkers_reset / cm3_checkers_step.  Exit code 0 = both pass.
//   hipcc -std=c++17 -I include examples/kat_p1.cpp -L cm3_amd -lcm3_hip -Wl,-rpath,$PWD/cm3_amd -o examples/kat_p1
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "cm3_amd.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_CM3(x) do { int r_ = (x); if (r_ != CM3_OK) { printf("%s: %s\n", #x, cm3_last_error()); return 3; } } while (0)

// KAT-C1: Checkers stage 2 (3 x 8 band, n_obs 2, agents at (0,8) and (2,8)), goals = eye(2),
// values recorded from env/checkers.py: reset, then actions [4,1], [3,4], [3,3].  Integer / exact-decimal quantities: equality.
static int kat_c1() {
  const int E = 1, N = 2;
  cm3_checkers_desc d;
  memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.n_rows = 3; d.n_columns = 8; d.n_obs = 2; d.max_steps = 33; d.seed = 12341;
  d.grid_stride = 56; d.obs_self_t_stride = 152;                    // 4-byte padded records (the multi-lane kernel)
  d.agents_r[0] = 0; d.agents_r[1] = 2; d.agents_c[0] = 8; d.agents_c[1] = 8;
  cm3_checkers_bufs b;
  memset(&b, 0, sizeof(b));
  CHECK_HIP(hipMalloc((void **)&b.mask, 8)); CHECK_HIP(hipMalloc((void **)&b.agents, 4 * N)); CHECK_HIP(hipMalloc((void **)&b.steps, 4));
  CHECK_HIP(hipMalloc((void **)&b.episode, 4)); CHECK_HIP(hipMalloc((void **)&b.goals, N)); CHECK_HIP(hipMalloc((void **)&b.actions, 4 * N));
  CHECK_HIP(hipMalloc((void **)&b.grid, 56)); CHECK_HIP(hipMalloc((void **)&b.vec, 16 * N)); CHECK_HIP(hipMalloc((void **)&b.obs_others, 16 * N));
  CHECK_HIP(hipMalloc((void **)&b.obs_self_t, 152)); CHECK_HIP(hipMalloc((void **)&b.obs_self_v, 32 * N));
  CHECK_HIP(hipMalloc((void **)&b.local_rewards, 8 * N)); CHECK_HIP(hipMalloc((void **)&b.reward, 8)); CHECK_HIP(hipMalloc((void **)&b.done, 1));
  CHECK_HIP(hipMemset(b.episode, 0, 4));
  const uint8_t goal_idx[2] = {0, 1};                                // np.eye(2): agent 0 wants green, agent 1 orange
  CHECK_HIP(hipMemcpy(b.goals, goal_idx, 2, hipMemcpyHostToDevice));
  CHECK_CM3(cm3_checkers_reset(&d, &b, NULL, NULL));
  int32_t vec[8]; double oo[4], ov[8], loc[2], tot;
  CHECK_HIP(hipMemcpy(vec, b.vec, sizeof(vec), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(oo, b.obs_others, sizeof(oo), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(ov, b.obs_self_v, sizeof(ov), hipMemcpyDeviceToHost));
  const int32_t vec0[8] = {2, 10, 0, 0, 4, 10, 0, 0};
  if (memcmp(vec, vec0, sizeof(vec)) != 0) { printf("KAT-C1 reset: vec mismatch\n"); return 1; }
  // normalize(): (r - 7/2) / 7, (c - 13/2) / 13 in float64 (checkers.py:112-125)
  const double oo0[4] = {(4 - 3.5) / 7.0, (10 - 6.5) / 13.0, (2 - 3.5) / 7.0, (10 - 6.5) / 13.0};
  if (memcmp(oo, oo0, sizeof(oo)) != 0 || ov[0] != (2 - 3.5) / 7.0 || ov[1] != (10 - 6.5) / 13.0 || ov[2] != 0.0 || ov[3] != 0.0) {
    printf("KAT-C1 reset: obs mismatch\n");
    return 1;
  }
  const int32_t acts[3][2] = {{4, 1}, {3, 4}, {3, 3}};
  const double want_tot[3] = {-0.1, -0.6, 0.5}, want_loc[3][2] = {{-0.1, 0.0}, {-0.5, -0.1}, {1.0, -0.5}};
  const int32_t want_vec[3][8] = {{2, 10, 0, 0, 3, 10, 0, 0}, {2, 9, 0, 1, 3, 10, 0, 0}, {2, 8, 1, 1, 3, 9, 1, 0}};
  for (int t = 0; t < 3; ++t) {
    CHECK_HIP(hipMemcpy(b.actions, acts[t], sizeof(acts[t]), hipMemcpyHostToDevice));
    CHECK_CM3(cm3_checkers_step(&d, &b, NULL));
    CHECK_HIP(hipMemcpy(vec, b.vec, sizeof(vec), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(loc, b.local_rewards, sizeof(loc), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&tot, b.reward, sizeof(tot), hipMemcpyDeviceToHost));
    if (memcmp(vec, want_vec[t], sizeof(vec)) != 0 || loc[0] != want_loc[t][0] || loc[1] != want_loc[t][1] || tot != want_loc[t][0] + want_loc[t][1] ||
        fabs(tot - want_tot[t]) > 1e-15) {
      printf("KAT-C1 tick %d mismatch: vec %d %d %d %d | %d %d %d %d, local %g %g, total %g\n", t, vec[0], vec[1], vec[2], vec[3], vec[4],
             vec[5], vec[6], vec[7], loc[0], loc[1], tot);
      return 1;
    }
  }
  printf("KAT-C1 through the C ABI: exact\n");
  return 0;
}

int main() {
  const int E = 1, N = 2, L = 4;
  cm3_particle_desc d;
  memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.max_steps = 25; d.seed = 12341; d.prob_random = 0.0; d.initial_std = 0.0;
  d.agents_x[0] = -0.2; d.agents_x[1] = 0.2; d.landmarks_x[0] = 0.9; d.landmarks_x[1] = -0.9;   // all y = 0

  double *state, *goals, *obs, *rewn, *rew; int32_t *meta, *episode, *actions; uint8_t *done;
  CHECK_HIP(hipMalloc((void **)&state, sizeof(double) * N * E * 4)); CHECK_HIP(hipMalloc((void **)&goals, sizeof(double) * N * E * 2));
  CHECK_HIP(hipMalloc((void **)&obs, sizeof(double) * E * N * L)); CHECK_HIP(hipMalloc((void **)&rewn, sizeof(double) * E * N));
  CHECK_HIP(hipMalloc((void **)&rew, sizeof(double) * E)); CHECK_HIP(hipMalloc((void **)&meta, sizeof(int32_t) * E * 2));
  CHECK_HIP(hipMalloc((void **)&episode, sizeof(int32_t) * E)); CHECK_HIP(hipMalloc((void **)&actions, sizeof(int32_t) * E * N));
  CHECK_HIP(hipMalloc((void **)&done, E));
  CHECK_HIP(hipMemset(episode, 0, sizeof(int32_t) * E)); CHECK_HIP(hipMemset(meta, 0, sizeof(int32_t) * E * 2));

  cm3_particle_bufs b;
  memset(&b, 0, sizeof(b));
  b.state_in = b.state_out = state; b.goals_in = b.goals_out = goals; b.meta_in = b.meta_out = meta; b.episode = episode;
  b.actions = actions; b.obs_others = obs; b.reward_n = rewn; b.reward = rew; b.done = done;
  CHECK_CM3(cm3_particle_reset_f64(&d, &b, NULL, NULL));

  // per tick: actions, expected global_state rows (vx, vy, px, py) of both agents, reward, reward_n[0], collisions
  const int acts[4][2] = {{2, 1}, {2, 1}, {2, 1}, {0, 0}};
  const double want_v[4] = {0.5, 0.8680685281944008, -0.5850856602430008, -1.0047799810850504};
  const double want_p[4] = {-0.15000000000000002, -0.06319314718055993, -0.12170171320486001, -0.22217971131336506};
  const double want_r[4] = {-2.1, -3.92638629436112, -4.04340342640972, -2.2443594226267303};
  const double want_rn[4] = {-1.05, -1.96319314718056, -2.02170171320486, -1.1221797113133651};
  const int want_coll[4] = {0, 2, 4, 4};
  double worst = 0.0;
  for (int t = 0; t < 4; ++t) {
    CHECK_HIP(hipMemcpy(actions, acts[t], sizeof(int32_t) * 2, hipMemcpyHostToDevice));
    CHECK_CM3(cm3_particle_step_f64(&d, &b, NULL));
    double s[8], r, rn[2], o[8]; int32_t m[2]; uint8_t dn;
    CHECK_HIP(hipMemcpy(s, state, sizeof(s), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(&r, rew, sizeof(r), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(rn, rewn, sizeof(rn), hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(m, meta, sizeof(m), hipMemcpyDeviceToHost));
    CHECK_HIP(hipMemcpy(&dn, done, 1, hipMemcpyDeviceToHost)); CHECK_HIP(hipMemcpy(o, obs, sizeof(o), hipMemcpyDeviceToHost));
    // state layout [N][E][4]: agent 0 then agent 1; the two agents mirror each other
    const double err[] = {s[0] - want_v[t], s[1], s[2] - want_p[t], s[3], s[4] + want_v[t], s[5], s[6] + want_p[t], s[7],
                          r - want_r[t], rn[0] - want_rn[t], rn[1] - want_rn[t]};
    for (double e : err) worst = fmax(worst, fabs(e));
    if (m[0] != t + 1 || m[1] != want_coll[t] || dn != 0) { printf("tick %d: steps %d collisions %d done %d\n", t, m[0], m[1], dn); return 1; }
    if (t == 3) {  // obs_others[0] = (v_1 - v_0, p_1 - p_0)
      const double eo[] = {o[0] - 2.009559962170101, o[1], o[2] - 0.4443594226267301, o[3]};
      for (double e : eo) worst = fmax(worst, fabs(e));
    }
  }
  printf("KAT-P1 through the C ABI: max |error| = %.3e (abi %d)\n", worst, cm3_abi_version());
  if (!(worst < 1e-12)) return 1;
  return kat_c1();
}
