// Diagnostic: per-wave shader-clock timeline of the LAST tick of k_ck_policy_rollout<2> (csrc/policy_checkers.hip), E envs x 2 agents,
// T ticks (build with -DCM3_STAMPS).  usage: ck_policy_timeline [E = 8192] [T = 33]
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#define CM3_X3_CONV_STAMP 7
#include "../../cm3_amd/csrc/checkers.hip"
#include "../../cm3_amd/csrc/actor_checkers.hip"
#define CM3_POLICY_CHECKERS_BODY_ONLY 1
#include "../../cm3_amd/csrc/policy_checkers.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <vector>
#define CK(x) do { if ((x) != 0) { printf("%s:%d %s\n", __FILE__, __LINE__, cm3_last_error()); return 1; } } while (0)
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 8192, T = argc > 2 ? atoi(argv[2]) : 33, N = 2, Lo = 2;
  const int gstride = 56, ostride = 152;
  auto dmalloc = [](size_t n) { void *p; hipMalloc(&p, n); hipMemset(p, 0, n); return p; };
  const int waves = ((E * 2 + 63) / 64) * 8;
  long long *stamps = (long long *)dmalloc((size_t)waves * 16 * 8 + 4096);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_checkers_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.n_rows = 3; d.n_columns = 8; d.n_obs = 2; d.max_steps = 33; d.seed = 12341;
  d.grid_stride = gstride; d.obs_self_t_stride = ostride;
  d.agents_r[0] = 0; d.agents_r[1] = 2; d.agents_c[0] = 8; d.agents_c[1] = 8;
  cm3_checkers_traj t; memset(&t, 0, sizeof(t));
  t.mask = (uint64_t *)dmalloc((size_t)E * 8); t.agents = (uint32_t *)dmalloc((size_t)E * N * 4); t.steps = (int32_t *)dmalloc((size_t)E * 4);
  t.episode = (int32_t *)dmalloc((size_t)E * 4); t.goals = (uint8_t *)dmalloc((size_t)E * N);
  { std::vector<uint8_t> g((size_t)E * N); for (size_t k = 0; k < g.size(); ++k) g[k] = k & 1; hipMemcpy(t.goals, g.data(), g.size(), hipMemcpyHostToDevice); }
#define SLOT(field, stride_field, bytes, slots) t.stride_field = (bytes); t.field = (decltype(t.field))dmalloc((size_t)(bytes) * (slots))
  SLOT(actions, actions_stride, (size_t)E * N * 4, T);
  SLOT(grid, grid_slot_stride, (size_t)E * gstride, T + 1);
  SLOT(vec, vec_stride, (size_t)E * N * 16, T + 1);
  SLOT(obs_others, obs_others_stride, (size_t)E * N * Lo * 8, T + 1);
  SLOT(obs_self_t, obs_self_t_slot_stride, (size_t)E * ostride, T + 1);
  SLOT(obs_self_v, obs_self_v_stride, (size_t)E * N * 32, T + 1);
  SLOT(local_rewards, local_rewards_stride, (size_t)E * N * 8, T);
  SLOT(reward, reward_stride, (size_t)E * 8, T);
  SLOT(done, done_stride, (size_t)E, T);
  // reset through the library's own entry point (slot 0)
  cm3_checkers_bufs b; memset(&b, 0, sizeof(b));
  b.mask = t.mask; b.agents = t.agents; b.steps = t.steps; b.episode = t.episode; b.goals = t.goals;
  b.grid = t.grid; b.vec = t.vec; b.obs_others = t.obs_others; b.obs_self_t = t.obs_self_t; b.obs_self_v = t.obs_self_v;
  CK(cm3_checkers_reset(&d, &b, nullptr, nullptr));
  const size_t nw = 162 + 6 + 150 * 32 + 32 + 43 * 256 + 256 + 65536 + Lo * 256 + 256 + 65536 + 256 + 1280 + 5;
  std::vector<float> hw(nw);
  unsigned lcg = 12345u;
  for (auto &x : hw) { lcg = lcg * 1664525u + 1013904223u; x = ((int)(lcg >> 8) % 2001 - 1000) * 1e-4f; }
  float *w = (float *)dmalloc(nw * 4); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  cm3_actor_checkers_desc ad; memset(&ad, 0, sizeof(ad));
  ad.n_envs = E; ad.n_agents = N; ad.stage = 2; ad.n_obs = 2; ad.conv_f = 6; ad.n_conv_linear = 32; ad.n_h1 = 256; ad.n_h2 = 256; ad.n_actions = 5;
  ad.epsilon = 0.1f; ad.obs_self_t_stride = ostride; ad.precision = 2; ad.seed = d.seed;
  cm3_actor_checkers_weights wt; float *q = w;
  wt.conv_w = q; q += 162; wt.conv_b = q; q += 6; wt.lin_w = q; q += 4800; wt.lin_b = q; q += 32; wt.self_w = q; q += 43 * 256; wt.self_b = q; q += 256;
  wt.w_self_h2 = q; q += 65536; wt.others_w = q; q += Lo * 256; wt.others_b = q; q += 256; wt.w_others_h2 = q; q += 65536; wt.b_h2 = q; q += 256;
  wt.out_w = q; q += 1280; wt.out_b = q;
  void *packed = dmalloc(cm3_actor_checkers_packed_bytes()); wt.packed = packed;
  CK(cm3_actor_checkers_pack(&ad, &wt, packed, nullptr));
  hipDeviceSynchronize();
  hipStream_t s; hipStreamCreate(&s);
  for (int r = 0; r < 3; ++r) { CK(cm3_checkers_reset(&d, &b, nullptr, s)); CK(cm3_policy_rollout_checkers(&d, &t, &ad, &wt, nullptr, nullptr, nullptr, 0, nullptr, nullptr, T, s)); }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float tot = 0;
  const int reps = 10;
  for (int r = 0; r < reps; ++r) {
    CK(cm3_checkers_reset(&d, &b, nullptr, s));
    hipEventRecord(e0, s);
    CK(cm3_policy_rollout_checkers(&d, &t, &ad, &wt, nullptr, nullptr, nullptr, 0, nullptr, nullptr, T, s));
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); tot += ms;
  }
  printf("E=%d T=%d: %.3f us per tick (one launch per rollout)\n", E, T, tot * 1e3 / (reps * T));
#ifdef CM3_STAMPS
  std::vector<long long> h((size_t)waves * 16);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  struct Seg { int a, b; const char *name; bool env_only; };
  const Seg segs[] = {{0, 7, "conv (+ weight fetch) + store + barrier", false}, {7, 4, "conv_linear + store + barrier", false},
                      {4, 5, "table rows + branch_self + store + barrier", false}, {5, 6, "h2 pass (self)", false}, {6, 10, "barrier", false},
                      {10, 11, "h2 relu -> LDS + barrier", false}, {11, 13, "actor_out + softmax + sample + stores", false},
                      {13, 14, "barrier (actions)", false}, {14, 9, "plan fetch + agents act + reward stores", true},
                      {9, 12, "observation emit (global)", true}, {12, 15, "next inputs -> LDS (X0, tail, cell)", true}, {14, 8, "env phase incl. closing barrier", false}};
  for (const Seg &sg : segs) {
    double acc = 0; int n = 0;
    for (int wv = 0; wv < waves; ++wv) {

      acc += (double)(h[wv * 16 + sg.b] - h[wv * 16 + sg.a]); ++n;
    }
    printf("   %-52s %9.0f cycles%s\n", sg.name, acc / n, sg.env_only ? "" : "");
  }
  double tt = 0;
  for (int wv = 0; wv < waves; ++wv) tt += (double)(h[wv * 16 + 8] - h[wv * 16 + 0]);
  printf("   %-52s %9.0f cycles\n", "one tick", tt / waves);
#endif
  return 0;
}
