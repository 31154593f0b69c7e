// Diagnostic: per-wave shader-clock timeline of one tick of k_policy_rollout<4, f16x3 | f32, RT> (the whole-episode policy-driven
// collection) at 4096 envs, and the time per tick of a 33-tick launch.  Build with -DCM3_STAMPS for the timeline.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math [-DCM3_STAMPS] -o policy_timeline policy_timeline.hip
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/policy.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4096, N = 4, L = 12, T = 33;
  const int TL = argc > 3 ? atoi(argv[3]) : T;   // ticks per launch (1 = the launch-per-tick mode of the fused kernel)
  const int prec = argc > 2 ? atoi(argv[2]) : 2;
  float *w; long long *stamps;
  const size_t nw = 6 * 64 + 64 + 64 * 64 + L * 128 + 128 + 128 * 64 + 64 + 64 * 5 + 5;
  std::vector<float> hw(nw);
  unsigned seed = 12345;
  for (auto &v : hw) { seed = seed * 1664525u + 1013904223u; v = ((seed >> 8) / 16777216.0f - 0.5f) * 0.3f; }
  hipMalloc((void **)&w, nw * 4); hipMemcpy(w, hw.data(), nw * 4, hipMemcpyHostToDevice);
  auto dalloc = [](size_t bytes) { void *p; hipMalloc(&p, bytes); hipMemset(p, 0, bytes); return p; };
  cm3_particle_traj t; memset(&t, 0, sizeof(t));
  t.state = dalloc((size_t)(T + 1) * N * E * 16); t.state_stride = (size_t)N * E * 16;
  t.goals = dalloc((size_t)(T + 1) * N * E * 8); t.goals_stride = (size_t)N * E * 8;
  t.obs_others = dalloc((size_t)(T + 1) * E * N * L * 4); t.obs_others_stride = (size_t)E * N * L * 4;
  t.actions = (int32_t *)dalloc((size_t)T * E * N * 4); t.actions_stride = (size_t)E * N * 4;
  t.reward_n = dalloc((size_t)T * E * N * 4); t.reward_n_stride = (size_t)E * N * 4;
  t.reward = dalloc((size_t)T * E * 4); t.reward_stride = (size_t)E * 4;
  t.done = (uint8_t *)dalloc((size_t)T * E); t.done_stride = (size_t)E;
  t.meta = (int32_t *)dalloc((size_t)E * 8); t.episode = (int32_t *)dalloc((size_t)E * 4);
  t.term_state = dalloc((size_t)T * N * E * 16); t.term_state_stride = (size_t)N * E * 16;
  t.term_obs_others = dalloc((size_t)T * E * N * L * 4); t.term_obs_others_stride = (size_t)E * N * L * 4;
  t.collisions = (int32_t *)dalloc((size_t)T * E * 4); t.collisions_stride = (size_t)E * 4;
  const int max_wg = (E * N + 15) / 16;
  hipMalloc((void **)&stamps, (size_t)max_wg * 4 * 16 * 8 + 4096);
  hipMemset(stamps, 0, (size_t)max_wg * 4 * 16 * 8);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_particle_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.max_steps = 33; d.flags = CM3_FLAG_AUTO_RESET; d.seed = 7; d.prob_random = 0.2; d.initial_std = 0.0;
  const double ax[4] = {0.7, -0.7, 0.0, 0.0}, ay[4] = {0.0, 0.0, 0.7, -0.7};
  for (int i = 0; i < 4; ++i) { d.agents_x[i] = ax[i]; d.agents_y[i] = ay[i]; d.landmarks_x[i] = -ax[i]; d.landmarks_y[i] = -ay[i]; }
  // a reset through the C ABI puts a real episode start into slot 0
  cm3_particle_bufs rb; memset(&rb, 0, sizeof(rb));
  rb.state_in = rb.state_out = t.state; rb.goals_in = rb.goals_out = t.goals; rb.meta_in = rb.meta_out = t.meta; rb.episode = t.episode;
  rb.obs_others = t.obs_others; rb.actions = t.actions; rb.reward_n = t.reward_n; rb.reward = t.reward; rb.done = t.done;
  if (cm3::particle_call<float>(&d, &rb, cm3::kReset, nullptr, nullptr)) { printf("%s\n", cm3_last_error()); return 1; }
  cm3_actor_particle_desc ad; memset(&ad, 0, sizeof(ad));
  ad.n_envs = E; ad.n_agents = N; ad.stage = 2; ad.n_h1_self = 64; ad.n_h1_others = 128; ad.n_h2 = 64; ad.n_actions = 5; ad.epsilon = 0.1f;
  ad.precision = prec; ad.seed = d.seed;
  cm3_actor_particle_weights wt; float *q = w;
  wt.w_self = q; q += 6 * 64; wt.b_self = q; q += 64; wt.w_self_h2 = q; q += 64 * 64; wt.w_others = q; q += L * 128; wt.b_others = q; q += 128;
  wt.w_others_h2 = q; q += 128 * 64; wt.b_h2 = q; q += 64; wt.w_out = q; q += 64 * 5; wt.b_out = q;
  void *packed; hipMalloc(&packed, cm3::packed_floats<4>() * sizeof(float)); wt.packed = packed;
  {  // (policy.hip compiles actor.hip without its entry points: the body of cm3_actor_particle_pack)
    cm3::ActorParams ap; memset(&ap, 0, sizeof(ap));
    ap.stage = ad.stage;
    ap.w_self = (const float *)wt.w_self; ap.b_self = (const float *)wt.b_self; ap.w_self_h2 = (const float *)wt.w_self_h2;
    ap.w_oth = (const float *)wt.w_others; ap.b_oth = (const float *)wt.b_others; ap.w_oth_h2 = (const float *)wt.w_others_h2;
    ap.b_h2 = (const float *)wt.b_h2; ap.w_out = (const float *)wt.w_out; ap.b_out = (const float *)wt.b_out;
    if (cm3::pack_launch<4>(ap, (float *)packed, nullptr)) { printf("%s\n", cm3_last_error()); return 1; }
  }
  hipDeviceSynchronize();
  hipStream_t s; hipStreamCreate(&s);
  for (int k = 0; k < 3; ++k) if (cm3_policy_rollout_f32(&d, &t, &ad, &wt, nullptr, 0, TL, s)) { printf("%s\n", cm3_last_error()); return 1; }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  const int reps = 20;
  for (int k = 0; k < reps; ++k) cm3_policy_rollout_f32(&d, &t, &ad, &wt, nullptr, 0, TL, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("E=%d prec=%d %s: %.3f us per tick (%d-tick launches back to back)\n", E, prec, cm3_last_kernel_variant(), ms * 1e3 / (reps * TL), TL);
#ifdef CM3_STAMPS
  const char *v = cm3_last_kernel_variant();
  const char *g = strstr(v, "g="); const int rt = g ? atoi(g + 2) : 4;
  const int wgs = (E * N + 16 * rt - 1) / (16 * rt), waves = wgs * 4;
  std::vector<long long> h((size_t)waves * 16);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  // stamps of the LAST tick: 0 tick start | 3 phase A done | 4 barrier | 5 phase B done | 6 h2 stored + barrier | 7 head: action picked
  // | 8 physics integrated + exchanged | 9 rewards / done stored | 10 reset handled | 11 trajectory + tile stores | 12 end barrier
  const int order[] = {0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12};
  const char *names[] = {"phase A (first layers + split)", "barrier", "phase B (second layer)", "h2 -> LDS + barrier", "head: logits, softmax, pick",
                         "physics: forces, integrate, exchange", "rewards / done / stores", "reset (if any)", "trajectory + tile stores", "end barrier"};
  {  // launch prologue / epilogue: 13 kernel entry | 14 staged | 0 first... (slot 0 holds the LAST tick's start: prologue end only when TL == 1) | 15 out
    double pro = 0, stage = 0, epi = 0, life = 0; int c = 0;
    for (int wv = 0; wv < waves; ++wv) {
      const long long *q = &h[(size_t)wv * 16];
      if (!q[13] || !q[14] || !q[15] || !q[12]) continue;
      stage += (double)(q[14] - q[13]); if (TL == 1) pro += (double)(q[0] - q[13]); epi += (double)(q[15] - q[12]); life += (double)(q[15] - q[13]); c++;
    }
    if (c) printf("  launch: entry -> staged %.0f | entry -> first tick %.0f (TL == 1 only) | last tick end -> out %.0f | wave life %.0f cycles (%d waves)\n", stage / c, pro / c, epi / c, life / c, c);
  }
  for (int which = 0; which < 2; ++which) {
    double seg[10] = {0}; int c = 0;
    for (int wv = 0; wv < waves; ++wv) {
      const bool rowwave = (wv & 3) < rt;
      if (rowwave != (which == 0)) continue;
      bool ok = true;
      for (int k = 0; k < 11; ++k) if (!rowwave && order[k] >= 7 && order[k] <= 11) continue; else if (h[wv * 16 + order[k]] == 0) ok = false;
      if (!ok) continue;
      if (rowwave) for (int k = 0; k < 10; ++k) seg[k] += (double)(h[wv * 16 + order[k + 1]] - h[wv * 16 + order[k]]);
      else { for (int k = 0; k < 4; ++k) seg[k] += (double)(h[wv * 16 + order[k + 1]] - h[wv * 16 + order[k]]); seg[9] += (double)(h[wv * 16 + 12] - h[wv * 16 + 6]); }
      c++;
    }
    if (!c) continue;
    printf("  %s (%d waves), cycles of the last tick:\n", which == 0 ? "waves with rows" : "waves without rows (w >= RT)", c);
    double tot = 0;
    for (int k = 0; k < 10; ++k) { printf("     %-40s %8.0f\n", names[k], seg[k] / c); tot += seg[k] / c; }
    printf("     %-40s %8.0f\n", "tick", tot);
  }
#endif
  return 0;
}
