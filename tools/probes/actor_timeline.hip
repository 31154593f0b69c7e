// Diagnostic: per-wave shader-clock timeline of k_actor_particle<4> at 4096 envs (build with -DCM3_STAMPS).
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/actor.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <algorithm>
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4096, N = 4, L = 12;
  const int bf16 = argc > 2 ? atoi(argv[2]) : 0;
  float *obs, *state, *goals, *w; int32_t *meta, *episode, *actions; long long *stamps;
  hipMalloc((void **)&obs, (size_t)E * N * L * 4); hipMalloc((void **)&state, (size_t)N * E * 16); hipMalloc((void **)&goals, (size_t)N * E * 8);
  hipMalloc((void **)&meta, (size_t)E * 8); hipMalloc((void **)&episode, (size_t)E * 4); hipMalloc((void **)&actions, (size_t)E * N * 4);
  hipMemset(obs, 0, (size_t)E * N * L * 4); hipMemset(state, 0, (size_t)N * E * 16); hipMemset(goals, 0, (size_t)N * E * 8);
  hipMemset(meta, 0, (size_t)E * 8); hipMemset(episode, 0, (size_t)E * 4);
  const size_t nw = 6 * 64 + 64 + 64 * 64 + L * 128 + 128 + 128 * 64 + 64 + 64 * 5 + 5;
  hipMalloc((void **)&w, nw * 4); hipMemset(w, 0, nw * 4);
  const int waves = ((E * N + 63) / 64) * 4;
  hipMalloc((void **)&stamps, (size_t)waves * 16 * 8 + 4096);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_actor_particle_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.stage = 2; d.n_h1_self = 64; d.n_h1_others = 128; d.n_h2 = 64; d.n_actions = 5; d.epsilon = 0.1f; d.precision = bf16;
  cm3_actor_particle_weights wt; float *q = w;
  wt.w_self = q; q += 6 * 64; wt.b_self = q; q += 64; wt.w_self_h2 = q; q += 64 * 64; wt.w_others = q; q += L * 128; wt.b_others = q; q += 128;
  wt.w_others_h2 = q; q += 128 * 64; wt.b_h2 = q; q += 64; wt.w_out = q; q += 64 * 5; wt.b_out = q;
  void *packed; hipMalloc(&packed, cm3_actor_particle_packed_bytes(N)); wt.packed = packed;
  if (cm3_actor_particle_pack(&d, &wt, packed, nullptr)) { printf("%s\n", cm3_last_error()); return 1; }
  hipDeviceSynchronize();
  cm3_actor_particle_bufs b; memset(&b, 0, sizeof(b));
  b.obs_others = obs; b.state = state; b.goals = goals; b.meta = meta; b.episode = episode; b.actions = actions;
  hipStream_t s; hipStreamCreate(&s);
  for (int t = 0; t < 20; ++t) if (cm3_actor_particle_f32(&d, &wt, &b, s)) { printf("%s\n", cm3_last_error()); return 1; }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int t = 0; t < 200; ++t) cm3_actor_particle_f32(&d, &wt, &b, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("E=%d bf16=%d: %.3f us per actor launch (back-to-back eager)\n", E, bf16, ms * 1e3 / 200);
#ifdef CM3_STAMPS
  std::vector<long long> h((size_t)waves * 16);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"stage weights + input/B loads", "barrier 1", "phase A (first layers)", "barrier 2", "phase B (MFMA)", "h2 -> LDS + barrier 3", "head (all waves, 16 rows each)"};
  double seg[7] = {0}; int cnt[7] = {0};
  for (int wv = 0; wv < waves; ++wv) for (int k = 0; k < 7; ++k) {
    seg[k] += (double)(h[wv * 16 + k + 1] - h[wv * 16 + k]); cnt[k]++;
  }
  for (int k = 0; k < 7; ++k) printf("   %-32s %9.0f cycles\n", names[k], seg[k] / cnt[k]);
  // inside segment 0 (non-draining stamps 8, 9, 10), averaged separately over wave 0 of each workgroup and the other waves
  const char *sub[] = {"entry -> table copy issued+stored", "-> input rows staged (wave 0)", "-> W2 loads issued", "-> all loads landed (drain)"};
  const int order[] = {0, 8, 9, 10, 1};
  for (int which = 0; which < 2; ++which) {
    double a[4] = {0}; int c = 0;
    for (int wv = 0; wv < waves; ++wv) {
      if (((wv & 3) == 0) != (which == 0)) continue;
      for (int k = 0; k < 4; ++k) a[k] += (double)(h[wv * 16 + order[k + 1]] - h[wv * 16 + order[k]]);
      c++;
    }
    printf("   segment 0, %s:\n", which == 0 ? "wave 0 of each workgroup" : "waves 1-3");
    for (int k = 0; k < 4; ++k) printf("      %-38s %9.0f cycles\n", sub[k], a[k] / c);
  }
#endif
  return 0;
}
