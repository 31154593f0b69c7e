// Diagnostic: per-wave shader-clock timeline of k_particle_step_pairs<float, 4, 4> at the C2 size (build with -DCM3_STAMPS).
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/particle.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4096, N = 4;
  void *state, *goals, *obs, *rewn, *rew;
  int32_t *meta, *episode, *actions; uint8_t *done; long long *stamps;
  hipMalloc(&state, (size_t)N * E * 16); hipMalloc(&goals, (size_t)N * E * 8); hipMalloc(&obs, (size_t)E * N * 48);
  hipMalloc(&rewn, (size_t)E * N * 4); hipMalloc(&rew, (size_t)E * 4); hipMalloc((void **)&meta, (size_t)E * 8);
  hipMalloc((void **)&episode, (size_t)E * 4); hipMalloc((void **)&actions, (size_t)E * N * 4); hipMalloc((void **)&done, E);
  const int nw = (E + 3) / 4;
  hipMalloc((void **)&stamps, (size_t)(nw + 8) * 16 * 8);
  hipMemset(episode, 0, (size_t)E * 4);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_particle_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.max_steps = 33; d.seed = 12341; d.prob_random = 0.2;
  const double ax[4] = {-0.9, 0.9, -0.9, 0.9}, ay[4] = {-0.9, 0.9, 0.9, -0.9}, lx[4] = {0.9, -0.9, 0.9, -0.9}, ly[4] = {0.9, -0.9, -0.9, 0.9};
  for (int i = 0; i < 4; ++i) { d.agents_x[i] = ax[i]; d.agents_y[i] = ay[i]; d.landmarks_x[i] = lx[i]; d.landmarks_y[i] = ly[i]; }
  cm3_particle_bufs b; memset(&b, 0, sizeof(b));
  b.state_in = b.state_out = state; b.goals_in = b.goals_out = goals; b.meta_in = b.meta_out = meta; b.episode = episode;
  b.actions = actions; b.obs_others = obs; b.reward_n = rewn; b.reward = rew; b.done = done;
  hipStream_t s; hipStreamCreate(&s);
  if (cm3_particle_reset_f32(&d, &b, nullptr, s)) { printf("reset: %s\n", cm3_last_error()); return 1; }
  d.flags = CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS;
  cm3::ParticleParams pp;
  cm3::fill_params(&d, &b, cm3::kStep, nullptr, pp);
  for (int t = 0; t < 50; ++t) cm3::launch_pairs<float, 4, 4>(pp, s);   // mid-episode (tick 50 mod 33 = 17)
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int t = 0; t < 330; ++t) cm3::launch_pairs<float, 4, 4>(pp, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("pairs<4> W=4 E=%d: %.3f us per launch (eager back to back; stamped build)\n", E, ms * 1e3 / 330);
#ifdef CM3_STAMPS
  // the last launch above was tick (50 + 330) mod 33 = 17 of an episode: no reset
  std::vector<long long> h((size_t)nw * 16);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"loads (drained)", "action draw (Philox) + store", "own contact force", "gather forces + integrate + post-step shuffle (drained)",
                         "reward / collisions / ballots / sums + small stores (drained)", "reset section", "state + obs stores (issue)", "tail (drain)"};
  double tot = 0;
  for (int k = 0; k < 8; ++k) {
    double seg = 0;
    for (int w = 0; w < nw; ++w) seg += (double)(h[w * 16 + k + 1] - h[w * 16 + k]);
    printf("   %-64s %8.0f cycles\n", names[k], seg / nw);
    tot += seg / nw;
  }
  printf("   %-64s %8.0f cycles\n", "total per wave", tot);
#endif
  return 0;
}
