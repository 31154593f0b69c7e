// Diagnostic: per-wave timeline (shader-clock stamps) of the lane-per-env particle step kernel at the C2
// size, launched back to back like the bench does.  Build: hipcc -DCM3_STAMPS ... ; run on the GPU box.
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/particle.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <algorithm>
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 4096, N = 4;
  void *state, *goals, *obs, *rewn, *rew, *term = nullptr;
  int32_t *meta, *episode, *actions; uint8_t *done; long long *stamps;
  hipMalloc(&state, (size_t)N * E * 16); hipMalloc(&goals, (size_t)N * E * 8); hipMalloc(&obs, (size_t)E * N * 48);
  hipMalloc(&rewn, (size_t)E * N * 4); hipMalloc(&rew, (size_t)E * 4); hipMalloc((void **)&meta, (size_t)E * 8);
  hipMalloc((void **)&episode, (size_t)E * 4); hipMalloc((void **)&actions, (size_t)E * N * 4); hipMalloc((void **)&done, E);
  const int waves = (E + 63) / 64;
  hipMalloc((void **)&stamps, (size_t)(E + 3) / 4 * 16 * 8 + 4096);
  hipMemset(episode, 0, (size_t)E * 4);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_particle_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.max_steps = 33; d.seed = 12341; d.prob_random = argc > 2 ? atof(argv[2]) : 0.2;
  double ax[4] = {-0.9, 0.9, -0.9, 0.9}, ay[4] = {-0.9, 0.9, 0.9, -0.9}, lx[4] = {0.9, -0.9, 0.9, -0.9}, ly[4] = {0.9, -0.9, -0.9, 0.9};
  for (int i = 0; i < 4; ++i) { d.agents_x[i] = ax[i]; d.agents_y[i] = ay[i]; d.landmarks_x[i] = lx[i]; d.landmarks_y[i] = ly[i]; }
  cm3_particle_bufs b; memset(&b, 0, sizeof(b));
  b.state_in = b.state_out = state; b.goals_in = b.goals_out = goals; b.meta_in = b.meta_out = meta; b.episode = episode;
  b.actions = actions; b.obs_others = obs; b.reward_n = rewn; b.reward = rew; b.done = done;
  hipStream_t s; hipStreamCreate(&s);
  d.flags = 0;
  if (cm3_particle_reset_f32(&d, &b, nullptr, s)) { printf("reset: %s\n", cm3_last_error()); return 1; }
  d.flags = CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS;
  cm3::ParticleParams pp;
  cm3::fill_params(&d, &b, cm3::kStep, nullptr, pp);
  const char *variant[] = {"lane-per-env W=1", "pairs W=1", "pairs W=2", "pairs W=4"};
  for (int v = 0; v < 4; ++v) {
    auto launch = [&]() {
      switch (v) {
        case 0: cm3::launch_one<float, 4, 1>(pp, cm3::kStep, s); break;
        case 1: cm3::launch_pairs<float, 4, 1>(pp, s); break;
        case 2: cm3::launch_pairs<float, 4, 2>(pp, s); break;
        case 3: cm3::launch_pairs<float, 4, 4>(pp, s); break;
      }
    };
    const int nw = v == 0 ? waves : (E + 3) / 4;
    for (int t = 0; t < 40; ++t) launch();
    hipStreamSynchronize(s);
    // graph of 33 launches, replayed: average launch time like the bench measures it
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int t = 0; t < 33; ++t) launch();
    hipStreamEndCapture(s, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < 30; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h((size_t)nw * 16);
    launch(); hipStreamSynchronize(s);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = h[0], tmax = 0; double seg[6] = {0, 0, 0, 0, 0, 0};
    for (int w = 0; w < nw; ++w) {
      tmin = std::min(tmin, h[w * 16]); tmax = std::max(tmax, h[w * 16 + 6]);
      for (int k = 0; k < 6; ++k) seg[k] += (double)(h[w * 16 + k + 1] - h[w * 16 + k]) / nw;
    }
    printf("%-18s E=%d: %.3f us/launch (graph); waves=%d first-entry->last-exit %.2f us; per-wave segments (cycles):",
           variant[v], E, ms * 1e3 / (33 * 30), nw, (double)(tmax - tmin) / 100.0 * 1.0);
    for (int k = 0; k < 6; ++k) printf(" %.0f", seg[k]);
    printf("\n");
  }
  return 0;
}
