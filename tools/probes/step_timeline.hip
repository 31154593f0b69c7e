// Diagnostic: per-wave timeline (shader-clock stamps) of the lane-per-env particle step kernel at the C2
// size, launched back to back like the bench does.  Build: hipcc -DCM3_STAMPS ... ; run on the GPU box.
#define CM3_STAMPS 1
__device__ long long *cm3_stamp_buf;
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
  hipMalloc((void **)&stamps, (size_t)waves * 16 * 8);
  hipMemset(episode, 0, (size_t)E * 4);
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
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
  d.flags = CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS | CM3_FLAG_KERNEL_LANE_PER_ENV;
  for (int t = 0; t < 40; ++t) cm3_particle_step_f32(&d, &b, s);
  hipStreamSynchronize(s);
  std::vector<long long> h((size_t)waves * 16);
  const char *names[] = {"entry->loads done", "contact forces+integrate", "reward/collisions", "small stores issue+reset", "obs_others LDS staging", "final drain"};
  for (int t = 0; t < 3; ++t) {
    cm3_particle_step_f32(&d, &b, s);
    hipStreamSynchronize(s);
    hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
    long long tmin = h[0], tmax = 0;
    double seg[6] = {0, 0, 0, 0, 0, 0};
    double sub[3] = {0, 0, 0};
    for (int w = 0; w < waves; ++w) {
      tmin = std::min(tmin, h[w * 16]); tmax = std::max(tmax, h[w * 16 + 6]);
      for (int k = 0; k < 6; ++k) seg[k] += (double)(h[w * 16 + k + 1] - h[w * 16 + k]) / waves;
      sub[0] += (double)(h[w * 16 + 8] - h[w * 16 + 3]) / waves;
      sub[1] += (double)(h[w * 16 + 9] - h[w * 16 + 8]) / waves;
      sub[2] += (double)(h[w * 16 + 4] - h[w * 16 + 9]) / waves;
    }
    printf("tick %d: waves=%d first-entry -> last-exit = %lld cycles; mean per-wave segments (cycles):\n", t, waves, tmax - tmin);
    for (int k = 0; k < 6; ++k) printf("   %-28s %9.0f\n", names[k], seg[k]);
    printf("      [3->4 split] reward stores %.0f | reset branch %.0f | state/goal/meta stores %.0f\n", sub[0], sub[1], sub[2]);
    printf("      wall: %.2f us at 2.4 GHz\n", (double)(tmax - tmin) / 2400.0);
  }
  return 0;
}
