// Diagnostic: per-wave shader-clock timeline of k_checkers_step_fast<2> (8 lanes per env, in place) at the C3 size -- build with
// -DCM3_STAMPS.  Stamps inflate the kernel (read proportions, see README.md).
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/checkers.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 8192, N = 2;
  cm3_checkers_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.n_rows = 3; d.n_columns = 8; d.n_obs = 2; d.max_steps = 33; d.seed = 12341;
  d.grid_stride = 56; d.obs_self_t_stride = 152; d.agents_r[0] = 0; d.agents_r[1] = 2; d.agents_c[0] = 8; d.agents_c[1] = 8;
  cm3_checkers_bufs b; memset(&b, 0, sizeof(b));
  auto dev = [](size_t bytes) { void *p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
  b.mask = (uint64_t *)dev((size_t)E * 8); b.agents = (uint32_t *)dev((size_t)E * N * 4); b.steps = (int32_t *)dev((size_t)E * 4);
  b.episode = (int32_t *)dev((size_t)E * 4); b.goals = (uint8_t *)dev((size_t)E * N); b.actions = (int32_t *)dev((size_t)E * N * 4);
  b.grid = (int8_t *)dev((size_t)E * 56); b.vec = (int32_t *)dev((size_t)E * N * 16); b.obs_others = (double *)dev((size_t)E * N * 16);
  b.obs_self_t = (int8_t *)dev((size_t)E * 152); b.obs_self_v = (double *)dev((size_t)E * N * 32);
  b.local_rewards = (double *)dev((size_t)E * N * 8); b.reward = (double *)dev((size_t)E * 8); b.done = (uint8_t *)dev(E);
  std::vector<uint8_t> g((size_t)E * N);
  for (int e = 0; e < E; ++e) { g[2 * e] = 0; g[2 * e + 1] = 1; }
  (void)hipMemcpy(b.goals, g.data(), g.size(), hipMemcpyHostToDevice);
  const int waves = (E + 7) / 8;
  long long *stamps = (long long *)dev((size_t)(waves + 8) * 16 * 8);
#ifdef CM3_STAMPS
  (void)hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  hipStream_t s; (void)hipStreamCreate(&s);
  if (cm3_checkers_reset(&d, &b, nullptr, s)) { printf("reset: %s\n", cm3_last_error()); return 1; }
  d.flags = CM3_FLAG_AUTO_RESET | CM3_FLAG_GEN_ACTIONS;
  for (int t = 0; t < 50; ++t) if (cm3_checkers_step(&d, &b, s)) { printf("step: %s\n", cm3_last_error()); return 1; }
  (void)hipStreamSynchronize(s);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, s);
  for (int t = 0; t < 330; ++t) cm3_checkers_step(&d, &b, s);
  (void)hipEventRecord(e1, s); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  printf("checkers fast<2>, 8 lanes per env, E=%d: %.3f us per launch (eager back to back%s)\n", E, ms * 1e3 / 330,
#ifdef CM3_STAMPS
         "; stamped build"
#else
         ""
#endif
  );
#ifdef CM3_STAMPS
  std::vector<long long> h((size_t)waves * 16);
  (void)hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"loads (drained)", "action draw (Philox) + store", "agents act + rewards", "done + per-env stores",
                         "emit: grid", "emit: obs_self_t (4 cells per lane)", "emit: vec", "emit: normalised doubles (1 division per lane)",
                         "state stores + tail (drained)"};
  for (int k = 0; k < 9; ++k) {
    double seg = 0;
    for (int w = 0; w < waves; ++w) seg += (double)(h[w * 16 + k + 1] - h[w * 16 + k]);
    printf("   %-52s %8.0f cycles\n", names[k], seg / waves);
  }
#endif
  return 0;
}
