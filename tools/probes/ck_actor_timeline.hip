// Diagnostic: per-wave shader-clock timeline of k_ck_actor (Checkers actor) at E envs x 2 agents (build with -DCM3_STAMPS).
#ifdef CM3_STAMPS
__device__ long long *cm3_stamp_buf;
#endif
#include "../../cm3_amd/csrc/actor_checkers.hip"
#include "../../cm3_amd/csrc/util.hip"
#include <vector>
int main(int argc, char **argv) {
  const int E = argc > 1 ? atoi(argv[1]) : 8192, N = 2, Lo = 2, prec = argc > 2 ? atoi(argv[2]) : 0;   // prec: 0 f32, 1 bf16, 2 f16x3
  const int nwaves = argc > 3 ? atoi(argv[3]) : 8;   // waves per workgroup of the f16x3 kernel (4 | 8)
  if (prec == 2 && cm3_actor_checkers_force_waves(nwaves)) { printf("%s\n", cm3_last_error()); return 1; }
  int8_t *obst; double *obsv, *obso; uint8_t *goals; int32_t *steps, *episode, *actions; float *w; long long *stamps;
  hipMalloc((void **)&obst, (size_t)E * 152); hipMalloc((void **)&obsv, (size_t)E * N * 32); hipMalloc((void **)&obso, (size_t)E * N * Lo * 8);
  hipMalloc((void **)&goals, (size_t)E * N); hipMalloc((void **)&steps, (size_t)E * 4); hipMalloc((void **)&episode, (size_t)E * 4);
  hipMalloc((void **)&actions, (size_t)E * N * 4);
  hipMemset(obst, 1, (size_t)E * 152); hipMemset(obsv, 0, (size_t)E * N * 32); hipMemset(obso, 0, (size_t)E * N * Lo * 8);
  hipMemset(goals, 0, (size_t)E * N); hipMemset(steps, 0, (size_t)E * 4); hipMemset(episode, 0, (size_t)E * 4);
  const size_t nw = 162 + 6 + 150 * 32 + 32 + 43 * 256 + 256 + 65536 + Lo * 256 + 256 + 65536 + 256 + 1280 + 5;
  hipMalloc((void **)&w, nw * 4); hipMemset(w, 0, nw * 4);
  const int waves = ((E * N + 63) / 64) * (prec == 2 ? nwaves : 4);
  hipMalloc((void **)&stamps, (size_t)waves * 16 * 8 + 4096);
#ifdef CM3_STAMPS
  hipMemcpyToSymbol(HIP_SYMBOL(cm3_stamp_buf), &stamps, sizeof(stamps));
#endif
  cm3_actor_checkers_desc d; memset(&d, 0, sizeof(d));
  d.n_envs = E; d.n_agents = N; d.stage = 2; d.n_obs = 2; d.conv_f = 6; d.n_conv_linear = 32; d.n_h1 = 256; d.n_h2 = 256; d.n_actions = 5;
  d.epsilon = 0.1f; d.obs_self_t_stride = 152; d.precision = prec;
  cm3_actor_checkers_weights wt; float *q = w;
  wt.conv_w = q; q += 162; wt.conv_b = q; q += 6; wt.lin_w = q; q += 4800; wt.lin_b = q; q += 32; wt.self_w = q; q += 43 * 256; wt.self_b = q; q += 256;
  wt.w_self_h2 = q; q += 65536; wt.others_w = q; q += Lo * 256; wt.others_b = q; q += 256; wt.w_others_h2 = q; q += 65536; wt.b_h2 = q; q += 256;
  wt.out_w = q; q += 1280; wt.out_b = q;
  void *packed; hipMalloc(&packed, cm3_actor_checkers_packed_bytes()); wt.packed = packed;
  if (cm3_actor_checkers_pack(&d, &wt, packed, nullptr)) { printf("%s\n", cm3_last_error()); return 1; }
  hipDeviceSynchronize();
  cm3_actor_checkers_bufs b; memset(&b, 0, sizeof(b));
  b.obs_self_t = obst; b.obs_self_v = obsv; b.obs_others = obso; b.goals = goals; b.steps = steps; b.episode = episode; b.actions = actions;
  hipStream_t s; hipStreamCreate(&s);
  for (int t = 0; t < 10; ++t) if (cm3_actor_checkers_f32(&d, &wt, &b, s)) { printf("%s\n", cm3_last_error()); return 1; }
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0, s);
  for (int t = 0; t < 100; ++t) cm3_actor_checkers_f32(&d, &wt, &b, s);
  hipEventRecord(e1, s); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("E=%d precision=%d waves=%d: %.3f us per actor launch (back-to-back eager)\n", E, prec, nwaves, ms * 1e3 / 100);
#ifdef CM3_STAMPS
  std::vector<long long> h((size_t)waves * 16);
  hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost);
  const char *names[] = {"stage inputs (bytes -> floats)", "barrier", "conv (Toeplitz) + store", "conv_linear + store", "branch_self + store",
                         "h2 pass 1 (self)", "barrier", "branch_others + store", "h2 pass 2 (others)", "barrier", "h2 relu -> LDS + barrier",
                         "actor_out + softmax + sample"};
  for (int k = 0; k < 12; ++k) {
    double seg = 0;
    for (int wv = 0; wv < waves; ++wv) seg += (double)(h[wv * 16 + k + 1] - h[wv * 16 + k]);
    printf("   %-34s %9.0f cycles\n", names[k], seg / waves);
  }
  double tot = 0;
  for (int wv = 0; wv < waves; ++wv) tot += (double)(h[wv * 16 + 12] - h[wv * 16 + 0]);
  printf("   %-34s %9.0f cycles (shader clock = 100 MHz x ? -- compare ratios)\n", "total", tot / waves);
#endif
  return 0;
}
