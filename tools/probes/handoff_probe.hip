// Diagnostic (round 4): can the per-tick launches of a rollout be spread over Q hardware queues (tick t in queue t mod Q) with a wave
// of tick t + 1 waiting only for the SAME wave of tick t (the env -> wave map never changes) -- a per-wave progress word in memory
// instead of the queue's barrier bit?  This probe runs the bare protocol: every launch = 256 workgroups x 4 waves; a wave polls its
// progress word until the count of completed ticks is congruent to its launch's tick (mod Q), loads its 64 x 16-byte state with
// agent-scope loads, checks it against the value tick count -> state must have, spins ~`work` dependent VALU instructions, stores the
// next state write-through, waits for the acknowledgement, bumps the progress word, and then does `tail` more instructions of
// off-chain work.  Printed: errors (stale or torn state / timeouts) and us per tick for Q = 1 .. 4.
//   hipcc --offload-arch=gfx950 -O3 -o handoff_probe handoff_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint32_t u4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t ld_flag(const uint32_t *p) {
  uint32_t v;
  asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <bool SC1> __device__ __forceinline__ u4 ld_state(const u4 *p) {
  u4 v;
  if (SC1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}
template <bool SC1> __device__ __forceinline__ void st_state(u4 *p, u4 v) {
  if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(v) : "memory");
}

template <bool SC1>
__global__ void __launch_bounds__(256) k_tick(u4 *state, uint32_t *flags, uint32_t *errors, uint32_t *sink, int q, int Q, int work, int tail,
                                              int use_flags) {
  const int lane = threadIdx.x & 63, wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  uint32_t count = 0;
  if (use_flags && ld_flag(errors + 1) < 64u) {   // (after a few timeouts nothing waits any more: a broken protocol must not cost minutes)
    int polls = 0;
    for (;;) {
      count = ld_flag(flags + wave);
      if ((int)(count % (uint32_t)Q) == q) break;
      if (++polls > (1 << 16)) {          // never hang the box: give up, count it
        if (lane == 0) atomicAdd(errors + 1, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
  } else {
    count = ld_flag(flags + wave);
  }
  u4 *p = state + (size_t)wave * 64 + lane;
  u4 s = ld_state<SC1>(p);
  // the state after `count` ticks is (count, count * 3 + lane, wave, count ^ lane)
  const bool ok = s.x == count && s.y == count * 3u + (uint32_t)lane && s.z == (uint32_t)wave && s.w == (count ^ (uint32_t)lane);
  if (use_flags && !ok) atomicAdd(errors, 1u);
  uint32_t a = s.y;
  for (int k = 0; k < work; ++k) a = a * 1664525u + 1013904223u;   // dependent chain: ~2 instructions per step
  u4 n;
  n.x = count + 1u;
  n.y = (count + 1u) * 3u + (uint32_t)lane;
  n.z = (uint32_t)wave;
  n.w = (count + 1u) ^ (uint32_t)lane;
  if (a == 0x12345u) n.x += 1u;   // (keeps the chain alive)
  st_state<SC1>(p, n);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) asm volatile("global_store_dword %0, %1, off sc1" ::"v"(flags + wave), "v"(count + 1u) : "memory");
  uint32_t b = a ^ 77u;
  for (int k = 0; k < tail; ++k) b = b * 22695477u + 1u;
  sink[(size_t)wave * 64 + lane] = b;   // plain off-chain output
}

int main(int argc, char **argv) {
  const int T = 330, reps = 20, WG = 256, waves = WG * 4;
  const int work = argc > 1 ? atoi(argv[1]) : 150, tail = argc > 2 ? atoi(argv[2]) : 250;
  u4 *state; uint32_t *flags, *errors, *sink;
  hipMalloc((void **)&state, (size_t)waves * 64 * 16); hipMalloc((void **)&flags, waves * 4); hipMalloc((void **)&errors, 8);
  hipMalloc((void **)&sink, (size_t)waves * 64 * 4);
  for (int sc1 = 1; sc1 >= 0; --sc1)
    for (int Q = 1; Q <= 4; ++Q) {
      if (T % Q) continue;
      std::vector<uint32_t> h((size_t)waves * 64 * 4);
      for (int w = 0; w < waves; ++w) for (int l = 0; l < 64; ++l) { uint32_t *e = &h[((size_t)w * 64 + l) * 4]; e[0] = 0; e[1] = l; e[2] = w; e[3] = l; }
      hipMemcpy(state, h.data(), h.size() * 4, hipMemcpyHostToDevice);
      hipMemset(flags, 0, waves * 4); hipMemset(errors, 0, 8);
      std::vector<hipStream_t> st(Q); std::vector<hipGraphExec_t> ge(Q);
      for (int q = 0; q < Q; ++q) {
        hipStreamCreateWithFlags(&st[q], hipStreamNonBlocking);
        hipGraph_t g;
        hipStreamBeginCapture(st[q], hipStreamCaptureModeThreadLocal);
        for (int t = q; t < T; t += Q) {
          if (sc1) hipLaunchKernelGGL(k_tick<true>, dim3(WG), dim3(256), 0, st[q], state, flags, errors, sink, q, Q, work, tail, 1);
          else hipLaunchKernelGGL(k_tick<false>, dim3(WG), dim3(256), 0, st[q], state, flags, errors, sink, q, Q, work, tail, 1);
        }
        hipStreamEndCapture(st[q], &g);
        hipGraphInstantiate(&ge[q], g, nullptr, nullptr, 0);
        hipGraphDestroy(g);
      }
      auto run = [&]() { for (int q = 0; q < Q; ++q) hipGraphLaunch(ge[q], st[q]); };
      run(); run();
      hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      // wall clock over all queues: host timer around a full synchronisation
      struct timespec a, b; clock_gettime(CLOCK_MONOTONIC, &a);
      for (int r = 0; r < reps; ++r) run();
      hipDeviceSynchronize();
      clock_gettime(CLOCK_MONOTONIC, &b);
      const double us = ((b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3) / (reps * T);
      uint32_t err[2]; hipMemcpy(err, errors, 8, hipMemcpyDeviceToHost);
      std::vector<uint32_t> f(waves); hipMemcpy(f.data(), flags, waves * 4, hipMemcpyDeviceToHost);
      uint32_t bad = 0; for (int w = 0; w < waves; ++w) bad += f[w] != (uint32_t)((reps + 2) * T);
      printf("%s Q=%d work=%d tail=%d: %.3f us per tick   state errors %u, timeouts %u, waves with a wrong final count %u\n",
             sc1 ? "sc1  " : "plain", Q, work, tail, us, err[0], err[1], bad);
      for (int q = 0; q < Q; ++q) { hipGraphExecDestroy(ge[q]); hipStreamDestroy(st[q]); }
    }
  return 0;
}
