// Does the size of the kernel-argument segment move the dependent-launch boundary?  330 launches per hipGraph, 256 workgroups x 256
// threads, each thread one load + one store; the kernel takes a by-value struct of 64 .. 4032 bytes and reads ONE dword of it
// (chosen at run time, so that the whole struct stays an argument).
//   hipcc --offload-arch=gfx950 -O3 -o kernarg_gap_probe kernarg_gap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
template <int DW> struct Args { unsigned v[DW]; };
template <int DW> __global__ void __launch_bounds__(256) k(Args<DW> a, unsigned *buf, int sel) {
  const unsigned i = blockIdx.x * 256 + threadIdx.x;
  buf[i] = buf[i] + a.v[sel & (DW - 1)];
}
template <int DW> static void run(unsigned *buf) {
  Args<DW> a;
  for (int q = 0; q < DW; ++q) a.v[q] = q;
  hipStream_t s; hipStreamCreate(&s);
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int t = 0; t < 330; ++t) hipLaunchKernelGGL((k<DW>), dim3(256), dim3(256), 0, s, a, buf, t);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int w = 0; w < 3; ++w) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, s);
    for (int r = 0; r < 10; ++r) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("kernel arguments %5d bytes (+16): %.3f us per launch\n", DW * 4, best * 1e3f / 3300.0f);
  hipGraphExecDestroy(ge); hipGraphDestroy(g); hipStreamDestroy(s);
}
int main() {
  unsigned *buf; hipMalloc((void **)&buf, 65536 * 4); hipMemset(buf, 0, 65536 * 4);
  for (int rep = 0; rep < 2; ++rep) { run<16>(buf); run<64>(buf); run<128>(buf); run<256>(buf); run<512>(buf); run<1008>(buf); }
  return 0;
}
