// Floor of one kernel node in a replayed hipGraph (the harness bench.py uses): empty kernel, a kernel that only touches
// its arguments, and a kernel that loads 2 KiB per workgroup and stores 8 bytes -- all with 256 workgroups of 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k_empty() {}
__global__ void k_touch(const double* x, double* y) { if (threadIdx.x == 0) y[blockIdx.x] = x[blockIdx.x]; }
__global__ void k_stage(const double* x, double* y) {
  __shared__ double s[256];
  s[threadIdx.x] = x[blockIdx.x * 256 + threadIdx.x];
  __syncthreads();
  if (threadIdx.x == 0) { double a = 0; for (int i = 0; i < 256; ++i) a += s[i]; y[blockIdx.x] = a; }
}
template <class F> double run(F launch) {
  hipStream_t st; (void)hipStreamCreate(&st);
  hipGraph_t g; hipGraphExec_t ge;
  (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
  for (int i = 0; i < 100; ++i) launch(st);
  (void)hipStreamEndCapture(st, &g);
  (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipStreamSynchronize(st);
  (void)hipEventRecord(a, st);
  for (int i = 0; i < 20; ++i) (void)hipGraphLaunch(ge, st);
  (void)hipEventRecord(b, st);
  (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b);
  return 1e3 * ms / 2000.0;
}
int main() {
  double *x, *y; (void)hipMalloc(&x, 256 * 256 * 8); (void)hipMalloc(&y, 256 * 8); (void)hipMemset(x, 0, 256 * 256 * 8);
  printf("empty kernel      %.2f us per node\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_empty, 256, 256, 0, s); }));
  printf("touch arguments   %.2f us per node\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_touch, 256, 256, 0, s, x, y); }));
  printf("stage + barrier   %.2f us per node\n", run([&](hipStream_t s) { hipLaunchKernelGGL(k_stage, 256, 256, 0, s, x, y); }));
  return 0;
}
