// Latency of the FIRST global load of a kernel (L2 invalidated at the kernel boundary, TLBs cold?) vs a second,
// dependent-address load to a different line, vs a third to a line in the same 2-MiB page.  256 WGs x 256 threads.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(const double* __restrict__ a, const double* __restrict__ b, double* out, double* sink) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  double v0 = a[threadIdx.x];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double v1 = a[65536 + threadIdx.x + (int)(v0 * 1e-300)];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t2 = __builtin_amdgcn_s_memtime();
  double v2 = b[threadIdx.x + blockIdx.x * 256 + (int)(v1 * 1e-300)];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned long long t3 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 3] = (double)(t1 - t0); out[blockIdx.x * 3 + 1] = (double)(t2 - t1); out[blockIdx.x * 3 + 2] = (double)(t3 - t2); }
  sink[blockIdx.x * 256 + threadIdx.x] = v0 + v1 + v2;
}
int main() {
  double *a, *b, *out, *sink;
  (void)hipMalloc(&a, 8 << 20); (void)hipMalloc(&b, 8 << 20); (void)hipMalloc(&out, 256 * 24); (void)hipMalloc(&sink, 256 * 256 * 8);
  (void)hipMemset(a, 0, 8 << 20); (void)hipMemset(b, 0, 8 << 20);
  for (int rep = 0; rep < 6; ++rep) {
    hipLaunchKernelGGL(k, 256, 256, 0, 0, a, b, out, sink); (void)hipDeviceSynchronize();
    std::vector<double> h(768); (void)hipMemcpy(h.data(), out, 768 * 8, hipMemcpyDeviceToHost);
    double s[3] = {0, 0, 0}, mx[3] = {0, 0, 0};
    for (int g = 0; g < 256; ++g) for (int i = 0; i < 3; ++i) { s[i] += h[g * 3 + i]; mx[i] = h[g * 3 + i] > mx[i] ? h[g * 3 + i] : mx[i]; }
    printf("rep %d: first load %6.0f (max %6.0f)  second (other line, same table) %6.0f (max %6.0f)  third (other buffer, per-WG line) %6.0f (max %6.0f) cycles\n",
           rep, s[0] / 256, mx[0], s[1] / 256, mx[1], s[2] / 256, mx[2]);
  }
  return 0;
}
