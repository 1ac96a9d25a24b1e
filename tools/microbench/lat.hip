// Microbenchmarks for the per-column dependency chain of the column-sweep TRSV (gfx950).
// Each kernel runs ONE wave; cycles per iteration from s_memtime.  Build: hipcc -O3 --offload-arch=gfx950 lat.hip -o lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ double readlane64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
#define T0 unsigned long long t0 = __builtin_amdgcn_s_memtime(); unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#define T1 unsigned long long t1 = __builtin_amdgcn_s_memtime(); unsigned long long r1 = __builtin_amdgcn_s_memrealtime(); if (threadIdx.x == 0) { out[0] = (double)(t1 - t0); out[1] = (double)(r1 - r0); }

// A: dependent fma chain
__global__ void k_fma_chain(double* out, double* sink, int iters, double a) {
  double x = threadIdx.x * 1e-3;
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) x = fma(a, x, 1.0);
  }
  T1
  sink[threadIdx.x] = x;
}
// B: readlane -> fma chain (one accumulator row) : exactly the critical path of the sweep for R=1
__global__ void k_rl_fma(double* out, double* sink, int iters, const double* l) {
  double x = threadIdx.x * 1e-3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { double z = readlane64(x, u + 1); x = fma(-lv, z, x); }
  }
  T1
  sink[threadIdx.x] = x;
}
// C: readlane -> 4 fma (R=4 rows), first fma feeds next readlane
__global__ void k_rl_fma4(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { double z = readlane64(x0, u + 1); x0 = fma(-lv, z, x0); x1 = fma(-lv, z, x1); x2 = fma(-lv, z, x2); x3 = fma(-lv, z, x3); }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// D: independent readlanes (throughput): 16 readlane64 of a fixed register, summed by fma with independent accs
__global__ void k_rl_tput(double* out, double* sink, int iters, const double* l) {
  double x = threadIdx.x * 1e-3; double lv = l[threadIdx.x]; double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; u += 4) {
      a0 = fma(lv, readlane64(x, u), a0); a1 = fma(lv, readlane64(x, u + 1), a1);
      a2 = fma(lv, readlane64(x, u + 2), a2); a3 = fma(lv, readlane64(x, u + 3), a3);
    }
  }
  T1
  sink[threadIdx.x] = a0 + a1 + a2 + a3;
}
// E: independent fma throughput (8 accumulators)
__global__ void k_fma_tput(double* out, double* sink, int iters, double a) {
  double x[8]; for (int k = 0; k < 8; ++k) x[k] = threadIdx.x * 1e-3 + k;
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int k = 0; k < 8; ++k) x[k] = fma(a, x[k], 1.0);
  }
  T1
  double s = 0; for (int k = 0; k < 8; ++k) s += x[k];
  sink[threadIdx.x] = s;
}
// F: LDS broadcast chain: lane j writes, all read
__global__ void k_lds_bcast(double* out, double* sink, int iters, const double* l) {
  __shared__ double sh[64];
  double x = threadIdx.x * 1e-3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { sh[threadIdx.x] = x; __builtin_amdgcn_wave_barrier(); double z = sh[u + 1]; x = fma(-lv, z, x); }
  }
  T1
  sink[threadIdx.x] = x;
}
// G: ds_bpermute broadcast chain (__shfl)
__global__ void k_shfl_chain(double* out, double* sink, int iters, const double* l) {
  double x = threadIdx.x * 1e-3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { double z = __shfl(x, u + 1); x = fma(-lv, z, x); }
  }
  T1
  sink[threadIdx.x] = x;
}
// H: 32-bit float readlane chain for comparison
__global__ void k_rl_fma_f32(double* out, double* sink, int iters, const double* l) {
  float x = threadIdx.x * 1e-3f; float lv = (float)l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { float z = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), u + 1)); x = fmaf(-lv, z, x); }
  }
  T1
  sink[threadIdx.x] = x;
}

template <class F> void run(const char* name, F launch, int iters, int per_iter) {
  double *out, *sink; hipMalloc(&out, 16); hipMalloc(&sink, 64 * 8 * 4);
  launch(out, sink, 10); hipDeviceSynchronize();
  double best = 1e30, bestr = 0;
  for (int rep = 0; rep < 5; ++rep) { launch(out, sink, iters); hipDeviceSynchronize(); double h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost); if (h[0] < best) { best = h[0]; bestr = h[1]; } }
  printf("%-14s %8.2f cycles/op   (clock %.0f MHz)\n", name, best / ((double)iters * per_iter), best / bestr * 100.0);
  hipFree(out); hipFree(sink);
}
int main() {
  double* l; hipMalloc(&l, 64 * 8); std::vector<double> h(64, 1e-9); hipMemcpy(l, h.data(), 512, hipMemcpyHostToDevice);
  const int it = 20000;
  run("fma_chain", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_fma_chain, 1, 64, 0, 0, o, s, n, 0.999); }, it, 16);
  run("fma_tput8", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_fma_tput, 1, 64, 0, 0, o, s, n, 0.999); }, it, 16);
  run("rl_fma", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_rl_fma, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("rl_fma4", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_rl_fma4, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("rl_tput", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_rl_tput, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("lds_bcast", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_lds_bcast, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("shfl_chain", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_shfl_chain, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("rl_fma_f32", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_rl_fma_f32, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  return 0;
}
