// Column-step orderings for the sweep's critical path (R = 4 row blocks): cycles per column, one wave.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ double readlane64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
#define SB __builtin_amdgcn_sched_barrier(0)
#define T0 unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define T1 unsigned long long t1 = __builtin_amdgcn_s_memtime(); if (threadIdx.x == 0) out[0] = (double)(t1 - t0);

// V1: rl, rl, fma x4 (what the compiler emits today)
__global__ void k_v1(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { double z = readlane64(x0, u + 1); SB; x0 = fma(-lv, z, x0); x1 = fma(-lv, z, x1); x2 = fma(-lv, z, x2); x3 = fma(-lv, z, x3); SB; }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V2: software-pipelined: fma(k0), fma(k1), rl(next), fma(k2), fma(k3)
__global__ void k_v2(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  double z = readlane64(x0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = fma(-lv, z, x0); SB; x1 = fma(-lv, z, x1); SB;
      double zn = readlane64(x0, u + 1); SB;
      x2 = fma(-lv, z, x2); SB; x3 = fma(-lv, z, x3); SB;
      z = zn;
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V2r: as V2 but the lane select is a run-time SGPR (base | u) and every FMA reads its own factor register
__global__ void __launch_bounds__(64) k_v2r(double* out, double* sink, int iters, const double* l, int base) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  double lv[16][4];
  for (int u = 0; u < 16; ++u) for (int k = 0; k < 4; ++k) lv[u][k] = l[(threadIdx.x + u * 4 + k) & 63];
  T0
  double z = readlane64(x0, base);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = fma(-lv[u][0], z, x0); SB; x1 = fma(-lv[u][1], z, x1); SB;
      double zn = readlane64(x0, base | ((u + 1) & 15)); SB;
      x2 = fma(-lv[u][2], z, x2); SB; x3 = fma(-lv[u][3], z, x3); SB;
      z = zn;
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V2s: run-time lane select, same factor register
__global__ void k_v2s(double* out, double* sink, int iters, const double* l, int base) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  double z = readlane64(x0, base);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = fma(-lv, z, x0); SB; x1 = fma(-lv, z, x1); SB;
      double zn = readlane64(x0, base | ((u + 1) & 15)); SB;
      x2 = fma(-lv, z, x2); SB; x3 = fma(-lv, z, x3); SB;
      z = zn;
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V2d: compile-time lane select, distinct factor registers
__global__ void __launch_bounds__(64) k_v2d(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
  double lv[16][4];
  for (int u = 0; u < 16; ++u) for (int k = 0; k < 4; ++k) lv[u][k] = l[(threadIdx.x + u * 4 + k) & 63];
  T0
  double z = readlane64(x0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = fma(-lv[u][0], z, x0); SB; x1 = fma(-lv[u][1], z, x1); SB;
      double zn = readlane64(x0, (u + 1) & 15); SB;
      x2 = fma(-lv[u][2], z, x2); SB; x3 = fma(-lv[u][3], z, x3); SB;
      z = zn;
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V3: fma(k0), rl(next) immediately, then fma(k1..k3)
__global__ void k_v3(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  double z = readlane64(x0, 0);
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      x0 = fma(-lv, z, x0); SB;
      double zn = readlane64(x0, u + 1); SB;
      x1 = fma(-lv, z, x1); SB; x2 = fma(-lv, z, x2); SB; x3 = fma(-lv, z, x3); SB;
      z = zn;
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V4: pure readlane throughput: 16 readlane64 (32 v_readlane) of a fixed register, scalar-summed
__global__ void k_rl_only(double* out, double* sink, int iters, const double* l) {
  int x = threadIdx.x * 3 + 1; int acc = 0;
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 32; ++u) acc += __builtin_amdgcn_readlane(x, u);
  }
  T1
  sink[threadIdx.x] = acc;
}
// V5: fma x4 only (no broadcast), R=4 independent accumulators with a uniform SGPR multiplier
__global__ void k_fma4(double* out, double* sink, int iters, const double* l, double zs) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) { x0 = fma(-lv, zs, x0); x1 = fma(-lv, zs, x1); x2 = fma(-lv, zs, x2); x3 = fma(-lv, zs, x3); }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
// V6: v_mov_b64 DPP row_newbcast + fma x4 with VGPR z (16-lane rows)
__global__ void k_dpp(double* out, double* sink, int iters, const double* l) {
  double x0 = threadIdx.x * 1e-3, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3; double lv = l[threadIdx.x];
  T0
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      double z;
      asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(z) : "v"(x0), "n"(u & 15));
      x0 = fma(-lv, z, x0); x1 = fma(-lv, z, x1); x2 = fma(-lv, z, x2); x3 = fma(-lv, z, x3);
    }
  }
  T1
  sink[threadIdx.x] = x0 + x1 + x2 + x3;
}
template <class F> void run(const char* name, F launch, int iters, int per_iter) {
  double *out, *sink; (void)hipMalloc(&out, 16); (void)hipMalloc(&sink, 64 * 8 * 4);
  launch(out, sink, 10); (void)hipDeviceSynchronize();
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) { launch(out, sink, iters); (void)hipDeviceSynchronize(); double h[2]; (void)hipMemcpy(h, out, 8, hipMemcpyDeviceToHost); if (h[0] < best) best = h[0]; }
  printf("%-10s %8.2f cycles/column\n", name, best / ((double)iters * per_iter));
  (void)hipFree(out); (void)hipFree(sink);
}
int main() {
  double* l; (void)hipMalloc(&l, 64 * 8); std::vector<double> h(64, 1e-9); (void)hipMemcpy(l, h.data(), 512, hipMemcpyHostToDevice);
  const int it = 20000;
  run("v1_naive", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v1, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("v2_pipe", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v2, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("v2_rt_sel", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v2r, 1, 64, 0, 0, o, s, n, l, 16); }, it, 16);
  run("v2_rt_same", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v2s, 1, 64, 0, 0, o, s, n, l, 16); }, it, 16);
  run("v2_ct_dist", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v2d, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("v3_pipe", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_v3, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("rl_only/2", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_rl_only, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  run("fma4_only", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_fma4, 1, 64, 0, 0, o, s, n, l, 0.5); }, it, 16);
  run("dpp_fma4", [&](double* o, double* s, int n) { hipLaunchKernelGGL(k_dpp, 1, 64, 0, 0, o, s, n, l); }, it, 16);
  return 0;
}
