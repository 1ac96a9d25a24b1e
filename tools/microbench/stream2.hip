// Workgroup-level L2 -> CU ingest: W waves per workgroup split a table of `units` 1-KiB units (unit u -> wave u % W),
// D loads in flight per wave, every workgroup reads the SAME table (like the factor stream).  Reports cycles per unit
// per workgroup for the first pass after kernel start (L2 invalidated -> Infinity Cache) and for a second pass (L2 warm).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double d2 __attribute__((ext_vector_type(2)));

template <int D>
__global__ void k(const d2* __restrict__ src, int units, double* out, double* sink, int rot) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, W = blockDim.x >> 6;
  d2 acc = {0, 0};
  const int r0 = rot ? (int)((blockIdx.x * 2654435761u) % (unsigned)units) : 0;   // de-phase the workgroups
  unsigned long long t[3];
  for (int pass = 0; pass < 2; ++pass) {
    __syncthreads();
    t[pass] = __builtin_amdgcn_s_memtime();
    d2 v[D];
#pragma unroll
    for (int i = 0; i < D; ++i) v[i] = src[(size_t)((wave + i * W + r0) % units) * 64 + lane];
    for (int u = wave; u < units; u += W * D) {
#pragma unroll
      for (int i = 0; i < D; ++i) {
        acc += v[i];
        __builtin_amdgcn_sched_barrier(0);
        int nu = u + (D + i) * W; nu = nu < units ? nu : units - 1;
        v[i] = src[(size_t)((nu + r0) % units) * 64 + lane];
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) acc += v[i];
  }
  __syncthreads();
  t[2] = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[blockIdx.x * 2] = (double)(t[1] - t[0]); out[blockIdx.x * 2 + 1] = (double)(t[2] - t[1]); }
  sink[blockIdx.x * blockDim.x + threadIdx.x] = acc.x + acc.y;
}

template <int D> void run(const d2* src, int units, int grid, int W, int rot = 0) {
  double *out, *sink; (void)hipMalloc(&out, grid * 16); (void)hipMalloc(&sink, grid * W * 64 * 8);
  double best0 = 1e30, best1 = 1e30;
  for (int rep = 0; rep < 4; ++rep) {
    hipLaunchKernelGGL(k<D>, grid, 64 * W, 0, 0, src, units, out, sink, rot); (void)hipDeviceSynchronize();
    std::vector<double> h(grid * 2); (void)hipMemcpy(h.data(), out, grid * 16, hipMemcpyDeviceToHost);
    double m0 = 0, m1 = 0; for (int g = 0; g < grid; ++g) { m0 = h[2 * g] > m0 ? h[2 * g] : m0; m1 = h[2 * g + 1] > m1 ? h[2 * g + 1] : m1; }
    if (rep > 0) { best0 = m0 < best0 ? m0 : best0; best1 = m1 < best1 ? m1 : best1; }
  }
  printf("rot %d units %5d grid %4d W %d D %2d : cold %6.1f cyc/unit (%5.1f B/clk/CU)   warm %6.1f cyc/unit (%5.1f B/clk/CU)\n", rot, units, grid, W, D,
         best0 / units, 1024.0 * units / best0, best1 / units, 1024.0 * units / best1);
  (void)hipFree(out); (void)hipFree(sink);
}
int main() {
  const int maxu = 4352;
  d2* src; (void)hipMalloc(&src, (size_t)maxu * 1024); (void)hipMemset(src, 0, (size_t)maxu * 1024);
  for (int rot : {0, 1}) for (int units : {320, 4352}) for (int grid : {1, 32, 256, 512}) for (int W : {8}) { run<8>(src, units, grid, W, rot); run<16>(src, units, grid, W, rot); }
  return 0;
}
