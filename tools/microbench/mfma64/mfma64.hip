// Issue rate and dependent latency of v_mfma_f64_16x16x4_f64 on gfx950: cycles (s_memtime) per instruction for chains of
// 1, 2, 4 accumulators per wave, with 1 or 2 waves per SIMD.  Build: hipcc -O3 --offload-arch=gfx950 mfma64.hip -o mfma64
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void __launch_bounds__(512) k(double* out, unsigned long long* cyc, int n, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = a0 + threadIdx.x, b = b0 + threadIdx.x;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int j = 0; j < n; j += NACC) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    d4 s = acc[0];
    for (int i = 1; i < NACC; ++i) s += acc[i];
    const double keep = s[0] + s[1] + s[2] + s[3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = keep;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC>
void run(int threads, int n)
{
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8);
    hipMalloc(&cyc, 256 * 8 * 8);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, n, 1.0, 2.0);
    hipDeviceSynchronize();
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    printf("acc chains %d, waves/SIMD %d, %4d MFMAs per wave: %6.1f cycles per MFMA per wave -> %6.1f per SIMD\n", NACC, threads / 256, n,
           (double)h[0] / n, (double)h[0] / n / (threads / 256));
    hipFree(out);
    hipFree(cyc);
}

// wall clock (HIP events) of a long run: ns per MFMA per SIMD, whatever s_memtime counts
template <int NACC>
void wall(int threads, int n)
{
    double* out;
    unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 8);
    hipMalloc(&cyc, 256 * 8 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, n, 1.0, 2.0);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<NACC>, dim3(256), dim3(threads), 0, 0, out, cyc, n, 1.0, 2.0);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    hipMemcpy(h, cyc, sizeof h, hipMemcpyDeviceToHost);
    const double per_simd = (double)n * (threads / 256);
    printf("wall: acc chains %d, waves/SIMD %d, %d MFMAs per wave: %.3f ms -> %.2f ns per MFMA per SIMD = %.1f TFLOP/s fp64 on 1024 SIMDs; s_memtime ticks per ns %.3f\n",
           NACC, threads / 256, n, ms, 1e6 * ms / per_simd, 2048.0 * per_simd * 1024 / (ms * 1e-3) * 1e-12, (double)h[0] / (1e6 * ms));
    hipFree(out);
    hipFree(cyc);
}

int main()
{
    for (int threads : {256, 512}) {
        wall<1>(threads, 400000);
        wall<4>(threads, 400000);
    }
    for (int n : {12, 48, 480}) {
        for (int threads : {256, 512}) {
            run<1>(threads, n);
            run<2>(threads, n);
            run<4>(threads, n);
        }
    }
    return 0;
}
