// Microbenchmark: fp64 MFMA 16x16x4 and fp64 VALU FMA issue rates on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

template<int NACC>
__global__ __launch_bounds__(256) void mfma_kernel(double* out, int iters) {
  double4_t acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = double4_t{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
  }
  double s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template<int NACC>
__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters) {
  double acc[NACC];
  for (int k = 0; k < NACC; ++k) acc[k] = k;
  double a = threadIdx.x * 1e-3 + 0.5, b = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int k = 0; k < NACC; ++k) acc[k] = __builtin_fma(acc[k], a, b);
  }
  double s = 0;
  for (int k = 0; k < NACC; ++k) s += acc[k];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double* d; CK(hipMalloc(&d, 1 << 24));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  int iters = 20000;
  for (int wpc : {4, 8, 16}) {
    int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(mfma_kernel<4>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CK(hipEventRecord(t0));
    hipLaunchKernelGGL(mfma_kernel<4>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    double n_mfma_per_simd = (double)iters * 4 * (wpc / 4.0);
    double flops = (double)blocks * 4 * iters * 4 * 2048.0;
    printf("MFMA f64 16x16x4: %2d waves/CU: %.3f ms, %.1f TFLOP/s, %.1f ns per MFMA per SIMD (=%.1f cyc @2.4GHz)\n", wpc, ms, flops / ms * 1e-9, ms * 1e6 / n_mfma_per_simd, ms * 1e6 / n_mfma_per_simd * 2.4);
  }
  for (int wpc : {4, 8, 16}) {
    int blocks = 256 * wpc / 4;
    hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(256), 0, 0, d, 10);
    CK(hipEventRecord(t0));
    hipLaunchKernelGGL(fma_kernel<8>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    double n_per_simd = (double)iters * 8 * (wpc / 4.0);
    double flops = (double)blocks * 256 * iters * 8 * 2.0;
    printf("VALU f64 FMA    : %2d waves/CU: %.3f ms, %.1f TFLOP/s, %.2f ns per wave-FMA per SIMD (=%.1f cyc @2.4GHz)\n", wpc, ms, flops / ms * 1e-9, ms * 1e6 / n_per_simd, ms * 1e6 / n_per_simd * 2.4);
  }
  return 0;
}
