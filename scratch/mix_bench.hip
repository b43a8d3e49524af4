// Do fp64 MFMA and fp64 VALU FMA overlap on a gfx950 SIMD?  Mixed kernel: in each 512-thread block
// (8 waves = 2 per SIMD) the waves of the first half issue MFMAs, those of the second half FMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef double double4_t __attribute__((ext_vector_type(4)));

// mode: 0 = all waves MFMA, 1 = all waves FMA, 2 = waves 0..3 MFMA and 4..7 FMA, 3 = as 2 but FMA half idle, 4 = MFMA half idle
__global__ __launch_bounds__(512) void mix_kernel(double* out, int iters_mfma, int iters_fma, int mode) {
  const int wave = threadIdx.x >> 6;
  const bool do_mfma = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
  const bool do_fma = mode == 1 || ((mode == 2 || mode == 4) && wave >= 4);
  double s = 0;
  if (do_mfma) {
    double4_t acc[4];
    for (int k = 0; k < 4; ++k) acc[k] = double4_t{0, 0, 0, 0};
    double a = threadIdx.x * 1e-3, b = threadIdx.x * 2e-3 + 1.0;
    for (int it = 0; it < iters_mfma; ++it) {
#pragma unroll
      for (int k = 0; k < 4; ++k) acc[k] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[k], 0, 0, 0);
    }
    for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  } else if (do_fma) {
    double acc[16];
    for (int k = 0; k < 16; ++k) acc[k] = k;
    double a = threadIdx.x * 1e-3 + 0.5, b = 1e-9;
    for (int it = 0; it < iters_fma; ++it) {
#pragma unroll
      for (int k = 0; k < 16; ++k) acc[k] = __builtin_fma(acc[k], a, b);
    }
    for (int k = 0; k < 16; ++k) s += acc[k];
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  double* d; CK(hipMalloc(&d, 1 << 24));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int blocks = 256;  // one block per CU
  const int im = 20000, ifm = 20000;   // per wave: 80000 MFMAs (64 cyc each) / 320000 FMAs (4 cyc each)
  const char* names[] = {"all 8 waves MFMA", "all 8 waves FMA", "4 waves MFMA + 4 waves FMA", "4 waves MFMA (others idle)", "4 waves FMA (others idle)"};
  for (int mode = 0; mode < 5; ++mode) {
    hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 0, 0, d, 10, 10, mode);
    CK(hipEventRecord(t0));
    hipLaunchKernelGGL(mix_kernel, dim3(blocks), dim3(512), 0, 0, d, im, ifm, mode);
    CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
    float ms; CK(hipEventElapsedTime(&ms, t0, t1));
    const double mf = (mode == 0 ? 8 : (mode == 2 || mode == 3) ? 4 : 0) * (double)blocks * im * 4 * 2048.0;
    const double ff = (mode == 1 ? 8 : (mode == 2 || mode == 4) ? 4 : 0) * (double)blocks * 64 * (double)ifm * 16 * 2.0;
    printf("%-32s %.3f ms   MFMA %.1f TFLOP/s + FMA %.1f TFLOP/s\n", names[mode], ms, mf / ms * 1e-9, ff / ms * 1e-9);
  }
  return 0;
}
