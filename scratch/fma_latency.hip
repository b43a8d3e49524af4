// Dependent-issue latency of the fp64 vector instructions on gfx950: N independent accumulators, each instruction depending on
// the one N instructions earlier; cycles per instruction by s_memtime, one wave per SIMD.  latency ~ N x (cycles per
// instruction) while that exceeds the issue cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
__device__ inline unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}
// KIND 0: v_fma_f64 acc = acc * c0 + c1 (VGPR operands); 1: v_fmac_f64 acc += s * v (SGPR coefficient); 2: v_mul_f64 acc = acc * c0
template<int N, int KIND>
__global__ void chain(unsigned long long* out, double* sink, int iters, double sc) {
  double f[N];
  for (int k = 0; k < N; ++k) f[k] = 1.0 + k * 1e-3 + threadIdx.x * 1e-6;
  const double c0 = 0.999 + threadIdx.x * 1e-9, c1 = 1e-9;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 32 / N; ++rep)
#pragma unroll
      for (int k = 0; k < N; ++k) {
        if constexpr (KIND == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[k]) : "v"(c0), "v"(c1));
        if constexpr (KIND == 1) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(f[k]) : "s"(sc), "v"(c1));
        if constexpr (KIND == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[k]) : "v"(c0));
      }
  }
  unsigned long long t1 = now();
  double s = 0;
  for (int k = 0; k < N; ++k) s += f[k];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}
template<int N, int KIND>
static void run(unsigned long long* d_out, double* d_sink) {
  const int iters = 2000, blocks = 256, threads = 256, waves = blocks * threads / 64;
  hipLaunchKernelGGL((chain<N, KIND>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 10, 1.0000001);
  hipLaunchKernelGGL((chain<N, KIND>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, iters, 1.0000001);
  CK(hipDeviceSynchronize());
  std::vector<unsigned long long> h(waves);
  CK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  std::sort(h.begin(), h.end());
  const double per = (double)h[waves / 2] / (iters * 32.0);
  printf("kind %d (%s), %2d independent chains: %6.2f cycles per instruction -> dependent latency <= %6.1f cycles\n", KIND,
         KIND == 0 ? "v_fma_f64 vgpr" : (KIND == 1 ? "v_fmac_f64 sgpr x vgpr" : "v_mul_f64"), N, per, per * N);
}
int main() {
  unsigned long long* d_out; double* d_sink;
  CK(hipMalloc(&d_out, 1 << 20)); CK(hipMalloc(&d_sink, 64));
  run<1, 0>(d_out, d_sink); run<2, 0>(d_out, d_sink); run<4, 0>(d_out, d_sink); run<8, 0>(d_out, d_sink); run<16, 0>(d_out, d_sink); run<32, 0>(d_out, d_sink);
  run<1, 1>(d_out, d_sink); run<4, 1>(d_out, d_sink); run<8, 1>(d_out, d_sink); run<16, 1>(d_out, d_sink); run<32, 1>(d_out, d_sink);
  run<1, 2>(d_out, d_sink); run<8, 2>(d_out, d_sink); run<16, 2>(d_out, d_sink);
  return 0;
}
