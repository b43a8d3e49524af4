// How fast does the chip read scattered runs of R bytes?  (phase 2 of the p = 2 assembly reads rows of 648 B and 216 B out
// of 17 KB pieces.)  One wave per "row group": NR runs of R bytes each at scattered offsets, summed into a register.
// hipcc --offload-arch=gfx950 -O3 -o scratch/run_bench scratch/run_bench.hip && scratch/run_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ __launch_bounds__(256) void read_runs(const double* __restrict__ buf, const unsigned* __restrict__ offs, int runs_per_wave,
                                                 int run_doubles, double* __restrict__ out, long n_waves) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= n_waves) return;
  double acc = 0.0;
  for (int k = 0; k < runs_per_wave; k += 3) {
    double v[3][4];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const double* p = buf + (size_t)offs[w * runs_per_wave + (k + c < runs_per_wave ? k + c : k)] * 8;   // offsets in units of 64 B
#pragma unroll
      for (int t = 0; t < 4; ++t) v[c][t] = (lane + 64 * t < run_doubles) ? p[lane + 64 * t] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t) acc += v[c][t];
  }
  if (acc == 1.2345e300) out[w] = acc;
}

int main() {
  const size_t bytes = 8ull << 30;
  double* buf;
  hipMalloc(&buf, bytes);
  hipMemset(buf, 0, bytes);
  double* out;
  hipMalloc(&out, 1 << 24);
  const int sizes[] = {27, 81, 243, 256};
  for (int rd : sizes) {
    // the same total volume for every run length: 8 GiB
    const long n_runs = (long)(bytes / (rd * 8));
    const int rpw = rd <= 81 ? 81 : 27;
    const long n_waves = n_runs / rpw;
    std::vector<unsigned> h((size_t)n_waves * rpw);
    // pieces of 17496 B; a wave reads one run from each of rpw different pieces (as the gather does), runs do not overlap
    const size_t slots = bytes / 64;
    unsigned long long x = 88172645463325252ull;
    for (size_t i = 0; i < h.size(); ++i) {
      x ^= x << 13; x ^= x >> 7; x ^= x << 17;
      size_t o = (x % (slots - 64));
      h[i] = (unsigned)o;
    }
    unsigned* d;
    hipMalloc(&d, h.size() * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(a);
      hipLaunchKernelGGL(read_runs, dim3((unsigned)((n_waves + 3) / 4)), dim3(256), 0, 0, buf, d, rpw, rd, out, n_waves);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (rep == 2) printf("runs of %4d B (random 64 B-aligned offsets): %.2f ms for %.2f GB -> %.2f TB/s\n", rd * 8, ms, n_runs * rd * 8 / 1e9, n_runs * rd * 8 / 1e9 / ms);
    }
    hipFree(d);
  }
  return 0;
}
