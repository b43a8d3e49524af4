// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE for the access widths of the assembly kernels:
// streaming reads of 8 B per lane and 16 B per lane, streaming writes of 8 B per lane, 2 GiB each.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ void read8(const double* p, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  if (s == 1.2345) out[0] = s;
}
__global__ void read16(const double2* p, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { double2 v = p[i]; s += v.x + v.y; }
  if (s == 1.2345) out[0] = s;
}
__global__ void write8(double* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 1.0;
}
int main() {
  const size_t bytes = (size_t)2 << 30;
  double* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes));
  double* o; CK(hipMalloc(&o, 8));
  hipLaunchKernelGGL(read8, dim3(256 * 16), dim3(256), 0, 0, d, bytes / 8, o);
  hipLaunchKernelGGL(read16, dim3(256 * 16), dim3(256), 0, 0, (const double2*)d, bytes / 16, o);
  hipLaunchKernelGGL(write8, dim3(256 * 16), dim3(256), 0, 0, d, bytes / 8);
  CK(hipDeviceSynchronize());
  printf("each kernel moves %zu bytes = %.0f KB\n", bytes, bytes / 1024.0);
  return 0;
}
