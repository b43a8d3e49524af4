#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned lane = threadIdx.x;
  unsigned a = 1000 + lane, b = 2000 + lane;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  out[lane] = r[0];
  out[64 + lane] = r[1];
  // rot32 of a single value x: swap(x, x)
  unsigned x = 3000 + lane;
  auto q = __builtin_amdgcn_permlane32_swap(x, x, false, false);
  out[128 + lane] = q[0];
  out[192 + lane] = q[1];
}
int main() {
  unsigned* d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  unsigned h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int s = 0; s < 4; ++s) { printf("out%d:", s); for (int l = 0; l < 64; l += 8) printf(" [%d]=%u", l, h[s * 64 + l]); printf("\n"); }
  return 0;
}
