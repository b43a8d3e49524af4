// What `row_newbcast:N` does to a 64-bit DPP operand on gfx950: every lane of a row of 16 must read lane N of ITS row.
// Build: hipcc --offload-arch=gfx950 -O3 -o dpp_bcast scratch/dpp_bcast.hip ; prints "ok" or the first mismatch.
#include <hip/hip_runtime.h>
#include <cstdio>
template<int N>
__global__ void k(double* out) {
  const int lane = threadIdx.x;
  double tab = 1000.0 * (lane >> 4) + (lane & 15);   // row r, lane n of the row -> 1000 r + n
  double e = 0.5, w = 2.0, c = 0.0;
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(e) : "v"(tab), "v"(w), "n"(N));
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(tab), "n"(N));
  out[lane] = e;
  out[64 + lane] = c;
}
template<int N>
int check(double* d) {
  hipLaunchKernelGGL(k<N>, dim3(1), dim3(64), 0, 0, d);
  double h[128];
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  for (int l = 0; l < 64; ++l) {
    const double t = 1000.0 * (l >> 4) + N;
    if (h[l] != 0.5 + t * 2.0 || h[64 + l] != t) { printf("N %d lane %d: fmac %g mov %g, expected %g %g\n", N, l, h[l], h[64 + l], 0.5 + 2 * t, t); return 1; }
  }
  return 0;
}
int main() {
  double* d;
  if (hipMalloc(&d, 128 * sizeof(double)) != hipSuccess) return 1;
  int bad = check<0>(d) + check<3>(d) + check<7>(d) + check<12>(d) + check<15>(d);
  printf(bad ? "MISMATCH\n" : "ok: row_newbcast:N gives every lane the value of lane N of its own row of 16 (v_fmac_f64_dpp, v_mov_b64_dpp)\n");
  return bad;
}
