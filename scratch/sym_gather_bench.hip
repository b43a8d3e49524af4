// VERDICT round 3, item 5a, as a microbenchmark: what phase 2 of the degree-2 assembly (tensor_p2_kernel: one wave per
// node, the <= 27 element blocks around it) would read if every off-diagonal (i > j) block of the element matrix were stored
// ONCE instead of twice.
//   layout A (shipped): per (element, local node a) the three rows (a, i = 0..2) are one run of 243 doubles
//                       [rows a2 = 0: 9 x 243 | rows a2 >= 1, b2 = 0: 18 x 81 | carried rows, column top only: 18 x 162]
//   layout B (stored once): a row (a, i) keeps the columns j <= i: 27 + 54 + 81 = 162 of the 243; the 81 entries with j > i
//                       are K[(b, j), (a, i)] and sit in the rows of the OTHER 27 nodes b of the same element block: per b
//                       three 8-byte reads, (b, 1)[j = 0][a], (b, 2)[j = 0][a], (b, 2)[j = 1][a]
// Same grid (128 x 128 x 16 elements, 130 x 130 x 18 nodes), same walk over the elements, same number of useful doubles; the
// waves only add what they read (the LDS row image and the CSR write of the real kernel are the same in both layouts).
// Build: hipcc --offload-arch=gfx950 -O3 -o scratch/sym_gather_bench scratch/sym_gather_bench.hip ; run: ./sym_gather_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
constexpr int NX = 128, NY = 128, NZ = 16;
constexpr int BLK_A = 2187 + 18 * 81, TOP_A = 18 * 162;      // 3645 (+ 2916 at the top element of a column)
constexpr int BLK_B = 9 * 162 + 18 * 54, TOP_B = 18 * 108;   // 2430 (+ 1944)

template<int LAYOUT>
__global__ __launch_bounds__(256) void gather(const double* __restrict__ blocks, const int64_t* __restrict__ base, double* __restrict__ out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t n = (int64_t)blockIdx.x * 4 + wave;
  const int MX = NX + 2, MY = NY + 2, MZ = NZ + 2;
  if (n >= (int64_t)MX * MY * MZ) return;
  const int A0 = n % MX, A1 = (n / MX) % MY, A2 = n / ((int64_t)MX * MY);
  double acc = 0.0;
  for (int ez = max(A2 - 2, 0); ez <= min(A2, NZ - 1); ++ez)
    for (int ey = max(A1 - 2, 0); ey <= min(A1, NY - 1); ++ey)
      for (int ex = max(A0 - 2, 0); ex <= min(A0, NX - 1); ++ex) {
        const int a = (A0 - ex) + 3 * (A1 - ey) + 9 * (A2 - ez);
        const int64_t e = ex + (int64_t)NX * (ey + (int64_t)NY * ez);
        const int64_t etop = ex + (int64_t)NX * (ey + (int64_t)NY * (NZ - 1));
        const double* blk = blocks + base[e];
        const double* top = blocks + base[etop];
        if (LAYOUT == 0) {
          if (a < 9) {
            const double* run = blk + a * 243;
            for (int t = lane; t < 243; t += 64) acc += run[t];
          } else {
            const double* r1 = blk + 2187 + (a - 9) * 81;
            const double* r2 = top + BLK_A + (a - 9) * 162;     // (the carried rows: written by the column's top element)
            for (int t = lane; t < 81; t += 64) acc += r1[t];
            for (int t = lane; t < 162; t += 64) acc += r2[t];
          }
        } else {
          // own rows: the columns j <= i
          if (a < 9) {
            const double* run = blk + a * 162;
            for (int t = lane; t < 162; t += 64) acc += run[t];
          } else {
            const double* r1 = blk + 1458 + (a - 9) * 54;
            const double* r2 = top + BLK_B + (a - 9) * 108;
            for (int t = lane; t < 54; t += 64) acc += r1[t];
            for (int t = lane; t < 108; t += 64) acc += r2[t];
          }
          // the columns j > i: from the rows of every node b of the element -- rows (b, 1) and (b, 2), three entries each
          for (int t = lane; t < 81; t += 64) {
            const int b = t / 3, k = t % 3;
            // row (b, i) starts at b * 162 + {0, 27, 81}[i] (rows b < 9; the others are as far apart); entry [j][a]
            const int pos = b < 9 ? b * 162 + (k == 0 ? 27 + a : (k == 1 ? 81 + a : 81 + 27 + a))
                                  : 1458 + (b - 9) * 54 + (k == 0 ? 9 + (a % 9) : (k == 1 ? 27 + (a % 9) : 36 + (a % 9)));
            acc += blk[pos];
          }
        }
      }
  out[n * 64 + lane] = acc;
}

int main() {
  const int64_t n_el = (int64_t)NX * NY * NZ, n_nodes = (int64_t)(NX + 2) * (NY + 2) * (NZ + 2);
  for (int layout = 0; layout < 2; ++layout) {
    const int blk = layout ? BLK_B : BLK_A, topx = layout ? TOP_B : TOP_A;
    int64_t* hb = (int64_t*)malloc(n_el * sizeof(int64_t));
    int64_t total = 0;
    for (int64_t e = 0; e < n_el; ++e) {
      hb[e] = total;
      total += blk + ((e / ((int64_t)NX * NY)) == NZ - 1 ? topx : 0);
    }
    double *d_blocks, *d_out;
    int64_t* d_base;
    CK(hipMalloc(&d_blocks, total * sizeof(double)));
    CK(hipMemset(d_blocks, 0, total * sizeof(double)));
    CK(hipMalloc(&d_out, n_nodes * 64 * sizeof(double)));
    CK(hipMalloc(&d_base, n_el * sizeof(int64_t)));
    CK(hipMemcpy(d_base, hb, n_el * sizeof(int64_t), hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((n_nodes + 3) / 4);
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipEventRecord(e0));
      if (layout == 0) hipLaunchKernelGGL(gather<0>, dim3(grid), dim3(256), 0, 0, d_blocks, d_base, d_out);
      else hipLaunchKernelGGL(gather<1>, dim3(grid), dim3(256), 0, 0, d_blocks, d_base, d_out);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (rep && ms < best) best = ms;
    }
    printf("layout %s: pieces %.2f GB, gather %.3f ms = %.2f TB/s of piece bytes\n", layout ? "B (off-diagonal blocks stored once)" : "A (shipped)",
           total * 8e-9, best, total * 8e-9 / best);
    CK(hipFree(d_blocks));
    CK(hipFree(d_out));
    CK(hipFree(d_base));
    free(hb);
  }
  return 0;
}
