// Microbenchmark: cost of scattering dense 81x81 element blocks into a structured CSR
// (128x128x16 p=2 block), by different write strategies.  Not part of the product.
#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstdint>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Mesh { int m[3]; int n[3]; };

__device__ inline int wid(int A, int n) { int lo = A - 2 < 0 ? 0 : A - 2; int hi = A + 2 > n - 1 ? n - 1 : A + 2; return hi - lo + 1; }

// position of (B,0) in row (A,i): rowptr computed arithmetically is expensive; we precompute rowptr.
__device__ inline int64_t entry_pos(const Mesh& M, const int64_t* rowptr, int A0, int A1, int A2, int i, int B0, int B1, int B2) {
  int lo0 = A0 - 2 < 0 ? 0 : A0 - 2, lo1 = A1 - 2 < 0 ? 0 : A1 - 2, lo2 = A2 - 2 < 0 ? 0 : A2 - 2;
  int w0 = wid(A0, M.n[0]), w1 = wid(A1, M.n[1]);
  int64_t A = A0 + (int64_t)M.n[0] * (A1 + (int64_t)M.n[1] * A2);
  int nb = (B0 - lo0) + w0 * ((B1 - lo1) + w1 * (B2 - lo2));
  return rowptr[A * 3 + i] + (int64_t)nb * 3;
}

// MODE 0: atomic add, lane = (a,b) pair, 9 entries each (v1 mapping)
// MODE 1: atomic add, lane = entry (row-major over the 81 columns of a row; 81 lanes... use 2 waves)
// MODE 2: plain RMW same mapping as 1 (racy, bandwidth only)
// MODE 3: plain store same mapping as 1
// MODE 4: dense store of K_e to scratch [e][6561] (coalesced)
template<int MODE>
__global__ __launch_bounds__(256) void scatter_kernel(Mesh M, const int64_t* __restrict__ rowptr, double* __restrict__ A, double* __restrict__ scratch, int e_begin, int e_stride) {
  int e = e_begin + blockIdx.x * e_stride;
  int e0 = e % M.m[0], e1 = (e / M.m[0]) % M.m[1], e2 = e / (M.m[0] * M.m[1]);
  int tid = threadIdx.x;
  if (MODE == 0) {
    for (int pr = tid; pr < 729; pr += 256) {
      int b = pr % 27, a = pr / 27;
      int a0 = a % 3, a1 = (a / 3) % 3, a2 = a / 9, b0 = b % 3, b1 = (b / 3) % 3, b2 = b / 9;
      for (int i = 0; i < 3; ++i) {
        int64_t pos = entry_pos(M, rowptr, e0 + a0, e1 + a1, e2 + a2, i, e0 + b0, e1 + b1, e2 + b2);
        for (int j = 0; j < 3; ++j) unsafeAtomicAdd(&A[pos + j], 1.0 + pr);
      }
    }
  } else if (MODE == 4) {
    for (int k = tid; k < 6561; k += 256) scratch[(int64_t)e * 6561 + k] = 1.0 + k;
  } else {
    // entry k = row*81 + c ; row = (a,i) a-major ; c = (b1b2 segment)*9 + (b0*3+j)
    for (int k = tid; k < 6561; k += 256) {
      int row = k / 81, c = k % 81;
      int a = row / 3, i = row % 3;
      int seg = c / 9, within = c % 9;
      int a0 = a % 3, a1 = (a / 3) % 3, a2 = a / 9;
      int b1 = seg % 3, b2 = seg / 3;
      int64_t pos = entry_pos(M, rowptr, e0 + a0, e1 + a1, e2 + a2, i, e0 + 0, e1 + b1, e2 + b2) + within;
      if (MODE == 1) unsafeAtomicAdd(&A[pos], 1.0 + k);
      if (MODE == 2) A[pos] += 1.0 + k;
      if (MODE == 3) A[pos] = 1.0 + k;
    }
  }
}

int main() {
  Mesh M; M.m[0] = 128; M.m[1] = 128; M.m[2] = 16;
  for (int d = 0; d < 3; ++d) M.n[d] = M.m[d] + 2;
  int64_t n_nodes = (int64_t)M.n[0] * M.n[1] * M.n[2];
  std::vector<int64_t> rp(n_nodes * 3 + 1);
  int64_t acc = 0;
  auto wd = [&](int A, int n) { int lo = A - 2 < 0 ? 0 : A - 2; int hi = A + 2 > n - 1 ? n - 1 : A + 2; return hi - lo + 1; };
  for (int64_t A = 0; A < n_nodes; ++A) {
    int A0 = A % M.n[0], A1 = (A / M.n[0]) % M.n[1], A2 = A / ((int64_t)M.n[0] * M.n[1]);
    int64_t len = 3LL * wd(A0, M.n[0]) * wd(A1, M.n[1]) * wd(A2, M.n[2]);
    for (int i = 0; i < 3; ++i) { rp[A * 3 + i] = acc; acc += len; }
  }
  rp[n_nodes * 3] = acc;
  int64_t nnz = acc;
  int n_el = M.m[0] * M.m[1] * M.m[2];
  printf("n_el %d nnz %lld (%.2f GB values), K_e scatter %.2f GB\n", n_el, (long long)nnz, nnz * 8e-9, n_el * 6561.0 * 8e-9);
  int64_t* d_rp; double* d_A; double* d_scr;
  CK(hipMalloc(&d_rp, rp.size() * 8)); CK(hipMemcpy(d_rp, rp.data(), rp.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&d_A, nnz * 8)); CK(hipMemset(d_A, 0, nnz * 8));
  CK(hipMalloc(&d_scr, (size_t)n_el * 6561 * 8));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  auto run = [&](const char* name, auto kernel, int colours) {
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(t0));
      if (colours == 1) {
        hipLaunchKernelGGL(kernel, dim3(n_el), dim3(256), 0, 0, M, d_rp, d_A, d_scr, 0, 1);
      } else {
        // 27 colours: elements with (e0%3,e1%3,e2%3) == c ; emulate by strided launch over a permuted index:
        // here simply launch 27 times each 1/27 of the elements with stride 27 (not conflict-free, traffic only)
        for (int c = 0; c < 27; ++c) hipLaunchKernelGGL(kernel, dim3(n_el / 27), dim3(256), 0, 0, M, d_rp, d_A, d_scr, c, 27);
      }
      CK(hipEventRecord(t1)); CK(hipEventSynchronize(t1));
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 2) printf("%-40s %8.3f ms  -> %6.1f GB/s of K_e bytes, %6.2f M el/s\n", name, ms, n_el * 6561.0 * 8e-6 / ms, n_el / ms * 1e-3);
    }
  };
  run("atomic f64, lane=(a,b) pair [v1]", scatter_kernel<0>, 1);
  run("atomic f64, lane=entry (72B runs)", scatter_kernel<1>, 1);
  run("plain RMW, lane=entry, 1 launch", scatter_kernel<2>, 1);
  run("plain RMW, lane=entry, 27 launches", scatter_kernel<2>, 27);
  run("plain store, lane=entry", scatter_kernel<3>, 1);
  run("dense scratch store", scatter_kernel<4>, 1);
  return 0;
}
