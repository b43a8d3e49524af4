// Device-resident linear solve of the Newton step and the caller-side eliminations around it (SURVEY 8 rows a10,
// f-4): what the reference does after the assembly returns.
//
//   forms::Nonlinear::AddMult / AddMultGrad  (forms/nonlinear.hpp:76-80,112-115): r[ess] = 0;
//       grad.EliminateRowCol(ess, DIAG_ONE) -- serial loops over the essential dofs there, one kernel each here.
//   PyNonlinearSolid::Setup, "use_iterative_solver" (py/py_nonlinear_solid.cpp:329-339): mfem::GMRESSolver with an
//       mfem::DSmoother (Jacobi) preconditioner, rel 1e-8, abs 1e-12, 300 iterations.  MFEM is an absent, un-pinned
//       submodule of the reference; what is restated here is its published algorithm (mfem linalg/solvers.cpp,
//       GMRESSolver::Mult): left-preconditioned restarted GMRES(m = 50), modified Gram-Schmidt, Givens rotations,
//       convergence on the preconditioned residual  |s_{i+1}| <= max(rel_tol * ||M r_0||, abs_tol).
//
// Everything vector-sized stays in HBM; per Arnoldi step the host sees one Hessenberg column.  Reductions are
// deterministic: a fixed grid writes per-block partial sums, every consumer adds them in the same order.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <memory>
#include <string>
#include <vector>

#include "../../include/mimi_hip.h"
#include "common.hpp"

namespace mimi_hip {

constexpr int KR_BLOCKS = 512;    // reduction grid (two blocks per CU), also the number of partial sums per dot
constexpr int KR_THREADS = 256;

__device__ __forceinline__ double kr_wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// sum over the block, result valid in thread 0
__device__ __forceinline__ double kr_block_sum(double v, double* sh) {
  v = kr_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int w = 0; w < KR_THREADS / 64; ++w) t += sh[w];
  __syncthreads();
  return t;
}

// the scalar behind KR_BLOCKS partial sums, same order for every reader (every thread gets it)
__device__ __forceinline__ double kr_total(const double* partial, double* sh) {
  double v = 0.0;
  for (int k = threadIdx.x; k < KR_BLOCKS; k += KR_THREADS) v += partial[k];
  v = kr_wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < KR_THREADS / 64; ++w) t += sh[w];
  __syncthreads();
  return t;
}

// The products of one wave: ROWS consecutive rows that share ONE column list (ROWS = 3: the three dofs of a node in the
// byVDIM numbering of py_nonlinear_solid.cpp:63, whose CSR rows have identical columns; ROWS = 1: any CSR matrix).  The
// column indices and the gathered x are read once for the ROWS rows; four entries per lane are in flight per trip.  Every
// lane adds its entries in increasing position and the lanes are combined by the same shuffle tree for both values of
// ROWS, so a row's sum does not depend on which form ran.
// NODECOL (ROWS == 3 only): the columns of a row come in triples 3 c, 3 c + 1, 3 c + 2 (the dofs of node c), and `col` is the
// list of the nodes c instead -- a third of the index bytes of the group form, a ninth of the plain one.
template<int ROWS, bool NODECOL>
__device__ __forceinline__ void kr_row_products(int64_t row0, int lane, const int64_t* __restrict__ rowptr,
                                                const int32_t* __restrict__ col, const double* __restrict__ val,
                                                const double* __restrict__ x, double (&s)[ROWS]) {
  const double* v[ROWS];
  const int64_t beg = rowptr[row0];
  const int len = (int)(rowptr[row0 + 1] - beg);
#pragma unroll
  for (int j = 0; j < ROWS; ++j) {
    v[j] = val + (j == 0 ? beg : rowptr[row0 + j]);
    s[j] = 0.0;
  }
  const int32_t* c = col + (NODECOL ? beg / 9 : beg);
  // four entries per lane and trip, every load of the trip issued before the first product
  for (int k0 = lane; k0 < len; k0 += 256) {
    int32_t cc[4];
    double a[4][ROWS], xx[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int k = k0 + 64 * u;
      if constexpr (NODECOL) cc[u] = k < len ? 3 * c[k / 3] + k % 3 : 0;
      else cc[u] = k < len ? c[k] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int j = 0; j < ROWS; ++j) a[u][j] = k0 + 64 * u < len ? v[j][k0 + 64 * u] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u) xx[u] = k0 + 64 * u < len ? x[cc[u]] : 0.0;
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (k0 + 64 * u < len) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) s[j] = __builtin_fma(a[u][j], xx[u], s[j]);
      }
  }
#pragma unroll
  for (int j = 0; j < ROWS; ++j) s[j] = kr_wave_sum(s[j]);
}

// y = Dinv (A x) (or A x when dinv == nullptr; y = Dinv (b - A x) when b != nullptr): one wave per unit of ROWS rows
template<int ROWS, bool NODECOL>
__global__ __launch_bounds__(256) void kr_spmv_kernel(int64_t n_units, const int64_t* __restrict__ rowptr,
                                                      const int32_t* __restrict__ col, const double* __restrict__ val,
                                                      const double* __restrict__ x, const double* __restrict__ b,
                                                      const double* __restrict__ dinv, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int64_t row0 = unit * ROWS;
  double s[ROWS];
  kr_row_products<ROWS, NODECOL>(row0, lane, rowptr, col, val, x, s);
  if (lane == 0) {   // kr_wave_sum leaves the sums in lane 0
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
      const double t = b ? b[row0 + j] - s[j] : s[j];
      y[row0 + j] = dinv ? dinv[row0 + j] * t : t;
    }
  }
}

// y += alpha A x (mfem::SparseMatrix::AddMult)
template<int ROWS, bool NODECOL>
__global__ __launch_bounds__(256) void kr_add_mult_kernel(int64_t n_units, const int64_t* __restrict__ rowptr,
                                                          const int32_t* __restrict__ col, const double* __restrict__ val,
                                                          const double* __restrict__ x, double alpha, double* __restrict__ y) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int64_t row0 = unit * ROWS;
  double s[ROWS];
  kr_row_products<ROWS, NODECOL>(row0, lane, rowptr, col, val, x, s);
  if (lane == 0) {
#pragma unroll
    for (int j = 0; j < ROWS; ++j) y[row0 + j] += alpha * s[j];
  }
}

// do rows g b, ..., g b + g - 1 hold the same column list for every b?  One wave per row that is not the first of its
// group; a difference sets `bit` of *status
__global__ __launch_bounds__(256) void kr_groups_kernel(int64_t n, int g, const int64_t* __restrict__ rowptr,
                                                        const int32_t* __restrict__ col, int bit, int* __restrict__ status) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n || row % g == 0) return;
  const int64_t first = row - row % g;
  const int64_t beg = rowptr[row], beg0 = rowptr[first];
  const int64_t len = rowptr[row + 1] - beg;
  bool differ = len != rowptr[first + 1] - beg0;
  if (!differ)
    for (int64_t k = lane; k < len; k += 64) differ |= col[beg + k] != col[beg0 + k];
  if (differ) atomicOr(status, bit);
}

// rows in groups of three with one column list: is that list made of triples 3 c, 3 c + 1, 3 c + 2?  (one wave per group;
// a difference sets `bit` of *status); and the list of the c, at rowptr[3 u] / 9
__global__ __launch_bounds__(256) void kr_nodecol_kernel(int64_t n_units, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                         int bit, int* __restrict__ status, int32_t* __restrict__ ncol) {
  const int lane = threadIdx.x & 63;
  const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (unit >= n_units) return;
  const int64_t beg = rowptr[3 * unit];
  const int len = (int)(rowptr[3 * unit + 1] - beg);
  if (!ncol) {
    bool differ = len % 3 != 0 || beg % 9 != 0;
    for (int m = lane; 3 * m + 2 < len; m += 64) {
      const int32_t c0 = col[beg + 3 * m];
      differ |= c0 % 3 != 0 || col[beg + 3 * m + 1] != c0 + 1 || col[beg + 3 * m + 2] != c0 + 2;
    }
    if (differ) atomicOr(status, bit);
  } else {
    for (int m = lane; m < len / 3; m += 64) ncol[beg / 9 + m] = col[beg + 3 * m] / 3;
  }
}

// dinv[row] = 1 / A(row, row)   (mfem::DSmoother, type 0, scale 1)
__global__ void kr_diag_kernel(int64_t n, const int64_t* __restrict__ rowptr, const int64_t* __restrict__ diag_pos,
                               const double* __restrict__ val, double* __restrict__ dinv) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row < n) dinv[row] = 1.0 / val[rowptr[row] + diag_pos[row]];
}

__global__ void kr_diag_pos_kernel(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                   int64_t* __restrict__ diag_pos, int* __restrict__ status) {
  const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= n) return;
  int64_t lo = rowptr[row], hi = rowptr[row + 1];
  const int64_t base = lo;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (col[mid] < row) lo = mid + 1; else hi = mid;
  }
  if (lo >= rowptr[row + 1] || col[lo] != row) {
    atomicOr(status, 1);
    diag_pos[row] = 0;
    return;
  }
  diag_pos[row] = lo - base;
}

// one modified Gram-Schmidt stage: w -= h_prev * v_prev (h_prev = total of partial_prev; skipped when v_prev is null),
// then partial_out[block] = sum_block w * v_next (v_next == w itself gives the squared norm)
__global__ __launch_bounds__(KR_THREADS) void kr_mgs_kernel(int64_t n, double* __restrict__ w, const double* __restrict__ v_prev,
                                                            const double* __restrict__ partial_prev, const double* v_next,
                                                            double* __restrict__ partial_out) {
  __shared__ double sh[KR_THREADS / 64];
  constexpr int B = 8;   // entries of a thread whose loads are in flight together (n <= 1 M: all of them)
  const int64_t stride = (int64_t)KR_BLOCKS * KR_THREADS;
  const bool self = (v_next == w);
  double h = 0.0, acc = 0.0;
  const int64_t first = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x;
  int64_t base = first;
  do {   // (every thread makes the first trip, entries or not: the total below holds block barriers)
    double wv[B], pv[B], nv[B];
#pragma unroll
    for (int e = 0; e < B; ++e) {
      const int64_t i = base + e * stride;
      wv[e] = i < n ? w[i] : 0.0;
      pv[e] = (v_prev && i < n) ? v_prev[i] : 0.0;
      nv[e] = (!self && i < n) ? v_next[i] : 0.0;
    }
    if (v_prev && base == first) h = kr_total(partial_prev, sh);   // after the loads are issued
#pragma unroll
    for (int e = 0; e < B; ++e) {
      const int64_t i = base + e * stride;
      if (i < n) {
        double wi = wv[e];
        if (v_prev) {
          wi -= h * pv[e];
          w[i] = wi;
        }
        acc += wi * (self ? wi : nv[e]);
      }
    }
    base += B * stride;
  } while (base < n);
  const double t = kr_block_sum(acc, sh);
  if (threadIdx.x == 0) partial_out[blockIdx.x] = t;
}

// v = w / sqrt(total(partial)); scalars[slot] for the host: the totals of n_cols partial arrays
__global__ __launch_bounds__(KR_THREADS) void kr_normalize_kernel(int64_t n, const double* __restrict__ w,
                                                                  const double* __restrict__ partial_norm2, double* __restrict__ v) {
  __shared__ double sh[KR_THREADS / 64];
  const double nrm = sqrt(kr_total(partial_norm2, sh));
  const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
  for (int64_t i = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x; i < n; i += (int64_t)KR_BLOCKS * KR_THREADS) v[i] = w[i] * inv;
}

// totals[c] = total of partial array c, c < n_cols (one block per column; same order as kr_total)
__global__ __launch_bounds__(KR_THREADS) void kr_totals_kernel(const double* __restrict__ partials, double* __restrict__ totals) {
  __shared__ double sh[KR_THREADS / 64];
  const double t = kr_total(partials + (int64_t)blockIdx.x * KR_BLOCKS, sh);
  if (threadIdx.x == 0) totals[blockIdx.x] = t;
}

// x += sum_k y[k] V[k]
__global__ __launch_bounds__(KR_THREADS) void kr_update_kernel(int64_t n, int k_count, const double* __restrict__ V, int64_t ldv,
                                                               const double* __restrict__ y, double* __restrict__ x) {
  for (int64_t i = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * KR_THREADS) {
    double s = x[i];
    for (int k = 0; k < k_count; ++k) s += y[k] * V[(int64_t)k * ldv + i];
    x[i] = s;
  }
}

// conjugate gradients: z = dinv o r (or r), partial_out = partial sums of r . z
__global__ __launch_bounds__(KR_THREADS) void kr_cg_precond_kernel(int64_t n, const double* __restrict__ r, const double* __restrict__ dinv,
                                                                   double* __restrict__ z, double* __restrict__ partial_out) {
  __shared__ double sh[KR_THREADS / 64];
  double acc = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x; i < n; i += (int64_t)KR_BLOCKS * KR_THREADS) {
    const double ri = r[i], zi = dinv ? dinv[i] * ri : ri;
    z[i] = zi;
    acc += ri * zi;
  }
  const double t = kr_block_sum(acc, sh);
  if (threadIdx.x == 0) partial_out[blockIdx.x] = t;
}

// x += alpha d; r -= alpha q
__global__ __launch_bounds__(KR_THREADS) void kr_cg_update_kernel(int64_t n, double alpha, const double* __restrict__ d,
                                                                  const double* __restrict__ q, double* __restrict__ x, double* __restrict__ r) {
  for (int64_t i = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * KR_THREADS) {
    x[i] += alpha * d[i];
    r[i] -= alpha * q[i];
  }
}

// d = z + beta d
__global__ __launch_bounds__(KR_THREADS) void kr_cg_direction_kernel(int64_t n, double beta, const double* __restrict__ z, double* __restrict__ d) {
  for (int64_t i = (int64_t)blockIdx.x * KR_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * KR_THREADS) d[i] = z[i] + beta * d[i];
}

// forms/nonlinear.hpp:76-80: r[ess] = 0
__global__ void kr_zero_entries_kernel(int64_t n_ess, const int64_t* __restrict__ ess, double* __restrict__ r) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_ess) r[ess[k]] = 0.0;
}

// SparseMatrix::EliminateRowCol(rc, DIAG_ONE) for every essential dof (forms/nonlinear.hpp:112-115): one wave per row;
// an entry goes when its row or its column is essential, the diagonal of an essential row becomes 1
__global__ __launch_bounds__(256) void kr_eliminate_kernel(int64_t n, const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                                           const unsigned char* __restrict__ is_ess, double* __restrict__ val) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= n) return;
  const bool row_ess = is_ess[row];
  for (int64_t k = rowptr[row] + lane; k < rowptr[row + 1]; k += 64) {
    const int32_t c = col[k];
    if (row_ess) val[k] = (c == row) ? 1.0 : 0.0;
    else if (is_ess[c]) val[k] = 0.0;
  }
}

__global__ void kr_mark_kernel(int64_t n_ess, const int64_t* __restrict__ ess, unsigned char* __restrict__ is_ess) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_ess) is_ess[ess[k]] = 1;
}

}  // namespace mimi_hip

using namespace mimi_hip;

struct mimi_hip_linear_s {
  int device = 0;
  int64_t n = 0, nnz = 0;
  int kdim = 50;
  hipStream_t stream = nullptr, own_stream = nullptr;
  DeviceBuffer<int64_t> rowptr_own, diag_pos, ess;
  DeviceBuffer<int32_t> col_own;
  const int64_t* rowptr = nullptr;
  const int32_t* col = nullptr;
  DeviceBuffer<unsigned char> is_ess;
  int64_t n_ess = 0;
  int group = 1;          // rows g b .. g b + g - 1 share their column list (kr_groups_kernel; g = 3, 2 or 1): read once per group
  bool nodecol = false;   // ... and, g = 3, the list is made of node triples: `ncol` holds the nodes (kr_nodecol_kernel)
  DeviceBuffer<int32_t> ncol;
  DeviceBuffer<double> dinv, V, w, r, partials, totals, ycoef, stage_val, stage_b, stage_x;
  int* status_dev = nullptr;
  double* column_host[2] = {nullptr, nullptr};   // pinned: the Hessenberg column of the step before last and of the last one
  hipEvent_t column_ready[2] = {nullptr, nullptr};
  int column_cap = 0;
  void reserve_columns(int count) {
    if (count <= column_cap) return;
    for (int k = 0; k < 2; ++k) {
      if (column_host[k]) MH_HIP(hipHostFree(column_host[k]));
      column_host[k] = nullptr;
      MH_HIP(hipHostMalloc(reinterpret_cast<void**>(&column_host[k]), sizeof(double) * count, hipHostMallocDefault));
      if (!column_ready[k]) MH_HIP(hipEventCreateWithFlags(&column_ready[k], hipEventDisableTiming));
    }
    column_cap = count;
  }
  ~mimi_hip_linear_s() {
    for (int k = 0; k < 2; ++k) {
      if (column_host[k]) (void)hipHostFree(column_host[k]);
      if (column_ready[k]) (void)hipEventDestroy(column_ready[k]);
    }
    if (status_dev) (void)hipFree(status_dev);
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};

namespace {

template<typename F>
int guarded_k(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  } catch (...) {
    set_last_error("unknown error");
    return 1;
  }
}

void spmv(mimi_hip_linear_s* h, const double* val, const double* x, const double* b, const double* dinv, double* y) {
  if (h->nodecol)
    hipLaunchKernelGGL((kr_spmv_kernel<3, true>), dim3((unsigned)((h->n / 3 + 3) / 4)), dim3(256), 0, h->stream, h->n / 3, h->rowptr,
                       h->ncol.ptr, val, x, b, dinv, y);
  else if (h->group == 3)
    hipLaunchKernelGGL((kr_spmv_kernel<3, false>), dim3((unsigned)((h->n / 3 + 3) / 4)), dim3(256), 0, h->stream, h->n / 3, h->rowptr,
                       h->col, val, x, b, dinv, y);
  else if (h->group == 2)
    hipLaunchKernelGGL((kr_spmv_kernel<2, false>), dim3((unsigned)((h->n / 2 + 3) / 4)), dim3(256), 0, h->stream, h->n / 2, h->rowptr,
                       h->col, val, x, b, dinv, y);
  else
    hipLaunchKernelGGL((kr_spmv_kernel<1, false>), dim3((unsigned)((h->n + 3) / 4)), dim3(256), 0, h->stream, h->n, h->rowptr, h->col,
                       val, x, b, dinv, y);
  MH_HIP(hipGetLastError());
}

}  // namespace

extern "C" {

int mimi_hip_linear_create(int64_t n, const int64_t* csr_rowptr, const int32_t* csr_col, const int64_t* ess_dofs,
                           int64_t n_ess, int device, mimi_hip_linear_t* out) {
  return guarded_k([&] {
    if (!out || !csr_rowptr || !csr_col || n < 1) fail("null / empty argument");
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
      fail("libmimi_hip: no HIP device visible -- this library has no CPU fallback");
    if (device < 0 || device >= count) fail("device %d out of range (%d visible)", device, count);
    auto h = std::make_unique<mimi_hip_linear_s>();
    h->device = device;
    h->n = n;
    MH_HIP(hipSetDevice(device));
    MH_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    if (is_device_pointer(csr_rowptr)) {
      h->rowptr = csr_rowptr;
    } else {
      h->rowptr_own.assign(csr_rowptr, (size_t)n + 1, h->stream);
      h->rowptr = h->rowptr_own.ptr;
    }
    int64_t nnz = 0;
    MH_HIP(hipMemcpy(&nnz, h->rowptr + n, sizeof(int64_t), hipMemcpyDeviceToHost));
    h->nnz = nnz;
    if (is_device_pointer(csr_col)) {
      h->col = csr_col;
    } else {
      h->col_own.assign(csr_col, (size_t)nnz, h->stream);
      h->col = h->col_own.ptr;
    }
    MH_HIP(hipMalloc(reinterpret_cast<void**>(&h->status_dev), sizeof(int)));
    MH_HIP(hipMemsetAsync(h->status_dev, 0, sizeof(int), h->stream));
    h->diag_pos.resize((size_t)n);
    hipLaunchKernelGGL(kr_diag_pos_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, h->stream, n, h->rowptr, h->col,
                       h->diag_pos.ptr, h->status_dev);
    MH_HIP(hipGetLastError());
    int status = 0;
    MH_HIP(hipMemcpyAsync(&status, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
    if (status) fail("CSR pattern has a row without a diagonal entry");
    // the dofs of a node (byVDIM numbering) have identical rows in the pattern the integrators assemble into
    for (int g = 3; g >= 2; --g)
      if (n % g == 0)
        hipLaunchKernelGGL(kr_groups_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, h->stream, n, g, h->rowptr, h->col, 1 << g,
                           h->status_dev);
    MH_HIP(hipGetLastError());
    MH_HIP(hipMemcpyAsync(&status, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
    h->group = (n % 3 == 0 && !(status & 8)) ? 3 : (n % 2 == 0 && !(status & 4)) ? 2 : 1;
    if (h->group == 3 && nnz % 9 == 0) {
      hipLaunchKernelGGL(kr_nodecol_kernel, dim3((unsigned)((n / 3 + 3) / 4)), dim3(256), 0, h->stream, n / 3, h->rowptr, h->col, 16,
                         h->status_dev, (int32_t*)nullptr);
      MH_HIP(hipGetLastError());
      MH_HIP(hipMemcpyAsync(&status, h->status_dev, sizeof(int), hipMemcpyDeviceToHost, h->stream));
      MH_HIP(hipStreamSynchronize(h->stream));
      if (!(status & 16)) {
        h->ncol.resize((size_t)(nnz / 9));
        hipLaunchKernelGGL(kr_nodecol_kernel, dim3((unsigned)((n / 3 + 3) / 4)), dim3(256), 0, h->stream, n / 3, h->rowptr, h->col, 16,
                           h->status_dev, h->ncol.ptr);
        MH_HIP(hipGetLastError());
        h->nodecol = true;
      }
    }
    h->is_ess.resize((size_t)n);
    MH_HIP(hipMemsetAsync(h->is_ess.ptr, 0, (size_t)n, h->stream));
    h->n_ess = n_ess;
    if (n_ess > 0) {
      if (!ess_dofs) fail("null essential dof list");
      h->ess.assign(ess_dofs, (size_t)n_ess, h->stream);
      hipLaunchKernelGGL(kr_mark_kernel, dim3((unsigned)((n_ess + 255) / 256)), dim3(256), 0, h->stream, n_ess, h->ess.ptr,
                         h->is_ess.ptr);
      MH_HIP(hipGetLastError());
    }
    MH_HIP(hipStreamSynchronize(h->stream));
    *out = h.release();
  });
}

int mimi_hip_linear_destroy(mimi_hip_linear_t h) {
  return guarded_k([&] {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
  });
}

int mimi_hip_linear_set_stream(mimi_hip_linear_t h, void* stream) {
  return guarded_k([&] {
    if (!h) fail("null handle");
    h->stream = stream == MIMI_HIP_STREAM_NULL ? nullptr : (stream ? reinterpret_cast<hipStream_t>(stream) : h->own_stream);
  });
}

int64_t mimi_hip_linear_info(mimi_hip_linear_t h, int what) {
  if (!h) return -1;
  switch (what) {
    case 0: return h->n;
    case 1: return h->nnz;
    case 2: return h->group;
    case 3: return h->nodecol ? 1 : 0;
    default: return -1;
  }
}

int mimi_hip_linear_eliminate(mimi_hip_linear_t h, double* r, double* A_values) {
  return guarded_k([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    if (h->n_ess == 0) return;
    if (r) {
      Mirror<double> mr = Mirror<double>::inout(r, (size_t)h->n, h->stage_b, h->stream);
      hipLaunchKernelGGL(kr_zero_entries_kernel, dim3((unsigned)((h->n_ess + 255) / 256)), dim3(256), 0, h->stream, h->n_ess,
                         h->ess.ptr, mr.dev);
      MH_HIP(hipGetLastError());
      mr.finish(h->stream);
      if (mr.host) MH_HIP(hipStreamSynchronize(h->stream));
    }
    if (A_values) {
      Mirror<double> mA = Mirror<double>::inout(A_values, (size_t)h->nnz, h->stage_val, h->stream);
      hipLaunchKernelGGL(kr_eliminate_kernel, dim3((unsigned)((h->n + 3) / 4)), dim3(256), 0, h->stream, h->n, h->rowptr, h->col,
                         h->is_ess.ptr, mA.dev);
      MH_HIP(hipGetLastError());
      mA.finish(h->stream);
      if (mA.host) MH_HIP(hipStreamSynchronize(h->stream));
    }
  });
}

int mimi_hip_linear_add_mult(mimi_hip_linear_t h, const double* A_values, const double* x, double alpha, double* y) {
  return guarded_k([&] {
    if (!h) fail("null handle");
    if (!A_values || !x || !y) fail("null vector argument");
    MH_HIP(hipSetDevice(h->device));
    Mirror<double> mA = Mirror<double>::in(A_values, (size_t)h->nnz, h->stage_val, h->stream);
    Mirror<double> mx = Mirror<double>::in(x, (size_t)h->n, h->stage_x, h->stream);
    Mirror<double> my = Mirror<double>::inout(y, (size_t)h->n, h->stage_b, h->stream);
    if (h->nodecol)
      hipLaunchKernelGGL((kr_add_mult_kernel<3, true>), dim3((unsigned)((h->n / 3 + 3) / 4)), dim3(256), 0, h->stream, h->n / 3,
                         h->rowptr, h->ncol.ptr, mA.dev, mx.dev, alpha, my.dev);
    else if (h->group == 3)
      hipLaunchKernelGGL((kr_add_mult_kernel<3, false>), dim3((unsigned)((h->n / 3 + 3) / 4)), dim3(256), 0, h->stream, h->n / 3,
                         h->rowptr, h->col, mA.dev, mx.dev, alpha, my.dev);
    else if (h->group == 2)
      hipLaunchKernelGGL((kr_add_mult_kernel<2, false>), dim3((unsigned)((h->n / 2 + 3) / 4)), dim3(256), 0, h->stream, h->n / 2,
                         h->rowptr, h->col, mA.dev, mx.dev, alpha, my.dev);
    else
      hipLaunchKernelGGL((kr_add_mult_kernel<1, false>), dim3((unsigned)((h->n + 3) / 4)), dim3(256), 0, h->stream, h->n, h->rowptr,
                         h->col, mA.dev, mx.dev, alpha, my.dev);
    MH_HIP(hipGetLastError());
    my.finish(h->stream);
    if (mA.host || mx.host || my.host) MH_HIP(hipStreamSynchronize(h->stream));
  });
}

int mimi_hip_linear_gmres(mimi_hip_linear_t h, const double* A_values, const double* b, double* x, double rel_tol, double abs_tol,
                          int max_iter, int kdim, int use_jacobi, int32_t* iterations, double* final_norm, int32_t* converged) {
  return guarded_k([&] {
    if (!h || !A_values || !b || !x) fail("null argument");
    if (kdim < 1) kdim = 50;   // mfem::GMRESSolver default m
    MH_HIP(hipSetDevice(h->device));
    const int64_t n = h->n;
    hipStream_t s = h->stream;
    Mirror<double> mA = Mirror<double>::in(A_values, (size_t)h->nnz, h->stage_val, s);
    Mirror<double> mb = Mirror<double>::in(b, (size_t)n, h->stage_b, s);
    Mirror<double> mx = Mirror<double>::inout(x, (size_t)n, h->stage_x, s);   // overwritten: iterative_mode == false
    h->V.resize((size_t)(kdim + 1) * n);
    h->w.resize((size_t)n);
    h->r.resize((size_t)n);
    h->partials.resize((size_t)(kdim + 2) * KR_BLOCKS);
    h->totals.resize((size_t)(kdim + 2));
    h->ycoef.resize((size_t)kdim);
    h->reserve_columns(kdim + 2);
    const double* dinv = nullptr;
    if (use_jacobi) {
      h->dinv.resize((size_t)n);
      hipLaunchKernelGGL(kr_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, h->rowptr, h->diag_pos.ptr, mA.dev,
                         h->dinv.ptr);
      MH_HIP(hipGetLastError());
      dinv = h->dinv.ptr;
    }
    double* V = h->V.ptr;
    double* w = h->w.ptr;
    double* r = h->r.ptr;
    double* part = h->partials.ptr;
    std::vector<double> H((size_t)(kdim + 1) * kdim, 0.0), sv(kdim + 1, 0.0), cs(kdim + 1, 0.0), sn(kdim + 1, 0.0),
        y(kdim);
    auto Hat = [&](int i, int j) -> double& { return H[(size_t)i + (size_t)j * (kdim + 1)]; };
    auto norm_of = [&](const double* vec) -> double {   // vec also goes to `part[0..)` as its squared norm
      hipLaunchKernelGGL(kr_mgs_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, const_cast<double*>(vec), (const double*)nullptr,
                         (const double*)nullptr, vec, part);
      hipLaunchKernelGGL(kr_totals_kernel, dim3(1), dim3(KR_THREADS), 0, s, part, h->totals.ptr);
      MH_HIP(hipGetLastError());
      double t = 0.0;
      MH_HIP(hipMemcpyAsync(&t, h->totals.ptr, sizeof(double), hipMemcpyDeviceToHost, s));
      MH_HIP(hipStreamSynchronize(s));
      return std::sqrt(t);
    };
    auto update = [&](int k_count) {   // GMRESSolver: Update(x, k, H, s, v)
      for (int i = k_count - 1; i >= 0; --i) {
        double t = sv[i];
        for (int j = i + 1; j < k_count; ++j) t -= Hat(i, j) * y[j];
        y[i] = t / Hat(i, i);
      }
      MH_HIP(hipMemcpyAsync(h->ycoef.ptr, y.data(), sizeof(double) * k_count, hipMemcpyHostToDevice, s));
      hipLaunchKernelGGL(kr_update_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, k_count, V, n, h->ycoef.ptr, mx.dev);
      MH_HIP(hipGetLastError());
      MH_HIP(hipStreamSynchronize(s));   // y is host memory reused by the next cycle
    };
    auto finish = [&](int it, double nrm, bool conv) {
      if (iterations) *iterations = it;
      if (final_norm) *final_norm = nrm;
      if (converged) *converged = conv ? 1 : 0;
      mx.finish(s);
      MH_HIP(hipStreamSynchronize(s));
    };

    // x = 0; r = M b
    MH_HIP(hipMemsetAsync(mx.dev, 0, sizeof(double) * n, s));
    {
      // r = M (b - A x) with x = 0
      spmv(h, mA.dev, mx.dev, mb.dev, dinv, r);
    }
    double beta = norm_of(r);
    double goal = std::fmax(rel_tol * beta, abs_tol);
    if (beta <= goal) {
      finish(0, beta, true);
      return;
    }
    // One Arnoldi step on the stream: w = M A v_i, the i + 2 modified Gram-Schmidt stages (stage k subtracts
    // h_{k-1} v_{k-1} and forms h_k = w . v_k; the last one forms ||w||^2), v_{i+1}, and the Hessenberg column into
    // pinned host memory.  Nothing here waits for the host, so step i + 1 is launched BEFORE the host reads column i:
    // the round trip of the column is hidden behind the next step, and when column i ends the solve the step that was
    // launched ahead is simply not used (the update reads v_0 .. v_i only).
    auto launch_step = [&](int i) {
      spmv(h, mA.dev, V + (int64_t)i * n, nullptr, dinv, w);
      for (int k = 0; k <= i + 1; ++k) {
        const double* v_prev = k > 0 ? V + (int64_t)(k - 1) * n : nullptr;
        const double* p_prev = k > 0 ? part + (int64_t)(k - 1) * KR_BLOCKS : nullptr;
        const double* v_next = k <= i ? V + (int64_t)k * n : w;
        hipLaunchKernelGGL(kr_mgs_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, w, v_prev, p_prev, v_next,
                           part + (int64_t)k * KR_BLOCKS);
      }
      hipLaunchKernelGGL(kr_normalize_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, w, part + (int64_t)(i + 1) * KR_BLOCKS,
                         V + (int64_t)(i + 1) * n);
      hipLaunchKernelGGL(kr_totals_kernel, dim3(i + 2), dim3(KR_THREADS), 0, s, part, h->totals.ptr);
      MH_HIP(hipGetLastError());
      MH_HIP(hipMemcpyAsync(h->column_host[i & 1], h->totals.ptr, sizeof(double) * (i + 2), hipMemcpyDeviceToHost, s));
      MH_HIP(hipEventRecord(h->column_ready[i & 1], s));
    };
    int j = 1;
    while (j <= max_iter) {
      // v_0 = r / beta (partials of ||r||^2 are in part[0..))
      hipLaunchKernelGGL(kr_normalize_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, r, part, V);
      std::fill(sv.begin(), sv.end(), 0.0);
      sv[0] = beta;
      const int steps = std::min(kdim, max_iter - j + 1);   // of this cycle, unless one of them converges
      launch_step(0);
      int i = 0;
      for (; i < steps; ++i, ++j) {
        if (i + 1 < steps) launch_step(i + 1);
        MH_HIP(hipEventSynchronize(h->column_ready[i & 1]));
        const double* col = h->column_host[i & 1];
        for (int k = 0; k <= i; ++k) Hat(k, i) = col[k];
        Hat(i + 1, i) = std::sqrt(col[i + 1]);
        // Givens rotations (GMRESSolver: ApplyPlaneRotation / GeneratePlaneRotation)
        for (int k = 0; k < i; ++k) {
          const double t = cs[k] * Hat(k, i) + sn[k] * Hat(k + 1, i);
          Hat(k + 1, i) = -sn[k] * Hat(k, i) + cs[k] * Hat(k + 1, i);
          Hat(k, i) = t;
        }
        {
          const double dx = Hat(i, i), dy = Hat(i + 1, i);
          if (dy == 0.0) {
            cs[i] = 1.0;
            sn[i] = 0.0;
          } else if (std::fabs(dy) > std::fabs(dx)) {
            const double t = dx / dy;
            sn[i] = 1.0 / std::sqrt(1.0 + t * t);
            cs[i] = t * sn[i];
          } else {
            const double t = dy / dx;
            cs[i] = 1.0 / std::sqrt(1.0 + t * t);
            sn[i] = t * cs[i];
          }
          Hat(i, i) = cs[i] * dx + sn[i] * dy;
          Hat(i + 1, i) = 0.0;
          sv[i + 1] = -sn[i] * sv[i];
          sv[i] = cs[i] * sv[i];
        }
        const double resid = std::fabs(sv[i + 1]);
        if (resid <= goal) {
          update(i + 1);
          finish(j, resid, true);
          return;
        }
      }
      update(i);
      // r = M (b - A x)
      spmv(h, mA.dev, mx.dev, mb.dev, dinv, r);
      beta = norm_of(r);
      if (beta <= goal) {
        finish(j - 1, beta, true);
        return;
      }
    }
    finish(max_iter, beta, false);
  });
}

/* mfem::CGSolver + mfem::DSmoother as operators::NonlinearSolid configures its mass solve (operators/nonlinear_solid.cpp:
 * 39-50, .hpp:38-42: rel 1e-8, abs 1e-12, 1000 iterations, iterative_mode false): preconditioned conjugate gradients,
 * stops when (r, M r) <= max(rel_tol^2 (r0, M r0), abs_tol^2)  (mfem linalg/solvers.cpp, CGSolver::Mult) */
int mimi_hip_linear_cg(mimi_hip_linear_t h, const double* A_values, const double* b, double* x, double rel_tol, double abs_tol,
                       int max_iter, int use_jacobi, int32_t* iterations, double* final_norm, int32_t* converged) {
  return guarded_k([&] {
    if (!h || !A_values || !b || !x) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    const int64_t n = h->n;
    hipStream_t s = h->stream;
    Mirror<double> mA = Mirror<double>::in(A_values, (size_t)h->nnz, h->stage_val, s);
    Mirror<double> mb = Mirror<double>::in(b, (size_t)n, h->stage_b, s);
    Mirror<double> mx = Mirror<double>::inout(x, (size_t)n, h->stage_x, s);
    h->V.resize((size_t)3 * n);            // d, z, q
    h->r.resize((size_t)n);
    h->partials.resize((size_t)2 * KR_BLOCKS);
    h->totals.resize(2);
    const double* dinv = nullptr;
    if (use_jacobi) {
      h->dinv.resize((size_t)n);
      hipLaunchKernelGGL(kr_diag_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, h->rowptr, h->diag_pos.ptr, mA.dev,
                         h->dinv.ptr);
      dinv = h->dinv.ptr;
    }
    double* d = h->V.ptr;
    double* z = d + n;
    double* q = z + n;
    double* r = h->r.ptr;
    double* part = h->partials.ptr;
    auto total_of = [&]() -> double {
      hipLaunchKernelGGL(kr_totals_kernel, dim3(1), dim3(KR_THREADS), 0, s, part, h->totals.ptr);
      MH_HIP(hipGetLastError());
      double t = 0.0;
      MH_HIP(hipMemcpyAsync(&t, h->totals.ptr, sizeof(double), hipMemcpyDeviceToHost, s));
      MH_HIP(hipStreamSynchronize(s));
      return t;
    };
    auto finish = [&](int it, double nom, bool conv) {
      if (iterations) *iterations = it;
      if (final_norm) *final_norm = std::sqrt(std::fabs(nom));
      if (converged) *converged = conv ? 1 : 0;
      mx.finish(s);
      MH_HIP(hipStreamSynchronize(s));
    };
    // x = 0, r = b, z = M r, d = z
    MH_HIP(hipMemsetAsync(mx.dev, 0, sizeof(double) * n, s));
    MH_HIP(hipMemcpyAsync(r, mb.dev, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    hipLaunchKernelGGL(kr_cg_precond_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, r, dinv, z, part);
    MH_HIP(hipMemcpyAsync(d, z, sizeof(double) * n, hipMemcpyDeviceToDevice, s));
    double nom = total_of();
    const double r0 = std::fmax(nom * rel_tol * rel_tol, abs_tol * abs_tol);
    if (nom <= r0) {
      finish(0, nom, true);
      return;
    }
    spmv(h, mA.dev, d, nullptr, nullptr, q);
    hipLaunchKernelGGL(kr_mgs_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, q, (const double*)nullptr, (const double*)nullptr,
                       (const double*)d, part);
    double den = total_of();
    if (!(den > 0.0)) {   // not positive definite: mfem warns and stops
      finish(0, nom, false);
      return;
    }
    int it = 1;
    for (;; ++it) {
      const double alpha = nom / den;
      hipLaunchKernelGGL(kr_cg_update_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, alpha, d, q, mx.dev, r);
      hipLaunchKernelGGL(kr_cg_precond_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, r, dinv, z, part);
      const double betanom = total_of();
      if (betanom <= r0) {
        finish(it, betanom, true);
        return;
      }
      if (it >= max_iter) {
        finish(it, betanom, false);
        return;
      }
      const double beta = betanom / nom;
      hipLaunchKernelGGL(kr_cg_direction_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, beta, z, d);
      spmv(h, mA.dev, d, nullptr, nullptr, q);
      hipLaunchKernelGGL(kr_mgs_kernel, dim3(KR_BLOCKS), dim3(KR_THREADS), 0, s, n, q, (const double*)nullptr, (const double*)nullptr,
                         (const double*)d, part);
      den = total_of();
      if (!(den > 0.0)) {
        finish(it, betanom, false);
        return;
      }
      nom = betanom;
    }
  });
}

}  // extern "C"
