// Setup kernels for a tensor-product B-spline patch:
//  * per-point geometry  dxi/dX and w*det  (IsoparametricTransformation::Jacobian/Weight and
//    CalcInverse of utils/precomputed.cpp:302,320)
//  * expansion to the reference's per-point dN_dX tables (precomputed.cpp:316-321) for the
//    general kernels
//  * structured CSR pattern (PrecomputedData::PrepareSparsity, precomputed.cpp:151-174)
#pragma once

#include <hip/hip_runtime.h>

#include "materials.hpp"

namespace mimi_hip {

struct PatchDev {
  int dim;
  int p[3], nq[3], n_ctrl[3];
  int box_begin[3], box_n[3];       // element (span) box integrated by this handle
  const double* B[3];               // [n_spans][p+1][nq]
  const double* D[3];
  const double* W[3];               // [nq]
  const int32_t* first[3];          // [n_spans] first non-zero basis of the span
  const double* ctrl;               // [n_nodes][dim] lexicographic
  const int64_t* node_ids;          // lexicographic -> global or nullptr
  int n_dof, n_q, n_el;
};

MH_DEV void split3(int idx, const int* n, int* out) {
  out[0] = idx % n[0];
  idx /= n[0];
  out[1] = idx % n[1];
  out[2] = idx / n[1];
}

// geo[e][k][q]: k = d*DIM + J -> dxi_d/dX_J ; k = DIM*DIM -> w*det
template<int DIM>
__global__ void geometry_kernel(PatchDev P, double* __restrict__ geo, int* __restrict__ status) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)P.n_el * P.n_q) return;
  const int q = idx % P.n_q;
  const int e = idx / P.n_q;
  int el[3] = {0, 0, 0}, qi[3] = {0, 0, 0};
  int bn[3] = {P.box_n[0], P.box_n[1], DIM == 3 ? P.box_n[2] : 1};
  int nq[3] = {P.nq[0], P.nq[1], DIM == 3 ? P.nq[2] : 1};
  split3(e, bn, el);
  split3(q, nq, qi);
  int span[3], first[3];
  for (int d = 0; d < DIM; ++d) {
    span[d] = P.box_begin[d] + el[d];
    first[d] = P.first[d][span[d]];
  }
  double Jm[DIM * DIM];  // Jm(I,d) column-major: dX_I/dxi_d
  for (int k = 0; k < DIM * DIM; ++k) Jm[k] = 0.0;
  const int na2 = DIM == 3 ? P.p[2] + 1 : 1;
  for (int a2 = 0; a2 < na2; ++a2)
    for (int a1 = 0; a1 <= P.p[1]; ++a1)
      for (int a0 = 0; a0 <= P.p[0]; ++a0) {
        const int al[3] = {a0, a1, a2};
        double b[3] = {1, 1, 1}, dd[3] = {0, 0, 0};
        int64_t node = 0, stride = 1;
        for (int d = 0; d < DIM; ++d) {
          const size_t o = ((size_t)span[d] * (P.p[d] + 1) + al[d]) * P.nq[d] + qi[d];
          b[d] = P.B[d][o];
          dd[d] = P.D[d][o];
          node += (int64_t)(first[d] + al[d]) * stride;
          stride *= P.n_ctrl[d];
        }
        double dN[3];
        dN[0] = dd[0] * b[1] * b[2];
        dN[1] = b[0] * dd[1] * b[2];
        dN[2] = b[0] * b[1] * dd[2];
        for (int I = 0; I < DIM; ++I) {
          const double X = P.ctrl[node * DIM + I];
          for (int d = 0; d < DIM; ++d) MH_M(Jm, I, d) += X * dN[d];
        }
      }
  const double det = det_of<DIM>(Jm);
  if (!(det > 0.0)) atomicOr(status, 8);  // inverted / degenerate geometry map
  double Ji[DIM * DIM];                   // Ji(d,J) = dxi_d/dX_J
  inverse_of<DIM>(Jm, det, Ji);
  double w = 1.0;
  for (int d = 0; d < DIM; ++d) w *= P.W[d][qi[d]];
  double* g = geo + (int64_t)e * (DIM * DIM + 1) * P.n_q + q;
  for (int d = 0; d < DIM; ++d)
    for (int J = 0; J < DIM; ++J) g[(int64_t)(d * DIM + J) * P.n_q] = MH_M(Ji, d, J);
  g[(int64_t)(DIM * DIM) * P.n_q] = w * det;
}

// element connectivity + reference-layout tables for the general kernels
template<int DIM>
__global__ void expand_tables_kernel(PatchDev P, const double* __restrict__ geo, int32_t* __restrict__ dofs,
                                     double* __restrict__ dN_dX, double* __restrict__ wdet) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)P.n_el * P.n_q * P.n_dof;
  if (idx >= total) return;
  const int a = idx % P.n_dof;
  const int q = (idx / P.n_dof) % P.n_q;
  const int e = idx / ((int64_t)P.n_dof * P.n_q);
  int el[3] = {0, 0, 0}, qi[3] = {0, 0, 0}, al[3] = {0, 0, 0};
  int bn[3] = {P.box_n[0], P.box_n[1], DIM == 3 ? P.box_n[2] : 1};
  int nq[3] = {P.nq[0], P.nq[1], DIM == 3 ? P.nq[2] : 1};
  int np[3] = {P.p[0] + 1, P.p[1] + 1, DIM == 3 ? P.p[2] + 1 : 1};
  split3(e, bn, el);
  split3(q, nq, qi);
  split3(a, np, al);
  double b[3] = {1, 1, 1}, dd[3] = {0, 0, 0};
  int64_t node = 0, stride = 1;
  for (int d = 0; d < DIM; ++d) {
    const int span = P.box_begin[d] + el[d];
    const size_t o = ((size_t)span * (P.p[d] + 1) + al[d]) * P.nq[d] + qi[d];
    b[d] = P.B[d][o];
    dd[d] = P.D[d][o];
    node += (int64_t)(P.first[d][span] + al[d]) * stride;
    stride *= P.n_ctrl[d];
  }
  double dN[3];
  dN[0] = dd[0] * b[1] * b[2];
  dN[1] = b[0] * dd[1] * b[2];
  dN[2] = b[0] * b[1] * dd[2];
  const double* g = geo + (int64_t)e * (DIM * DIM + 1) * P.n_q + q;
  for (int J = 0; J < DIM; ++J) {
    double s = 0;
    for (int d = 0; d < DIM; ++d) s += dN[d] * g[(int64_t)(d * DIM + J) * P.n_q];
    dN_dX[(((int64_t)e * P.n_q + q) * DIM + J) * P.n_dof + a] = s;
  }
  if (a == 0) wdet[(int64_t)e * P.n_q + q] = g[(int64_t)(DIM * DIM) * P.n_q];
  if (q == 0) dofs[(int64_t)e * P.n_dof + a] = (int32_t)(P.node_ids ? P.node_ids[node] : node);
}

// element connectivity only (ElementData::dofs, precomputed.cpp:83)
__global__ void connectivity_kernel(PatchDev P, int32_t* __restrict__ dofs) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)P.n_el * P.n_dof) return;
  const int a = idx % P.n_dof;
  const int e = idx / P.n_dof;
  int el[3], al[3];
  int bn[3] = {P.box_n[0], P.box_n[1], P.dim == 3 ? P.box_n[2] : 1};
  int np[3] = {P.p[0] + 1, P.p[1] + 1, P.dim == 3 ? P.p[2] + 1 : 1};
  split3(e, bn, el);
  split3(a, np, al);
  int64_t node = 0, stride = 1;
  for (int d = 0; d < P.dim; ++d) {
    const int span = P.box_begin[d] + el[d];
    node += (int64_t)(P.first[d][span] + al[d]) * stride;
    stride *= P.n_ctrl[d];
  }
  dofs[idx] = (int32_t)(P.node_ids ? P.node_ids[node] : node);
}

// ---- structured sparsity ------------------------------------------------------------
struct SparsityDev {
  int dim;
  int n[3], p[3];
  const int64_t* prefix[3];  // [n_d + 1] exclusive prefix sums of the 1-D stencil widths
  // row-sliced patterns (mimi_hip_bspline_sparsity_rows): rows of nodes outside [req_lo, req_hi) may be absent
  // (zero length); partial == 0: every row must be there
  int partial;
  int req_lo[3], req_hi[3];
};

MH_DEV int width_1d(int A, int n, int p) {
  const int lo = A - p < 0 ? 0 : A - p;
  const int hi = A + p > n - 1 ? n - 1 : A + p;
  return hi - lo + 1;
}

// rowptr[(A*dim + i)] in closed form; one thread per node
__global__ void structured_rowptr_kernel(SparsityDev S, int64_t n_nodes, int64_t* __restrict__ rowptr) {
  const int64_t A = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (A > n_nodes) return;
  const int dim = S.dim;
  const int64_t W0 = S.prefix[0][S.n[0]], W1 = S.prefix[1][S.n[1]];
  if (A == n_nodes) {
    const int64_t W2 = dim == 3 ? S.prefix[2][S.n[2]] : 1;
    rowptr[n_nodes * dim] = (int64_t)dim * dim * W0 * W1 * W2;
    return;
  }
  const int A0 = A % S.n[0];
  const int A1 = (A / S.n[0]) % S.n[1];
  const int A2 = A / ((int64_t)S.n[0] * S.n[1]);
  const int64_t w0 = width_1d(A0, S.n[0], S.p[0]);
  const int64_t w1 = width_1d(A1, S.n[1], S.p[1]);
  const int64_t w2 = dim == 3 ? width_1d(A2, S.n[2], S.p[2]) : 1;
  const int64_t P2 = dim == 3 ? S.prefix[2][A2] : 0;
  const int64_t before = W0 * W1 * P2 + W0 * S.prefix[1][A1] * w2 + S.prefix[0][A0] * w1 * w2;
  const int64_t len = dim * w0 * w1 * w2;
  for (int i = 0; i < dim; ++i) rowptr[A * dim + i] = (int64_t)dim * dim * before + i * len;
}

// one wave per row; lanes stride over the row's entries
__global__ void structured_col_kernel(SparsityDev S, int64_t n_rows, const int64_t* __restrict__ rowptr,
                                      int32_t* __restrict__ col, int check_only, int* __restrict__ mismatch) {
  const int64_t row = (int64_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
  const int lane = threadIdx.x & 63;
  if (row >= n_rows) return;
  const int dim = S.dim;
  const int64_t A = row / dim;
  const int Am[3] = {(int)(A % S.n[0]), (int)((A / S.n[0]) % S.n[1]), (int)(A / ((int64_t)S.n[0] * S.n[1]))};
  int lo[3] = {0, 0, 0}, w[3] = {1, 1, 1};
  for (int d = 0; d < dim; ++d) {
    lo[d] = Am[d] - S.p[d] < 0 ? 0 : Am[d] - S.p[d];
    w[d] = width_1d(Am[d], S.n[d], S.p[d]);
  }
  const int64_t start = rowptr[row];
  const int len = dim * w[0] * w[1] * w[2];
  const int64_t have = rowptr[row + 1] - start;
  if (have == 0) {
    // a row this (row-sliced) pattern does not hold: fine unless the handle's elements touch it
    bool required = check_only && !S.partial;
    if (check_only && S.partial) {
      required = true;
      for (int d = 0; d < dim; ++d) required = required && Am[d] >= S.req_lo[d] && Am[d] < S.req_hi[d];
    }
    if (required && lane == 0) atomicOr(mismatch, 1);
    return;
  }
  if (check_only && have != len) {
    if (lane == 0) atomicOr(mismatch, 1);
    return;
  }
  for (int k = lane; k < len; k += 64) {
    const int j = k % dim;
    int nb = k / dim;
    const int t0 = nb % w[0];
    nb /= w[0];
    const int t1 = nb % w[1];
    const int t2 = nb / w[1];
    const int64_t B = (lo[0] + t0) + (int64_t)S.n[0] * ((lo[1] + t1) + (int64_t)S.n[1] * (lo[2] + t2));
    const int32_t c = (int32_t)(B * dim + j);
    if (check_only) {
      if (col[start + k] != c) atomicOr(mismatch, 1);
    } else {
      col[start + k] = c;
    }
  }
}

// Permuted numbering (node_ids = lexicographic -> caller's node id, e.g. MFEM's NURBS dof map): the CSR row of
// node perm[A] holds the columns perm[B] of A's (2p+1)^3 window in ascending order of perm[B].  One wave per
// lexicographic node A: nbr_pos[A][t] = rank of perm[B_t] inside the window (t = lexicographic window index),
// and a check that the caller's CSR is exactly that pattern.  3-D; RankT / WMAX: uint8 ranks in windows of <= 125 = 5^3
// entries (p <= 2), uint16 ranks in windows of <= 343 = 7^3 (p = 3).
template<typename RankT, int WMAX>
__global__ void permuted_window_kernel(SparsityDev S, int64_t n_nodes, const int64_t* __restrict__ perm,
                                       const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                       RankT* __restrict__ nbr_pos, int* __restrict__ mismatch) {
  __shared__ int64_t vals_all[4][(WMAX + 7) / 8 * 8];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t A = (int64_t)blockIdx.x * 4 + wave;
  if (A >= n_nodes) return;
  int64_t* vals = vals_all[wave];
  const int Am[3] = {(int)(A % S.n[0]), (int)((A / S.n[0]) % S.n[1]), (int)(A / ((int64_t)S.n[0] * S.n[1]))};
  int lo[3], w[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = Am[d] - S.p[d] < 0 ? 0 : Am[d] - S.p[d];
    w[d] = width_1d(Am[d], S.n[d], S.p[d]);
  }
  const int nw = w[0] * w[1] * w[2];
  for (int t = lane; t < nw; t += 64) {
    const int t0 = t % w[0], t1 = (t / w[0]) % w[1], t2 = t / (w[0] * w[1]);
    const int64_t B = (lo[0] + t0) + (int64_t)S.n[0] * ((lo[1] + t1) + (int64_t)S.n[1] * (lo[2] + t2));
    vals[t] = perm[B];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int64_t gA = perm[A];
  for (int t = lane; t < nw; t += 64) {
    const int64_t v = vals[t];
    int rank = 0;
    for (int s2 = 0; s2 < nw; ++s2) rank += vals[s2] < v ? 1 : 0;
    nbr_pos[A * WMAX + t] = (RankT)rank;
    for (int i = 0; i < 3; ++i) {
      const int64_t row = gA * 3 + i;
      const int64_t start = rowptr[row];
      if (rowptr[row + 1] - start != 3 * nw) {
        atomicOr(mismatch, 1);
      } else {
        for (int j = 0; j < 3; ++j)
          if (col[start + 3 * rank + j] != (int32_t)(v * 3 + j)) atomicOr(mismatch, 1);
      }
    }
  }
}

// Columns of the permuted structured pattern from the window ranks (the inverse of permuted_window_kernel's check):
// one wave per lexicographic node A, col[rowptr[perm[A] 3 + i] + 3 rank(t) + j] = perm[B_t] 3 + j.
template<typename RankT, int WMAX>
__global__ void permuted_col_kernel(SparsityDev S, int64_t n_nodes, const int64_t* __restrict__ perm,
                                    const int64_t* __restrict__ rowptr, const RankT* __restrict__ nbr_pos,
                                    int32_t* __restrict__ col) {
  const int lane = threadIdx.x & 63;
  const int64_t A = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (A >= n_nodes) return;
  const int Am[3] = {(int)(A % S.n[0]), (int)((A / S.n[0]) % S.n[1]), (int)(A / ((int64_t)S.n[0] * S.n[1]))};
  int lo[3], w[3];
  for (int d = 0; d < 3; ++d) {
    lo[d] = Am[d] - S.p[d] < 0 ? 0 : Am[d] - S.p[d];
    w[d] = width_1d(Am[d], S.n[d], S.p[d]);
  }
  const int nw = w[0] * w[1] * w[2];
  const int64_t gA = perm[A];
  for (int t = lane; t < nw; t += 64) {
    const int t0 = t % w[0], t1 = (t / w[0]) % w[1], t2 = t / (w[0] * w[1]);
    const int64_t B = (lo[0] + t0) + (int64_t)S.n[0] * ((lo[1] + t1) + (int64_t)S.n[1] * (lo[2] + t2));
    const int64_t v = perm[B];
    const int rank = nbr_pos[A * WMAX + t];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) col[rowptr[gA * 3 + i] + 3 * rank + j] = (int32_t)(v * 3 + j);
  }
}

}  // namespace mimi_hip
