// Tensor-product (sum-factorised) element kernel for SMALL elements: 2-D patches of degree 1..3 and 3-D patches of degree 1
// (4..16 nodes, 9..27 Gauss points) -- what the reference's own examples and solver tests are (2-D).  Same quantities as
// kernels_tensor.hpp (integrators/nonlinear_solid.hpp:65-87, nonlinear_solid.cpp:48-149), from the 1-D tables and the
// per-point inverse geometry Jacobian instead of the reference's flat dN/dX tables (6.4 KB -> 1 KB per 2-D p = 3 element).
//
// One WAVE per element, four elements per workgroup, everything staged in wave-private LDS:
//   lane = quadrature point   F, material, Phat_i[m] and (tangent) Ahat_i[m][j][n] -> LDS
//   lanes over (a, i)         element residual  R_i[a] = sum_q sum_m dN_a/dxi_m Phat_i[m]
//   per (i, j) block          K[(a, i), (b, j)] = sum_q sum_mn dN_a/dxi_m Ahat_i[m][j][n] dN_b/dxi_n, one parametric
//                             direction at a time (lanes over the outputs of each stage)
// The element block and the element residual vector go out densely (scratch_k[e][(a, i)][(j, b)], scratch_r[e][i][a]) and
// general_gather_kernel (kernels_general.hpp: one wave per CSR row, fixed summation order) adds them into r / A: no atomics.
// MODE 0: residual; 1: residual + tangent; 2: DomainPostTimeAdvance (state commit, nonlinear_solid.cpp:179-199).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor.hpp"

namespace mimi_hip {

template<int DIM, int P>
struct SmallShape {
  static constexpr int NB = P + 1, NQ = P + 2;
  static constexpr int ND = DIM == 2 ? NB * NB : NB * NB * NB;
  static constexpr int NPT = DIM == 2 ? NQ * NQ : NQ * NQ * NQ;
  static constexpr int NT = ND * DIM, DD = DIM * DIM, D4 = DD * DD;
  static constexpr int NB2 = NB * NB;
  // wave-private LDS carve, in doubles
  static constexpr int off_ue = 0;                                // [DIM][ND]
  static constexpr int off_tab = off_ue + NT;                     // [DIM dir][B, D][NB][NQ]
  static constexpr int off_ph = off_tab + DIM * 2 * NB * NQ;      // [DD (i, m)][NPT]
  static constexpr int off_ah = off_ph + DD * NPT;                // [D4 (i, m, j, n)][NPT]            (MODE 1)
  static constexpr int c1 = DD * NB2 * (DIM == 2 ? NQ : NQ * NQ); // first stage of one (i, j) block
  static constexpr int c2 = DIM == 3 ? DD * NB2 * NB2 * NQ : 0;   // second stage (3-D)
  static constexpr int off_c1 = off_ah + D4 * NPT;
  static constexpr int off_c2 = off_c1 + c1;
  static constexpr int total1 = off_c2 + c2;                      // MODE 1
  static constexpr int total0 = off_ah;                           // MODE 0 / 2
};

// dN_a/dxi_m at point q from the 1-D tables (tab = [dir][B, D][NB][NQ])
template<int DIM, int P>
MH_DEV double small_grad(const double* tab, int a, int q, int m) {
  using S = SmallShape<DIM, P>;
  double v = 1.0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const int ad = d == 0 ? a % S::NB : (d == 1 ? (a / S::NB) % S::NB : a / S::NB2);
    const int qd = d == 0 ? q % S::NQ : (d == 1 ? (q / S::NQ) % S::NQ : q / (S::NQ * S::NQ));
    v *= tab[((d * 2 + (d == m ? 1 : 0)) * S::NB + ad) * S::NQ + qd];
  }
  return v;
}

template<int DIM, int P, int FAMILY, int MODE>
__global__ __launch_bounds__(256) void tensor_small_kernel(TensorArgs p, int n_el) {
  using S = SmallShape<DIM, P>;
  constexpr int NB = S::NB, NQ = S::NQ, ND = S::ND, NPT = S::NPT, NT = S::NT, DD = S::DD, D4 = S::D4, NB2 = S::NB2;
  static_assert(NPT <= 64 && NT <= 64, "one wave per element: at most 64 quadrature points and element dofs");
  extern __shared__ __align__(16) double smem_small[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= n_el) return;   // (the whole wave leaves; only wave barriers below)
  double* lds = smem_small + wave * (MODE == 1 ? S::total1 : S::total0);
  double* ue = lds + S::off_ue;
  double* tab = lds + S::off_tab;
  double* PH = lds + S::off_ph;
  double* AH = lds + S::off_ah;
  int el[3] = {0, 0, 0};
  el[0] = (int)(e % p.box_n[0]);
  el[1] = (int)((e / p.box_n[0]) % p.box_n[1]);
  if (DIM == 3) el[2] = (int)(e / ((int64_t)p.box_n[0] * p.box_n[1]));
  if (lane < ND) {
    const int64_t node = p.dofs[e * ND + lane];
#pragma unroll
    for (int c = 0; c < DIM; ++c) ue[c * ND + lane] = p.u[node * DIM + c];
  }
  for (int t = lane; t < DIM * 2 * NB * NQ; t += 64) {
    const int dir = t / (2 * NB * NQ), rem = t % (2 * NB * NQ), isD = rem / (NB * NQ), k = rem % (NB * NQ);
    const int span = p.box_begin[dir] + el[dir];
    tab[t] = ((isD ? p.tabD[dir] : p.tabB[dir]) + (int64_t)span * NB * NQ)[k];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();

  // ---- quadrature-point stage: lane = point ------------------------------------------------------------------------
  int status = 0;
  if (lane < NPT) {
    double H[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) H[k] = 0.0;
    for (int a = 0; a < ND; ++a) {
      double g[DIM];
#pragma unroll
      for (int m = 0; m < DIM; ++m) g[m] = small_grad<DIM, P>(tab, a, lane, m);
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double uu = ue[i * ND + a];
#pragma unroll
        for (int m = 0; m < DIM; ++m) H[i * DIM + m] += uu * g[m];
      }
    }
    const double* gq = p.geo + e * (int64_t)((DD + 1) * NPT) + lane;
    double Ji[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) Ji[k] = gq[(int64_t)k * NPT];       // dxi_m / dX_J at (m DIM + J)
    const double wd = gq[(int64_t)DD * NPT];
    double F[DD];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int J = 0; J < DIM; ++J) {
        double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < DIM; ++m) sf += H[i * DIM + m] * Ji[m * DIM + J];
        F[i + J * DIM] = sf;
      }
    const int64_t pt = e * NPT + lane;
    if constexpr (MODE == 2) {
      if constexpr (FAMILY != 0) status |= accumulate_other<DIM, (FAMILY >= 2 ? FAMILY : -1)>(p.mat, p.dt, p.state, pt, F);
      else status |= accumulate_state<DIM>(p.mat, p.dt, p.state, pt, F);
    } else {
      double Pk[DD], A[MODE == 1 ? D4 : 1];
      if constexpr (FAMILY != 0) {     // (2..5: that material as a compile-time constant, as in the other kernel families)
        status |= evaluate_other<DIM, (FAMILY >= 2 ? FAMILY : -1)>(p.mat, p.dt, p.state, pt, F, Pk, MODE == 1 ? A : nullptr, 1.0);
      } else {
        PointResult<DIM> w;
        status |= evaluate_pk1<DIM>(p.mat, p.dt, p.state, pt, F, w);
#pragma unroll
        for (int k = 0; k < DD; ++k) Pk[k] = w.P[k];
        if constexpr (MODE == 1) tangent_of<DIM>(p.mat.m, w, A);
      }
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int m = 0; m < DIM; ++m) {
          double t = 0.0;
#pragma unroll
          for (int J = 0; J < DIM; ++J) t += Pk[i + J * DIM] * Ji[m * DIM + J];
          PH[(i * DIM + m) * NPT + lane] = wd * t;
        }
      if constexpr (MODE == 1) {
        // Ahat_i[m][j][n] = wd sum_JL Jinv[m][J] A_iJjL Jinv[n][L]
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) {
            double B[DD];
#pragma unroll
            for (int m = 0; m < DIM; ++m)
#pragma unroll
              for (int L = 0; L < DIM; ++L) {
                double t = 0.0;
#pragma unroll
                for (int J = 0; J < DIM; ++J) t += Ji[m * DIM + J] * A[((i * DIM + J) * DIM + j) * DIM + L];
                B[m * DIM + L] = t;
              }
#pragma unroll
            for (int m = 0; m < DIM; ++m)
#pragma unroll
              for (int n = 0; n < DIM; ++n) {
                double t = 0.0;
#pragma unroll
                for (int L = 0; L < DIM; ++L) t += B[m * DIM + L] * Ji[n * DIM + L];
                AH[(((i * DIM + m) * DIM + j) * DIM + n) * NPT + lane] = wd * t;
              }
          }
      }
    }
  }
  if (status) atomicOr(p.status, status);
  if constexpr (MODE == 2) return;
  __builtin_amdgcn_wave_barrier();

  // ---- element residual: lane = (i, a) ------------------------------------------------------------------------------
  if (lane < NT) {
    const int i = lane / ND, a = lane % ND;
    double sr = 0.0;
    for (int q = 0; q < NPT; ++q)
#pragma unroll
      for (int m = 0; m < DIM; ++m) sr += small_grad<DIM, P>(tab, a, q, m) * PH[(i * DIM + m) * NPT + q];
    p.scratch_r[e * NT + lane] = sr;       // [i][a], as the general kernels
  }
  if constexpr (MODE == 0) return;

  // ---- element tangent, block by block --------------------------------------------------------------------------------
  double* C1 = lds + S::off_c1;
  double* C2 = lds + S::off_c2;
  double* Ke = p.scratch_k + e * (int64_t)(NT * NT);
  const double* tB[3] = {tab, tab + 2 * NB * NQ, tab + (DIM == 3 ? 4 : 2) * NB * NQ};
  auto T = [&](int dir, int isD, int a, int q) { return tB[dir][(isD * NB + a) * NQ + q]; };
  for (int ij = 0; ij < DD; ++ij) {
    const int i = ij / DIM, j = ij % DIM;
    __builtin_amdgcn_wave_barrier();
    if constexpr (DIM == 2) {
      // C1[mn][(a1 b1)][q0] = sum_q1 T1^m_a1 T1^n_b1 Ahat_mn[q0 + NQ q1]
      for (int t = lane; t < DD * NB2 * NQ; t += 64) {
        const int q0 = t % NQ, pr = (t / NQ) % NB2, mn = t / (NQ * NB2), m = mn / DIM, n = mn % DIM;
        const int a1 = pr / NB, b1 = pr % NB;
        const double* ah = AH + (((i * DIM + m) * DIM + j) * DIM + n) * NPT + q0;
        double s = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < NQ; ++q1) s += T(1, m == 1, a1, q1) * T(1, n == 1, b1, q1) * ah[NQ * q1];
        C1[t] = s;
      }
      __builtin_amdgcn_wave_barrier();
      // K[(a0, a1), (b0, b1)] = sum_mn sum_q0 T0^m_a0 T0^n_b0 C1[mn][(a1 b1)][q0]
      for (int t = lane; t < ND * ND; t += 64) {
        const int a = t / ND, b = t % ND, a0 = a % NB, a1 = a / NB, b0 = b % NB, b1 = b / NB;
        double s = 0.0;
#pragma unroll
        for (int mn = 0; mn < DD; ++mn) {
          const int m = mn / DIM, n = mn % DIM;
          const double* c = C1 + (mn * NB2 + a1 * NB + b1) * NQ;
#pragma unroll
          for (int q0 = 0; q0 < NQ; ++q0) s += T(0, m == 0, a0, q0) * T(0, n == 0, b0, q0) * c[q0];
        }
        Ke[(a * DIM + i) * NT + j * ND + b] = s;
      }
    } else {
      // C1[mn][(a2 b2)][q01] = sum_q2 T2^m_a2 T2^n_b2 Ahat_mn[q01 + NQ^2 q2]
      for (int t = lane; t < DD * NB2 * NQ * NQ; t += 64) {
        const int q01 = t % (NQ * NQ), pr = (t / (NQ * NQ)) % NB2, mn = t / (NQ * NQ * NB2), m = mn / DIM, n = mn % DIM;
        const int a2 = pr / NB, b2 = pr % NB;
        const double* ah = AH + (((i * DIM + m) * DIM + j) * DIM + n) * NPT + q01;
        double s = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) s += T(2, m == 2, a2, q2) * T(2, n == 2, b2, q2) * ah[NQ * NQ * q2];
        C1[t] = s;
      }
      __builtin_amdgcn_wave_barrier();
      // C2[mn][(a1 b1)][(a2 b2)][q0] = sum_q1 T1^m_a1 T1^n_b1 C1[mn][(a2 b2)][q0 + NQ q1]
      for (int t = lane; t < DD * NB2 * NB2 * NQ; t += 64) {
        const int q0 = t % NQ, pr2 = (t / NQ) % NB2, pr1 = (t / (NQ * NB2)) % NB2, mn = t / (NQ * NB2 * NB2);
        const int m = mn / DIM, n = mn % DIM, a1 = pr1 / NB, b1 = pr1 % NB;
        const double* c = C1 + (mn * NB2 + pr2) * NQ * NQ + q0;
        double s = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < NQ; ++q1) s += T(1, m == 1, a1, q1) * T(1, n == 1, b1, q1) * c[NQ * q1];
        C2[t] = s;
      }
      __builtin_amdgcn_wave_barrier();
      // K[a, b] = sum_mn sum_q0 T0^m_a0 T0^n_b0 C2[mn][(a1 b1)][(a2 b2)][q0]
      for (int t = lane; t < ND * ND; t += 64) {
        const int a = t / ND, b = t % ND;
        const int a0 = a % NB, a1 = (a / NB) % NB, a2 = a / NB2, b0 = b % NB, b1 = (b / NB) % NB, b2 = b / NB2;
        double s = 0.0;
#pragma unroll
        for (int mn = 0; mn < DD; ++mn) {
          const int m = mn / DIM, n = mn % DIM;
          const double* c = C2 + ((mn * NB2 + a1 * NB + b1) * NB2 + a2 * NB + b2) * NQ;
#pragma unroll
          for (int q0 = 0; q0 < NQ; ++q0) s += T(0, m == 0, a0, q0) * T(0, n == 0, b0, q0) * c[q0];
        }
        Ke[(a * DIM + i) * NT + j * ND + b] = s;
      }
    }
  }
}

// shapes with this kernel: 2-D degree 1..3, 3-D degree 1
inline bool tensor_small_shape(int dim, const int* degree, int nq) {
  const int p = degree[0];
  for (int d = 1; d < dim; ++d)
    if (degree[d] != p) return false;
  if (nq != p + 2) return false;
  return (dim == 2 && p >= 1 && p <= 3) || (dim == 3 && p == 1);
}

inline bool tensor_small(const mimi_hip_domain_s* h) { return h->path == 1 && tensor_small_shape(h->dim, h->degree, h->nq1[0]); }

}  // namespace mimi_hip
