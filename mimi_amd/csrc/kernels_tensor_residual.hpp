// Residual-only assembly (AddDomainResidual; the line search of the reference's Newton calls it several times
// per iteration) in the two-phase style: no colouring, no atomics, bitwise reproducible.
//
//   phase 1  tensor_residual_kernel: one wave per element, lane = quadrature point: gather u, F, P(F),
//            then per component i the element residual piece by sum factorisation (through LDS)
//            -> scratch_r[element][i][a].
//   phase 2  tensor_residual_gather_kernel: one wave per node: lane = element of the 3 x 3 x 3
//            neighbourhood, fixed-shape tree sum, r[node, i] += sum.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor_wgs.hpp"

namespace mimi_hip {

struct ResidualLds {
  static constexpr int ND = 27, NB = 3, NQ = 4, NQ3 = 64, NB2 = 9;
  static constexpr int off_ue = 0;                      // [3][27] (+1)
  static constexpr int off_tab = off_ue + 3 * ND + 1;   // [3][2][3][4]
  static constexpr int off_r = off_tab + 6 * NB * NQ;   // stage-R scratch
  static constexpr int r_size = 3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ;
  static constexpr int per_wave = off_r + r_size;       // 598 doubles
};

template<int KIND>
__global__ __launch_bounds__(256) void tensor_residual_kernel(TensorArgs p, int n_el) {
  using L = ResidualLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64;
  __shared__ double lds_all[4][L::per_wave];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= n_el) return;
  double* ue = lds_all[wave] + L::off_ue;
  double* tab = lds_all[wave] + L::off_tab;
  double* RS = lds_all[wave] + L::off_r;
  int el[3];
  el[0] = e % p.box_n[0];
  el[1] = (e / p.box_n[0]) % p.box_n[1];
  el[2] = e / ((int64_t)p.box_n[0] * p.box_n[1]);
  if (lane < ND) {
    const int64_t node = p.dofs[e * ND + lane];
#pragma unroll
    for (int c = 0; c < 3; ++c) ue[c * ND + lane] = p.u[node * 3 + c];
  }
  for (int t = lane; t < 6 * NB * NQ; t += 64) {
    const int dir = t / (2 * NB * NQ), rem = t % (2 * NB * NQ), isD = rem / (NB * NQ), k = rem % (NB * NQ);
    const int span = p.box_begin[dir] + el[dir];
    tab[t] = ((isD ? p.tabD[dir] : p.tabB[dir]) + (int64_t)span * NB * NQ)[k];
  }
  double Ji[9];
  const double* g = p.geo + e * 10 * NQ3 + lane;
#pragma unroll
  for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NQ3];
  const double wd = g[(int64_t)9 * NQ3];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // F at the quadrature point of this lane, q = q0 + 4 q1 + 16 q2
  double F[9];
  {
    const int q0 = lane & 3, q1 = (lane >> 2) & 3, q2 = lane >> 4;
    double b0[NB], d0[NB], b1[NB], d1[NB], b2[NB], d2[NB];
#pragma unroll
    for (int a = 0; a < NB; ++a) {
      b0[a] = tab_ptr<P>(tab, 0, 0)[a * NQ + q0];
      d0[a] = tab_ptr<P>(tab, 0, 1)[a * NQ + q0];
      b1[a] = tab_ptr<P>(tab, 1, 0)[a * NQ + q1];
      d1[a] = tab_ptr<P>(tab, 1, 1)[a * NQ + q1];
      b2[a] = tab_ptr<P>(tab, 2, 0)[a * NQ + q2];
      d2[a] = tab_ptr<P>(tab, 2, 1)[a * NQ + q2];
    }
    double H[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) H[k] = 0.0;
#pragma unroll
    for (int a2 = 0; a2 < NB; ++a2)
#pragma unroll
      for (int a1 = 0; a1 < NB; ++a1) {
        const double tbb = b1[a1] * b2[a2], tdb = d1[a1] * b2[a2], tbd = b1[a1] * d2[a2];
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0) {
          const int a = a0 + NB * (a1 + NB * a2);
          const double dn0 = d0[a0] * tbb, dn1 = b0[a0] * tdb, dn2 = b0[a0] * tbd;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const double uu = ue[i * ND + a];
            H[i * 3 + 0] += uu * dn0;
            H[i * 3 + 1] += uu * dn1;
            H[i * 3 + 2] += uu * dn2;
          }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(H[k]) : : "memory");
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * Ji[m * 3 + J];
        F[i + J * 3] = sf;
      }
  }
  PointResult<3> w;
  int status;
  if constexpr (KIND == WGS_KIND_RECORD) {
    status = evaluate_other<3>(p.mat, p.dt, p.state, e * NQ3 + lane, F, w.P, nullptr, 1.0);
  } else {
    MaterialDev mat = p.mat;
    mat.m.kind = KIND;
    status = evaluate_pk1<3>(mat, p.dt, p.state, e * NQ3 + lane, F, w);
  }
  if (status) atomicOr(p.status, status);
  // element residual piece of every component by sum factorisation (as kernels_tensor_2phase.hpp, stage R)
  double* PH = RS;                   // [3 m][64]
  double* V = PH + 3 * NQ3;          // [3 m][3 a2][16]
  double* W = V + 3 * NB * NQ * NQ;  // [3 m][9 a1a2][4]
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double sp = 0.0;
#pragma unroll
      for (int J = 0; J < 3; ++J) sp += w.P[I + J * 3] * Ji[m * 3 + J];
      PH[m * NQ3 + lane] = wd * sp;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NB * NQ * NQ) {
      const int q01 = lane % (NQ * NQ), a2 = lane / (NQ * NQ);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T2 = tab_ptr<P>(tab, 2, m == 2 ? 1 : 0) + a2 * NQ;
        double sv = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) sv += T2[q2] * PH[m * NQ3 + q01 + NQ * NQ * q2];
        V[(m * NB + a2) * NQ * NQ + q01] = sv;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NB2 * NQ) {
      const int q0 = lane % NQ, a12 = lane / NQ, a1 = a12 % NB, a2 = a12 / NB;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T1 = tab_ptr<P>(tab, 1, m == 1 ? 1 : 0) + a1 * NQ;
        double sw = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < NQ; ++q1) sw += T1[q1] * V[(m * NB + a2) * NQ * NQ + q0 + NQ * q1];
        W[(m * NB2 + a12) * NQ + q0] = sw;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ND) {
      const int a0 = lane % NB, a12 = lane / NB;
      double sr = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T0 = tab_ptr<P>(tab, 0, m == 0 ? 1 : 0) + a0 * NQ;
#pragma unroll
        for (int q0 = 0; q0 < NQ; ++q0) sr += T0[q0] * W[(m * NB2 + a12) * NQ + q0];
      }
      p.scratch_r[(e * 3 + I) * ND + lane] = sr;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one wave per node: lane = element (dz, dy, dx) of the 3 x 3 x 3 neighbourhood, fixed-shape tree sum
__global__ __launch_bounds__(256) void tensor_residual_gather_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, NB = 3, ND = 27;
  const int64_t Al = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // node index inside the shard's node box
  const int lane = threadIdx.x & 63;
  if (Al >= n_nodes) return;
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1];
  const int m0 = p.box_n[0] + P, m1 = p.box_n[1] + P;
  const int A0 = p.box_begin[0] + (int)(Al % m0), A1 = p.box_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.box_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  const int bx0 = p.box_begin[0], bx1 = p.box_begin[1], bx2 = p.box_begin[2];
  const int ex_lo = max(A0 - P, bx0), ex_hi = min(A0, bx0 + p.box_n[0] - 1);
  const int ey_lo = max(A1 - P, bx1), ey_hi = min(A1, bx1 + p.box_n[1] - 1);
  const int ez_lo = max(A2 - P, bx2), ez_hi = min(A2, bx2 + p.box_n[2] - 1);
  if (ex_lo > ex_hi || ey_lo > ey_hi || ez_lo > ez_hi) return;
  const int dz = lane / 9, dy = (lane / 3) % 3, dx = lane % 3;
  const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
  const bool in = lane < ND && ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
  const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
  const int64_t e = in ? (ex - bx0) + (int64_t)p.box_n[0] * ((ey - bx1) + (int64_t)p.box_n[1] * (ez - bx2)) : 0;
  double rs[3];
#pragma unroll
  for (int I = 0; I < 3; ++I) rs[I] = in ? p.scratch_r[(e * 3 + I) * ND + a] : 0.0;
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) rs[I] += __shfl_down(rs[I], off, 32);
  }
  if (lane == 0) {
    const int64_t gA = p.perm ? p.perm[A] : A;
#pragma unroll
    for (int I = 0; I < 3; ++I) p.r[gA * 3 + I] += rs[I];
  }
}

inline void launch_tensor_residual(mimi_hip_domain_s* h, TensorArgs a) {
  h->scratch_r.resize((size_t)h->n_el * 3 * 27);
  a.scratch_r = h->scratch_r.ptr;
  const unsigned blocks = (unsigned)((h->n_el + 3) / 4);
  if (h->mat.m.kind != MIMI_HIP_MAT_NEOHOOKEAN && h->mat.m.kind != MIMI_HIP_MAT_J2)
    hipLaunchKernelGGL(tensor_residual_kernel<WGS_KIND_RECORD>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  else if (h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN)
    hipLaunchKernelGGL(tensor_residual_kernel<MIMI_HIP_MAT_NEOHOOKEAN>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  else
    hipLaunchKernelGGL(tensor_residual_kernel<MIMI_HIP_MAT_J2>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  MH_HIP(hipGetLastError());
  const int64_t n_nodes = (int64_t)(a.box_n[0] + 2) * (a.box_n[1] + 2) * (a.box_n[2] + 2);   // nodes of the shard
  hipLaunchKernelGGL(tensor_residual_gather_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, a, n_nodes);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
