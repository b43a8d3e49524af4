// Phase 1 of the two-phase tangent assembly as a ROLE-SPECIALISED workgroup (p = 2, 3-D).
//
// Why: a single wave doing everything for an (element column, row i) -- the first phase-1 kernel of this path --
// needs the whole register file of a SIMD (constitutive stage + contraction tiles + prefetch live together) and
// 31 KB of LDS, so it runs alone per SIMD and every LDS / matrix-pipe / memory latency is exposed (measured:
// ~46 k cycles per (element, i) for ~20 k cycles of issued work).  Here the work of one element column is
// split over four waves of one workgroup so that each wave fits in half a register file and two workgroups
// (8 waves) share a CU:
//
//   wave 0  "X"   quadrature-point stage: gathers u, evaluates F and the material ONCE per point
//                 (the single-wave kernel repeated this for every row i), then per row i the
//                 residual piece (sum factorisation) and the pulled-back tangent row
//                 Ahat_i[m][j][n] -> LDS (lane = quadrature point, private slots).
//   wave 1+i "Y_i" contraction of row i:  S1 (matrix pipe, q2) -> S2 (vector pipe, q1, wave-uniform
//                 coefficients) -> S3 (matrix pipe, q0), with the tile TRANSPOSED with respect to
//                 the single-wave kernel: rows = (a2,b2) pair index, so that the entries shared with
//                 the next element of the column, (a2,b2) -> (a2-1,b2-1) = pair index - 4, are the
//                 NEXT ACCUMULATOR REGISTER OF THE SAME LANE.  The carry along the column therefore
//                 lives in registers (27 doubles per lane) instead of a 17.5 KB LDS tile with
//                 read-modify-write; LDS is only used to transpose the finished entries for
//                 coalesced stores (write once, read once).
//
// Per element X evaluates the points of the NEXT element and then writes its three Ahat rows; Y_i contracts the
// column components j = 0, 1, 2 of row i.  Only Ahat is shared, with a single LDS buffer:
//     Y: [tables, j = 0, j = 1: free running]  [read the j = 2 operands] barrier [j = 2, flush] barrier
//     X: [point stage of the next element: free running]               barrier [its rows]       barrier
// (what runs free reads data written before the previous barrier and writes nothing another wave reads before
// the next one; the store-transposition buffer of a contraction wave is private).  Every wave executes 2 n + 1
// barriers.  Scratch layout, phase 2 and the bitwise reproducibility of the result: kernels_tensor_2phase.hpp.
#pragma once

#include <hip/hip_runtime.h>

#include "materials_other.hpp"

#include "kernels_tensor_2phase.hpp"

namespace mimi_hip {

struct WgsLds {
  static constexpr int NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81;
  static constexpr int off_ue = 0;                       // [3][27] (+1 pad)           X private
  static constexpr int off_tab = off_ue + 3 * ND + 1;    // [2 parity][3 dir][2][3][4]  X -> Y
  static constexpr int off_r = off_tab + 2 * 6 * NB * NQ;  // stage-R scratch            X private
  static constexpr int r_size = 3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ;
  static constexpr int off_ah = off_r + r_size;          // [3 i][27 (m,j,n)][64]        X -> Y_i
  static constexpr int n_final = 9 * NROW + 18 * ND;     // 1215 entries stored by a non-last element
  static constexpr int n_carry = 18 * 2 * ND;            // 972 more by the last element of a column
  static constexpr int st_size = n_final + 1;
  static constexpr int off_st = off_ah + 3 * ND * NQ3;   // [3 i][1216] store transposition     Y_i private
  static constexpr int total = off_st + 3 * st_size;
};

#define WGS_PIN(x) asm volatile("" : "+v"(x))

MH_DEV void wgs_barrier() {
  // LDS hand-off only: outstanding global loads / stores need not drain here
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// v_permlane32_swap(a, b) -> ([a.lo, b.lo], [a.hi, b.hi]) (halves of 32 lanes; checked on gfx950,
// scratch/pl/pl_test.hip).  One pair of swaps gives both rotations the carry needs:
//   c_hi_in_lo : lanes 0..31 hold c of lanes 32..63      o_lo_in_hi : lanes 32..63 hold o of lanes 0..31
MH_DEV void swap32_f64(double c, double o, double& c_hi_in_lo, double& o_lo_in_hi) {
  const unsigned long long uc = __double_as_longlong(c), uo = __double_as_longlong(o);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)uc, (unsigned)uo, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(uc >> 32), (unsigned)(uo >> 32), false, false);
  o_lo_in_hi = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
  c_hi_in_lo = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
}

// ------------------------------------------------------------------------------------------------
// J2: material pre-pass.  The return mapping (scalar Newton with dual-number Johnson-Cook hardening:
// pow / log in fp64) costs tens of thousands of cycles per point; inside the workgroup it would run in
// the single point wave and bound the whole kernel.  Here every wave of the chip takes one element
// (lane = quadrature point) and leaves the PointResult in scratch_pt[element][field][point].
// ------------------------------------------------------------------------------------------------
// record of a point: Finv 9, detF, sigma (symmetric: 6), trial deviator (symmetric: 6), beta, gamma of the algorithmic
// tangent (materials.hpp tangent_row_of) -- P = J sigma F^-T is recomputed by the reader
constexpr int WGS_PT_FIELDS = 24;

MH_DEV void wgs_point_store(double* rec, int lane, const mimi_hip_material& m, const PointResult<3>& w) {
  constexpr int NQ3 = 64;
  constexpr int sym_i[6] = {0, 1, 2, 1, 2, 2}, sym_j[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
  for (int k = 0; k < 9; ++k) rec[k * NQ3 + lane] = w.Finv[k];
  rec[9 * NQ3 + lane] = w.detF;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    rec[(10 + k) * NQ3 + lane] = w.sigma[sym_i[k] + 3 * sym_j[k]];
    rec[(16 + k) * NQ3 + lane] = w.s_trial[sym_i[k] + 3 * sym_j[k]];
  }
  double beta = 1.0, gamma = 0.0;
  if (w.plastic) {
    const double q = w.q, G = m.G;
    beta = 1.0 - 3.0 * G * w.delta / q;
    gamma = 3.0 * G * (1.5 / q) * (1.0 / ((3.0 * G + w.hprime) * q) - w.delta / (q * q));
  }
  rec[22 * NQ3 + lane] = beta;
  rec[23 * NQ3 + lane] = gamma;
}

// w gets Finv, detF, sigma, s_trial and (recomputed as the material does, pk1_from_cauchy) P
MH_DEV void wgs_point_load(const double* rec, int lane, PointResult<3>& w, double& beta, double& gamma) {
  constexpr int NQ3 = 64;
  constexpr int sym_i[6] = {0, 1, 2, 1, 2, 2}, sym_j[6] = {0, 0, 0, 1, 1, 2};
#pragma unroll
  for (int k = 0; k < 9; ++k) w.Finv[k] = rec[k * NQ3 + lane];
  w.detF = rec[9 * NQ3 + lane];
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const double sg = rec[(10 + k) * NQ3 + lane], st = rec[(16 + k) * NQ3 + lane];
    w.sigma[sym_i[k] + 3 * sym_j[k]] = sg;
    w.sigma[sym_j[k] + 3 * sym_i[k]] = sg;
    w.s_trial[sym_i[k] + 3 * sym_j[k]] = st;
    w.s_trial[sym_j[k] + 3 * sym_i[k]] = st;
  }
  beta = rec[22 * NQ3 + lane];
  gamma = rec[23 * NQ3 + lane];
  pk1_from_cauchy<3>(w);
}

// The other materials (materials_other.hpp): no closed form of the pulled-back tangent -- the pre-pass leaves the
// tangent itself, pulled back to the reference element and weighted, in the record:
//   field i*27 + (m*3 + j)*3 + n : Ahat_i[m][j][n] = wd sum_JL Jinv[m][J] dP_iJ/dF_jL Jinv[n][L]
//   field 81 + i*3 + m           : Phat_i[m]       = wd sum_J  P_iJ Jinv[m][J]
// (46 KB per element; wave X of the nine-block kernel then only copies its row into LDS).
constexpr int WGS_REC_FIELDS = 90;
constexpr int WGS_KIND_RECORD = 100;   // compile-time "material kind" of the kernels that read such records

// FAMILY 0: J2 (24-field record, closed forms in wave X); 2..5: that one of the other materials (90-field record) as a
// compile-time constant (all four in one kernel spilled 168 .. 298 registers, one at a time none).
// COMMIT 1: DomainPostTimeAdvance (nonlinear_solid.cpp:179-199) -- F at the points as for an assembly, then the material's
// state commit and nothing else (round 4: the separate commit kernel's unrolled 27-node sum took 432 registers, one wave per
// SIMD: 4.8 ms for the 16.8 M points of the north-star mesh with J2)
template<int FAMILY, int COMMIT = 0>
__global__ __launch_bounds__(256) void tensor_point_kernel(TensorArgs p, int n_el) {
  constexpr int P = 2, NB = 3, NQ = 4, ND = 27, NQ3 = 64;
  constexpr int FK = FAMILY >= 2 ? FAMILY : -1;
  __shared__ double ue_all[4][3 * ND];
  __shared__ double tab_all[4][6 * NB * NQ];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= n_el) return;
  double* ue = ue_all[wave];
  double* tab = tab_all[wave];
  int el[3];
  el[0] = e % p.box_n[0];
  el[1] = (e / p.box_n[0]) % p.box_n[1];
  el[2] = e / ((int64_t)p.box_n[0] * p.box_n[1]);
  const int32_t* dofs = p.dofs + e * ND;
  for (int a = lane; a < ND; a += 64) {
    const int64_t node = dofs[a];
    for (int c = 0; c < 3; ++c) ue[c * ND + a] = p.u[node * 3 + c];
  }
  for (int t = lane; t < 6 * NB * NQ; t += 64) {
    const int dir = t / (2 * NB * NQ), rem = t % (2 * NB * NQ), isD = rem / (NB * NQ), k = rem % (NB * NQ);
    const int span = p.box_begin[dir] + el[dir];
    tab[t] = ((isD ? p.tabD[dir] : p.tabB[dir]) + (int64_t)span * NB * NQ)[k];
  }
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  const int q0 = lane % NQ, q1 = (lane / NQ) % NQ, q2 = lane / (NQ * NQ);
  double H[9];
  for (int k = 0; k < 9; ++k) H[k] = 0.0;
#pragma unroll 1
  for (int a2 = 0; a2 < NB; ++a2)
#pragma unroll 1
    for (int a1 = 0; a1 < NB; ++a1)
      for (int a0 = 0; a0 < NB; ++a0) {
        const int a = a0 + NB * (a1 + NB * a2);
        const double b0 = tab_ptr<P>(tab, 0, 0)[a0 * NQ + q0], d0 = tab_ptr<P>(tab, 0, 1)[a0 * NQ + q0];
        const double b1 = tab_ptr<P>(tab, 1, 0)[a1 * NQ + q1], d1 = tab_ptr<P>(tab, 1, 1)[a1 * NQ + q1];
        const double b2 = tab_ptr<P>(tab, 2, 0)[a2 * NQ + q2], d2 = tab_ptr<P>(tab, 2, 1)[a2 * NQ + q2];
        // same association as the point wave: d0 * (b1 * b2), b0 * (d1 * b2), b0 * (b1 * d2)
        const double dn0 = d0 * (b1 * b2), dn1 = b0 * (d1 * b2), dn2 = b0 * (b1 * d2);
        for (int i = 0; i < 3; ++i) {
          const double uu = ue[i * ND + a];
          H[i * 3 + 0] += uu * dn0;
          H[i * 3 + 1] += uu * dn1;
          H[i * 3 + 2] += uu * dn2;
        }
      }
  const double* g = p.geo + e * 10 * NQ3 + lane;
  double F[9];
  for (int i = 0; i < 3; ++i)
    for (int J = 0; J < 3; ++J) {
      double sf = (i == J) ? 1.0 : 0.0;
      for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * g[(int64_t)(m * 3 + J) * NQ3];
      F[i + J * 3] = sf;
    }
  if constexpr (COMMIT) {
    int st;
    if constexpr (FAMILY != 0) st = accumulate_other<3, FK>(p.mat, p.dt, p.state, e * NQ3 + lane, F);
    else st = accumulate_state<3>(p.mat, p.dt, p.state, e * NQ3 + lane, F);
    if (st) atomicOr(p.status, st);
    return;
  }
  if constexpr (FAMILY != 0) {
    // the tangent one direction (j, L) at a time, pulled back and accumulated over L as it arrives (27 accumulators per j
    // instead of all 81 entries of dP/dF: see tp3_point_kernel, tensor_p3.hip)
    double Pk[9];
    OtherTangent<3> ot;
    const int status = other_tangent_begin<3, FK>(p.mat, p.dt, p.state, e * NQ3 + lane, F, Pk, ot);
    double* rec = p.scratch_pt + e * (int64_t)(WGS_REC_FIELDS * NQ3) + lane;
    const double wd = g[(int64_t)9 * NQ3];
    double Ji[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NQ3];
#pragma unroll 1
    for (int j = 0; j < 3; ++j) {
      double acc[27];
#pragma unroll
      for (int k = 0; k < 27; ++k) acc[k] = 0.0;
#pragma unroll 1
      for (int L = 0; L < 3; ++L) {
        double dP[9];
        other_tangent_dir<3, FK>(p.mat, p.dt, F, ot, j, L, dP);
        double jl[3];     // Jinv[n][L]
#pragma unroll
        for (int n = 0; n < 3; ++n) jl[n] = L == 0 ? Ji[n * 3] : (L == 1 ? Ji[n * 3 + 1] : Ji[n * 3 + 2]);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            double b = 0.0;
#pragma unroll
            for (int J = 0; J < 3; ++J) b += Ji[m * 3 + J] * dP[i + J * 3];
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[(i * 3 + m) * 3 + n] += b * jl[n];
          }
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) rec[(int64_t)(i * 27 + (m * 3 + j) * 3 + n) * NQ3] = wd * acc[(i * 3 + m) * 3 + n];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double t = 0.0;
#pragma unroll
        for (int J = 0; J < 3; ++J) t += Pk[i + J * 3] * Ji[m * 3 + J];
        rec[(int64_t)(81 + i * 3 + m) * NQ3] = wd * t;
      }
    if (status) atomicOr(p.status, status);
  } else {
    PointResult<3> w;
    const int status = evaluate_pk1<3>(p.mat, p.dt, p.state, e * NQ3 + lane, F, w);
    wgs_point_store(p.scratch_pt + e * (int64_t)(WGS_PT_FIELDS * NQ3), lane, p.mat.m, w);
    if (status) atomicOr(p.status, status);
  }
}

// ------------------------------------------------------------------------------------------------
// wave X
// ------------------------------------------------------------------------------------------------
// What wave X keeps per quadrature point between its three steps.  KIND is the material kind as a
// compile-time constant: the fields the other material needs are then dead, and so is its code.
template<int KIND>
struct WgsPoint;

// Neo-Hookean, closed form of the pulled-back tangent: with G = Jinv F^-1 (G[m][i] = sum_J Jinv[m][J] Finv[J][i])
// and M = Jinv Jinv^T, from dP_iJ/dF_jL = mu d_ij d_JL - c1 Finv_Li Finv_Jj + c2 Finv_Lj Finv_Ji
// (materials.hpp tangent_of):
//   Ahat_i[m][j][n] = wd (mu d_ij M[m][n] - c1 G[n][i] G[m][j] + c2 G[n][j] G[m][i])
template<>
struct WgsPoint<MIMI_HIP_MAT_NEOHOOKEAN> {
  double G[9], M[6], Phat[9];  // Phat[i*3 + m]
  double mu_w, c1_w, c2_w;     // coefficients times wd
};

// J2 (small-strain return mapping inside finite kinematics, materials.hpp tangent_row_of): with G = Jinv F^-1,
// N = G Jinv^T, Q[n][j] = sum_L s_jL Jinv[n][L], S_i[m] = sum_k sigma_ik G[m][k], Ts_i[m] = sum_k s_ik G[m][k]
// (s = trial deviator), the pulled-back tangent row is
//   Ahat_i[m][j][n] = wd J ( G[n][j] S_i[m] - G[m][j] S_i[n] + (K - beta 2G/3) G[m][i] Jinv[n][j]
//                            + beta G (d_ij N[m][n] + G[m][j] Jinv[n][i]) - 2G gamma Ts_i[m] Q[n][j] )
template<>
struct WgsPoint<MIMI_HIP_MAT_J2> {
  double G[9], Ji[9], N[9], Q[9], sig[9], st[9], Phat[9];
  double wdJ, Kc, hb, gg;   // wd J, K - beta 2G/3, beta G, 2G gamma
};

template<>
struct WgsPoint<WGS_KIND_RECORD> {
  const double* rec;   // this lane's column of the element's record
};

// fills WgsPoint<J2> from the PointResult of the material pre-pass
MH_DEV void wgs_j2_point(const mimi_hip_material& m, const PointResult<3>& w, double beta, double gamma, const double* Ji,
                         double wd, WgsPoint<MIMI_HIP_MAT_J2>& s) {
  const double G2 = 2.0 * m.G;
  s.wdJ = wd * w.detF;
  s.Kc = m.K - beta * G2 / 3.0;
  s.hb = 0.5 * beta * G2;
  s.gg = G2 * gamma;
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    s.Ji[k] = Ji[k];
    s.sig[k] = w.sigma[k];
    s.st[k] = w.s_trial[k];
  }
#pragma unroll
  for (int mm = 0; mm < 3; ++mm)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double g = 0.0;
#pragma unroll
      for (int Jx = 0; Jx < 3; ++Jx) g += Ji[mm * 3 + Jx] * w.Finv[Jx + c * 3];
      s.G[mm * 3 + c] = g;
    }
#pragma unroll
  for (int mm = 0; mm < 3; ++mm)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      double v = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) v += s.G[mm * 3 + k] * Ji[n * 3 + k];
      s.N[mm * 3 + n] = v;
    }
#pragma unroll
  for (int n = 0; n < 3; ++n)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double v = 0.0;
#pragma unroll
      for (int Lx = 0; Lx < 3; ++Lx) v += w.s_trial[j + Lx * 3] * Ji[n * 3 + Lx];
      s.Q[n * 3 + j] = v;
    }
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) {
      double sp = 0.0;
#pragma unroll
      for (int Jx = 0; Jx < 3; ++Jx) sp += w.P[i + Jx * 3] * Ji[mm * 3 + Jx];
      s.Phat[i * 3 + mm] = wd * sp;
    }
}

template<int KIND>
MH_DEV int wgs_x_point(const TensorArgs& p, int64_t pt, const double* F, const double* Ji, double wd, WgsPoint<KIND>& s) {
  if constexpr (KIND == MIMI_HIP_MAT_NEOHOOKEAN) {
    PointResult<3> w;
    neo_hookean_stress<3>(p.mat.m, F, w);
    const double J = w.detF;
    s.mu_w = wd * p.mat.m.mu;
    s.c1_w = wd * (p.mat.m.lambda * J * (J - 1.) - p.mat.m.mu);
    s.c2_w = wd * (p.mat.m.lambda * (2. * J - 1.) * J);
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double g = 0.0;
#pragma unroll
        for (int Jx = 0; Jx < 3; ++Jx) g += Ji[m * 3 + Jx] * w.Finv[Jx + i * 3];
        s.G[m * 3 + i] = g;
      }
    {
      int c = 0;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = m; n < 3; ++n) {
          double v = 0.0;
#pragma unroll
          for (int Jx = 0; Jx < 3; ++Jx) v += Ji[m * 3 + Jx] * Ji[n * 3 + Jx];
          s.M[c++] = v;
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double sp = 0.0;
#pragma unroll
        for (int Jx = 0; Jx < 3; ++Jx) sp += w.P[i + Jx * 3] * Ji[m * 3 + Jx];
        s.Phat[i * 3 + m] = wd * sp;
      }
    return 0;
  } else {
    MaterialDev mat = p.mat;
    mat.m.kind = KIND;
    PointResult<3> w;
    const int status = evaluate_pk1<3>(mat, p.dt, p.state, pt, F, w);
    double beta = 1.0, gamma = 0.0;
    if (w.plastic) {
      const double q = w.q, G = p.mat.m.G;
      beta = 1.0 - 3.0 * G * w.delta / q;
      gamma = 3.0 * G * (1.5 / q) * (1.0 / ((3.0 * G + w.hprime) * q) - w.delta / (q * q));
    }
    wgs_j2_point(p.mat.m, w, beta, gamma, Ji, wd, s);
    return status;
  }
}

// row I of one element: Ahat_I -> LDS (lane = quadrature point), residual piece -> scratch_r
template<int KIND, int I>
MH_DEV void wgs_x_row(const TensorArgs& p, double* lds, int lane, int64_t e, int par, const WgsPoint<KIND>& s) {
  using L = WgsLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64;
  const double* tab = lds + L::off_tab + par * 6 * NB * NQ;
  double* AH = lds + L::off_ah + I * ND * NQ3;
  double* RS = lds + L::off_r;
  double Phat[3];
  if constexpr (KIND == WGS_KIND_RECORD) {
#pragma unroll
    for (int m = 0; m < 3; ++m) Phat[m] = s.rec[(int64_t)(81 + I * 3 + m) * NQ3];
    double v[ND];
#pragma unroll
    for (int k = 0; k < ND; ++k) v[k] = s.rec[(int64_t)(I * ND + k) * NQ3];
#pragma unroll
    for (int k = 0; k < ND; ++k) AH[k * NQ3 + lane] = v[k];
  } else if constexpr (KIND == MIMI_HIP_MAT_NEOHOOKEAN) {
#pragma unroll
    for (int m = 0; m < 3; ++m) Phat[m] = s.Phat[I * 3 + m];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const double c2gm = s.c2_w * s.G[m * 3 + I];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const double c1gm = s.c1_w * s.G[m * 3 + j];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          double v = c2gm * s.G[n * 3 + j] - c1gm * s.G[n * 3 + I];
          if (I == j) {
            const int lo = m < n ? m : n, hi = m < n ? n : m;
            v += s.mu_w * s.M[lo * 3 - lo * (lo - 1) / 2 + (hi - lo)];
          }
          AH[((m * 3 + j) * 3 + n) * NQ3 + lane] = v;
        }
      }
    }
  } else {
#pragma unroll
    for (int m = 0; m < 3; ++m) Phat[m] = s.Phat[I * 3 + m];
    double S[3], Ts[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a += s.sig[I + k * 3] * s.G[m * 3 + k];
        b += s.st[I + k * 3] * s.G[m * 3 + k];
      }
      S[m] = a;
      Ts[m] = b;
    }
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          double v = s.G[n * 3 + j] * S[m] - s.G[m * 3 + j] * S[n];
          v += s.Kc * s.G[m * 3 + I] * s.Ji[n * 3 + j];
          v += s.hb * ((I == j ? s.N[m * 3 + n] : 0.0) + s.G[m * 3 + j] * s.Ji[n * 3 + I]);
          v -= s.gg * Ts[m] * s.Q[n * 3 + j];
          AH[((m * 3 + j) * 3 + n) * NQ3 + lane] = s.wdJ * v;
        }
  }
  // residual row I by sum factorisation (as kernels_tensor_2phase.hpp)
  double* PH = RS;                   // [3 m][64]
  double* V = PH + 3 * NQ3;          // [3 m][3 a2][16]
  double* W = V + 3 * NB * NQ * NQ;  // [3 m][9 a1a2][4]
#pragma unroll
  for (int m = 0; m < 3; ++m) PH[m * NQ3 + lane] = Phat[m];
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
    p.scratch_r[(e * ND + lane) * 3 + I] = sr;
  }
  __builtin_amdgcn_wave_barrier();
}

template<int KIND>
MH_DEV void wgs_x_loop(const TensorArgs& p, double* lds, int eu, int ev, int& status) {
  using L = WgsLds;
  constexpr int P = 2, NB = 3, NQ = 4, ND = 27, NQ3 = 64;
  constexpr int TROUNDS = 2;  // 72 table values
  const int lane = threadIdx.x & 63;
  double* ue = lds + L::off_ue;
  const int n_seq = p.box_n[2];
  auto element_at = [&](int es) -> int64_t { return eu + (int64_t)p.box_n[0] * (ev + (int64_t)p.box_n[1] * es); };
  auto table_src = [&](int es, int t) -> const double* {
    const int dir = t / (2 * NB * NQ);
    const int rem = t % (2 * NB * NQ);
    const int isD = rem / (NB * NQ);
    const int k = rem % (NB * NQ);
    const int span = (dir == 0 ? p.box_begin[0] + eu : dir == 1 ? p.box_begin[1] + ev : p.box_begin[2] + es);
    return (isD ? (dir == 0 ? p.tabD[0] : dir == 1 ? p.tabD[1] : p.tabD[2])
                : (dir == 0 ? p.tabB[0] : dir == 1 ? p.tabB[1] : p.tabB[2])) + (int64_t)span * NB * NQ + k;
  };
  // The operands of an element are requested at the END of the previous element's last step (they are
  // in flight across the two barriers, not across the row computations, which need the registers);
  // the connectivity one element earlier still.
  int32_t node_n = lane < ND ? p.dofs[element_at(0) * ND + lane] : 0;
  double ue_r[3], tab_r[TROUNDS], geo_r[10];
  auto request = [&](int es) {
    const int64_t e_n = element_at(es);
#pragma unroll
    for (int c = 0; c < 3; ++c) ue_r[c] = p.u[(int64_t)node_n * 3 + c];
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      tab_r[rd] = *table_src(es, t < 6 * NB * NQ ? t : 0);
    }
    const double* g = p.geo + e_n * 10 * NQ3 + lane;
#pragma unroll
    for (int k = 0; k < 10; ++k) geo_r[k] = g[(int64_t)k * NQ3];
    if (es + 1 < n_seq) node_n = lane < ND ? p.dofs[element_at(es + 1) * ND + lane] : 0;
  };
  request(0);

  WgsPoint<KIND> s;
  // quadrature-point stage of element es from the requested operands (tables -> LDS parity es & 1)
  auto point_stage = [&](int es) {
    const int par = es & 1;
    const int64_t e = element_at(es);
    double* tab = lds + L::off_tab + par * 6 * NB * NQ;
    if (lane < ND) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ue[c * ND + lane] = ue_r[c];
    }
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      if (t < 6 * NB * NQ) tab[t] = tab_r[rd];
    }
    double Ji[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ji[k] = geo_r[k];
    const double wd = geo_r[9];
    __builtin_amdgcn_wave_barrier();
    if constexpr (KIND == MIMI_HIP_MAT_NEOHOOKEAN) {
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
            // keep the LDS reads of later (a1, a2) where they are (hoisted together they need 162 registers)
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
      status |= wgs_x_point<KIND>(p, e * NQ3 + lane, F, Ji, wd, s);
    } else if constexpr (KIND == WGS_KIND_RECORD) {
      s.rec = p.scratch_pt + e * (int64_t)(WGS_REC_FIELDS * NQ3) + lane;
    } else {
      // J2: the material was evaluated by tensor_point_kernel
      PointResult<3> w;
      double beta, gamma;
      wgs_point_load(p.scratch_pt + e * (int64_t)(WGS_PT_FIELDS * NQ3), lane, w, beta, gamma);
      wgs_j2_point(p.mat.m, w, beta, gamma, Ji, wd, s);
    }
    __builtin_amdgcn_wave_barrier();
  };
  auto rows = [&](int es) {
    wgs_x_row<KIND, 0>(p, lds, lane, element_at(es), es & 1, s);
    wgs_x_row<KIND, 1>(p, lds, lane, element_at(es), es & 1, s);
    wgs_x_row<KIND, 2>(p, lds, lane, element_at(es), es & 1, s);
  };
  // prologue: point stage and rows of element 0
  point_stage(0);
  if (1 < n_seq) request(1);
  rows(0);
  wgs_barrier();
  for (int it = 0; it < n_seq; ++it) {
    // free running: point stage of element it + 1 (its own ue / point data, the table buffer of the OTHER parity)
    if (it + 1 < n_seq) {
      point_stage(it + 1);
      if (it + 2 < n_seq) request(it + 2);
    }
    // lock step: rows of element it + 1, once every contraction wave holds its last operands of element it
    wgs_barrier();
    if (it + 1 < n_seq) rows(it + 1);
    wgs_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// contraction waves: shared pieces (also used by kernels_tensor_wgsym.hpp)
// ------------------------------------------------------------------------------------------------
struct WgsLane {
  int lane, grp;
  bool col_ok;
  // 1.0 where the carried value (as held / rotated by 32 lanes) belongs into the first / second result register of this
  // lane group, else 0.0: the carry is added by ONE fused multiply-add instead of a select and an add
  double take_c0, take_c1;
  // compact store-transposition slots (kernels_tensor_wgs.hpp): as computed ...
  int base0, stride0, base1, basec;
  // ... and with the roles of a and b exchanged
  int baseT0, strideT0, baseT1, basecT;
};

// for_dump: the lanes / registers without an entry get zero offsets and strides (what they index is the dump region of
// wgs_contract_block, never a staging buffer)
MH_DEV WgsLane wgs_lane_constants(bool for_dump = false) {
  constexpr int NB = 3, ND = 27, NROW = 81;
  WgsLane c;
  c.lane = threadIdx.x & 63;
  const int col = c.lane & 15;
  c.grp = c.lane >> 4;
  c.col_ok = col < 9;
  c.take_c0 = c.grp != 2 ? 1.0 : 0.0;
  c.take_c1 = c.grp == 0 ? 1.0 : 0.0;
  const int a0 = c.col_ok ? col / NB : 0, b0 = c.col_ok ? col % NB : 0;
  const int grp = c.grp;
  // rows of register 0: (a2,b2) = (0,0) (0,1) (0,2) (1,0) for grp 0..3; register 1, grp 2: (2,0)
  c.base0 = grp < 3 ? a0 * NROW + grp * ND + b0 * 3 : 9 * NROW + a0 * ND + b0 * 3;
  c.stride0 = grp < 3 ? 3 * NROW : 3 * ND;   // per a1
  c.base1 = 9 * NROW + (a0 + 9) * ND + b0 * 3;  // per a1: 3 * ND
  // carried rows in the element block (P2Block: row a - 9 at 162 (a - 9), this piece's 54 values inside at 54 I -- the
  // caller's pointer --, then (b2 - 1) 27 + b1 9 + b0 3 + j)
  c.basec = a0 * 162 + b0 * 3 + (grp == 0 ? 0 : grp == 1 ? ND : grp == 3 ? 9 * 162 : 9 * 162 + ND);  // per a1: 486
  // transposed: node a' = b, node b' = a.  a2' = b2 == 0 (grp 0, 3; row 6): s = (b0 + 3 b1) 81 + a2 27 + a1 9 + a0 3 + i
  //             a2' = b2 >= 1 (grp 1, 2):      s = 729 + (b0 + 3 b1 + 9 (b2 - 1)) 27 + a1 9 + a0 3 + i
  c.baseT0 = grp == 0 ? b0 * NROW + a0 * 3
           : grp == 3 ? b0 * NROW + ND + a0 * 3
           : grp == 1 ? 9 * NROW + b0 * ND + a0 * 3
                      : 9 * NROW + (b0 + 9) * ND + a0 * 3;
  c.strideT0 = (grp == 0 || grp == 3) ? 3 * NROW : 3 * ND;   // per b1; per a1: 9
  c.baseT1 = b0 * NROW + 2 * ND + a0 * 3;                      // row 6 (2,0) -> (0,2): per b1 3 * NROW; per a1 9
  // carried rows: grp 0 row 4 (1,1) -> (1,1); grp 1 row 5 (1,2) -> (2,1); grp 3 row 7 (2,1) -> (1,2); grp 2 row 8 (2,2)
  //   s' = (a' - 9) 162 + (b2' - 1) 27 + b1' 9 + b0' 3 + i,  a' = b0 + 3 b1 + 9 b2,  b2' = a2
  c.basecT = b0 * 162 + a0 * 3 + (grp == 0 ? 0 : grp == 1 ? 9 * 162 : grp == 3 ? ND : 9 * 162 + ND);   // per b1: 486; per a1: 9
  if (for_dump) {
    if (!c.col_ok) c.base0 = c.stride0 = c.baseT0 = c.strideT0 = 0;
    if (!c.col_ok || grp != 2) c.base1 = c.baseT1 = 0;
  }
  return c;
}

// One (i, j) block of one element in a contraction wave: S1, then S2 / S3 pipelined over b1, the
// carry in registers, finished entries into the store-transposition buffer(s).
// st_n / jn: buffer of piece i and the column component j; st_t / jt: buffer of piece j and i.
// MODE 0: plain block.  MODE 1 (off-diagonal block, i > j): every entry is also stored transposed.
// MODE 2 (diagonal block, i == j): the block is symmetric itself, K[(a1 ..), (b1 ..)] = K[(b1 ..), (a1 ..)]^T, so
// only the six chains with a1 >= b1 are contracted and those with a1 > b1 are also stored transposed
// (st_t == st_n, jt == jn).
// DUMP (round 4, symmetric-half kernel): the staging stores carry no execution mask -- the lanes that hold no
// entry (7 of every 16 columns; for the second register every lane group but 2) store to a 512-double dump region instead
// (their lane constants are zero, wgs_lane_constants(true)).  Thirty mask regions per element and contraction wave
// (s_and_saveexec / s_cbranch_execz / s_or: ~90 scalar instructions and as many scheduling boundaries) are gone.
template<int MODE, bool DUMP = false>
MH_DEV void wgs_contract_block(const WgsLane& lc, const double (&ah)[9], const double (&aS0)[4], const double (&aS2)[4],
                               const double (&uB1)[3][4], const double (&uD1)[3][4], double (&C)[9],
                               double* st_n, int jn, double* st_t, int jt, double* dump = nullptr) {
  constexpr int NB = 3, NQ = 4, ND = 27, NROW = 81;
  const mh_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  const int grp = lc.grp;
  double* const dn0 = lc.col_ok ? st_n : dump;
  double* const dn1 = (lc.col_ok && grp == 2) ? st_n : dump;
  double* const dt0 = lc.col_ok ? st_t : dump;
  double* const dt1 = (lc.col_ok && grp == 2) ? st_t : dump;
  mh_d4 D1[9];
#pragma unroll
  for (int mn = 0; mn < 9; ++mn) {
    const int m = mn / 3, n = mn % 3;
    const int v2 = (m == 2 ? 1 : 0) + (n == 2 ? 2 : 0);
    // (diagonal block: Ahat_(i m)(i n) is symmetric in (m, n) and (0, 1), (1, 0) share the table variant: one product)
    if (MODE == 2 && mn == 3) D1[3] = D1[1];
    else D1[mn] = __builtin_amdgcn_mfma_f64_16x16x4f64(ah[mn], aS2[v2], zero4, 0, 0, 0);
  }
  mh_d4 Kt[NB];
  auto carry_and_stage = [&](int b1) {
#pragma unroll
    for (int a1 = 0; a1 < NB; ++a1) {
      if (MODE == 2 && a1 < b1) continue;
      const int a1b1 = a1 * NB + b1;
      const double cin = C[a1b1];
      double c_rot, o2_rot;
      swap32_f64(cin, Kt[a1][2], c_rot, o2_rot);
      // (the nine-block kernel, MODE 0, has no registers for the two lane masks: it keeps the select + add form)
      const double out0 = MODE == 0 ? Kt[a1][0] + (grp != 2 ? cin : 0.0) : __builtin_fma(cin, lc.take_c0, Kt[a1][0]);
      const double out1 = MODE == 0 ? Kt[a1][1] + (grp == 0 ? c_rot : 0.0) : __builtin_fma(c_rot, lc.take_c1, Kt[a1][1]);
      if constexpr (DUMP) {
        dn0[lc.base0 + a1 * lc.stride0 + b1 * 9 + jn] = out0;
        dn1[lc.base1 + a1 * (3 * ND) + b1 * 9 + jn] = out1;
        if (MODE == 1 || (MODE == 2 && a1 > b1)) {
          dt0[lc.baseT0 + b1 * lc.strideT0 + a1 * 9 + jt] = out0;
          dt1[lc.baseT1 + b1 * (3 * NROW) + a1 * 9 + jt] = out1;
        }
      } else {
        if (lc.col_ok) st_n[lc.base0 + a1 * lc.stride0 + b1 * 9 + jn] = out0;
        if (lc.col_ok && grp == 2) st_n[lc.base1 + a1 * (3 * ND) + b1 * 9 + jn] = out1;
        if (MODE == 1 || (MODE == 2 && a1 > b1)) {
          if (lc.col_ok) st_t[lc.baseT0 + b1 * lc.strideT0 + a1 * 9 + jt] = out0;
          if (lc.col_ok && grp == 2) st_t[lc.baseT1 + b1 * (3 * NROW) + a1 * 9 + jt] = out1;
        }
      }
      C[a1b1] = grp == 2 ? o2_rot : out1;
    }
  };
#pragma unroll
  for (int b1 = 0; b1 < NB; ++b1) {
    // S2.  The nine (m, n) terms are merged by the table variant of their a-side BEFORE the a-side multiplication
    // (g = 3: (0,0); g = 1: (0,1) (0,2); g = 2: (1,0) | (2,0); g = 0: (1,1) (1,2) | (2,1) (2,2)): 9 + 6 x (pairs) instead
    // of 9 + 9 x (pairs) instructions per (b1, q1), every accumulation ONE fused multiply-add (round 3: on this chip
    // every vector instruction costs its ~5 cycles of the SIMD whatever else is in flight, scratch/issue_bench.hip).
    double Ec[4][NB];
#pragma unroll
    for (int q1 = 0; q1 < NQ; ++q1) {
      const double cbB = uB1[b1][q1], cbD = uD1[b1][q1];
      const double W3 = cbB * D1[0][q1];
      const double W1 = __builtin_fma(cbD, D1[1][q1], cbB * D1[2][q1]);
      const double W2a = cbB * D1[3][q1], W2b = cbB * D1[6][q1];
      const double W0a = __builtin_fma(cbD, D1[4][q1], cbB * D1[5][q1]);
      const double W0b = __builtin_fma(cbD, D1[7][q1], cbB * D1[8][q1]);
#pragma unroll
      for (int a1 = 0; a1 < NB; ++a1) {
        if (MODE == 2 && a1 < b1) continue;
        const double caB = uB1[a1][q1], caD = uD1[a1][q1];
        if (q1 == 0) {   // (no accumulator starts from 0.0: x + 0.0 is an instruction)
          Ec[3][a1] = caB * W3;
          Ec[1][a1] = caB * W1;
          Ec[2][a1] = __builtin_fma(caD, W2a, caB * W2b);
          Ec[0][a1] = __builtin_fma(caD, W0a, caB * W0b);
        } else {
          Ec[3][a1] = __builtin_fma(caB, W3, Ec[3][a1]);
          Ec[1][a1] = __builtin_fma(caB, W1, Ec[1][a1]);
          Ec[2][a1] = __builtin_fma(caD, W2a, __builtin_fma(caB, W2b, Ec[2][a1]));
          Ec[0][a1] = __builtin_fma(caD, W0a, __builtin_fma(caB, W0b, Ec[0][a1]));
        }
      }
    }
    if (b1 > 0) carry_and_stage(b1 - 1);
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int a1 = 0; a1 < NB; ++a1)
        if (!(MODE == 2 && a1 < b1)) WGS_PIN(Ec[g][a1]);
#pragma unroll
    for (int a1 = 0; a1 < NB; ++a1)
      if (!(MODE == 2 && a1 < b1)) Kt[a1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ec[0][a1], aS0[0], zero4, 0, 0, 0);
#pragma unroll
    for (int g = 1; g < 4; ++g)
#pragma unroll
      for (int a1 = 0; a1 < NB; ++a1)
        if (!(MODE == 2 && a1 < b1)) Kt[a1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ec[g][a1], aS0[g], Kt[a1], 0, 0, 0);
  }
  carry_and_stage(NB - 1);
}

// carried rows of the last element of a column -> the third part of the element block (st_n / st_t: P2Block::carry_of(E, i))
template<int MODE>
MH_DEV void wgs_stage_carry(const WgsLane& lc, const double (&C)[9], double* st_n, int jn, double* st_t, int jt) {
  constexpr int NB = 3;
#pragma unroll
  for (int a1b1 = 0; a1b1 < 9; ++a1b1) {
    const int a1 = a1b1 / NB, b1 = a1b1 % NB;
    if (MODE == 2 && a1 < b1) continue;
    if (lc.col_ok) st_n[lc.basec + a1 * 486 + b1 * 9 + jn] = C[a1b1];
    if ((MODE == 1 || (MODE == 2 && a1 > b1)) && lc.col_ok) st_t[lc.basecT + b1 * 486 + a1 * 9 + jt] = C[a1b1];
  }
}

// (the pieces stored with the non-temporal hint: step 7.18 -> 7.09 ms on one box, but WRITE_SIZE of this kernel 8.2 -> 9.9 GB
// -- the runs of 81 / 27 doubles are not whole lines and the hint takes them past the write combining of L2 --: not kept,
// profiles/r04_northstar_nt_pieces.txt)
#define WGS_PIECE_STORE(ptr, v) (*(ptr) = (v))
// buffer (compact: nine rows of 81, eighteen rows of 27) -> this piece's runs in the element block E (P2Block)
MH_DEV void wgs_flush_final(int lane, const double* ST, double* E, int I) {
  constexpr int NROW = 81, ND = 27;
  {
    // rows a2 = 0: E[a 243 + I 81 + k], k < 81: the first 64 values of a row per instruction, then the 17 others of
    // three rows together (lanes 0..16, 17..33, 34..50)
    const int g3 = lane / 17, k3 = lane - 17 * g3;
    const bool ok3 = lane < 51;
    const int src3 = ok3 ? g3 * NROW + 64 + k3 : 0, dst3 = g3 * 3 * NROW + 64 + k3;
    double v0[9], v1[3];
#pragma unroll
    for (int a = 0; a < 9; ++a) v0[a] = ST[a * NROW + lane];
#pragma unroll
    for (int q = 0; q < 3; ++q) v1[q] = ST[q * 3 * NROW + src3];
    double* d = E + I * NROW;
#pragma unroll
    for (int a = 0; a < 9; ++a) WGS_PIECE_STORE(&d[(unsigned)(a * 3 * NROW + lane)], v0[a]);
#pragma unroll
    for (int q = 0; q < 3; ++q)
      if (ok3) WGS_PIECE_STORE(&d[(unsigned)(q * 9 * NROW + dst3)], v1[q]);
  }
  {
    // rows a2 >= 1, b2 = 0: E[2187 + (a - 9) 81 + I 27 + k], k < 27: two rows per instruction (lanes 0..26, 32..58)
    const int half = lane >> 5, k = lane & 31;
    const bool ok = k < ND;
    double w[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) w[r] = ST[9 * NROW + (2 * r + half) * ND + (ok ? k : 0)];
    double* d = E + P2Block::off_b20 + I * ND + half * NROW + k;
#pragma unroll
    for (int r = 0; r < 9; ++r)
      if (ok) WGS_PIECE_STORE(&d[(unsigned)(2 * r * NROW)], w[r]);
  }
}

// ------------------------------------------------------------------------------------------------
// wave Y_I of the nine-block kernel: row I, one column component j per step
// ------------------------------------------------------------------------------------------------
template<int I>
MH_DEV void wgs_y_loop(const TensorArgs& p, double* lds, int eu, int ev) {
  using L = WgsLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81, NK = ND * NROW;
  const WgsLane lc = wgs_lane_constants();
  const int lane = lc.lane;
  double* AH = lds + L::off_ah + I * ND * NQ3;
  double* ST = lds + L::off_st + I * L::st_size;
  const int n_seq = p.box_n[2];
  auto block_of = [&](int es) -> double* {
    return p.scratch_k + (eu + (int64_t)p.box_n[0] * (ev + (int64_t)p.box_n[1] * es)) * (int64_t)P2Block::size;
  };
  // matrix-operand lane constants: pair index on bits 3:0, contraction index on bits 5:4
  const int mrow = lane & 15, mk = lane >> 4;
  const bool mrow_ok = mrow < NB2;
  const int mra = mrow_ok ? mrow / NB : 0, mrb = mrow_ok ? mrow % NB : 0;

  double C[3][NB2];  // packed carry [j][a1b1]
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < NB2; ++k) C[j][k] = 0.0;
  double aS0[4], aS2[4];            // pair tables of directions 0 and 2 (variants B.B, D.B, B.D, D.D)
  double uB1[NB][NQ], uD1[NB][NQ];  // wave-uniform direction-1 tables
#pragma unroll
  for (int v = 0; v < 4; ++v) aS0[v] = aS2[v] = 0.0;
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int q = 0; q < NQ; ++q) uB1[a][q] = uD1[a][q] = 0.0;
  auto load_slice = [&](int j, double (&ah)[9]) {
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) ah[m * 3 + n] = AH[((m * 3 + j) * 3 + n) * NQ3 + lane];
  };
  // prologue: X writes the rows of element 0
  wgs_barrier();
  for (int it = 0; it < n_seq; ++it) {
    // ---- free running: tables, column components j = 0, 1 (the store-transposition buffer of this wave is private;
    // the operands read here were written before the previous barrier and are rewritten after the next one) -----
    {
      const double* tab = lds + L::off_tab + (it & 1) * 6 * NB * NQ;
      {
        const double Ba = tab_ptr<P>(tab, 0, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 0, 1)[mra * NQ + mk];
        const double Bb = tab_ptr<P>(tab, 0, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 0, 1)[mrb * NQ + mk];
        aS0[0] = mrow_ok ? Ba * Bb : 0.0;
        aS0[1] = mrow_ok ? Da * Bb : 0.0;
        aS0[2] = mrow_ok ? Ba * Db : 0.0;
        aS0[3] = mrow_ok ? Da * Db : 0.0;
      }
      {
        const double Ba = tab_ptr<P>(tab, 2, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 2, 1)[mra * NQ + mk];
        const double Bb = tab_ptr<P>(tab, 2, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 2, 1)[mrb * NQ + mk];
        aS2[0] = mrow_ok ? Ba * Bb : 0.0;
        aS2[1] = mrow_ok ? Da * Bb : 0.0;
        aS2[2] = mrow_ok ? Ba * Db : 0.0;
        aS2[3] = mrow_ok ? Da * Db : 0.0;
      }
#pragma unroll
      for (int a = 0; a < NB; ++a)
#pragma unroll
        for (int q1 = 0; q1 < NQ; ++q1) {
          const unsigned long long vb = __double_as_longlong(tab_ptr<P>(tab, 1, 0)[a * NQ + q1]);
          const unsigned long long vd = __double_as_longlong(tab_ptr<P>(tab, 1, 1)[a * NQ + q1]);
          const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)vb), bhi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
          const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)vd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(vd >> 32));
          uB1[a][q1] = __longlong_as_double(((unsigned long long)bhi << 32) | blo);
          uD1[a][q1] = __longlong_as_double(((unsigned long long)dhi << 32) | dlo);
        }
    }
    {
      double ah[9];
      load_slice(0, ah);
      wgs_contract_block<0>(lc, ah, aS0, aS2, uB1, uD1, C[0], ST, 0, ST, 0);
    }
    {
      double ah[9];
      load_slice(1, ah);
      wgs_contract_block<0>(lc, ah, aS0, aS2, uB1, uD1, C[1], ST, 1, ST, 1);
    }
    // ---- lock step: column component j = 2 while X rewrites the rows for the next element ----------------------
    {
      double ah[9];
      load_slice(2, ah);
      wgs_barrier();
      wgs_contract_block<0>(lc, ah, aS0, aS2, uB1, uD1, C[2], ST, 2, ST, 2);
      // this wave's piece of the element is complete: buffer -> scratch
      __builtin_amdgcn_wave_barrier();
      wgs_flush_final(lane, ST, block_of(it), I);
      __builtin_amdgcn_wave_barrier();
      wgs_barrier();
    }
  }
  // the carried rows of the last element have no successor: straight from the registers into the third part of the piece
  {
    double* Sc = P2Block::carry_of(block_of(n_seq - 1), I);
#pragma unroll
    for (int jj = 0; jj < 3; ++jj) wgs_stage_carry<0>(lc, C[jj], Sc, jj, Sc, jj);
  }
}

#define WGS_Y_ARGS p, smem_wgs, eu, ev

// Two workgroups per CU (256 registers per wave) for both materials.
template<int KIND>
__global__ __launch_bounds__(256, 2) void tensor_wgs_kernel(TensorArgs p) {
  extern __shared__ __align__(16) double smem_wgs[];
  // Wave w of a workgroup lands on SIMD w; the point wave idles more than the contraction waves, so
  // the role of a wave rotates with the workgroup and every SIMD hosts a mix of roles.
#ifndef WGS_ROT
#define WGS_ROT(b) ((b) >> 3)
#endif
  const int role = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + WGS_ROT(blockIdx.x)) & 3);
  const int unit = blockIdx.x;
  const int eu = unit % p.box_n[0], ev = unit / p.box_n[0];
  if (role == 0) {
    int status = 0;
    wgs_x_loop<KIND>(p, smem_wgs, eu, ev, status);
    if (status) atomicOr(p.status, status);
  } else if (role == 1) {
    wgs_y_loop<0>(WGS_Y_ARGS);
  } else if (role == 2) {
    wgs_y_loop<1>(WGS_Y_ARGS);
  } else {
    wgs_y_loop<2>(WGS_Y_ARGS);
  }
}

inline void launch_tensor_wgs(mimi_hip_domain_s* h, TensorArgs a) {
  h->scratch_k.resize((size_t)h->n_el * P2Block::size);
  h->scratch_r.resize((size_t)h->n_el * 3 * 27);
  a.scratch_k = h->scratch_k.ptr;
  a.scratch_r = h->scratch_r.ptr;
  a.n_units_u = a.box_n[0];
  a.n_units_v = a.box_n[1];
  const size_t lds = WgsLds::total * sizeof(double);
  const int kind = h->mat.m.kind;
  const bool record = kind != MIMI_HIP_MAT_NEOHOOKEAN && kind != MIMI_HIP_MAT_J2;
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[0], h->stream));   // phase 1 = material pre-pass + nine-block kernel
  if (kind != MIMI_HIP_MAT_NEOHOOKEAN) {
    h->scratch_pt.resize((size_t)h->n_el * (record ? WGS_REC_FIELDS : WGS_PT_FIELDS) * 64);
    a.scratch_pt = h->scratch_pt.ptr;
    if (h->phase_select != 2) {
      void (*point_kernel)(TensorArgs, int) =
          kind == MIMI_HIP_MAT_J2 ? tensor_point_kernel<0>
          : kind == MIMI_HIP_MAT_STVK ? tensor_point_kernel<MIMI_HIP_MAT_STVK>
          : kind == MIMI_HIP_MAT_J2LINEAR ? tensor_point_kernel<MIMI_HIP_MAT_J2LINEAR>
          : kind == MIMI_HIP_MAT_J2SIMO ? tensor_point_kernel<MIMI_HIP_MAT_J2SIMO> : tensor_point_kernel<MIMI_HIP_MAT_J2LOG>;
      hipLaunchKernelGGL(point_kernel, dim3((unsigned)((h->n_el + 3) / 4)), dim3(256), 0, h->stream, a, (int)h->n_el);
      MH_HIP(hipGetLastError());
    }
  }
  h->phase_has_prepass = kind != MIMI_HIP_MAT_NEOHOOKEAN;
  if (h->phase_timing && h->phase_has_prepass) MH_HIP(hipEventRecord(h->phase_ev[3], h->stream));
  auto kernel = kind == MIMI_HIP_MAT_NEOHOOKEAN ? tensor_wgs_kernel<MIMI_HIP_MAT_NEOHOOKEAN>
                : record ? tensor_wgs_kernel<WGS_KIND_RECORD> : tensor_wgs_kernel<MIMI_HIP_MAT_J2>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
  if (h->phase_select != 2) {
    hipLaunchKernelGGL(kernel, dim3(a.box_n[0] * a.box_n[1]), dim3(256), lds, h->stream, a);
    MH_HIP(hipGetLastError());
  }
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[1], h->stream));
  if (h->phase_select != 1) launch_tensor_p2(h, a);
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[2], h->stream));
}

// DomainPostTimeAdvance at degree 2: the pre-pass kernel in its commit mode
inline void launch_tensor_p2_post(mimi_hip_domain_s* h, TensorArgs a) {
  const int kind = h->mat.m.kind;
  void (*kernel)(TensorArgs, int) =
      (kind == MIMI_HIP_MAT_J2 || kind == MIMI_HIP_MAT_NEOHOOKEAN) ? tensor_point_kernel<0, 1>
      : kind == MIMI_HIP_MAT_STVK ? tensor_point_kernel<MIMI_HIP_MAT_STVK, 1>
      : kind == MIMI_HIP_MAT_J2LINEAR ? tensor_point_kernel<MIMI_HIP_MAT_J2LINEAR, 1>
      : kind == MIMI_HIP_MAT_J2SIMO ? tensor_point_kernel<MIMI_HIP_MAT_J2SIMO, 1> : tensor_point_kernel<MIMI_HIP_MAT_J2LOG, 1>;
  hipLaunchKernelGGL(kernel, dim3((unsigned)((h->n_el + 3) / 4)), dim3(256), 0, h->stream, a, (int)h->n_el);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
