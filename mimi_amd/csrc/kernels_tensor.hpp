// Tensor-product (sum-factorised) domain kernels for a 3-D B-spline patch of uniform degree P
// with NQ = P + 2 Gauss points per direction.  See DESIGN.md, "tensor path".
//
// Replaces the reference's per-element quadrature loop + FD Jacobian + thread-local assembly
// (integrators/nonlinear_solid.hpp:65-87, nonlinear_solid.cpp:48-149, nonlinear_base.hpp:90-151)
// for structured patches.  What is computed (exact same quantities as the general kernels):
//   H_im(q)  = sum_a u_ai dN_a/dxi_m(q)              (tensor-product basis from 1-D tables)
//   F        = I + H dxi/dX ;  P(F), A = dP/dF        (materials.hpp)
//   Phat_im  = w det * P_iJ  dxi_m/dX_J ;  Ahat_im,jn = w det * dxi_m/dX_J A_iJjL dxi_n/dX_L
//   R_ai     = sum_q sum_m dN_a/dxi_m Phat_im
//   K_ai,bj  = sum_q sum_mn dN_a/dxi_m Ahat_im,jn dN_b/dxi_n
// The two sums over q are done one parametric direction at a time (sum factorisation), which
// cuts the tangent from ~2.5 MFLOP to ~0.5 MFLOP per p=2 element.
//
// Parallel decomposition
//   * a "unit" is the column of elements along the patch's shortest axis; one wave owns
//     (unit, displacement component i) and walks the column element by element, so every CSR
//     row (A,i) is touched by exactly one wave of a launch at a time;
//   * units whose elements share nodes are separated into (P+1)^2 colours = launches; inside
//     a launch no two waves ever touch the same CSR entry, so the accumulation into r / A is a
//     plain load-add-store (no atomics: fp64 atomics run at ~1 TB/s on gfx950, 10x slower).
//     Sums are therefore bitwise reproducible run to run.
//   * lanes: quadrature points in the constitutive stage; (a2 b2, q0) in the first
//     contraction stage; (b0, a1 b1 a2 b2) in the last, which makes the CSR read-modify-write
//     run along the contiguous (b0, j) direction of a row.
#pragma once

#include <hip/hip_runtime.h>

#include "domain.hpp"
#include "materials.hpp"
#include "materials_other.hpp"

namespace mimi_hip {

struct TensorArgs {
  // patch / shard geometry
  int box_begin[3], box_n[3];   // element box of this handle
  int seq_axis, u_axis, v_axis; // walk axis and the two colour axes
  int colour_u, colour_v;       // colour of this launch
  int n_units_u, n_units_v;     // units of this colour along u and v
  const double* tabB[3];        // [n_spans][NB][NQ]
  const double* tabD[3];
  const double* geo;            // [n_el][10][NQ^3]
  const int32_t* dofs;          // [n_el][NB^3]
  const int32_t* pair_pos;      // [n_el][NB^3][NB^3]
  int structured;               // CSR positions computable arithmetically (lexicographic patch)
  int n_ctrl[3];                // control points per direction
  const int32_t* first[3];      // [n_spans] first basis index of a span
  const int64_t* rowptr;
  const double* u;
  double* r;
  double* A;
  const double* A_base;      // two-phase paths, phase 2: A[row] = A_base[row] + grad_factor * sum (A itself for the plain "+=";
                             // another array for mimi_hip_domain_add_residual_and_grad_from)
  double grad_factor, dt;
  MaterialDev mat;
  StateView state;
  int* status;
  double* scratch_k;         // two-phase path: [n_el][3][27*81] element row pieces
  double* scratch_r;         // two-phase path: [n_el][3][27] element residual pieces
  double* scratch_pt;        // two-phase path, J2: [n_el][24][n_q] material results per quadrature point
  const int64_t* perm;       // two-phase path: lexicographic -> caller's node id (nullptr = identity)
  int cols_per_wg;           // symmetric-half kernel: units (element columns / column segments) a workgroup walks back to back
  int seg_len;               // elements per unit along the walked direction (box_n[2] = whole columns); phase 2: an
                             // element at the end of a unit holds the carried rows (third part of its pieces)
  const unsigned char* nbr_pos;  // two-phase path, permuted numbering: [n_nodes][125] positions inside a CSR row
  const uint16_t* nbr_pos16; // p = 3 two-phase path, permuted numbering: [n_nodes][343] positions inside a CSR row
  double* scratch_tail;      // p = 3 two-phase path: [column][3][48*144] carried rows of the last element of a column
  const double* t2pack;      // p = 3 two-phase path: [span of direction 2][64 lanes][8] table values of the contraction's S1 B operands
  int win_begin[3], win_n[3];  // two-phase paths, phase 2: the nodes this launch gathers (global node indices per direction;
                               // default: every node the handle's elements touch; mimi_hip_domain_gather: a part of them)
};

// wave-private LDS carve, in doubles
template<int P>
struct TensorLds {
  static constexpr int NB = P + 1, NQ = P + 2, NB2 = NB * NB, ND = NB * NB * NB, NQ3 = NQ * NQ * NQ;
  static constexpr int NC = NB2 * NB2;
  static constexpr int off_ue = 0;                       // [3][ND]
  static constexpr int off_tab = off_ue + 3 * ND;        // [3 dir][2 (B,D)][NB][NQ]
  static constexpr int off_rs = off_tab + 6 * NB * NQ;   // [ND] row starts (as int64 bits)
  static constexpr int off_ah = off_rs + ND + (ND & 1);  // [3 j][9 mn][NQ q0][NQ*NQ]  Ahat of row I
  static constexpr int off_zb = off_ah + 27 * NQ3;       // [2 groups][NQ q0][NC]; also stage-R scratch
  static constexpr int zb_r = 3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ;
  static constexpr int zb_size = 2 * NQ * NC > zb_r ? 2 * NQ * NC : zb_r;
  static constexpr int off_cb = off_zb + zb_size + (zb_size & 1);   // [3 j][NB a0][n_carry] carry along the walk axis
  static constexpr int n_carry = (NB - 1) * NB * (NB - 1) * NB * NB;
  static constexpr int total = off_cb + 3 * NB * n_carry;
};

template<int P>
MH_DEV const double* tab_ptr(const double* tab, int dir, int isD) {
  return tab + ((dir * 2 + isD) * (P + 1)) * (P + 2);
}

// One (unit, i) wave.  GRAD: 0 residual only, 1 residual + tangent.
//
// The wave runs alone on its SIMD (the register file is spent on prefetch and on the
// contraction tiles), so every global access is software-pipelined: connectivity two elements
// ahead, u / row starts / geometry / 1-D tables one element ahead, the CSR entries of a block
// before its contraction starts.
template<int P, int GRAD, int I>
MH_DEV void tensor_wave_body(const TensorArgs& p, double* lds, int eu, int ev, int& status) {
  using L = TensorLds<P>;
  constexpr int NB = L::NB, NQ = L::NQ, NB2 = L::NB2, ND = L::ND, NQ3 = L::NQ3, NC = L::NC;
  constexpr int QROUNDS = (NQ3 + 63) / 64;
  constexpr int DROUNDS = (ND + 63) / 64;
  constexpr int TROUNDS = (6 * NB * NQ + 63) / 64;
  constexpr int S3R = (NB * NC + 63) / 64;
  const int lane = threadIdx.x & 63;
  double* ue = lds + L::off_ue;
  double* tab = lds + L::off_tab;
  int64_t* rs = reinterpret_cast<int64_t*>(lds + L::off_rs);
  double* AH = lds + L::off_ah;
  double* ZB = lds + L::off_zb;
  double* CB = lds + L::off_cb;
  // entries shared with the next element of the column are carried in LDS instead of being
  // written and re-read through memory (only when the walk axis is the third local direction)
  const bool use_carry = p.seq_axis == 2;

  const int n_seq = p.seq_axis == 0 ? p.box_n[0] : (p.seq_axis == 1 ? p.box_n[1] : p.box_n[2]);
  auto element_of = [&](int es, int* el) -> int64_t {
#pragma unroll
    for (int d = 0; d < 3; ++d) el[d] = (d == p.seq_axis) ? es : (d == p.u_axis ? eu : ev);
    return el[0] + (int64_t)p.box_n[0] * (el[1] + (int64_t)p.box_n[1] * el[2]);
  };
  auto load_nodes = [&](int64_t e, int32_t* node) {
#pragma unroll
    for (int rd = 0; rd < DROUNDS; ++rd) {
      const int a = rd * 64 + lane;
      node[rd] = a < ND ? p.dofs[e * ND + a] : 0;
    }
  };
  auto table_src = [&](const int* el, int t) -> const double* {
    const int dir = t / (2 * NB * NQ);
    const int rem = t % (2 * NB * NQ);
    const int isD = rem / (NB * NQ);
    const int k = rem % (NB * NQ);
    const int span = (dir == 0 ? p.box_begin[0] + el[0] : dir == 1 ? p.box_begin[1] + el[1] : p.box_begin[2] + el[2]);
    return (isD ? (dir == 0 ? p.tabD[0] : dir == 1 ? p.tabD[1] : p.tabD[2])
                : (dir == 0 ? p.tabB[0] : dir == 1 ? p.tabB[1] : p.tabB[2])) + (int64_t)span * NB * NQ + k;
  };

  // ---- pipeline prologue: element 0 fully, connectivity of element 1 -------------------------
  int el_c[3], el_n[3];
  int64_t e_cur = element_of(0, el_c);
  int32_t node_c[DROUNDS], node_n[DROUNDS];
  load_nodes(e_cur, node_c);
  double ue_r[DROUNDS][3];
  int64_t rs_r[DROUNDS];
  double tab_r[TROUNDS];
  double geo_r[QROUNDS][10];
#pragma unroll
  for (int rd = 0; rd < DROUNDS; ++rd) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ue_r[rd][c] = p.u[(int64_t)node_c[rd] * 3 + c];
    rs_r[rd] = p.rowptr[(int64_t)node_c[rd] * 3 + I];
  }
#pragma unroll
  for (int rd = 0; rd < TROUNDS; ++rd) {
    const int t = rd * 64 + lane;
    tab_r[rd] = *table_src(el_c, t < 6 * NB * NQ ? t : 0);
  }
#pragma unroll
  for (int rd = 0; rd < QROUNDS; ++rd) {
    const int q = rd * 64 + lane;
    const double* g = p.geo + e_cur * 10 * NQ3 + (q < NQ3 ? q : 0);
#pragma unroll
    for (int k = 0; k < 10; ++k) geo_r[rd][k] = g[(int64_t)k * NQ3];
  }
  if (n_seq > 1) {
    const int64_t e1 = element_of(1, el_n);
    load_nodes(e1, node_n);
  }

  for (int es = 0; es < n_seq; ++es) {
    // ---- stage 0: registers -> LDS, then issue the loads of the NEXT element -----------------
#pragma unroll
    for (int rd = 0; rd < DROUNDS; ++rd) {
      const int a = rd * 64 + lane;
      if (a < ND) {
#pragma unroll
        for (int c = 0; c < 3; ++c) ue[c * ND + a] = ue_r[rd][c];
        rs[a] = rs_r[rd];
      }
    }
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      if (t < 6 * NB * NQ) tab[t] = tab_r[rd];
    }
    double Ji_all[QROUNDS][10];
#pragma unroll
    for (int rd = 0; rd < QROUNDS; ++rd)
#pragma unroll
      for (int k = 0; k < 10; ++k) Ji_all[rd][k] = geo_r[rd][k];
    int32_t node_w[DROUNDS];  // nodes of the current element (for the residual rows)
#pragma unroll
    for (int rd = 0; rd < DROUNDS; ++rd) node_w[rd] = node_c[rd];
    const int64_t e = e_cur;
    // pair positions of this element: needed only at scatter time
    int32_t ppv[S3R][NB];
    if constexpr (GRAD == 1) {
      const int32_t* pp = p.pair_pos + e * ND * ND;
#pragma unroll
      for (int rd = 0; rd < S3R; ++rd) {
        const int t = rd * 64 + lane;
        const int tt = t < NB * NC ? t : 0;
        const int b0 = tt % NB, c = tt / NB;
        const int b1 = c % NB, b2 = (c / NB) % NB, a1 = (c / NB2) % NB, a2 = c / (NB2 * NB);
        const int b = b0 + NB * (b1 + NB * b2);
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0) ppv[rd][a0] = pp[(a0 + NB * (a1 + NB * a2)) * ND + b];
      }
    }
    if (es + 1 < n_seq) {
      e_cur = element_of(es + 1, el_c);
#pragma unroll
      for (int rd = 0; rd < DROUNDS; ++rd) {
        node_c[rd] = node_n[rd];
#pragma unroll
        for (int c = 0; c < 3; ++c) ue_r[rd][c] = p.u[(int64_t)node_c[rd] * 3 + c];
        rs_r[rd] = p.rowptr[(int64_t)node_c[rd] * 3 + I];
      }
#pragma unroll
      for (int rd = 0; rd < TROUNDS; ++rd) {
        const int t = rd * 64 + lane;
        tab_r[rd] = *table_src(el_c, t < 6 * NB * NQ ? t : 0);
      }
#pragma unroll
      for (int rd = 0; rd < QROUNDS; ++rd) {
        const int q = rd * 64 + lane;
        const double* g = p.geo + e_cur * 10 * NQ3 + (q < NQ3 ? q : 0);
#pragma unroll
        for (int k = 0; k < 10; ++k) geo_r[rd][k] = g[(int64_t)k * NQ3];
      }
      if (es + 2 < n_seq) {
        const int64_t e2 = element_of(es + 2, el_n);
        load_nodes(e2, node_n);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // ---- stage A: constitutive update, lane = quadrature point ----------------------------
    double Phat[QROUNDS][3];
#pragma unroll
    for (int rd = 0; rd < QROUNDS; ++rd) {
      const int q = rd * 64 + lane;
      const bool active = q < NQ3;
      const int qq = active ? q : 0;
      const int q0 = qq % NQ, q1 = (qq / NQ) % NQ, q2 = qq / (NQ * NQ);
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
      double H[9];  // H[i*3 + m]
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
        }
      // geometry: Ji[m*3 + J] = dxi_m/dX_J, wd = w*det
      const double* Ji = Ji_all[rd];
      const double wd = Ji_all[rd][9];
      double F[9];  // column-major F(i,J)
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int J = 0; J < 3; ++J) {
          double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * Ji[m * 3 + J];
          F[i + J * 3] = sf;
        }
      PointResult<3> w;
      status |= evaluate_pk1<3>(p.mat, p.dt, p.state, e * NQ3 + qq, F, w);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double sp = 0.0;
#pragma unroll
        for (int J = 0; J < 3; ++J) sp += w.P[I + J * 3] * Ji[m * 3 + J];
        Phat[rd][m] = active ? wd * sp : 0.0;
      }
      if constexpr (GRAD == 1) {
        double A[27];  // row I: A[(J*3 + j)*3 + L]
        tangent_row_of<3, I>(p.mat.m, w, A);
        // T[J][j][n] = sum_L A[I][J][j][L] Ji[n][L]
        double T[27];
#pragma unroll
        for (int J = 0; J < 3; ++J)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
              double st = 0.0;
#pragma unroll
              for (int Lx = 0; Lx < 3; ++Lx) st += A[(J * 3 + j) * 3 + Lx] * Ji[n * 3 + Lx];
              T[(J * 3 + j) * 3 + n] = st;
            }
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int n = 0; n < 3; ++n) {
              double sa = 0.0;
#pragma unroll
              for (int J = 0; J < 3; ++J) sa += Ji[m * 3 + J] * T[(J * 3 + j) * 3 + n];
              // AH[j][mn][q0][q1*NQ + q2]
              if (active) AH[(((j * 3 + m) * 3 + n) * NQ + q0) * NQ * NQ + q1 * NQ + q2] = wd * sa;
            }
      }
    }
    // ---- stage R: residual row I by sum factorisation (scratch aliases ZB) ------------------
    {
      double* PH = ZB;                   // [3 m][NQ3]  (q = q0 + NQ q1 + NQ^2 q2)
      double* V = PH + 3 * NQ3;          // [3 m][NB a2][NQ*NQ]
      double* W = V + 3 * NB * NQ * NQ;  // [3 m][NB2 a1a2][NQ]
#pragma unroll
      for (int rd = 0; rd < QROUNDS; ++rd) {
        const int q = rd * 64 + lane;
        if (q < NQ3) {
#pragma unroll
          for (int m = 0; m < 3; ++m) PH[m * NQ3 + q] = Phat[rd][m];
        }
      }
      __builtin_amdgcn_wave_barrier();
      for (int t = lane; t < NB * NQ * NQ; t += 64) {
        const int q01 = t % (NQ * NQ), a2 = t / (NQ * NQ);
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
      for (int t = lane; t < NB2 * NQ; t += 64) {
        const int q0 = t % NQ, a12 = t / NQ, a1 = a12 % NB, a2 = a12 / NB;
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
#pragma unroll
      for (int rd = 0; rd < DROUNDS; ++rd) {
        const int a = rd * 64 + lane;
        if (a < ND) {
          const int a0 = a % NB, a12 = a / NB;
          double sr = 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m) {
            const double* T0 = tab_ptr<P>(tab, 0, m == 0 ? 1 : 0) + a0 * NQ;
#pragma unroll
            for (int q0 = 0; q0 < NQ; ++q0) sr += T0[q0] * W[(m * NB2 + a12) * NQ + q0];
          }
          double* dst = p.r + (int64_t)node_w[rd] * 3 + I;
          *dst += sr;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
    if constexpr (GRAD == 1) {
      // ---- issue the loads of this element's CSR entries now (3 contiguous values per node
      // pair: j = 0..2); they are consumed after the contractions.  The entries touched by
      // one (element, I) are distinct, so nothing aliases.
      double* dstp[S3R][NB];
      double old[S3R][NB][3];
      double Kv[3][S3R][NB];
#pragma unroll
      for (int rd = 0; rd < S3R; ++rd) {
        const int t = rd * 64 + lane;
        const bool act = t < NB * NC;
        const int tt = act ? t : 0;
        const int c = tt / NB;
        const int b2 = (c / NB) % NB, a1 = (c / NB2) % NB, a2 = c / (NB2 * NB);
        const bool carried = use_carry && a2 >= 1 && b2 >= 1 && es + 1 < n_seq;
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0) {
          const int a = a0 + NB * (a1 + NB * a2);
          dstp[rd][a0] = (act && !carried) ? p.A + rs[a] + ppv[rd][a0] : nullptr;
        }
      }
#pragma unroll
      for (int rd = 0; rd < S3R; ++rd)
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0)
#pragma unroll
          for (int jj = 0; jj < 3; ++jj) old[rd][a0][jj] = dstp[rd][a0] ? dstp[rd][a0][jj] : 0.0;
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        const double* AHj = AH + j * 9 * NQ3;
        double Kt[S3R][NB];
#pragma unroll
        for (int rd = 0; rd < S3R; ++rd)
#pragma unroll
          for (int a0 = 0; a0 < NB; ++a0) Kt[rd][a0] = 0.0;
        // two passes: groups {g0 (B,B), g2 (B,D)} share the a-side table B0 in the last
        // contraction, groups {g1 (D,B), g3 (D,D)} share D0   [g = (m==0) + 2 (n==0)]
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          // ---- S1+S2: lane = (q0, a2 b2); contract q2 then q1 ------------------------------
          for (int t = lane; t < NB2 * NQ; t += 64) {
            const int q0 = t % NQ, ab2 = t / NQ, a2 = ab2 / NB, b2 = ab2 % NB;
            double tt2[4][NQ];  // BB, DB (m==2), BD (n==2), DD
#pragma unroll
            for (int q2 = 0; q2 < NQ; ++q2) {
              const double Ba = tab_ptr<P>(tab, 2, 0)[a2 * NQ + q2], Da = tab_ptr<P>(tab, 2, 1)[a2 * NQ + q2];
              const double Bb = tab_ptr<P>(tab, 2, 0)[b2 * NQ + q2], Db = tab_ptr<P>(tab, 2, 1)[b2 * NQ + q2];
              tt2[0][q2] = Ba * Bb;
              tt2[1][q2] = Da * Bb;
              tt2[2][q2] = Ba * Db;
              tt2[3][q2] = Da * Db;
            }
            double B1[NB][NQ], D1[NB][NQ];
#pragma unroll
            for (int a = 0; a < NB; ++a)
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1) {
                B1[a][q1] = tab_ptr<P>(tab, 1, 0)[a * NQ + q1];
                D1[a][q1] = tab_ptr<P>(tab, 1, 1)[a * NQ + q1];
              }
            // X^{mn}[q1] = sum_q2 T2^m[a2] T2^n[b2] Ahat_mn
            auto contract2 = [&](int m, int n, double* X) {
              const int v2 = (m == 2 ? 1 : 0) + (n == 2 ? 2 : 0);
              const double* src = AHj + ((m * 3 + n) * NQ + q0) * NQ * NQ;
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1) {
                double sx = 0.0;
#pragma unroll
                for (int q2 = 0; q2 < NQ; ++q2) sx += tt2[v2][q2] * src[q1 * NQ + q2];
                X[q1] = sx;
              }
            };
            double accA[NB2], accB[NB2];
#pragma unroll
            for (int k = 0; k < NB2; ++k) accA[k] = accB[k] = 0.0;
            if (pass == 0) {
              // g0: m,n in {1,2}:  acc += B1[a1](B1[b1] X22 + D1[b1] X21) + D1[a1](B1[b1] X12 + D1[b1] X11)
              double X11[NQ], X12[NQ], X21[NQ], X22[NQ];
              contract2(1, 1, X11);
              contract2(1, 2, X12);
              contract2(2, 1, X21);
              contract2(2, 2, X22);
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                for (int b1 = 0; b1 < NB; ++b1) {
                  const double wB = B1[b1][q1] * X22[q1] + D1[b1][q1] * X21[q1];
                  const double wD = B1[b1][q1] * X12[q1] + D1[b1][q1] * X11[q1];
#pragma unroll
                  for (int a1 = 0; a1 < NB; ++a1) accA[a1 * NB + b1] += B1[a1][q1] * wB + D1[a1][q1] * wD;
                }
              // g2: n == 0, m in {1,2}:  acc += (B1[a1] X20 + D1[a1] X10) B1[b1]
              double X10[NQ], X20[NQ];
              contract2(1, 0, X10);
              contract2(2, 0, X20);
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                for (int a1 = 0; a1 < NB; ++a1) {
                  const double wa = B1[a1][q1] * X20[q1] + D1[a1][q1] * X10[q1];
#pragma unroll
                  for (int b1 = 0; b1 < NB; ++b1) accB[a1 * NB + b1] += wa * B1[b1][q1];
                }
            } else {
              // g1: m == 0, n in {1,2}:  acc += B1[a1] (B1[b1] X02 + D1[b1] X01)
              double X01[NQ], X02[NQ];
              contract2(0, 1, X01);
              contract2(0, 2, X02);
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                for (int b1 = 0; b1 < NB; ++b1) {
                  const double wB = B1[b1][q1] * X02[q1] + D1[b1][q1] * X01[q1];
#pragma unroll
                  for (int a1 = 0; a1 < NB; ++a1) accA[a1 * NB + b1] += B1[a1][q1] * wB;
                }
              // g3: m == n == 0
              double X00[NQ];
              contract2(0, 0, X00);
#pragma unroll
              for (int q1 = 0; q1 < NQ; ++q1)
#pragma unroll
                for (int b1 = 0; b1 < NB; ++b1) {
                  const double wB = B1[b1][q1] * X00[q1];
#pragma unroll
                  for (int a1 = 0; a1 < NB; ++a1) accB[a1 * NB + b1] += B1[a1][q1] * wB;
                }
            }
            // ZB[h][q0][c], c = ((a2*NB + a1)*NB + b2)*NB + b1 ; h = 0: b-side B0, h = 1: b-side D0
#pragma unroll
            for (int a1 = 0; a1 < NB; ++a1)
#pragma unroll
              for (int b1 = 0; b1 < NB; ++b1) {
                const int c = ((a2 * NB + a1) * NB + b2) * NB + b1;
                ZB[(0 * NQ + q0) * NC + c] = accA[a1 * NB + b1];
                ZB[(1 * NQ + q0) * NC + c] = accB[a1 * NB + b1];
              }
          }
          __builtin_amdgcn_wave_barrier();
          // ---- S3 partial: lane = (b0, c):  K[a0] += T0a[a0] (B0[b0] Z_h0 + D0[b0] Z_h1) ----------
#pragma unroll
          for (int rd = 0; rd < S3R; ++rd) {
            const int t = rd * 64 + lane;
            const int tt = t < NB * NC ? t : 0;
            const int b0 = tt % NB, c = tt / NB;
#pragma unroll
            for (int q0 = 0; q0 < NQ; ++q0) {
              const double wv = tab_ptr<P>(tab, 0, 0)[b0 * NQ + q0] * ZB[(0 * NQ + q0) * NC + c]
                             + tab_ptr<P>(tab, 0, 1)[b0 * NQ + q0] * ZB[(1 * NQ + q0) * NC + c];
#pragma unroll
              for (int a0 = 0; a0 < NB; ++a0) Kt[rd][a0] += tab_ptr<P>(tab, 0, pass)[a0 * NQ + q0] * wv;
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        if (use_carry) {
          // incoming: the previous element's (a2+1, b2+1) entries are this element's (a2, b2)
          double* CBj = CB + j * NB * L::n_carry;
#pragma unroll
          for (int rd = 0; rd < S3R; ++rd) {
            const int t = rd * 64 + lane;
            const int tt = t < NB * NC ? t : 0;
            const int b0 = tt % NB, c = tt / NB;
            const int b1 = c % NB, b2 = (c / NB) % NB, a1 = (c / NB2) % NB, a2 = c / (NB2 * NB);
            if (t < NB * NC && es > 0 && a2 <= NB - 2 && b2 <= NB - 2) {
              const int u = (((a2 * NB + a1) * (NB - 1) + b2) * NB + b1) * NB + b0;
#pragma unroll
              for (int a0 = 0; a0 < NB; ++a0) Kt[rd][a0] += CBj[a0 * L::n_carry + u];
            }
          }
          __builtin_amdgcn_wave_barrier();
          // outgoing
#pragma unroll
          for (int rd = 0; rd < S3R; ++rd) {
            const int t = rd * 64 + lane;
            const int tt = t < NB * NC ? t : 0;
            const int b0 = tt % NB, c = tt / NB;
            const int b1 = c % NB, b2 = (c / NB) % NB, a1 = (c / NB2) % NB, a2 = c / (NB2 * NB);
            if (t < NB * NC && a2 >= 1 && b2 >= 1 && es + 1 < n_seq) {
              const int u = ((((a2 - 1) * NB + a1) * (NB - 1) + (b2 - 1)) * NB + b1) * NB + b0;
#pragma unroll
              for (int a0 = 0; a0 < NB; ++a0) CBj[a0 * L::n_carry + u] = Kt[rd][a0];
            }
          }
          __builtin_amdgcn_wave_barrier();
        }
        // keep the block in registers without dynamic indexing (the j loop stays rolled)
#pragma unroll
        for (int rd = 0; rd < S3R; ++rd)
#pragma unroll
          for (int a0 = 0; a0 < NB; ++a0)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) Kv[jj][rd][a0] = (jj == j) ? Kt[rd][a0] : Kv[jj][rd][a0];
      }
      // ---- CSR read-modify-write: adds and stores
#pragma unroll
      for (int rd = 0; rd < S3R; ++rd)
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0)
          if (dstp[rd][a0]) {
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) dstp[rd][a0][jj] = old[rd][a0][jj] + p.grad_factor * Kv[jj][rd][a0];
          }
    }
  }
}

template<int P, int GRAD>
__global__ __launch_bounds__(64) void tensor_domain_kernel(TensorArgs p) {
  extern __shared__ __align__(16) double smem_t[];
  double* lds = smem_t;
  const int comp = blockIdx.x % 3, unit = blockIdx.x / 3;
  const int ku = unit % p.n_units_u, kv = unit / p.n_units_u;
  const int eu = p.colour_u + (P + 1) * ku, ev = p.colour_v + (P + 1) * kv;
  int status = 0;
  if (comp == 0) tensor_wave_body<P, GRAD, 0>(p, lds, eu, ev, status);
  else if (comp == 1) tensor_wave_body<P, GRAD, 1>(p, lds, eu, ev, status);
  else tensor_wave_body<P, GRAD, 2>(p, lds, eu, ev, status);
  if (status) atomicOr(p.status, status);
}

inline bool tensor_supported(int dim, const int* degree, int nq) {
  // 3-D degree 2 (kernels_tensor_wgs*.hpp), 3-D degree 3 (tensor_p3.hip); 2-D degree 1..3 and 3-D degree 1
  // (kernels_tensor_small.hpp)
  for (int d = 1; d < dim; ++d)
    if (degree[d] != degree[0]) return false;
  if (nq != degree[0] + 2) return false;
  if (dim == 3) return degree[0] >= 1 && degree[0] <= 3;
  return dim == 2 && degree[0] >= 1 && degree[0] <= 3;
}

inline TensorArgs tensor_args(mimi_hip_domain_s* h, const double* u, double* r, double* A, double gf) {
  TensorArgs a{};
  for (int d = 0; d < 3; ++d) {
    a.box_begin[d] = h->el_begin[d];
    a.box_n[d] = h->el_end[d] - h->el_begin[d];
    a.tabB[d] = h->tab1d.ptr + h->tab_off_B[d];
    a.tabD[d] = h->tab1d.ptr + h->tab_off_D[d];
  }
  for (int d = 0; d < 3; ++d) {
    if (h->phase_select == 2) {
      a.win_begin[d] = h->gather_begin[d];
      a.win_n[d] = h->gather_end[d] - h->gather_begin[d];
    } else {
      a.win_begin[d] = a.box_begin[d];
      a.win_n[d] = a.box_n[d] + h->degree[d];
    }
  }
  a.seg_len = a.box_n[2];   // whole columns unless the launcher cuts them (launch_tensor_wgsym)
  // walk along the shortest axis (most units), colour over the other two
  int seq = 0;
  for (int d = 1; d < 3; ++d)
    if (a.box_n[d] <= a.box_n[seq]) seq = d;
  a.seq_axis = seq;
  a.u_axis = seq == 0 ? 1 : 0;
  a.v_axis = seq == 2 ? 1 : 2;
  a.geo = h->geo.ptr;
  a.dofs = h->dofs.ptr;
  a.pair_pos = h->pair_pos.ptr;
  a.structured = h->structured_csr ? 1 : 0;
  for (int d = 0; d < 3; ++d) {
    a.n_ctrl[d] = h->n_ctrl[d];
    a.first[d] = h->first1d.ptr + h->first_off[d];
  }
  a.rowptr = h->rowptr;
  a.u = u;
  a.r = r;
  a.A = A;
  a.A_base = (h->A_base && A) ? h->A_base : A;
  a.grad_factor = gf;
  a.dt = h->dt;
  a.mat = h->mat;
  a.state = StateView{h->eqps.ptr, h->temperature.ptr, h->plastic_strain.ptr, h->n_pts, h->state2.ptr};
  a.status = h->status_dev;
  a.perm = h->structured_perm ? h->node_ids.ptr : nullptr;
  a.nbr_pos = h->structured_perm ? h->nbr_pos.ptr : nullptr;
  a.nbr_pos16 = h->structured_perm ? h->nbr_pos16.ptr : nullptr;
  return a;
}

template<int P>
inline void launch_tensor_p(mimi_hip_domain_s* h, int grad, TensorArgs a) {
  const size_t lds = TensorLds<P>::total * sizeof(double);
  auto kernel = grad ? tensor_domain_kernel<P, 1> : tensor_domain_kernel<P, 0>;
  if (lds > 64 * 1024)
    ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
  if (grad) consume_base(h, a.A);     // (read-modify-write in place: the base array, if any, is copied first)
  const int nu = a.box_n[a.u_axis], nv = a.box_n[a.v_axis];
  for (int cv = 0; cv < P + 1; ++cv)
    for (int cu = 0; cu < P + 1; ++cu) {
      a.colour_u = cu;
      a.colour_v = cv;
      a.n_units_u = cu < nu ? (nu - cu + P) / (P + 1) : 0;
      a.n_units_v = cv < nv ? (nv - cv + P) / (P + 1) : 0;
      const int n_units = a.n_units_u * a.n_units_v;
      if (n_units == 0) continue;
      hipLaunchKernelGGL(kernel, dim3(n_units * 3), dim3(64), lds, h->stream, a);
      MH_HIP(hipGetLastError());
    }
}

}  // namespace mimi_hip
