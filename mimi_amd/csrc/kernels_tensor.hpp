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
// This header holds what the kernel families of the tensor path share: their argument block (TensorArgs), the 1-D table
// addressing and the dispatch predicates.  The kernels: kernels_tensor_wgsym.hpp / kernels_tensor_wgs.hpp (degree 2, phase 1),
// kernels_tensor_2phase.hpp (degree 2, phase 2), kernels_tensor_residual.hpp (degree 2, residual only), tensor_p3.hip
// (degree 3), kernels_tensor_small.hpp (2-D, 3-D degree 1).  The colour-partitioned read-modify-write kernel of round 1
// (one wave per (column, i), (P + 1)^2 launches; the fallback for a degree-2 patch whose CSR is not the structured
// pattern, 938 spilled registers) was removed in round 5: such patches take the general kernels (kernels_general.hpp),
// whose store + row gather has no atomics either.
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

template<int P>
MH_DEV const double* tab_ptr(const double* tab, int dir, int isD) {
  return tab + ((dir * 2 + isD) * (P + 1)) * (P + 2);
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

}  // namespace mimi_hip
