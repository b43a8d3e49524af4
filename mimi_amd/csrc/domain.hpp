// Host-side state of one domain integrator handle (= one integrators::NonlinearSolid,
// integrators/nonlinear_solid.hpp:15-50, bound to one HIP device and stream).
#pragma once

#include "bspline_host.hpp"
#include "common.hpp"

struct mimi_hip_domain_s {
  int device = 0;
  int dim = 0, n_el = 0, n_dof = 0, n_q = 0;
  int64_t n_nodes = 0, n_vdofs = 0, nnz = 0, n_pts = 0;
  int path = 0;  // 0 general tables, 1 tensor-product (sum factorisation)
  hipStream_t own_stream = nullptr, stream = nullptr;

  mimi_hip::MaterialDev mat{};
  double dt = 0.0, first_effective_dt = 0.0, second_effective_dt = 0.0;
  int tangent_mode = MIMI_HIP_TANGENT_ANALYTIC;

  // general path tables
  mimi_hip::DeviceBuffer<int32_t> dofs;      // [n_el][n_dof]
  mimi_hip::DeviceBuffer<double> dN_dX;      // [n_el][n_q][dim][n_dof]
  mimi_hip::DeviceBuffer<double> wdet;       // [n_el][n_q]
  mimi_hip::DeviceBuffer<int32_t> pair_pos;  // [n_el][n_dof][n_dof]
  mimi_hip::DeviceBuffer<int64_t> rowptr_own;
  mimi_hip::DeviceBuffer<int64_t> adj_ptr;   // two-phase general path: node -> incident (element, local index) list
  mimi_hip::DeviceBuffer<int32_t> adj;
  bool general_two_phase_failed = false;
  const int64_t* rowptr = nullptr;           // device

  // tensor path tables
  int degree[3] = {0, 0, 0}, nq1[3] = {1, 1, 1}, n_ctrl[3] = {1, 1, 1};
  int el_begin[3] = {0, 0, 0}, el_end[3] = {1, 1, 1}, el_total[3] = {1, 1, 1};
  mimi_hip::DeviceBuffer<double> tab1d;      // per direction B then D: [n_spans][p+1][nq]
  mimi_hip::DeviceBuffer<int32_t> first1d;   // per direction [n_spans]
  size_t tab_off_B[3] = {0, 0, 0}, tab_off_D[3] = {0, 0, 0}, tab_off_W[3] = {0, 0, 0}, first_off[3] = {0, 0, 0};
  mimi_hip::DeviceBuffer<double> geo;        // [n_el][dim*dim+1][n_q]: dxi/dX (d,J) then w*det
  mimi_hip::DeviceBuffer<int64_t> node_ids;  // lexicographic -> global, empty = identity
  bool structured_csr = false;               // CSR positions computable arithmetically
  bool structured_perm = false;              // permuted numbering whose CSR rows are the permuted structured pattern
  mimi_hip::DeviceBuffer<unsigned char> nbr_pos;  // [n_nodes][125] rank of each window neighbour inside the row (structured_perm, p <= 2)
  mimi_hip::DeviceBuffer<uint16_t> nbr_pos16;     // [n_nodes][343] the same for degree 3
  bool first_is_identity = false;            // span e's first basis function is e (no repeated interior knots)
  mimi_hip::DeviceBuffer<double> scratch_k, scratch_r, scratch_pt, scratch_tail;  // two-phase tangent path
  mimi_hip::DeviceBuffer<double> mat_rec;    // general path, other materials' tangent assemblies: [n_el][n_q][DIM^2 + DIM^4] (kernels_general.hpp)
  mimi_hip::DeviceBuffer<double> t3_t2pack;  // degree-3 contraction: direction-2 table values per (span, lane), [n_spans][64][8] (tensor_p3.hip)

  // J2 state, SoA over points
  mimi_hip::DeviceBuffer<double> eqps, temperature, plastic_strain, state2;

  // status word raised by kernels (ScalarSolve failures, bad pattern)
  int* status_dev = nullptr;
  int* status_host = nullptr;  // pinned
  // mimi_hip_domain_set_phase_timing: events around phase 1 (integration kernels) and phase 2 (gather) of the last
  // two-phase assembly, on the launch stream
  bool phase_timing = false;
  hipEvent_t phase_ev[4] = {nullptr, nullptr, nullptr, nullptr};   // [3]: end of the material pre-pass, when there is one
  bool phase_has_prepass = false;

  // mimi_hip_domain_integrate / _gather: 0 = an assembly runs both phases, 1 = phase 1 only, 2 = phase 2 only, over the
  // node window [gather_begin, gather_end)
  int phase_select = 0;
  int gather_begin[3] = {0, 0, 0}, gather_end[3] = {0, 0, 0};
  bool integrated = false;
  // mimi_hip_domain_add_residual_and_grad_from: the array the row gathers read the old values from during this call
  // (device; nullptr = the output array itself, the plain "+=").  Paths without a row gather take it by a copy
  // (consume_base below).
  const double* A_base = nullptr;
  // kernel family of the last assembly / state commit on this handle (mimi_hip_domain_info(h, 7)): 0 none yet,
  // 1 two-phase tensor degree 2, 2 two-phase tensor degree 3, 3 small-element tensor kernel, 4 general kernels
  int last_family = 0;

  // staging for host-resident u / r / A
  mimi_hip::DeviceBuffer<double> stage_u, stage_r, stage_A;

  ~mimi_hip_domain_s();
};

namespace mimi_hip {
// for an assembly route that adds into the value array in place (colour kernel, atomics): A <- A_base first, then "+="
inline void consume_base(mimi_hip_domain_s* h, double* A) {
  if (h->A_base && A && h->A_base != A)
    MH_HIP(hipMemcpyAsync(A, h->A_base, (size_t)h->nnz * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  h->A_base = nullptr;
}
}  // namespace mimi_hip
