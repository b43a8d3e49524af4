// Dispatch of a tensor-path assembly to its kernel family (included by domain.hip after every kernel header).
#pragma once

#include "kernels_tensor.hpp"
#include "kernels_tensor_small.hpp"
#include "tensor_p3.hpp"

namespace mimi_hip {

inline bool two_phase_supported(const mimi_hip_domain_s* h);                     // kernels_tensor_2phase.hpp
inline void launch_tensor_wgs(mimi_hip_domain_s* h, TensorArgs a);               // kernels_tensor_wgs.hpp
inline void launch_tensor_wgsym(mimi_hip_domain_s* h, TensorArgs a);             // kernels_tensor_wgsym.hpp
inline void launch_tensor_residual(mimi_hip_domain_s* h, TensorArgs a);          // kernels_tensor_residual.hpp
inline void launch_tensor_p2_post(mimi_hip_domain_s* h, TensorArgs a);           // kernels_tensor_wgs.hpp
static void ensure_pair_pos(mimi_hip_domain_s* h);                                // domain.hip

// can this handle's assembly run on the tensor kernels?  (p = 3 has the two-phase kernels only)
inline bool tensor_usable(const mimi_hip_domain_s* h) {
  return h->path == 1 && (tensor_small_shape(h->dim, h->degree, h->nq1[0]) || h->degree[0] != 3 || tensor_p3_ready(h));
}

// returns the kernel family that ran (mimi_hip_domain_s::last_family)
inline int launch_tensor(mimi_hip_domain_s* h, int grad, const double* u, double* r, double* A, double gf) {
  TensorArgs a = tensor_args(h, u, r, A, gf);
  if (h->degree[0] == 3) {
    launch_tensor_p3(h, grad, a);
    return 2;
  }
  // MIMI_HIP_TENSOR_VARIANT: default when supported = two-phase with role-specialised workgroups, the
  // symmetric-half kernel for hyperelastic materials; "wgs" forces the full nine-block kernel,
  // "valu" the colour-partitioned read-modify-write kernel
  static const char* variant = getenv("MIMI_HIP_TENSOR_VARIANT");
  const bool closed_form = h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN || h->mat.m.kind == MIMI_HIP_MAT_J2;
  const bool want_valu = variant && variant[0] == 'v' && closed_form;   // the colour kernel has the closed-form materials only
  if (grad && !want_valu && two_phase_supported(h)) {
    const bool want_full = variant && variant[0] == 'w';
    if (h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN && !want_full) launch_tensor_wgsym(h, a);
    else launch_tensor_wgs(h, a);
  }
  else if (!grad && !want_valu && two_phase_supported(h))
    launch_tensor_residual(h, a);
  else {
    if (grad) {
      ensure_pair_pos(h);       // domain.hip: the colour kernel scatters the tangent through the pair-position table
      a.pair_pos = h->pair_pos.ptr;
    }
    launch_tensor_p<2>(h, grad, a);
    return 5;
  }
  return 1;
}

inline void launch_tensor_post(mimi_hip_domain_s* h, const double* u) {
  TensorArgs a = tensor_args(h, u, nullptr, nullptr, 0.0);
  if (h->degree[0] == 3) {
    launch_tensor_p3_post(h, a);
    return;
  }
  launch_tensor_p2_post(h, a);
}

}  // namespace mimi_hip
