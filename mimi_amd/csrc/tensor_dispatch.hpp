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

// can this handle's assembly run on the tensor kernels?  (3-D degree 2 and 3 have the two-phase kernels only: a patch
// whose CSR is not the structured pattern, or with repeated interior knots, takes the general kernels)
inline bool tensor_usable(const mimi_hip_domain_s* h) {
  if (h->path != 1) return false;
  if (tensor_small_shape(h->dim, h->degree, h->nq1[0])) return true;
  return h->degree[0] == 3 ? tensor_p3_ready(h) : two_phase_supported(h);
}

// returns the kernel family that ran (mimi_hip_domain_s::last_family)
inline int launch_tensor(mimi_hip_domain_s* h, int grad, const double* u, double* r, double* A, double gf) {
  TensorArgs a = tensor_args(h, u, r, A, gf);
  if (h->degree[0] == 3) {
    launch_tensor_p3(h, grad, a);
    return 2;
  }
  // MIMI_HIP_TENSOR_VARIANT=wgs: the full nine-block kernel also for hyperelastic materials (default: symmetric half)
  static const char* variant = getenv("MIMI_HIP_TENSOR_VARIANT");
  if (grad) {
    const bool want_full = variant && variant[0] == 'w';
    if (h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN && !want_full) launch_tensor_wgsym(h, a);
    else launch_tensor_wgs(h, a);
  } else {
    launch_tensor_residual(h, a);
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
