// Tensor-product (sum-factorised) domain kernels -- see DESIGN.md.
#pragma once

#include "domain.hpp"
#include "materials.hpp"

namespace mimi_hip {

inline bool tensor_supported(int dim, const int* degree, int nq) {
  (void)dim; (void)degree; (void)nq;
  return false;
}

inline void launch_tensor(mimi_hip_domain_s*, int, const double*, double*, double*, double) {
  fail("tensor path not available");
}

inline void launch_tensor_post(mimi_hip_domain_s*, const double*) { fail("tensor path not available"); }

}  // namespace mimi_hip
