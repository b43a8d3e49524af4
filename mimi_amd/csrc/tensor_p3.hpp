// Two-phase sum-factorised assembly for structured 3-D patches of degree 3 (tensor_p3.hip).
#pragma once

#include "kernels_tensor.hpp"

namespace mimi_hip {

// the structured CSR pattern in lexicographic or permuted (node_ids) numbering, uniform degree 3, 5 Gauss points per
// direction, no repeated interior knots
bool tensor_p3_ready(const mimi_hip_domain_s* h);
// grad 0: r += R(u); 1: also A += grad_factor K(u)
void launch_tensor_p3(mimi_hip_domain_s* h, int grad, TensorArgs a);
// DomainPostTimeAdvance on the same handle: the pre-pass kernel in its commit mode
void launch_tensor_p3_post(mimi_hip_domain_s* h, TensorArgs a);

}  // namespace mimi_hip
