// C ABI of the contact integrator (integrators::MortarContact) -- implementation pending.
#include "common.hpp"

using namespace mimi_hip;

static int not_yet() {
  set_last_error("mimi_hip contact integrator: not implemented yet");
  return 1;
}

extern "C" {
int mimi_hip_contact_create(const mimi_hip_contact_tables*, int, mimi_hip_contact_t*) { return not_yet(); }
int mimi_hip_contact_destroy(mimi_hip_contact_t) { return not_yet(); }
int mimi_hip_contact_set_tangent_mode(mimi_hip_contact_t, int) { return not_yet(); }
int mimi_hip_contact_set_stream(mimi_hip_contact_t, void*) { return not_yet(); }
int mimi_hip_contact_synchronize(mimi_hip_contact_t) { return not_yet(); }
int mimi_hip_contact_add_residual(mimi_hip_contact_t, const double*, double*) { return not_yet(); }
int mimi_hip_contact_add_residual_and_grad(mimi_hip_contact_t, const double*, double, double*, double*) { return not_yet(); }
int mimi_hip_contact_gap_norm(mimi_hip_contact_t, const double*, double*) { return not_yet(); }
int mimi_hip_contact_last_history(mimi_hip_contact_t, double*) { return not_yet(); }
int mimi_hip_contact_get_pressure(mimi_hip_contact_t, double*, int64_t, int64_t*) { return not_yet(); }
}
