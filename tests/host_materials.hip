// TEST HARNESS ONLY: the device constitutive routines of mimi_amd/csrc (materials.hpp, materials_other.hpp) compiled
// for the HOST, so that their arithmetic can be checked against the oracle point by point without a GPU
// (tests/test_materials_host_cpu.py).  Nothing in mimi_amd builds or loads this.
#define MH_DEV __host__ __device__ inline
#include "../mimi_amd/csrc/materials_other.hpp"

using namespace mimi_hip;

// state arrays address ONE point (SoA with n_pts = 1 = plain column-major matrices)
extern "C" int host_point(const mimi_hip_material* m, double sigma_y_ref, int dim, double dt, const double* F, double* m1,
                          double* m2, double eqps, double T, double* P, double* A) {
  MaterialDev md{};
  md.m = *m;
  md.const_temperature_contribution = 1.0;
  md.sigma_y_ref = sigma_y_ref;
  StateView sv{&eqps, &T, m1, 1, m2};
  if (m->kind == MIMI_HIP_MAT_NEOHOOKEAN || m->kind == MIMI_HIP_MAT_J2) {
    int status;
    if (dim == 2) {
      PointResult<2> w;
      status = evaluate_pk1<2>(md, dt, sv, 0, F, w);
      for (int k = 0; k < 4; ++k) P[k] = w.P[k];
      tangent_of<2>(md.m, w, A);
    } else {
      PointResult<3> w;
      status = evaluate_pk1<3>(md, dt, sv, 0, F, w);
      for (int k = 0; k < 9; ++k) P[k] = w.P[k];
      tangent_of<3>(md.m, w, A);
    }
    return status;
  }
  if (dim == 2) return evaluate_other<2>(md, dt, sv, 0, F, P, A, 1.0);
  return evaluate_other<3>(md, dt, sv, 0, F, P, A, 1.0);
}

// DomainPostTimeAdvance at one point; state updated in place
extern "C" int host_accumulate(const mimi_hip_material* m, double sigma_y_ref, int dim, double dt, const double* F,
                               double* m1, double* m2, double* eqps, double* T) {
  MaterialDev md{};
  md.m = *m;
  md.const_temperature_contribution = 1.0;
  md.sigma_y_ref = sigma_y_ref;
  StateView sv{eqps, T, m1, 1, m2};
  if (m->kind == MIMI_HIP_MAT_NEOHOOKEAN || m->kind == MIMI_HIP_MAT_J2)
    return dim == 2 ? accumulate_state<2>(md, dt, sv, 0, F) : accumulate_state<3>(md, dt, sv, 0, F);
  return dim == 2 ? accumulate_other<2>(md, dt, sv, 0, F) : accumulate_other<3>(md, dt, sv, 0, F);
}

// x^q for x > 0 as the return-map Newton evaluates it (materials.hpp pow_positive)
extern "C" void host_pow_positive(int n, const double* x, const double* q, double* out) {
  for (int k = 0; k < n; ++k) out[k] = pow_positive(x[k], q[k]);
}

// x^q for any x (materials.hpp pow_any: what the hardening laws call -- no library pow behind it)
extern "C" void host_pow_any(int n, const double* x, const double* q, double* out) {
  for (int k = 0; k < n; ++k) out[k] = pow_any(x[k], q[k]);
}
