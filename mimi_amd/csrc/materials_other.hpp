// The reference's remaining materials on the device: StVenantKirchhoff, J2Linear, J2Simo, J2Log.
//
// Follows the arithmetic of (paths under /root/reference/src/mimi/):
//   StVenantKirchhoff::EvaluatePK1                     materials/materials.cpp:72-94
//   J2Linear::PlasticStress<accumulate>                materials/materials.hpp:185-236
//   J2Simo::PlasticStress<accumulate>                  materials/materials.hpp:452-545
//   J2Log::PlasticStress<accumulate>                   materials/materials.hpp:592-713, under the base
//       MaterialBase::EvaluatePK1 (materials.cpp:60-71) -- which takes alternative_stress_ = s + (p / det F) I as the
//       Cauchy stress and overwrites what PlasticStress left in tmp.stress_: P = det F (s + p/det F I) F^-T.  The
//       reference's golden series j2_log_h1_p2 is reproduced by exactly this, so it is kept.
//   LogarithmicStrain / Dev / Norm                     materials/material_utils.hpp:22-56,91-127
//   state creation                                     materials/materials.cpp:120-133,185-208,225-252
//
// The reference has no tangent for any of them (it differentiates the element residual numerically).  Here every stress
// routine is written once over a scalar type T: T = double gives the stress, T = Dual (value + one directional
// derivative) gives dP/dF in one direction -- DIM^2 passes build the consistent tangent.  The return-map increment is
// differentiated implicitly (residual(delta; F) = 0), the matrix logarithm by the Daleckii-Krein formula on the
// eigen-decomposition of the value part.
#pragma once

#include <type_traits>

#include "materials.hpp"

namespace mimi_hip {

// ---- Dual arithmetic ---------------------------------------------------------------------------
MH_DEV Dual operator+(Dual a, Dual b) { return Dual{a.v + b.v, a.d + b.d}; }
MH_DEV Dual operator-(Dual a, Dual b) { return Dual{a.v - b.v, a.d - b.d}; }
MH_DEV Dual operator*(Dual a, Dual b) { return Dual{a.v * b.v, a.d * b.v + a.v * b.d}; }
MH_DEV Dual operator/(Dual a, Dual b) {
  const double q = a.v / b.v;
  return Dual{q, (a.d - q * b.d) / b.v};
}
MH_DEV Dual operator-(Dual a) { return Dual{-a.v, -a.d}; }
MH_DEV Dual operator+(Dual a, double b) { return Dual{a.v + b, a.d}; }
MH_DEV Dual operator+(double a, Dual b) { return Dual{a + b.v, b.d}; }
MH_DEV Dual operator-(Dual a, double b) { return Dual{a.v - b, a.d}; }
MH_DEV Dual operator-(double a, Dual b) { return Dual{a - b.v, -b.d}; }
MH_DEV Dual operator*(Dual a, double b) { return Dual{a.v * b, a.d * b}; }
MH_DEV Dual operator*(double a, Dual b) { return Dual{a * b.v, a * b.d}; }
MH_DEV Dual operator/(Dual a, double b) { return Dual{a.v / b, a.d / b}; }
MH_DEV Dual operator/(double a, Dual b) {
  const double q = a / b.v;
  return Dual{q, -q * b.d / b.v};
}
MH_DEV Dual& operator+=(Dual& a, Dual b) {
  a.v += b.v;
  a.d += b.d;
  return a;
}

MH_DEV double ad_v(double x) { return x; }
MH_DEV double ad_v(Dual x) { return x.v; }
MH_DEV double ad_sqrt(double x) { return sqrt(x); }
MH_DEV Dual ad_sqrt(Dual x) {
  const double s = sqrt(x.v);
  return Dual{s, s > 0.0 ? x.d / (2.0 * s) : 0.0};
}
MH_DEV double ad_cbrt(double x) { return cbrt(x); }
MH_DEV Dual ad_cbrt(Dual x) {
  const double c = cbrt(x.v);
  return Dual{c, x.d * c / (3.0 * x.v)};
}
template<class T>
MH_DEV T ad_from(double v);
template<>
MH_DEV double ad_from<double>(double v) { return v; }
template<>
MH_DEV Dual ad_from<Dual>(double v) { return Dual{v, 0.0}; }

// ---- small dense algebra over T (column-major DIM x DIM) ------------------------------------------
template<int DIM, class T>
MH_DEV T det_t(const T* F) {
  if constexpr (DIM == 2) {
    return F[0] * F[3] - F[1] * F[2];
  } else {
    return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[3] * (F[1] * F[8] - F[2] * F[7]) + F[6] * (F[1] * F[5] - F[2] * F[4]);
  }
}

template<int DIM, class T>
MH_DEV void inverse_t(const T* F, T det, T* Fi) {
  const T t = 1.0 / det;
  if constexpr (DIM == 2) {
    Fi[0] = F[3] * t;
    Fi[1] = -F[1] * t;
    Fi[2] = -F[2] * t;
    Fi[3] = F[0] * t;
  } else {
    Fi[0] = (F[4] * F[8] - F[5] * F[7]) * t;
    Fi[1] = (F[2] * F[7] - F[1] * F[8]) * t;
    Fi[2] = (F[1] * F[5] - F[2] * F[4]) * t;
    Fi[3] = (F[5] * F[6] - F[3] * F[8]) * t;
    Fi[4] = (F[0] * F[8] - F[2] * F[6]) * t;
    Fi[5] = (F[2] * F[3] - F[0] * F[5]) * t;
    Fi[6] = (F[3] * F[7] - F[4] * F[6]) * t;
    Fi[7] = (F[1] * F[6] - F[0] * F[7]) * t;
    Fi[8] = (F[0] * F[4] - F[1] * F[3]) * t;
  }
}

// C = op(A) op(B): TA / TB transpose the operand
template<int DIM, bool TA, bool TB, class TL, class TR, class T>
MH_DEV void mat_mul_t(const TL* A, const TR* B, T* C) {
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      T s = ad_from<T>(0.0);
#pragma unroll
      for (int k = 0; k < DIM; ++k) s = s + (TA ? MH_M(A, k, i) : MH_M(A, i, k)) * (TB ? MH_M(B, j, k) : MH_M(B, k, j));
      MH_M(C, i, j) = s;
    }
}

template<int DIM, class T>
MH_DEV T trace_t(const T* A) {
  T tr = MH_M(A, 0, 0);
#pragma unroll
  for (int i = 1; i < DIM; ++i) tr = tr + MH_M(A, i, i);
  return tr;
}

// material_utils.hpp:22-56 Dev(A, dim, factor): the trace is divided by dim
template<int DIM, class T>
MH_DEV void dev_t(const T* A, double factor, T* out) {
  const T tr_over_dim = trace_t<DIM>(A) / (double)DIM;
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) out[i] = A[i] * factor;
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(out, i, i) = (MH_M(A, i, i) - tr_over_dim) * factor;
}

template<int DIM, class T>
MH_DEV T norm_t(const T* A) {
  T a = A[0] * A[0];
#pragma unroll
  for (int i = 1; i < DIM * DIM; ++i) a = a + A[i] * A[i];
  return ad_sqrt(a);
}

// ---- symmetric eigen-decomposition (mfem::DenseMatrix::CalcEigenvalues in the reference): cyclic Jacobi -----------
template<int DIM>
MH_DEV void sym_eig(const double* A, double* lam, double* Q) {
  double a[DIM * DIM];
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) a[i] = A[i];
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) MH_M(Q, i, j) = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int j = 0; j < DIM; ++j) {
        if (i != j) off += MH_M(a, i, j) * MH_M(a, i, j);
        else diag += MH_M(a, i, i) * MH_M(a, i, i);
      }
    if (off <= 1e-34 * diag || off == 0.0) break;
#pragma unroll
    for (int p = 0; p < DIM - 1; ++p)
#pragma unroll
      for (int q = p + 1; q < DIM; ++q) {
        const double apq = MH_M(a, p, q);
        if (apq == 0.0) continue;
        const double theta = (MH_M(a, q, q) - MH_M(a, p, p)) / (2.0 * apq);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const double akp = MH_M(a, k, p), akq = MH_M(a, k, q);
          MH_M(a, k, p) = c * akp - sn * akq;
          MH_M(a, k, q) = sn * akp + c * akq;
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const double apk = MH_M(a, p, k), aqk = MH_M(a, q, k);
          MH_M(a, p, k) = c * apk - sn * aqk;
          MH_M(a, q, k) = sn * apk + c * aqk;
        }
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          const double qkp = MH_M(Q, k, p), qkq = MH_M(Q, k, q);
          MH_M(Q, k, p) = c * qkp - sn * qkq;
          MH_M(Q, k, q) = sn * qkp + c * qkq;
        }
      }
  }
#pragma unroll
  for (int i = 0; i < DIM; ++i) lam[i] = MH_M(a, i, i);
}

// out = Q diag(f) Q^T  (mfem::MultADAt)
template<int DIM>
MH_DEV void q_diag_qt(const double* Q, const double* f, double* out) {
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      double t = 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) t += MH_M(Q, i, k) * f[k] * MH_M(Q, j, k);
      MH_M(out, i, j) = t;
    }
}

template<int DIM>
MH_DEV void sym_exp(const double* A, double* out) {
  double lam[DIM], Q[DIM * DIM];
  sym_eig<DIM>(A, lam, Q);
#pragma unroll
  for (int i = 0; i < DIM; ++i) lam[i] = exp(lam[i]);
  q_diag_qt<DIM>(Q, lam, out);
}

// What the value pass (T = double) of a point leaves for its derivative passes (T = Dual): the return-map increment
// (solved once) and the eigen-decomposition of the logarithmic strain's argument.
struct OtherCache {
  double delta, hprime;
  bool plastic;
  double lam[3], Q[9];
};

// E = 1/2 log(C), C symmetric positive definite (material_utils.hpp:91-114 LogarithmicStrain)
template<int DIM>
MH_DEV void half_log_sym(const double* C, double* E, OtherCache& cc) {
  double f[DIM];
  sym_eig<DIM>(C, cc.lam, cc.Q);
#pragma unroll
  for (int i = 0; i < DIM; ++i) f[i] = 0.5 * log(cc.lam[i]);
  q_diag_qt<DIM>(cc.Q, f, E);
}

// ... and its directional derivative: dE = Q [ (Q^T dC Q) o Gamma ] Q^T, Gamma_ab = (f(la) - f(lb)) / (la - lb)
template<int DIM>
MH_DEV void half_log_sym(const Dual* C, Dual* E, OtherCache& cc) {
  double Cv[DIM * DIM], dC[DIM * DIM], B[DIM * DIM], W[DIM * DIM], f[DIM];
  const double* lam = cc.lam;   // of the value part, from the value pass
  const double* Q = cc.Q;
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) dC[i] = C[i].d;
  mat_mul_t<DIM, true, false>(Q, dC, W);   // Q^T dC
  mat_mul_t<DIM, false, false>(W, Q, B);   // Q^T dC Q
#pragma unroll
  for (int a = 0; a < DIM; ++a)
#pragma unroll
    for (int b = 0; b < DIM; ++b) {
      const double x = (lam[a] - lam[b]) / lam[b];
      const double gamma = (a == b || x == 0.0) ? 1.0 / lam[b] : log1p(x) / (x * lam[b]);
      MH_M(B, a, b) *= 0.5 * gamma;
    }
  mat_mul_t<DIM, false, false>(Q, B, W);
  mat_mul_t<DIM, false, true>(W, Q, dC);   // Q B Q^T
#pragma unroll
  for (int i = 0; i < DIM; ++i) f[i] = 0.5 * log(lam[i]);
  q_diag_qt<DIM>(Q, f, Cv);
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) E[i] = Dual{Cv[i], dC[i]};
}

// ---- return map shared by J2Simo / J2Log: residual(delta) = a - b delta - H(eqps + delta) thermo rate(delta / dt) ----
MH_DEV double ad_implicit_delta(double delta, double, double, double, double, double) { return delta; }
MH_DEV Dual ad_implicit_delta(double delta, Dual a, Dual b, double hprime, double, double) {
  // d residual = da - db delta - (b + H') d delta = 0
  return Dual{delta, (a.d - b.d * delta) / (b.v + hprime)};
}

template<class T>
MH_DEV T return_map_increment(const MaterialDev& md, double dt, double eqps_old, double temperature, T a, T b,
                              bool& plastic, int& status, OtherCache& cc) {
  const mimi_hip_material& m = md.m;
  if constexpr (!std::is_same<T, double>::value) {
    // derivative pass: the increment of the value pass, differentiated implicitly
    plastic = cc.plastic;
    if (!plastic) return ad_from<T>(0.0);
    return ad_implicit_delta(cc.delta, a, b, cc.hprime, 0.0, 0.0);
  }
  ReturnMapCtx c{eqps_old, ad_v(a), thermo_contribution(md, temperature), dt, ad_v(b)};
  const double tolerance = md.sigma_y_ref * 1.e-10;
  plastic = false;
  const RmPoint at0 = rm_eval(m, c, 0.0);
  if (at0.R.v > tolerance) {
    const double upper = (c.q - at0.H.v * c.thermo) / c.slope;
    RmPoint sol;
    const double delta = scalar_solve(m, c, at0, 0.0, 0.0, upper, 1.e-10, tolerance, 100, status, sol);
    plastic = true;
    const Dual H = sol.H;            // (hardening and rate factor at the solution: what the last evaluation left)
    const double rc = sol.rc;
    const double hprime = H.d * rc * c.thermo + H.v * rate_contribution_derivative(m, delta / dt) / dt * c.thermo;
    cc.plastic = true;
    cc.delta = delta;
    cc.hprime = hprime;
    return ad_implicit_delta(delta, a, b, hprime, 0.0, 0.0);
  }
  cc.plastic = false;
  return ad_from<T>(0.0);
}

// ---- per-point state of these materials -------------------------------------------------------------------------
// first matrix  (StateView::plastic_strain): J2Linear plastic strain | J2Simo be_old | J2Log Fp_inv
// second matrix (StateView::state2)        : J2Linear beta           | J2Simo F_old
struct OtherState {
  double m1[9], m2[9], eqps, temperature;
};

template<int DIM>
MH_DEV void other_state_load(const MaterialDev& md, const StateView& st, int64_t pt, OtherState& s) {
  s.eqps = 0.0;
  s.temperature = 0.0;
  if (md.m.kind == MIMI_HIP_MAT_STVK) return;
#pragma unroll
  for (int c = 0; c < DIM * DIM; ++c) s.m1[c] = st.plastic_strain[c * st.n_pts + pt];
  if (md.m.kind != MIMI_HIP_MAT_J2LOG) {
#pragma unroll
    for (int c = 0; c < DIM * DIM; ++c) s.m2[c] = st.state2[c * st.n_pts + pt];
  }
  s.eqps = st.eqps[pt];
  s.temperature = st.temperature[pt];
}

template<int DIM>
MH_DEV void other_state_store(const MaterialDev& md, const StateView& st, int64_t pt, const OtherState& s) {
  if (md.m.kind == MIMI_HIP_MAT_STVK) return;
#pragma unroll
  for (int c = 0; c < DIM * DIM; ++c) st.plastic_strain[c * st.n_pts + pt] = s.m1[c];
  if (md.m.kind != MIMI_HIP_MAT_J2LOG) {
#pragma unroll
    for (int c = 0; c < DIM * DIM; ++c) st.state2[c * st.n_pts + pt] = s.m2[c];
  }
  st.eqps[pt] = s.eqps;
  st.temperature[pt] = s.temperature;
}

// ---- the stress routines, once over T ----------------------------------------------------------------------------
template<int DIM, class T>
MH_DEV void stvk_stress(const mimi_hip_material& m, const T* F, T* P) {
  T E[DIM * DIM], S[DIM * DIM];
  mat_mul_t<DIM, true, false>(F, F, E);   // C
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) E[i] = E[i] * 0.5;
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(E, i, i) = MH_M(E, i, i) - 0.5;
  const T ltr = trace_t<DIM>(E) * m.lambda;
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) S[i] = E[i] * (2 * m.mu);
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(S, i, i) = MH_M(S, i, i) + ltr;
  mat_mul_t<DIM, false, false>(F, S, P);
}

// P = det F sigma F^-T (materials.cpp:60-71)
template<int DIM, class T>
MH_DEV void pk1_from_cauchy_t(const T* sigma, const T* F, T* P) {
  const T det = det_t<DIM>(F);
  T Fi[DIM * DIM], tmp[DIM * DIM];
  inverse_t<DIM>(F, det, Fi);
  mat_mul_t<DIM, false, true>(sigma, Fi, tmp);
#pragma unroll
  for (int i = 0; i < DIM * DIM; ++i) P[i] = tmp[i] * det;
}

template<int DIM, bool ACCUMULATE, class T>
MH_DEV void j2linear_stress(const mimi_hip_material& m, const T* F, OtherState& st, T* P) {
  constexpr int DD = DIM * DIM;
  T eps[DD], s[DD], eta[DD];
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) MH_M(eps, i, j) = (MH_M(F, i, j) + MH_M(F, j, i)) * 0.5 - ((i == j ? 1.0 : 0.0) + MH_M(st.m1, i, j));
  const T p = trace_t<DIM>(eps) * m.K;
  dev_t<DIM>(eps, 2.0 * m.G, s);
#pragma unroll
  for (int i = 0; i < DD; ++i) eta[i] = s[i] - st.m2[i];
  const T eta_norm = norm_t<DIM>(eta);
  const T q = eta_norm * sqrt(3.0 / 2.0);
  const T phi = q - (m.sigma_y + m.lin_isotropic_hardening * st.eqps);
  if (ad_v(phi) > 0.) {
    const T inc = phi / (3. * m.G + m.lin_kinematic_hardening + m.lin_isotropic_hardening);
#pragma unroll
    for (int i = 0; i < DD; ++i) eta[i] = eta[i] / eta_norm;
    if constexpr (!ACCUMULATE) {
#pragma unroll
      for (int i = 0; i < DD; ++i) s[i] = s[i] - eta[i] * inc * (sqrt(6.0) * m.G);
    } else {
      st.eqps += ad_v(inc);
#pragma unroll
      for (int i = 0; i < DD; ++i) st.m1[i] += sqrt(3.0 / 2.0) * ad_v(inc) * ad_v(eta[i]);
#pragma unroll
      for (int i = 0; i < DD; ++i) st.m2[i] += sqrt(2.0 / 3.0) * m.lin_kinematic_hardening * ad_v(inc) * ad_v(eta[i]);
    }
  }
  if constexpr (!ACCUMULATE) {
#pragma unroll
    for (int i = 0; i < DIM; ++i) MH_M(s, i, i) = MH_M(s, i, i) + p;
    pk1_from_cauchy_t<DIM>(s, F, P);
  }
}

template<int DIM, bool ACCUMULATE, class T>
MH_DEV int j2simo_stress(const MaterialDev& md, double dt, const T* F, OtherState& st, T* P, OtherCache& cc) {
  const mimi_hip_material& m = md.m;
  constexpr int DD = DIM * DIM;
  int status = 0;
  const T detF = det_t<DIM>(F);
  T Finv[DD], w0[DD], f_bar[DD], be[DD], s[DD], Np[DD];
  inverse_t<DIM>(F, detF, Finv);
  mat_mul_t<DIM, false, false>(st.m2, Finv, w0);   // f_inv = F_old F^-1
  {
    const T d = det_t<DIM>(w0);
    inverse_t<DIM>(w0, d, f_bar);
    // materials.hpp:466-469: f_bar *= cbrt(det f_bar) -- multiplied, as written there
    const T c = ad_cbrt(det_t<DIM>(f_bar));
#pragma unroll
    for (int i = 0; i < DD; ++i) f_bar[i] = f_bar[i] * c;
  }
  mat_mul_t<DIM, false, false>(f_bar, st.m1, w0);   // f_bar be_old
  mat_mul_t<DIM, false, true>(w0, f_bar, be);
  dev_t<DIM>(be, m.G, s);
  const T s_norm = norm_t<DIM>(s);
  if (fabs(ad_v(s_norm)) < 2.220446049250313e-16) {
#pragma unroll
    for (int i = 0; i < DD; ++i) Np[i] = ad_from<T>(0.0);
#pragma unroll
    for (int i = 0; i < DIM; ++i) MH_M(Np, i, i) = ad_from<T>(sqrt(1. / 2.));
  } else {
#pragma unroll
    for (int i = 0; i < DD; ++i) Np[i] = s[i] * (sqrt(3. / 2.) / s_norm);
  }
  T s_effective = Np[0] * s[0];
#pragma unroll
  for (int i = 1; i < DD; ++i) s_effective = s_effective + Np[i] * s[i];
  const T be_trace = trace_t<DIM>(be);
  bool plastic;
  const T delta = return_map_increment<T>(md, dt, st.eqps, st.temperature, s_effective, be_trace * m.G, plastic, status, cc);
  if (plastic) {
#pragma unroll
    for (int i = 0; i < DD; ++i) be[i] = be[i] - Np[i] * (delta * be_trace * (2. / 3.));
    dev_t<DIM>(be, m.G, s);
    if constexpr (ACCUMULATE) {
      st.eqps += ad_v(delta);
      if (m.hardening == MIMI_HIP_HARD_JC_TEMP_RATE)
        st.temperature += m.heat_fraction * ad_v(s_effective) * ad_v(delta) / (m.density * m.specific_heat);
    }
  }
  if constexpr (!ACCUMULATE) {
    const T vol = (detF * detF - 1.) * (m.K * .5);
#pragma unroll
    for (int i = 0; i < DIM; ++i) MH_M(s, i, i) = MH_M(s, i, i) + vol;
    mat_mul_t<DIM, false, true>(s, Finv, P);   // tau F^-T
  } else {
#pragma unroll
    for (int i = 0; i < DD; ++i) {
      st.m2[i] = ad_v(F[i]);
      st.m1[i] = ad_v(be[i]);
    }
  }
  return status;
}

template<int DIM, bool ACCUMULATE, class T>
MH_DEV int j2log_stress(const MaterialDev& md, double dt, const T* F, OtherState& st, T* P, OtherCache& cc) {
  const mimi_hip_material& m = md.m;
  constexpr int DD = DIM * DIM;
  int status = 0;
  T F_e[DD], C_e[DD], E_e[DD], s[DD];
  mat_mul_t<DIM, false, false>(F, st.m1, F_e);
  mat_mul_t<DIM, true, false>(F_e, F_e, C_e);
  half_log_sym<DIM>(C_e, E_e, cc);
  const T p = trace_t<DIM>(E_e) * m.K;
  dev_t<DIM>(E_e, 2.0 * m.G, s);
  const T q = norm_t<DIM>(s) * sqrt(1.5);
  bool plastic;
  const T delta = return_map_increment<T>(md, dt, st.eqps, st.temperature, q, ad_from<T>(3.0 * m.G), plastic, status, cc);
  if (plastic) {
    // N_p = 1.5 / q s (trial);  s -= 2 G delta N_p
    if constexpr (ACCUMULATE) {
      double inc[DD], ex[DD], old[DD];
#pragma unroll
      for (int i = 0; i < DD; ++i) inc[i] = -ad_v(delta) * (1.5 / ad_v(q)) * ad_v(s[i]);
      sym_exp<DIM>(inc, ex);
      st.eqps += ad_v(delta);
#pragma unroll
      for (int i = 0; i < DD; ++i) old[i] = st.m1[i];
      mat_mul_t<DIM, false, false>(old, ex, st.m1);
    } else {
      const T fac = 1.0 - delta * (2.0 * m.G * 1.5) / q;
#pragma unroll
      for (int i = 0; i < DD; ++i) s[i] = s[i] * fac;
    }
  }
  if constexpr (!ACCUMULATE) {
    const T detF = det_t<DIM>(F);
    const T pj = p / detF;
#pragma unroll
    for (int i = 0; i < DIM; ++i) MH_M(s, i, i) = MH_M(s, i, i) + pj;
    pk1_from_cauchy_t<DIM>(s, F, P);
  }
  return status;
}

// FK >= 0: the material kind as a compile-time constant (the other materials' code is then dead: a kernel that holds all
// four spills registers the largest one alone does not need -- measured in round 4 on the pre-pass kernels); -1: md.m.kind
template<int DIM, bool ACCUMULATE, class T, int FK = -1>
MH_DEV int other_stress(const MaterialDev& md, double dt, const T* F, OtherState& st, T* P, OtherCache& cc) {
  switch (FK >= 0 ? FK : md.m.kind) {
  case MIMI_HIP_MAT_STVK:
    if constexpr (!ACCUMULATE) stvk_stress<DIM>(md.m, F, P);
    return 0;
  case MIMI_HIP_MAT_J2LINEAR: j2linear_stress<DIM, ACCUMULATE>(md.m, F, st, P); return 0;
  case MIMI_HIP_MAT_J2SIMO: return j2simo_stress<DIM, ACCUMULATE>(md, dt, F, st, P, cc);
  default: return j2log_stress<DIM, ACCUMULATE>(md, dt, F, st, P, cc);
  }
}

// One call per quadrature point: P (column-major) and, when A != nullptr, wd * dP_iJ/dF_jL into
// A[((i*DIM + J)*DIM + j)*DIM + L] (the layout of tangent_of)
template<int DIM, int FK = -1>
MH_DEV int evaluate_other(const MaterialDev& md, double dt, const StateView& sv, int64_t pt, const double* F, double* P,
                          double* A, double wd) {
  constexpr int DD = DIM * DIM;
  OtherState st;
  other_state_load<DIM>(md, sv, pt, st);
  OtherCache cc;
  int status = other_stress<DIM, false, double, FK>(md, dt, F, st, P, cc);
  if (A) {
    for (int jL = 0; jL < DD; ++jL) {   // F(j, L) is stored at j + L*DIM
      Dual Fd[DD], Pd[DD];
#pragma unroll
      for (int k = 0; k < DD; ++k) Fd[k] = Dual{F[k], k == jL ? 1.0 : 0.0};
      int ignored = 0;
      ignored |= other_stress<DIM, false, Dual, FK>(md, dt, Fd, st, Pd, cc);
      const int j = jL % DIM, L = jL / DIM;
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int J = 0; J < DIM; ++J) A[((i * DIM + J) * DIM + j) * DIM + L] = wd * MH_M(Pd, i, J).d;
    }
  }
  return status;
}

// The same, for callers that consume the tangent one direction at a time instead of holding all DIM^4 entries (the
// pre-pass kernels: 81 doubles per lane in 3-D were most of their spilled registers): other_tangent_begin leaves P and the
// state / cache the directional passes share; other_tangent_dir(j, L) returns dP_iJ / dF_jL for every (i, J), column-major.
template<int DIM>
struct OtherTangent {
  OtherState st;
  OtherCache cc;
};

template<int DIM, int FK = -1>
MH_DEV int other_tangent_begin(const MaterialDev& md, double dt, const StateView& sv, int64_t pt, const double* F, double* P,
                               OtherTangent<DIM>& t) {
  other_state_load<DIM>(md, sv, pt, t.st);
  return other_stress<DIM, false, double, FK>(md, dt, F, t.st, P, t.cc);
}

template<int DIM, int FK = -1>
MH_DEV void other_tangent_dir(const MaterialDev& md, double dt, const double* F, OtherTangent<DIM>& t, int j, int L, double* dP) {
  constexpr int DD = DIM * DIM;
  Dual Fd[DD], Pd[DD];
  const int jL = j + L * DIM;
#pragma unroll
  for (int k = 0; k < DD; ++k) Fd[k] = Dual{F[k], k == jL ? 1.0 : 0.0};
  (void)other_stress<DIM, false, Dual, FK>(md, dt, Fd, t.st, Pd, t.cc);
#pragma unroll
  for (int k = 0; k < DD; ++k) dP[k] = Pd[k].d;
}

template<int DIM, int FK = -1>
MH_DEV int accumulate_other(const MaterialDev& md, double dt, const StateView& sv, int64_t pt, const double* F) {
  if ((FK >= 0 ? FK : md.m.kind) == MIMI_HIP_MAT_STVK) return 0;
  OtherState st;
  other_state_load<DIM>(md, sv, pt, st);
  double unused[DIM * DIM];
  OtherCache cc;
  const int status = other_stress<DIM, true, double, FK>(md, dt, F, st, unused, cc);
  other_state_store<DIM>(md, sv, pt, st);
  return status;
}

}  // namespace mimi_hip
