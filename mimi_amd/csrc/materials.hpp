// Device-side constitutive updates: PK1 stress P(F) and consistent tangent dP/dF at one
// quadrature point, fp64, everything in registers.
//
// Follows the arithmetic of the reference (paths under /root/reference/src/mimi/):
//   CompressibleOgdenNeoHookean::EvaluateCauchy        materials/materials.cpp:96-118
//   MaterialBase::EvaluatePK1 (P = J sigma F^-T)       materials/materials.cpp:60-71
//   J2::PlasticStress<accumulate>                      materials/materials.hpp:311-391
//   ElasticStrain / Dev / Norm                         materials/material_utils.hpp:22-84,117-127
//   hardening laws (value + d/d eqps)                  materials/material_hardening.hpp:79-346
//   ScalarSolve (safeguarded Newton / bisection)       solvers/newton.hpp:53-169
// The reference has no analytic tangent (it differentiates the element residual
// numerically, integrators/nonlinear_solid.cpp:48-76); the closed forms below are new.
//
// Storage: all DIM x DIM tensors column-major, T(i,J) = T[i + J*DIM] (mfem::DenseMatrix).
// Tangent: A[((i*DIM + J)*DIM + j)*DIM + L] = dP_iJ / dF_jL.
#pragma once

#include <hip/hip_runtime.h>

#include "common.hpp"

namespace mimi_hip {

#ifndef MH_DEV
#define MH_DEV __device__ __forceinline__
#endif

template<int DIM>
MH_DEV double det_of(const double* F) {
  if constexpr (DIM == 2) {
    return F[0] * F[3] - F[1] * F[2];
  } else {
    return F[0] * (F[4] * F[8] - F[5] * F[7]) - F[3] * (F[1] * F[8] - F[2] * F[7])
           + F[6] * (F[1] * F[5] - F[2] * F[4]);
  }
}

template<int DIM>
MH_DEV void inverse_of(const double* F, double det, double* Fi) {
  const double t = 1.0 / det;
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

// ---- hardening -------------------------------------------------------------------
struct Dual {
  double v, d;
};

template<bool SC = false>
MH_DEV double pow_positive(double x, double q);

// material_hardening.hpp:75-77,261-279,326-333.  The homologous temperature's power is the library's pow only where
// pow_positive (below) is not defined: the library routine is ~ 350 instructions that every lane executes at every point of
// every assembly -- 8 % of the degree-3 pre-pass's vector instructions in its residual-only mode (round 5) --, and in
// the virgin state (T = T_ref at every point) it is handed 0, whose powers need no arithmetic at all.  (No call of the
// library's pow is left in this file: see pow_any.)
template<bool SC = false>
MH_DEV double thermo_contribution(const MaterialDev& md, double T) {
  const mimi_hip_material& m = md.m;
  if (m.hardening == MIMI_HIP_HARD_JC_TEMP_RATE) {
    double c = 1.0;
    if (T < m.reference_temperature) {
    } else if (T > m.melting_temperature) {
      c = 0.0;
    } else {
      const double base = (T - m.reference_temperature) / (m.melting_temperature - m.reference_temperature);   // in [0, 1]
      if (base > 0.0) c -= pow_positive<SC>(base, m.m);
      else if (base == 0.0) c -= m.m > 0.0 ? 0.0 : (m.m == 0.0 ? 1.0 : __builtin_huge_val());   // pow(0, m)
      else c -= m.m == 0.0 ? 1.0 : base;                                                       // (base is NaN here: as pow answers it)
    }
    return c;
  }
  if (m.hardening == MIMI_HIP_HARD_JC_CONST_TEMP) return md.const_temperature_contribution;
  return 1.0;
}

template<bool SC = false>
MH_DEV double ln_positive(double x);

template<bool SC = false>
MH_DEV double rate_contribution(const mimi_hip_material& m, double rate) {
  if (m.hardening >= MIMI_HIP_HARD_JC_RATE) {
    double v = 1.0;
    // (C == 0 -- the reference's tests never set C, SURVEY 8c: 1 + 0 log(..) is 1 to the bit, and the logarithm in every
    // trip of the return-map Newton is skipped.  C != 0: ln_positive, ~ 40 instructions and a few ulp, instead of the
    // library's log -- ~ 150 instructions inlined at every call site; tests/test_domain_gpu.py, rate-dependent parity)
    if (m.C != 0.0 && rate > m.eps0_dot) v += m.C * ln_positive<SC>(rate / m.eps0_dot);
    return v;
  }
  return 1.0;
}

MH_DEV double rate_contribution_derivative(const mimi_hip_material& m, double rate) {
  if (m.hardening >= MIMI_HIP_HARD_JC_RATE && rate > m.eps0_dot) return m.C / rate;
  return 0.0;
}

// x^q for x > 0 as exp(q ln x), both written out (round 3): the library's fp64 log and exp keep their error under one
// ulp with extended-precision steps -- several hundred instructions per call, inside every iteration of the return-map
// Newton, which is most of the degree-3 pre-pass.  Here: ln x = e ln 2 + 2 atanh((m - 1) / (m + 1)) with m in
// [sqrt(1/2), sqrt 2) (odd series to t^23, t^2 < 0.03: truncation < 1e-17), exp by k = rint(y / ln 2), the remainder in
// two steps (ln 2 split so that k ln2_hi is exact) and the Taylor polynomial to r^13 (|r| < 0.35: < 2e-17).  About 60
// instructions for the pair.  Error: a few ulp on ln x, hence |q ln x| x 3e-16 + 2e-16 relative on the power -- the
// conditioning of exp(q ln x) itself; < 1e-14 for the plastic strains that occur (>= 1e-13), far inside the 1e-9
// (state), 1e-11 (tangent) and 1e-12 (residual) bars the parity tests hold this path to.
// one Horner step p x + c as ONE vector instruction with the coefficient as an operand of its own.  Written as
// __builtin_fma the compiler forms the two-operand accumulate v_fmac_f64 and copies every (loop-invariant) coefficient into
// the accumulator first: two vector instructions per step, 24 copies per call in the return-map Newton (round 5: a
// quarter of pow_positive's instructions).  SC: the coefficient in a scalar register pair (two s_mov per step, which cost
// no vector issue) instead of a vector pair held across the Newton loop -- 48 vector registers less, what gives a kernel
// that parks its tensors (j2_stress, Park) a third / fourth wave per SIMD; without that extra wave the vector form is the
// faster one (profiles/r05_cfg3_horner_ab.txt).  Same instruction, same operands, same bits either way.
template<bool SC>
MH_DEV double horner_step(double p, double x, double c) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;
  if constexpr (SC) asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "s"(c));
  else asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(p), "v"(x), "v"(c));
  return r;
#else
  return __builtin_fma(p, x, c);
#endif
}

// ln x for x > 0 (+inf -> +inf), a few ulp: the first half of pow_positive, also what the Johnson-Cook rate term calls
template<bool SC>
MH_DEV double ln_positive(double x) {
  if (x > 1.79769313486231570815e+308) return x;
  int e;
  double m = __builtin_frexp(x, &e);   // [1/2, 1)
  if (m < 0.70710678118654752440) {
    m *= 2.0;
    e -= 1;
  }
  const double t = (m - 1.0) / (m + 1.0), t2 = t * t;
  double p = 1.0 / 23.0;
  p = horner_step<SC>(p, t2, 1.0 / 21.0);
  p = horner_step<SC>(p, t2, 1.0 / 19.0);
  p = horner_step<SC>(p, t2, 1.0 / 17.0);
  p = horner_step<SC>(p, t2, 1.0 / 15.0);
  p = horner_step<SC>(p, t2, 1.0 / 13.0);
  p = horner_step<SC>(p, t2, 1.0 / 11.0);
  p = horner_step<SC>(p, t2, 1.0 / 9.0);
  p = horner_step<SC>(p, t2, 1.0 / 7.0);
  p = horner_step<SC>(p, t2, 1.0 / 5.0);
  p = horner_step<SC>(p, t2, 1.0 / 3.0);
  p = __builtin_fma(p * t2, 2.0 * t, 2.0 * t);                      // ln m = 2 t (1 + t^2 p)
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double ed = (double)e;
  return __builtin_fma(ed, ln2_hi, p) + ed * ln2_lo;
}

template<bool SC>
MH_DEV double pow_positive(double x, double q) {
  // arguments outside the range the reduction below is written for, answered as pow() answers them (a diverging
  // return-map Newton can produce them): x = +inf here, |q ln x| beyond the exponent range below
  if (x > 1.79769313486231570815e+308) return q > 0.0 ? x : (q < 0.0 ? 0.0 : 1.0);
  const double lnx = ln_positive<SC>(x);
  constexpr double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  double y = q * lnx;
  // e^y over- / underflows long before +-1500: clamped so that k fits an int and the remainder stays small; ldexp then
  // returns inf / 0 as pow() does (a NaN passes through both comparisons and comes out as NaN)
  if (y > 1500.0) y = 1500.0;
  if (y < -1500.0) y = -1500.0;
  const double k = __builtin_rint(y * 1.44269504088896338700e+00);
  double r = __builtin_fma(-k, ln2_hi, y);
  r = __builtin_fma(-k, ln2_lo, r);
  double c = 1.0 / 6227020800.0;
  c = horner_step<SC>(c, r, 1.0 / 479001600.0);
  c = horner_step<SC>(c, r, 1.0 / 39916800.0);
  c = horner_step<SC>(c, r, 1.0 / 3628800.0);
  c = horner_step<SC>(c, r, 1.0 / 362880.0);
  c = horner_step<SC>(c, r, 1.0 / 40320.0);
  c = horner_step<SC>(c, r, 1.0 / 5040.0);
  c = horner_step<SC>(c, r, 1.0 / 720.0);
  c = horner_step<SC>(c, r, 1.0 / 120.0);
  c = horner_step<SC>(c, r, 1.0 / 24.0);
  c = horner_step<SC>(c, r, 1.0 / 6.0);
  c = horner_step<SC>(c, r, 0.5);
  c = __builtin_fma(c * r, r, r);                                   // e^r - 1
  return __builtin_ldexp(1.0 + c, (int)k);
}

// x^q for any x, as pow() answers it, WITHOUT the library routine: a return-map equation only ever asks for powers of a
// positive base (accumulated plastic strain + the increment, which the bracket keeps >= 0; 1 + strain / eps0), but the
// other cases must be answered, and the library's pow -- inlined into every kernel that evaluates a hardening law: ~ 230
// instructions and 25 registers at each of nine call sites -- was what set the register count of the degree-3 pre-pass
// (residual-only mode 145 -> 120 registers, a fourth wave per SIMD; commit 123 -> 98; 6 745 -> 4 795 instructions; round 5).
// Zero: pow's limits; negative base: +- |x|^q for an integer q (|x|^q by pow_positive: a few ulp, where the library rounds
// correctly), NaN otherwise; q = 0: 1.
template<bool SC = false>
MH_DEV double pow_any(double x, double q) {
  const double r = pow_positive<SC>(__builtin_fabs(x), q);     // (one inlined copy serves the negative bases too)
  if (x > 0.0) return r;
  if (q == 0.0) return 1.0;
  const bool integer = q == __builtin_rint(q);
  const bool odd = integer && (0.5 * q != __builtin_rint(0.5 * q));
  if (x == 0.0) {
    const double z = q < 0.0 ? __builtin_huge_val() : 0.0;
    return odd ? __builtin_copysign(z, x) : z;
  }
  if (x < 0.0 && integer) return odd ? -r : r;
  return __builtin_nan("");
}

// utils/ad.inl:263-279: pow(x, n) = x * x^(n-1), derivative n * x^(n-1) * x'
template<bool SC = false>
MH_DEV Dual dual_pow(Dual b, double power) {
  const double tmp = pow_any<SC>(b.v, power - 1.0);
  return Dual{b.v * tmp, b.d * (power * tmp)};
}

template<bool SC = false>
MH_DEV Dual hardening_evaluate(const mimi_hip_material& m, Dual eqps) {
  switch (m.hardening) {
  case MIMI_HIP_HARD_POWERLAW: {
    Dual p = dual_pow<SC>(Dual{1.0 + eqps.v / m.eps0, eqps.d / m.eps0}, 1.0 / m.n);
    return Dual{m.sigma_y * p.v, m.sigma_y * p.d};
  }
  case MIMI_HIP_HARD_VOCE: {
    const double e = exp(-eqps.v / m.strain_constant);
    return Dual{m.sigma_sat - (m.sigma_sat - m.sigma_y) * e,
                (m.sigma_sat - m.sigma_y) * e * (eqps.d / m.strain_constant)};
  }
  default: {
    if (fabs(eqps.v) < 1.e-13) return Dual{m.A, 0.0};
    Dual p = dual_pow<SC>(eqps, m.n);
    return Dual{m.A + m.B * p.v, m.B * p.d};
  }
  }
}

struct ReturnMapCtx {
  double eqps_old, q, thermo, dt;
  double slope;   // 3G (J2, J2Log: materials.hpp:345,622) or G tr(be) (J2Simo: materials.hpp:495)
};

// materials.hpp:343-349, evaluated once per point x with everything the caller needs afterwards: the residual and its
// derivative (dual seed 1), the hardening value / slope and the rate factor at x.  The reference evaluates the same
// function at the same x several times (admissibility check, both bracket ends, the Newton start, the hardening slope
// after the solve); identical arguments give identical results, so each distinct x is evaluated once here.
struct RmPoint {
  double x;
  Dual R;      // residual, d residual / d x
  Dual H;      // hardening at eqps_old + x
  double rc;   // rate contribution at x / dt
};

template<bool SC = false>
MH_DEV RmPoint rm_eval(const mimi_hip_material& m, const ReturnMapCtx& c, double x) {
  RmPoint e;
  e.x = x;
  e.H = hardening_evaluate<SC>(m, Dual{c.eqps_old + x, 1.0});
  // (the rate x / dt -- a division, 13 instructions in every iteration of the Newton below -- only where a rate term reads it)
  e.rc = (m.hardening >= MIMI_HIP_HARD_JC_RATE && m.C != 0.0) ? rate_contribution<SC>(m, x / c.dt) : 1.0;
  const double fac = e.rc * c.thermo;
  e.R = Dual{c.q - c.slope * x - e.H.v * fac, -c.slope - e.H.d * fac};
  return e;
}

// solvers/newton.hpp:53-169; status bit 1 = root not bracketed, bit 2 = not converged.  at_lower: the evaluation at
// `lower`, which the caller already has; the evaluation at the returned x is left in `last`.
template<bool SC = false>
MH_DEV double scalar_solve(const mimi_hip_material& m, const ReturnMapCtx& c, const RmPoint& at_lower, double x0, double lower,
                           double upper, double xtol, double rtol, int max_iter, int& status, RmPoint& last) {
  const double fl = at_lower.R.v;
  last = rm_eval<SC>(m, c, upper);
  const double fh = last.R.v;
  if (fabs(fl) < xtol) {
    last = at_lower;
    return lower;
  }
  if (fabs(fh) < xtol) return upper;
  if (fl * fh > 0.) {
    status |= 1;
    last = at_lower;
    return lower;
  }
  double xl = lower, xh = upper;
  if (fl > 0) {
    xl = upper;
    xh = lower;
  }
  if (x0 < lower || x0 > upper) x0 = 0.5 * (lower + upper);
  double x = x0;
  double delta_x_old = fabs(upper - lower);
  double delta_x = delta_x_old;
  last = x == lower ? at_lower : rm_eval<SC>(m, c, x);
  double fval = last.R.v, df_dx = last.R.d;
  bool converged = false;
  int iterations = 0;
  while (!converged) {
    if (iterations == max_iter) {
      status |= 2;
      break;
    }
    if ((x - xh) * df_dx - fval > 0 || (x - xl) * df_dx - fval < 0
        || fabs(2. * fval) > fabs(delta_x_old * df_dx)) {
      delta_x_old = delta_x;
      delta_x = 0.5 * (xh - xl);
      x = xl + delta_x;
    } else {
      delta_x_old = delta_x;
      delta_x = fval / df_dx;
      x -= delta_x;
    }
    last = rm_eval<SC>(m, c, x);
    fval = last.R.v;
    df_dx = last.R.d;
    converged = (fabs(delta_x) < xtol) || (fabs(fval) < rtol);
    if (fval < 0) xl = x; else xh = x;
    ++iterations;
  }
  return x;
}

// ---- per-point state view ---------------------------------------------------------
// J2 MaterialState (materials.hpp:278-286) stored SoA over points:
//   eqps[pt], temperature[pt], plastic_strain[c*n_pts + pt]
struct StateView {
  double* eqps;
  double* temperature;
  double* plastic_strain;   // first state matrix (J2, J2Linear: plastic strain; J2Simo: be_old; J2Log: Fp_inv)
  int64_t n_pts;
  double* state2;           // second state matrix (J2Linear: beta; J2Simo: F_old), same SoA layout
};

template<int DIM>
struct PointResult {
  double P[DIM * DIM];
  // by-products used by the tangent
  double Finv[DIM * DIM], detF;
  double sigma[DIM * DIM], s_trial[DIM * DIM];
  double q, delta, hprime;
  bool plastic;
};

#define MH_M(T, i, j) (T)[(i) + (j) * DIM]

template<int DIM>
MH_DEV void pk1_from_cauchy(PointResult<DIM>& w) {
  // materials.cpp:60-71: P = det(F) * sigma * F^-T
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int J = 0; J < DIM; ++J) {
      double s = 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) s += MH_M(w.sigma, i, k) * MH_M(w.Finv, J, k);
      MH_M(w.P, i, J) = s * w.detF;
    }
}

template<int DIM>
MH_DEV void neo_hookean_stress(const mimi_hip_material& m, const double* F, PointResult<DIM>& w) {
  w.detF = det_of<DIM>(F);
  inverse_of<DIM>(F, w.detF, w.Finv);
  const double mu_over = m.mu / w.detF;
  const double diag = -mu_over + m.lambda * (w.detF - 1.);
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) {
      double b = 0;
#pragma unroll
      for (int k = 0; k < DIM; ++k) b += MH_M(F, i, k) * MH_M(F, j, k);
      MH_M(w.sigma, i, j) = mu_over * b + (i == j ? diag : 0.0);
    }
  pk1_from_cauchy<DIM>(w);
}

// J2::PlasticStress<false> when ACCUMULATE == false; <true> otherwise (then the state
// arguments are updated in place and no stress is produced).
// Park: a caller's hook around the return mapping.  The scalar Newton solve needs none of the tensors of the trial state,
// but they are live across it (F^-1 and the trial deviator: 36 registers in 3-D); a kernel that wants a third wave per SIMD
// hands them to LDS for the duration of the solve (tp3_point_kernel, tensor_p3.hip).  NoPark: nothing happens.
struct NoPark {
  static constexpr bool scalar_coefficients = false;
  template<int DD> MH_DEV void save(const double (&)[DD], const double (&)[DD]) const {}
  template<int DD> MH_DEV void restore(double (&)[DD], double (&)[DD]) const {}
};

template<int DIM, bool ACCUMULATE, class Park = NoPark>
MH_DEV int j2_stress(const MaterialDev& md, double dt, const double* F, double* ep, double& eqps,
                     double& temperature, PointResult<DIM>& w, const Park& park = Park{}) {
  const mimi_hip_material& m = md.m;
  constexpr int DD = DIM * DIM;
  double eps[DD], s[DD];
  w.detF = det_of<DIM>(F);
  inverse_of<DIM>(F, w.detF, w.Finv);
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int j = 0; j < DIM; ++j) MH_M(eps, i, j) = 0.5 * (MH_M(F, i, j) + MH_M(F, j, i));
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(eps, i, i) -= 1.;
#pragma unroll
  for (int i = 0; i < DD; ++i) eps[i] -= ep[i];
  double tr = 0;
#pragma unroll
  for (int i = 0; i < DIM; ++i) tr += MH_M(eps, i, i);
  const double p = m.K * tr;
  const double tr_over_dim = tr / (double)DIM;  // Dev(): trace / dim (material_utils.hpp:33,44)
#pragma unroll
  for (int i = 0; i < DD; ++i) s[i] = eps[i] * (2.0 * m.G);
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(s, i, i) = (MH_M(eps, i, i) - tr_over_dim) * (2.0 * m.G);
  double nrm = 0;
#pragma unroll
  for (int i = 0; i < DD; ++i) nrm += s[i] * s[i];
  nrm = sqrt(nrm);
  const double q = sqrt(3.0 / 2.0) * nrm;
  w.q = q;
  w.plastic = false;
  w.delta = 0;
  w.hprime = 0;

  constexpr bool SC = Park::scalar_coefficients;     // (horner_step)
  ReturnMapCtx c{eqps, q, thermo_contribution<SC>(md, temperature), dt, 3.0 * m.G};
  const double tolerance = md.sigma_y_ref * 1.e-10;
  int status = 0;
  const RmPoint at0 = rm_eval<SC>(m, c, 0.0);
  const bool yields = at0.R.v > tolerance;
  RmPoint sol;
  double delta = 0.0;
  // (what is needed behind the solve: F^-1 and the deviator for the stress; the plastic strain and the deviator for a commit)
  if constexpr (ACCUMULATE) park.save(reinterpret_cast<double (&)[DD]>(*ep), s); else park.save(w.Finv, s);
  if (yields) {
    const double upper = (q - at0.H.v * c.thermo) / (3.0 * m.G);
    delta = scalar_solve<SC>(m, c, at0, 0.0, 0.0, upper, 1.e-10, tolerance, 100, status, sol);
  }
  if constexpr (ACCUMULATE) park.restore(reinterpret_cast<double (&)[DD]>(*ep), s); else park.restore(w.Finv, s);
#pragma unroll
  for (int i = 0; i < DD; ++i) w.s_trial[i] = s[i];
  if (yields) {
    w.plastic = true;
    w.delta = delta;
    const double npf = 1.5 / q;
    if constexpr (!ACCUMULATE) {
      const Dual H = sol.H;          // (hardening and rate factor at the solution: what the last evaluation left)
      const double rc = sol.rc;
      w.hprime = H.d * rc * c.thermo + H.v * rate_contribution_derivative(m, delta / dt) / dt * c.thermo;
#pragma unroll
      for (int i = 0; i < DD; ++i) s[i] += -2.0 * m.G * delta * (npf * s[i]);
    } else {
      eqps += delta;
#pragma unroll
      for (int i = 0; i < DD; ++i) ep[i] += delta * (npf * s[i]);
      if (m.hardening == MIMI_HIP_HARD_JC_TEMP_RATE) {
        temperature += m.heat_fraction * q * delta / (m.density * m.specific_heat);
      }
    }
  }
  if constexpr (!ACCUMULATE) {
#pragma unroll
    for (int i = 0; i < DD; ++i) w.sigma[i] = s[i];
#pragma unroll
    for (int i = 0; i < DIM; ++i) MH_M(w.sigma, i, i) += p;
    pk1_from_cauchy<DIM>(w);
  }
  return status;
}

// A_iJjL for the state left in `w` by the stress routines
template<int DIM>
MH_DEV void tangent_of(const mimi_hip_material& m, const PointResult<DIM>& w, double* A) {
  const double J = w.detF;
  const double* Fi = w.Finv;
  if (m.kind == MIMI_HIP_MAT_NEOHOOKEAN) {
    // P = mu F + (lambda J (J-1) - mu) F^-T
    const double c1 = m.lambda * J * (J - 1.) - m.mu;
    const double c2 = m.lambda * (2. * J - 1.) * J;
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int Jx = 0; Jx < DIM; ++Jx)
#pragma unroll
        for (int j = 0; j < DIM; ++j)
#pragma unroll
          for (int L = 0; L < DIM; ++L) {
            double v = (i == j && Jx == L) ? m.mu : 0.0;
            v += -c1 * MH_M(Fi, L, i) * MH_M(Fi, Jx, j);
            v += c2 * MH_M(Fi, L, j) * MH_M(Fi, Jx, i);
            A[((i * DIM + Jx) * DIM + j) * DIM + L] = v;
          }
    return;
  }
  // J2: P_iJ = J sigma_ik Finv_Jk, sigma = radial return of the small-strain trial state
  double beta = 1.0, gamma = 0.0;
  if (w.plastic) {
    const double q = w.q, G = m.G;
    beta = 1.0 - 3.0 * G * w.delta / q;
    gamma = 3.0 * G * (1.5 / q) * (1.0 / ((3.0 * G + w.hprime) * q) - w.delta / (q * q));
  }
  const double G2 = 2.0 * m.G;
#pragma unroll
  for (int i = 0; i < DIM; ++i)
#pragma unroll
    for (int Jx = 0; Jx < DIM; ++Jx)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int L = 0; L < DIM; ++L) {
          double v = 0.0;
#pragma unroll
          for (int k = 0; k < DIM; ++k) {
            v += MH_M(w.sigma, i, k) * J * (MH_M(Fi, L, j) * MH_M(Fi, Jx, k) - MH_M(Fi, Jx, j) * MH_M(Fi, L, k));
            double C = (i == k && j == L ? m.K : 0.0);
            C += beta * G2 * (0.5 * ((i == j && k == L ? 1.0 : 0.0) + (i == L && k == j ? 1.0 : 0.0))
                              - (i == k && j == L ? 1.0 / (double)DIM : 0.0));
            C -= G2 * gamma * MH_M(w.s_trial, i, k) * MH_M(w.s_trial, j, L);
            v += J * C * MH_M(Fi, Jx, k);
          }
          A[((i * DIM + Jx) * DIM + j) * DIM + L] = v;
        }
}

// Row I of the tangent only: Arow[(J*DIM + j)*DIM + L] = dP_IJ / dF_jL  (same closed forms)
template<int DIM, int I>
MH_DEV void tangent_row_of(const mimi_hip_material& m, const PointResult<DIM>& w, double* Arow) {
  const double J = w.detF;
  const double* Fi = w.Finv;
  if (m.kind == MIMI_HIP_MAT_NEOHOOKEAN) {
    const double c1 = m.lambda * J * (J - 1.) - m.mu;
    const double c2 = m.lambda * (2. * J - 1.) * J;
#pragma unroll
    for (int Jx = 0; Jx < DIM; ++Jx)
#pragma unroll
      for (int j = 0; j < DIM; ++j)
#pragma unroll
        for (int L = 0; L < DIM; ++L) {
          double v = (I == j && Jx == L) ? m.mu : 0.0;
          v += -c1 * MH_M(Fi, L, I) * MH_M(Fi, Jx, j);
          v += c2 * MH_M(Fi, L, j) * MH_M(Fi, Jx, I);
          Arow[(Jx * DIM + j) * DIM + L] = v;
        }
    return;
  }
  double beta = 1.0, gamma = 0.0;
  if (w.plastic) {
    const double q = w.q, G = m.G;
    beta = 1.0 - 3.0 * G * w.delta / q;
    gamma = 3.0 * G * (1.5 / q) * (1.0 / ((3.0 * G + w.hprime) * q) - w.delta / (q * q));
  }
  const double G2 = 2.0 * m.G;
#pragma unroll
  for (int Jx = 0; Jx < DIM; ++Jx)
#pragma unroll
    for (int j = 0; j < DIM; ++j)
#pragma unroll
      for (int L = 0; L < DIM; ++L) {
        double v = 0.0;
#pragma unroll
        for (int k = 0; k < DIM; ++k) {
          v += MH_M(w.sigma, I, k) * J * (MH_M(Fi, L, j) * MH_M(Fi, Jx, k) - MH_M(Fi, Jx, j) * MH_M(Fi, L, k));
          double C = (I == k && j == L ? m.K : 0.0);
          C += beta * G2 * (0.5 * ((I == j && k == L ? 1.0 : 0.0) + (I == L && k == j ? 1.0 : 0.0))
                            - (I == k && j == L ? 1.0 / (double)DIM : 0.0));
          C -= G2 * gamma * MH_M(w.s_trial, I, k) * MH_M(w.s_trial, j, L);
          v += J * C * MH_M(Fi, Jx, k);
        }
        Arow[(Jx * DIM + j) * DIM + L] = v;
      }
}

// One call per quadrature point.  pt indexes the SoA state; F is (i,J) column-major.
template<int DIM, class Park = NoPark>
MH_DEV int evaluate_pk1(const MaterialDev& md, double dt, const StateView& st, int64_t pt, const double* F,
                        PointResult<DIM>& w, const Park& park = Park{}) {
  if (md.m.kind == MIMI_HIP_MAT_NEOHOOKEAN) {
    neo_hookean_stress<DIM>(md.m, F, w);
    return 0;
  }
  double ep[DIM * DIM];
#pragma unroll
  for (int c = 0; c < DIM * DIM; ++c) ep[c] = st.plastic_strain[c * st.n_pts + pt];
  double eqps = st.eqps[pt], T = st.temperature[pt];
  return j2_stress<DIM, false, Park>(md, dt, F, ep, eqps, T, w, park);
}

template<int DIM, class Park = NoPark>
MH_DEV int accumulate_state(const MaterialDev& md, double dt, const StateView& st, int64_t pt, const double* F,
                            const Park& park = Park{}) {
  if (md.m.kind != MIMI_HIP_MAT_J2) return 0;
  double ep[DIM * DIM];
#pragma unroll
  for (int c = 0; c < DIM * DIM; ++c) ep[c] = st.plastic_strain[c * st.n_pts + pt];
  double eqps = st.eqps[pt], T = st.temperature[pt];
  PointResult<DIM> w;
  const int status = j2_stress<DIM, true, Park>(md, dt, F, ep, eqps, T, w, park);
  if (w.plastic) {
#pragma unroll
    for (int c = 0; c < DIM * DIM; ++c) st.plastic_strain[c * st.n_pts + pt] = ep[c];
    st.eqps[pt] = eqps;
    st.temperature[pt] = T;
  }
  return status;
}

}  // namespace mimi_hip
