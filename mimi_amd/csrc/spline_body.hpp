// Rigid spline body of the contact integrator: closest point on one B-spline / NURBS curve (2-D) or surface (3-D).
//
// Reference: coefficients::NearestDistanceToSplines (coefficients/nearest_distance.hpp:215-288) hands the query to
// splinepy (SplinepyPlantNewKdTreeForProximity for the initial guess, SplinepyVerboseProximity with aggressive bounds
// for the search), an absent, un-pinned third-party library.  What is restated here is the published scheme: nearest
// of res^para_dim sampled points as the initial guess, then Newton on the squared distance in the parametric
// coordinates, clipped to the parametric bounds, with step halving when the distance does not decrease -- **parity
// unpinned**, as all of contact.  Results as Results::{physical_, first_derivatives_} (nearest_distance.hpp:46-58);
// ComputeNormal<true> / NormalGap (:139-193) are applied by the caller.
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace mimi_hip {

#define SB_HD __host__ __device__ inline

constexpr int kMaxBodyDegree = 5;

struct SplineBodyDev {
  int para_dim, dim;
  int p[2], n_ctrl[2], n_knots[2];
  const double* knots[2];
  const double* ctrl_h;      // [n_ctrl0 * n_ctrl1][dim + 1]: (w x, w), first parametric direction fastest
  const double* sample_xi;   // [n_samples][para_dim]
  const double* sample_x;    // [n_samples][dim]
  int n_samples, max_iterations;
};

// span index i with U[i] <= xi < U[i+1] (the last non-empty span at the upper end)
SB_HD int sb_find_span(const double* U, int n_knots, int p, double xi) {
  const int n = n_knots - p - 1;
  if (xi >= U[n]) {
    int i = n - 1;
    while (i > p && U[i] == U[i + 1]) --i;
    return i;
  }
  int lo = p, hi = n;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (xi < U[mid]) hi = mid; else lo = mid;
  }
  return lo;
}

// the d + 1 B-splines of degree d that are non-zero on `span`: out[j] = N_{span-d+j, d}(xi)
SB_HD void sb_values(const double* U, int d, int span, double xi, double* out) {
  double left[kMaxBodyDegree + 1], right[kMaxBodyDegree + 1];
  out[0] = 1.0;
  for (int j = 1; j <= d; ++j) {
    left[j] = xi - U[span + 1 - j];
    right[j] = U[span + j] - xi;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      const double t = out[r] / (right[r + 1] + left[j - r]);
      out[r] = saved + right[r + 1] * t;
      saved = left[j - r] * t;
    }
    out[j] = saved;
  }
}

// first derivatives of the degree-d functions from the values of degree d - 1 (lower[j] = N_{span-d+1+j, d-1})
SB_HD void sb_derive(const double* U, int d, int span, const double* lower, double* out) {
  for (int j = 0; j <= d; ++j) {
    const int i = span - d + j;
    double v = 0.0;
    if (j >= 1) {
      const double den = U[i + d] - U[i];
      if (den > 0) v += d / den * lower[j - 1];
    }
    if (j <= d - 1) {
      const double den = U[i + d + 1] - U[i + 1];
      if (den > 0) v -= d / den * lower[j];
    }
    out[j] = v;
  }
}

// values, first and second derivatives of the p + 1 non-zero functions of degree p
SB_HD void sb_basis2(const double* U, int p, int span, double xi, double* N, double* D1, double* D2) {
  double v1[kMaxBodyDegree + 1], v2[kMaxBodyDegree + 1], d1low[kMaxBodyDegree + 1];
  sb_values(U, p, span, xi, N);
  for (int j = 0; j <= p; ++j) D1[j] = D2[j] = 0.0;
  if (p < 1) return;
  sb_values(U, p - 1, span, xi, v1);
  sb_derive(U, p, span, v1, D1);
  if (p < 2) return;
  sb_values(U, p - 2, span, xi, v2);
  sb_derive(U, p - 1, span, v2, d1low);   // first derivatives of the degree p - 1 functions
  sb_derive(U, p, span, d1low, D2);
}

// S, dS/dxi_k, d2S/dxi_k dxi_l at xi.  S1[k*dim + i], S2[(k*para_dim + l)*dim + i]
SB_HD void sb_evaluate(const SplineBodyDev& b, const double* xi, double* S, double* S1, double* S2) {
  const int dim = b.dim, hd = dim + 1;
  double N[2][kMaxBodyDegree + 1], D1[2][kMaxBodyDegree + 1], D2[2][kMaxBodyDegree + 1];
  int span[2] = {0, 0};
  for (int k = 0; k < 2; ++k) {
    N[k][0] = 1.0;
    D1[k][0] = D2[k][0] = 0.0;
  }
  for (int k = 0; k < b.para_dim; ++k) {
    span[k] = sb_find_span(b.knots[k], b.n_knots[k], b.p[k], xi[k]);
    sb_basis2(b.knots[k], b.p[k], span[k], xi[k], N[k], D1[k], D2[k]);
  }
  // homogeneous sums: A, A_0, A_1, A_00, A_01, A_11 (4 components at most)
  double A[6][4];
  for (int t = 0; t < 6; ++t)
    for (int c = 0; c < 4; ++c) A[t][c] = 0.0;
  const int p0 = b.p[0], p1 = b.para_dim == 2 ? b.p[1] : 0;
  for (int a1 = 0; a1 <= p1; ++a1)
    for (int a0 = 0; a0 <= p0; ++a0) {
      const int i0 = span[0] - p0 + a0;
      const int i1 = b.para_dim == 2 ? span[1] - p1 + a1 : 0;
      const double* cp = b.ctrl_h + (size_t)(i0 + (size_t)b.n_ctrl[0] * i1) * hd;
      const double n0 = N[0][a0], d0 = D1[0][a0], dd0 = D2[0][a0];
      const double n1 = b.para_dim == 2 ? N[1][a1] : 1.0, d1 = b.para_dim == 2 ? D1[1][a1] : 0.0,
                   dd1 = b.para_dim == 2 ? D2[1][a1] : 0.0;
      const double w[6] = {n0 * n1, d0 * n1, n0 * d1, dd0 * n1, d0 * d1, n0 * dd1};
      for (int t = 0; t < 6; ++t)
        for (int c = 0; c < hd; ++c) A[t][c] += w[t] * cp[c];
    }
  const double W = A[0][dim];
  for (int i = 0; i < dim; ++i) S[i] = A[0][i] / W;
  // first derivatives: S_k = (A_k - W_k S) / W
  for (int k = 0; k < b.para_dim; ++k)
    for (int i = 0; i < dim; ++i) S1[k * dim + i] = (A[1 + k][i] - A[1 + k][dim] * S[i]) / W;
  // second: S_kl = (A_kl - W_kl S - W_k S_l - W_l S_k) / W
  for (int k = 0; k < b.para_dim; ++k)
    for (int l = 0; l < b.para_dim; ++l) {
      const int t = (k == 0 && l == 0) ? 3 : (k == 1 && l == 1) ? 5 : 4;
      for (int i = 0; i < dim; ++i)
        S2[(k * b.para_dim + l) * dim + i] =
            (A[t][i] - A[t][dim] * S[i] - A[1 + k][dim] * S1[l * dim + i] - A[1 + l][dim] * S1[k * dim + i]) / W;
    }
}

// closest point: xi (para_dim), S (dim), S1 (para_dim x dim)
SB_HD void sb_closest_point(const SplineBodyDev& b, const double* xq, double* xi, double* S, double* S1) {
  const int dim = b.dim, pd = b.para_dim;
  // initial guess: nearest sample (lowest index among equals)
  {
    double best = 1e300;
    int ib = 0;
    for (int s = 0; s < b.n_samples; ++s) {
      double d2 = 0.0;
      for (int i = 0; i < dim; ++i) {
        const double t = b.sample_x[(size_t)s * dim + i] - xq[i];
        d2 += t * t;
      }
      if (d2 < best) {
        best = d2;
        ib = s;
      }
    }
    for (int k = 0; k < pd; ++k) xi[k] = b.sample_xi[(size_t)ib * pd + k];
  }
  double lo[2], hi[2];
  for (int k = 0; k < pd; ++k) {
    lo[k] = b.knots[k][b.p[k]];
    hi[k] = b.knots[k][b.n_knots[k] - b.p[k] - 1];
  }
  double S2[12];
  sb_evaluate(b, xi, S, S1, S2);
  auto dist2 = [&](const double* P) {
    double d2 = 0.0;
    for (int i = 0; i < dim; ++i) d2 += (P[i] - xq[i]) * (P[i] - xq[i]);
    return d2;
  };
  double f = dist2(S);
  const int max_it = b.max_iterations > 0 ? b.max_iterations : 50;
  for (int it = 0; it < max_it; ++it) {
    double g[2] = {0, 0}, H[4] = {0, 0, 0, 0}, GN[4] = {0, 0, 0, 0};
    for (int k = 0; k < pd; ++k)
      for (int i = 0; i < dim; ++i) g[k] += S1[k * dim + i] * (S[i] - xq[i]);
    for (int k = 0; k < pd; ++k)
      for (int l = 0; l < pd; ++l) {
        double gn = 0.0, cv = 0.0;
        for (int i = 0; i < dim; ++i) {
          gn += S1[k * dim + i] * S1[l * dim + i];
          cv += S2[(k * pd + l) * dim + i] * (S[i] - xq[i]);
        }
        GN[k * pd + l] = gn;
        H[k * pd + l] = gn + cv;
      }
    // a coordinate pinned at a bound with the gradient pointing outwards does not move
    bool freek[2] = {true, true};
    for (int k = 0; k < pd; ++k)
      if ((xi[k] <= lo[k] && g[k] > 0.0) || (xi[k] >= hi[k] && g[k] < 0.0)) freek[k] = false;
    double delta[2] = {0, 0};
    auto solve = [&](const double* M) -> bool {
      if (pd == 1) {
        if (!freek[0] || !(M[0] > 0.0)) return false;
        delta[0] = -g[0] / M[0];
        return true;
      }
      if (freek[0] && freek[1]) {
        const double det = M[0] * M[3] - M[1] * M[2];
        if (!(det > 0.0) || !(M[0] > 0.0)) return false;
        delta[0] = -(M[3] * g[0] - M[1] * g[1]) / det;
        delta[1] = -(M[0] * g[1] - M[2] * g[0]) / det;
        return true;
      }
      delta[0] = delta[1] = 0.0;
      if (freek[0]) {
        if (!(M[0] > 0.0)) return false;
        delta[0] = -g[0] / M[0];
      } else if (freek[1]) {
        if (!(M[3] > 0.0)) return false;
        delta[1] = -g[1] / M[3];
      }
      return true;
    };
    if (!freek[0] && (pd == 1 || !freek[1])) break;
    if (!solve(H) && !solve(GN)) break;   // Newton, else Gauss-Newton
    // step, clipped to the bounds, halved while the distance grows
    double xn[2], Sn[3], S1n[6], S2n[12], fn = f;
    bool moved = false;
    double scale = 1.0;
    for (int half = 0; half < 8; ++half, scale *= 0.5) {
      moved = false;
      for (int k = 0; k < pd; ++k) {
        double v = xi[k] + scale * delta[k];
        v = v < lo[k] ? lo[k] : (v > hi[k] ? hi[k] : v);
        if (v != xi[k]) moved = true;
        xn[k] = v;
      }
      if (!moved) break;
      sb_evaluate(b, xn, Sn, S1n, S2n);
      fn = dist2(Sn);
      if (fn <= f) break;
    }
    if (!moved || fn > f) break;
    double step = 0.0;
    for (int k = 0; k < pd; ++k) {
      const double t = fabs(xn[k] - xi[k]);
      step = t > step ? t : step;
      xi[k] = xn[k];
    }
    for (int i = 0; i < dim; ++i) S[i] = Sn[i];
    for (int i = 0; i < pd * dim; ++i) S1[i] = S1n[i];
    for (int i = 0; i < pd * pd * dim; ++i) S2[i] = S2n[i];
    f = fn;
    if (step < 1e-15 * (hi[0] - lo[0])) break;
  }
}

// NearestDistance + Results::ComputeNormal<true> + NormalGap (nearest_distance.hpp:139-193) for the spline body
SB_HD void sb_nearest(const SplineBodyDev& b, const double* xq, double& true_g, double& distance) {
  double xi[2], S[3], S1[6];
  sb_closest_point(b, xq, xi, S, S1);
  double n[3];
  if (b.dim == 2) {
    const double d0 = S1[0], d1 = S1[1];
    const double inv = 1.0 / sqrt(d0 * d0 + d1 * d1);
    n[0] = d1 * inv;
    n[1] = -d0 * inv;
  } else {
    const double d0 = S1[0], d1 = S1[1], d2 = S1[2], d3 = S1[3], d4 = S1[4], d5 = S1[5];
    const double n0 = d1 * d5 - d2 * d4, n1 = d2 * d3 - d0 * d5, n2 = d0 * d4 - d1 * d3;
    const double inv = 1.0 / sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    n[0] = n0 * inv;
    n[1] = n1 * inv;
    n[2] = n2 * inv;
  }
  double g = 0.0, d2 = 0.0;
  for (int i = 0; i < b.dim; ++i) {
    const double pmq = S[i] - xq[i];
    g -= n[i] * pmq;
    d2 += pmq * pmq;
  }
  true_g = g;
  distance = sqrt(d2);
}

}  // namespace mimi_hip
