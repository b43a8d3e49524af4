// Host-side 1-D B-spline tables for a tensor-product patch: per direction, per non-empty
// knot span, the p+1 non-zero basis functions and their derivatives (wrt the span's
// reference coordinate in [0,1]) at the Gauss points of the span.  These are the factors
// of what the reference asks MFEM for per element and quadrature point
// (CalcShape / CalcDShape, utils/precomputed.cpp:307-311) with quadrature order
// 2p+3 -> order/2+1 Gauss-Legendre points per direction (utils/precomputed.cpp:284-290).
#pragma once

#include <cmath>
#include <vector>

#include "common.hpp"

namespace mimi_hip {

// n-point Gauss-Legendre on [0,1] by Newton iteration on P_n
inline void gauss_legendre_01(int n, std::vector<double>& x, std::vector<double>& w) {
  x.assign(n, 0.0);
  w.assign(n, 0.0);
  const double pi = 3.14159265358979323846;
  for (int i = 0; i < (n + 1) / 2; ++i) {
    double z = std::cos(pi * (i + 0.75) / (n + 0.5));
    double pp = 1.0;
    for (int it = 0; it < 100; ++it) {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
      const double z1 = z;
      z = z1 - p1 / pp;
      if (std::fabs(z - z1) < 1e-16) break;
    }
    // one more evaluation so that pp matches the converged z
    {
      double p1 = 1.0, p2 = 0.0;
      for (int j = 1; j <= n; ++j) {
        const double p3 = p2;
        p2 = p1;
        p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
      }
      pp = n * (z * p1 - p2) / (z * z - 1.0);
    }
    const double wi = 2.0 / ((1.0 - z * z) * pp * pp);
    x[i] = 0.5 * (1.0 - z);
    x[n - 1 - i] = 0.5 * (1.0 + z);
    w[i] = w[n - 1 - i] = 0.5 * wi;
  }
}

// Cox-de Boor recursion, all degrees up to p on one span, triangular table.
// N[j] = N_{span-p+j, p}(xi), dN[j] = d/dxi of the same.
inline void bspline_basis(const double* U, int p, int span, double xi, double* N, double* dN) {
  // values of degree p-1 (for the derivative) and p
  std::vector<double> lo(p + 2, 0.0), cur(p + 2, 0.0);
  cur[0] = 1.0;  // degree 0 on [U[span], U[span+1])
  std::vector<double> prev;
  for (int d = 1; d <= p; ++d) {
    prev = cur;
    std::fill(cur.begin(), cur.end(), 0.0);
    // functions N_{span-d+j, d}, j = 0..d, from N_{span-d+1+j', d-1}, j' = 0..d-1 (prev[j'])
    for (int j = 0; j <= d; ++j) {
      const int i = span - d + j;
      double v = 0.0;
      if (j >= 1) {
        const double den = U[i + d] - U[i];
        if (den > 0) v += (xi - U[i]) / den * prev[j - 1];
      }
      if (j <= d - 1) {
        const double den = U[i + d + 1] - U[i + 1];
        if (den > 0) v += (U[i + d + 1] - xi) / den * prev[j];
      }
      cur[j] = v;
    }
    if (d == p - 1) lo = cur;
  }
  if (p == 1) {
    lo.assign(p + 2, 0.0);
    lo[0] = 1.0;
  }
  for (int j = 0; j <= p; ++j) {
    N[j] = cur[j];
    const int i = span - p + j;
    double d = 0.0;
    if (p >= 1) {
      // N'_{i,p} = p/(U[i+p]-U[i]) N_{i,p-1} - p/(U[i+p+1]-U[i+1]) N_{i+1,p-1}
      if (j >= 1) {
        const double den = U[i + p] - U[i];
        if (den > 0) d += p / den * lo[j - 1];
      }
      if (j <= p - 1) {
        const double den = U[i + p + 1] - U[i + 1];
        if (den > 0) d -= p / den * lo[j];
      }
    }
    dN[j] = d;
  }
}

struct Tables1D {
  int p = 0, n_ctrl = 0, n_spans = 0, nq = 0;
  std::vector<int> first;     // [n_spans] first non-zero basis index of the span
  std::vector<double> B, D;   // [n_spans][p+1][nq]
  std::vector<double> w;      // [nq]
};

// w1d (or nullptr): the 1-D weights of a patch whose NURBS weights are a tensor product w[a0,a1,a2] = w0[a0] w1[a1] w2[a2]
// -- the rational basis is then the tensor product of the 1-D rational bases R_a = w_a N_a / sum_b w_b N_b, and the
// tables hold R_a and dR_a/dxi.
inline Tables1D make_tables_1d(const double* knots, int n_knots, int p, int nq, const double* w1d = nullptr) {
  Tables1D t;
  t.p = p;
  t.nq = nq;
  t.n_ctrl = n_knots - p - 1;
  if (t.n_ctrl < p + 1) fail("knot vector too short for degree %d", p);
  std::vector<double> x;
  gauss_legendre_01(nq, x, t.w);
  for (int s = p; s < n_knots - p - 1; ++s) {
    const double h = knots[s + 1] - knots[s];
    if (!(h > 0)) continue;
    t.first.push_back(s - p);
    for (int a = 0; a <= p; ++a)
      for (int q = 0; q < nq; ++q) {
        t.B.push_back(0.0);
        t.D.push_back(0.0);
      }
    const size_t base = (t.first.size() - 1) * (size_t)(p + 1) * nq;
    std::vector<double> N(p + 1), dN(p + 1);
    for (int q = 0; q < nq; ++q) {
      bspline_basis(knots, p, s, knots[s] + x[q] * h, N.data(), dN.data());
      if (w1d) {
        double W = 0.0, dW = 0.0;
        for (int a = 0; a <= p; ++a) {
          W += w1d[s - p + a] * N[a];
          dW += w1d[s - p + a] * dN[a];
        }
        for (int a = 0; a <= p; ++a) {
          const double wa = w1d[s - p + a];
          const double Ra = wa * N[a] / W;
          dN[a] = wa * (dN[a] * W - N[a] * dW) / (W * W);
          N[a] = Ra;
        }
      }
      for (int a = 0; a <= p; ++a) {
        t.B[base + (size_t)a * nq + q] = N[a];
        t.D[base + (size_t)a * nq + q] = dN[a] * h;
      }
    }
  }
  t.n_spans = (int)t.first.size();
  return t;
}

}  // namespace mimi_hip
