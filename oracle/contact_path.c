/* TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.
 *
 * Plain-C restatement of the reference's penalty mortar contact integrator
 * (src/mimi/integrators/mortar_contact.{hpp,cpp}, MortarContactWorkData in
 * integrators/integrator_utils.{hpp,cpp}, NearestDistanceBase::Results in
 * coefficients/nearest_distance.hpp).
 *
 * PARITY UNPINNED: no reference test exercises contact, and the
 * reference's closest-point query is splinepy's `SplinepyVerboseProximity`
 * (coefficients/nearest_distance.hpp:268-279) -- an un-vendored, un-pinned
 * submodule.  This restatement therefore substitutes an ANALYTIC rigid body
 * (sphere or half-space) for the query and follows the reference's own
 * arithmetic for everything downstream of it.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAX_FDOF 16  /* (p+1)^(dim-1), p<=3 */
#define MAX_FTDOF 48

enum { BODY_SPHERE = 0, BODY_PLANE = 1 };
enum { TANGENT_FD = 0, TANGENT_EXACT = 1 };

typedef struct {
  int dim, n_faces, n_dof, n_q, n_vdofs, n_marked;
  const int* v_dofs;     /* [n_faces][n_tdof]   component-grouped global v_dofs      */
  const int* a_ids;      /* [n_faces][n_tdof^2] column-major -> CSR position, or NULL */
  const int* local_dofs; /* [n_faces][n_dof]    index into the nodal arrays (mortar_contact.cpp:118-124) */
  const double* N;       /* [n_faces][n_q][n_dof]                                    */
  const double* dN_dxi;  /* [n_faces][n_q][dim-1][n_dof]  (n_dof x (dim-1)) col-major */
  const double* weight;  /* [n_faces][n_q]                                           */
  const double* x_ref;   /* [n_faces][dim][n_dof]  (n_dof x dim) col-major (mortar_contact.cpp:103-111) */
  int body_kind;
  double body[8];        /* sphere: c[3], R ; plane: x0[3], n[3] (unit, out of the rigid body) */
  double penalty;        /* NearestDistanceBase::coefficient_ (nearest_distance.hpp:18) */
  double* area;          /* [n_marked]  area_           */
  double* gap;           /* [n_marked]  average_gap_    */
  double* pressure;      /* [n_marked]  average_pressure_ */
  double last_area, last_pressure, last_force[3];
  /* body_kind == BODY_SPLINE: one boundary spline (NearestDistanceToSplines, nearest_distance.hpp:215-288) */
  int sp_para_dim, sp_p[2], sp_n_knots[2];
  const double* sp_knots[2];
  const double* sp_ctrl;      /* [n_ctrl][dim] */
  const double* sp_weights;   /* [n_ctrl] or NULL */
  int sp_resolution, sp_max_iterations;
} oracle_contact;

/* analytic stand-in for NearestDistance + Results::ComputeNormal<true> + NormalGap
 * (nearest_distance.hpp:139-193): rigid unit normal n_r, true gap
 * g = -n_r . (x_rigid - x_query), distance = |x_rigid - x_query| */
#define BODY_SPLINE 2
#define SP_MAXP 5

/* Piegl & Tiller A2.1 */
static int sp_span(const double* U, int n_knots, int p, double u) {
  const int n = n_knots - p - 2; /* last control index */
  if (u >= U[n + 1]) {
    int i = n;
    while (i > p && U[i] == U[i + 1]) --i;
    return i;
  }
  int low = p, high = n + 1, mid = (low + high) / 2;
  while (u < U[mid] || u >= U[mid + 1]) {
    if (u < U[mid]) high = mid; else low = mid;
    mid = (low + high) / 2;
  }
  return mid;
}

/* Piegl & Tiller A2.3 DersBasisFuns for derivatives 0..2: ders[k][j] */
static void sp_ders(const double* U, int p, int i, double u, double ders[3][SP_MAXP + 1]) {
  double ndu[SP_MAXP + 1][SP_MAXP + 1], a[2][SP_MAXP + 1], left[SP_MAXP + 1], right[SP_MAXP + 1];
  ndu[0][0] = 1.0;
  for (int j = 1; j <= p; ++j) {
    left[j] = u - U[i + 1 - j];
    right[j] = U[i + j] - u;
    double saved = 0.0;
    for (int r = 0; r < j; ++r) {
      ndu[j][r] = right[r + 1] + left[j - r];
      const double temp = ndu[r][j - 1] / ndu[j][r];
      ndu[r][j] = saved + right[r + 1] * temp;
      saved = left[j - r] * temp;
    }
    ndu[j][j] = saved;
  }
  for (int j = 0; j <= p; ++j) ders[0][j] = ndu[j][p];
  for (int j = 0; j <= p; ++j) ders[1][j] = ders[2][j] = 0.0;
  const int nd = p < 2 ? p : 2;
  for (int r = 0; r <= p; ++r) {
    int s1 = 0, s2 = 1;
    a[0][0] = 1.0;
    for (int k = 1; k <= nd; ++k) {
      double d = 0.0;
      const int rk = r - k, pk = p - k;
      if (r >= k) {
        a[s2][0] = a[s1][0] / ndu[pk + 1][rk];
        d = a[s2][0] * ndu[rk][pk];
      }
      const int j1 = rk >= -1 ? 1 : -rk;
      const int j2 = (r - 1 <= pk) ? k - 1 : p - r;
      for (int j = j1; j <= j2; ++j) {
        a[s2][j] = (a[s1][j] - a[s1][j - 1]) / ndu[pk + 1][rk + j];
        d += a[s2][j] * ndu[rk + j][pk];
      }
      if (r <= pk) {
        a[s2][k] = -a[s1][k - 1] / ndu[pk + 1][r];
        d += a[s2][k] * ndu[r][pk];
      }
      ders[k][r] = d;
      const int t = s1;
      s1 = s2;
      s2 = t;
    }
  }
  double fac = p;
  for (int k = 1; k <= nd; ++k) {
    for (int j = 0; j <= p; ++j) ders[k][j] *= fac;
    fac *= (p - k);
  }
}

/* S, S_k, S_kl of the rational spline (quotient rule on the homogeneous sums) */
static void sp_eval(const oracle_contact* C, const double* xi, double* S, double* S1, double* S2) {
  const int dim = C->dim, pd = C->sp_para_dim;
  double d[2][3][SP_MAXP + 1];
  int span[2] = {0, 0}, p[2] = {C->sp_p[0], pd == 2 ? C->sp_p[1] : 0};
  for (int k = 0; k < pd; ++k) {
    span[k] = sp_span(C->sp_knots[k], C->sp_n_knots[k], p[k], xi[k]);
    sp_ders(C->sp_knots[k], p[k], span[k], xi[k], d[k]);
  }
  if (pd == 1) {
    d[1][0][0] = 1.0;
    d[1][1][0] = d[1][2][0] = 0.0;
  }
  const int n0 = C->sp_n_knots[0] - p[0] - 1;
  /* derivative orders (o0, o1) of the six sums */
  static const int ord[6][2] = {{0, 0}, {1, 0}, {0, 1}, {2, 0}, {1, 1}, {0, 2}};
  double A[6][4];
  memset(A, 0, sizeof(A));
  for (int a1 = 0; a1 <= p[1]; ++a1)
    for (int a0 = 0; a0 <= p[0]; ++a0) {
      const long node = (span[0] - p[0] + a0) + (long)n0 * (pd == 2 ? span[1] - p[1] + a1 : 0);
      const double w = C->sp_weights ? C->sp_weights[node] : 1.0;
      for (int t = 0; t < 6; ++t) {
        const double b = d[0][ord[t][0]][a0] * d[1][ord[t][1]][a1] * w;
        for (int i = 0; i < dim; ++i) A[t][i] += b * C->sp_ctrl[node * dim + i];
        A[t][dim] += b;
      }
    }
  const double W = A[0][dim];
  for (int i = 0; i < dim; ++i) S[i] = A[0][i] / W;
  for (int k = 0; k < pd; ++k)
    for (int i = 0; i < dim; ++i) S1[k * dim + i] = (A[1 + k][i] - A[1 + k][dim] * S[i]) / W;
  for (int k = 0; k < pd; ++k)
    for (int l = 0; l < pd; ++l) {
      const int t = (k == 0 && l == 0) ? 3 : (k == 1 && l == 1) ? 5 : 4;
      for (int i = 0; i < dim; ++i)
        S2[(k * pd + l) * dim + i] =
            (A[t][i] - A[t][dim] * S[i] - A[1 + k][dim] * S1[l * dim + i] - A[1 + l][dim] * S1[k * dim + i]) / W;
    }
}

/* the published scheme behind SplinepyVerboseProximity (nearest_distance.hpp:268-279; splinepy is absent): nearest of
 * resolution^para_dim samples, then Newton on the squared distance, clipped to the bounds, halved while it grows */
static void sp_nearest(const oracle_contact* C, const double* xq, double* true_g, double* distance) {
  const int dim = C->dim, pd = C->sp_para_dim, res = C->sp_resolution;
  double lo[2] = {0, 0}, hi[2] = {1, 1}, xi[2] = {0, 0}, S[3], S1[6], S2[12];
  for (int k = 0; k < pd; ++k) {
    lo[k] = C->sp_knots[k][C->sp_p[k]];
    hi[k] = C->sp_knots[k][C->sp_n_knots[k] - C->sp_p[k] - 1];
  }
  {
    const int n_s = pd == 2 ? res * res : res;
    double best = 1e300;
    for (int s = 0; s < n_s; ++s) {
      double x[2] = {0, 0};
      const int idx[2] = {s % res, s / res};
      for (int k = 0; k < pd; ++k) x[k] = lo[k] + (hi[k] - lo[k]) * idx[k] / (res - 1);
      sp_eval(C, x, S, S1, S2);
      double d2 = 0;
      for (int i = 0; i < dim; ++i) d2 += (S[i] - xq[i]) * (S[i] - xq[i]);
      if (d2 < best) {
        best = d2;
        xi[0] = x[0];
        xi[1] = x[1];
      }
    }
  }
  sp_eval(C, xi, S, S1, S2);
  double f = 0;
  for (int i = 0; i < dim; ++i) f += (S[i] - xq[i]) * (S[i] - xq[i]);
  const int max_it = C->sp_max_iterations > 0 ? C->sp_max_iterations : 50;
  for (int it = 0; it < max_it; ++it) {
    double g[2] = {0, 0}, H[4] = {0}, GN[4] = {0}, delta[2] = {0, 0};
    int fr[2] = {1, 1};
    for (int k = 0; k < pd; ++k)
      for (int i = 0; i < dim; ++i) g[k] += S1[k * dim + i] * (S[i] - xq[i]);
    for (int k = 0; k < pd; ++k)
      for (int l = 0; l < pd; ++l) {
        double gn = 0, cv = 0;
        for (int i = 0; i < dim; ++i) {
          gn += S1[k * dim + i] * S1[l * dim + i];
          cv += S2[(k * pd + l) * dim + i] * (S[i] - xq[i]);
        }
        GN[k * pd + l] = gn;
        H[k * pd + l] = gn + cv;
      }
    for (int k = 0; k < pd; ++k)
      if ((xi[k] <= lo[k] && g[k] > 0.0) || (xi[k] >= hi[k] && g[k] < 0.0)) fr[k] = 0;
    if (!fr[0] && (pd == 1 || !fr[1])) break;
    int ok = 0;
    for (int attempt = 0; attempt < 2 && !ok; ++attempt) {
      const double* Mx = attempt == 0 ? H : GN;
      delta[0] = delta[1] = 0.0;
      if (pd == 1) {
        if (Mx[0] > 0.0) {
          delta[0] = -g[0] / Mx[0];
          ok = 1;
        }
      } else if (fr[0] && fr[1]) {
        const double det = Mx[0] * Mx[3] - Mx[1] * Mx[2];
        if (det > 0.0 && Mx[0] > 0.0) {
          delta[0] = -(Mx[3] * g[0] - Mx[1] * g[1]) / det;
          delta[1] = -(Mx[0] * g[1] - Mx[2] * g[0]) / det;
          ok = 1;
        }
      } else if (fr[0]) {
        if (Mx[0] > 0.0) {
          delta[0] = -g[0] / Mx[0];
          ok = 1;
        }
      } else {
        if (Mx[3] > 0.0) {
          delta[1] = -g[1] / Mx[3];
          ok = 1;
        }
      }
    }
    if (!ok) break;
    double xn[2] = {0, 0}, Sn[3], S1n[6], S2n[12], fn = f, scale = 1.0;
    int moved = 0;
    for (int half = 0; half < 8; ++half, scale *= 0.5) {
      moved = 0;
      for (int k = 0; k < pd; ++k) {
        double v = xi[k] + scale * delta[k];
        v = v < lo[k] ? lo[k] : (v > hi[k] ? hi[k] : v);
        if (v != xi[k]) moved = 1;
        xn[k] = v;
      }
      if (!moved) break;
      sp_eval(C, xn, Sn, S1n, S2n);
      fn = 0;
      for (int i = 0; i < dim; ++i) fn += (Sn[i] - xq[i]) * (Sn[i] - xq[i]);
      if (fn <= f) break;
    }
    if (!moved || fn > f) break;
    double step = 0;
    for (int k = 0; k < pd; ++k) {
      if (fabs(xn[k] - xi[k]) > step) step = fabs(xn[k] - xi[k]);
      xi[k] = xn[k];
    }
    memcpy(S, Sn, sizeof(Sn));
    memcpy(S1, S1n, sizeof(S1n));
    memcpy(S2, S2n, sizeof(S2n));
    f = fn;
    if (step < 1e-15 * (hi[0] - lo[0])) break;
  }
  /* Results::ComputeNormal<true> (nearest_distance.hpp:139-184) and NormalGap (:186-193) */
  double n[3];
  if (dim == 2) {
    const double nn = sqrt(S1[0] * S1[0] + S1[1] * S1[1]);
    n[0] = S1[1] / nn;
    n[1] = -S1[0] / nn;
  } else {
    const double n0 = S1[1] * S1[5] - S1[2] * S1[4], n1 = S1[2] * S1[3] - S1[0] * S1[5], n2 = S1[0] * S1[4] - S1[1] * S1[3];
    const double nn = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
    n[0] = n0 / nn;
    n[1] = n1 / nn;
    n[2] = n2 / nn;
  }
  double g = 0, d2 = 0;
  for (int i = 0; i < dim; ++i) {
    g -= n[i] * (S[i] - xq[i]);
    d2 += (S[i] - xq[i]) * (S[i] - xq[i]);
  }
  *true_g = g;
  *distance = sqrt(d2);
}

static void nearest(const oracle_contact* C, const double* xq, double* true_g, double* distance) {
  const int dim = C->dim;
  if (C->body_kind == BODY_SPLINE) {
    sp_nearest(C, xq, true_g, distance);
    return;
  }
  if (C->body_kind == BODY_SPHERE) {
    double d[3], nrm = 0;
    for (int i = 0; i < dim; ++i) {
      d[i] = xq[i] - C->body[i];
      nrm += d[i] * d[i];
    }
    nrm = sqrt(nrm);
    const double R = C->body[3];
    double pmq[3], n[3], g = 0, dist = 0;
    for (int i = 0; i < dim; ++i) {
      n[i] = d[i] / nrm;
      pmq[i] = (C->body[i] + R * n[i]) - xq[i];
      g -= n[i] * pmq[i];
      dist += pmq[i] * pmq[i];
    }
    *true_g = g;
    *distance = sqrt(dist);
  } else {
    double s = 0, dist = 0, g = 0;
    const double* n = C->body + 3;
    for (int i = 0; i < dim; ++i) s += (xq[i] - C->body[i]) * n[i];
    for (int i = 0; i < dim; ++i) {
      const double pmq = -s * n[i];
      g -= n[i] * pmq;
      dist += pmq * pmq;
    }
    *true_g = g;
    *distance = sqrt(dist);
  }
}

/* integrator_utils.cpp:80-89: x_e = u[v_dofs] + X_ref */
static void current_x(const oracle_contact* C, int f, const double* u, double* x_e) {
  const int nt = C->n_dof * C->dim;
  const int* vd = C->v_dofs + (long)f * nt;
  const double* xr = C->x_ref + (long)f * nt;
  for (int k = 0; k < nt; ++k) x_e[k] = u[vd[k]] + xr[k];
}

/* integrator_utils.cpp:101-105 ComputeJ = x_e^T dN_dxi (dim x (dim-1)); J.Weight();
 * integrator_utils.hpp:216-251 ComputeUnitNormal.  Returns det_J, fills unit normal. */
static double jac_normal(int dim, int n_dof, const double* x_e, const double* dNdxi, double* normal) {
  double J[6];
  for (int k = 0; k < dim - 1; ++k)
    for (int i = 0; i < dim; ++i) {
      double s = 0;
      for (int a = 0; a < n_dof; ++a) s += x_e[a + i * n_dof] * dNdxi[a + k * n_dof];
      J[i + k * dim] = s;
    }
  if (dim == 2) {
    const double d0 = J[0], d1 = J[1];
    const double nrm = sqrt(d0 * d0 + d1 * d1);
    normal[0] = d1 / nrm;
    normal[1] = -d0 / nrm;
    return nrm;
  }
  const double n0 = J[1] * J[5] - J[2] * J[4];
  const double n1 = J[2] * J[3] - J[0] * J[5];
  const double n2 = J[0] * J[4] - J[1] * J[3];
  const double nrm = sqrt(n0 * n0 + n1 * n1 + n2 * n2);
  normal[0] = n0 / nrm;
  normal[1] = n1 / nrm;
  normal[2] = n2 / nrm;
  return nrm;
}

/* mortar_contact.cpp:148-193 ElementGapAndArea + :195-261 ComputePressure */
static void compute_pressure(oracle_contact* C, const double* u, double* area_total) {
  const int dim = C->dim, nd = C->n_dof;
  memset(C->area, 0, sizeof(double) * C->n_marked);
  memset(C->gap, 0, sizeof(double) * C->n_marked);
  memset(C->pressure, 0, sizeof(double) * C->n_marked);
  for (int f = 0; f < C->n_faces; ++f) {
    double x_e[MAX_FTDOF], la[MAX_FDOF], lg[MAX_FDOF], nrm[3];
    current_x(C, f, u, x_e);
    for (int i = 0; i < nd; ++i) la[i] = lg[i] = 0.0;
    for (int q = 0; q < C->n_q; ++q) {
      const long pt = (long)f * C->n_q + q;
      const double* N = C->N + pt * nd;
      double xq[3];
      for (int i = 0; i < dim; ++i) {
        double s = 0;
        for (int a = 0; a < nd; ++a) s += x_e[a + i * nd] * N[a];
        xq[i] = s;
      }
      double true_g, distance;
      nearest(C, xq, &true_g, &distance);
      double g = true_g < 0. ? true_g : 0.;
      const double detJ = jac_normal(dim, nd, x_e, C->dN_dxi + pt * nd * (dim - 1), nrm);
      const double fac = C->weight[pt] * detJ;
      *area_total += fac;
      const double ratio = fabs(true_g) / distance;
      if (acos(ratio < 1. ? ratio : 1.) > 1.e-5) g = 0.0;
      const double fac_g = fac * g;
      for (int i = 0; i < nd; ++i) {
        la[i] += fac * N[i];
        lg[i] += fac_g * N[i];
      }
    }
    const int* ld = C->local_dofs + (long)f * nd;
    for (int i = 0; i < nd; ++i) {
      C->area[ld[i]] += la[i];
      C->gap[ld[i]] += lg[i];
    }
  }
  for (int i = 0; i < C->n_marked; ++i) C->pressure[i] = C->gap[i] / C->area[i] * C->penalty;
}

/* mortar_contact.hpp:99-134 ElementResidual<record> */
static void element_residual(const oracle_contact* C, int f, const double* x_e, const double* p_e,
                             double* R_e, int record, double* force, double* pressure_integral) {
  const int dim = C->dim, nd = C->n_dof;
  memset(R_e, 0, sizeof(double) * nd * dim);
  for (int q = 0; q < C->n_q; ++q) {
    const long pt = (long)f * C->n_q + q;
    const double* N = C->N + pt * nd;
    double p = 0, nrm[3];
    for (int i = 0; i < nd; ++i) p += N[i] * p_e[i];
    const double detJ = jac_normal(dim, nd, x_e, C->dN_dxi + pt * nd * (dim - 1), nrm);
    const double fac = C->weight[pt] * detJ * p;
    for (int i = 0; i < dim; ++i) {
      const double aw = nrm[i] * (-fac);
      for (int a = 0; a < nd; ++a) R_e[a + i * nd] += aw * N[a];
    }
    if (record) {
      for (int i = 0; i < dim; ++i) force[i] += fac * nrm[i];
      *pressure_integral += fac;
    }
  }
}

/* exact derivative of the above wrt x_e with p frozen (the reference's FD also
 * freezes p: mortar_contact.cpp:281-294).  K_e column-major (n_tdof x n_tdof). */
static void element_grad_exact(const oracle_contact* C, int f, const double* x_e, const double* p_e,
                               double* K_e) {
  const int dim = C->dim, nd = C->n_dof, nt = nd * dim;
  memset(K_e, 0, sizeof(double) * nt * nt);
  for (int q = 0; q < C->n_q; ++q) {
    const long pt = (long)f * C->n_q + q;
    const double* N = C->N + pt * nd;
    const double* dN = C->dN_dxi + pt * nd * (dim - 1);
    double p = 0;
    for (int i = 0; i < nd; ++i) p += N[i] * p_e[i];
    const double wp = C->weight[pt] * p;
    double t[6];
    for (int k = 0; k < dim - 1; ++k)
      for (int i = 0; i < dim; ++i) {
        double s = 0;
        for (int a = 0; a < nd; ++a) s += x_e[a + i * nd] * dN[a + k * nd];
        t[i + k * dim] = s;
      }
    /* R(a,i) = -w p N_a m_i,  m = t1 x t2 (3-D)  or (t_y, -t_x) (2-D) */
    for (int b = 0; b < nd; ++b)
      for (int j = 0; j < dim; ++j) {
        double dm[3] = {0, 0, 0};
        if (dim == 2) {
          if (j == 1) dm[0] = dN[b];
          if (j == 0) dm[1] = -dN[b];
        } else {
          /* d(t1 x t2)/dx_bj = dN1_b (e_j x t2) + dN2_b (t1 x e_j) */
          double e[3] = {0, 0, 0};
          e[j] = 1.0;
          const double* t1 = t;
          const double* t2 = t + 3;
          const double d1 = dN[b], d2 = dN[b + nd];
          dm[0] = d1 * (e[1] * t2[2] - e[2] * t2[1]) + d2 * (t1[1] * e[2] - t1[2] * e[1]);
          dm[1] = d1 * (e[2] * t2[0] - e[0] * t2[2]) + d2 * (t1[2] * e[0] - t1[0] * e[2]);
          dm[2] = d1 * (e[0] * t2[1] - e[1] * t2[0]) + d2 * (t1[0] * e[1] - t1[1] * e[0]);
        }
        double* col = K_e + (long)(b + j * nd) * nt;
        for (int i = 0; i < dim; ++i)
          for (int a = 0; a < nd; ++a) col[a + i * nd] += -wp * N[a] * dm[i];
      }
  }
}

static int gather_pressure(const oracle_contact* C, int f, double* p_e) {
  const int* ld = C->local_dofs + (long)f * C->n_dof;
  int nonzero = 0;
  for (int i = 0; i < C->n_dof; ++i) {
    p_e[i] = C->pressure[ld[i]];
    if (p_e[i] != 0.0) nonzero = 1;
  }
  return nonzero; /* IsPressureZero, integrator_utils.cpp:112-119 */
}

/* mortar_contact.cpp:297-351 AddBoundaryResidual */
int oracle_contact_add_residual(oracle_contact* C, const double* u, double* r) {
  const int dim = C->dim, nt = C->n_dof * dim;
  C->last_area = 0.0;
  compute_pressure(C, u, &C->last_area);
  C->last_pressure = 0.0;
  for (int i = 0; i < 3; ++i) C->last_force[i] = 0.0;
  for (int f = 0; f < C->n_faces; ++f) {
    double p_e[MAX_FDOF], x_e[MAX_FTDOF], R_e[MAX_FTDOF];
    if (!gather_pressure(C, f, p_e)) continue;
    current_x(C, f, u, x_e);
    element_residual(C, f, x_e, p_e, R_e, 1, C->last_force, &C->last_pressure);
    const int* vd = C->v_dofs + (long)f * nt;
    for (int k = 0; k < nt; ++k) r[vd[k]] += R_e[k];
  }
  return 0;
}

/* mortar_contact.cpp:263-295 ElementResidualAndGrad (forward FD on the current
 * POSITION x_e = u_e + X_ref, step |x_i|*1e-8 or 1e-10, pressure frozen) and
 * :353-421 AddBoundaryResidualAndGrad */
int oracle_contact_add_residual_and_grad(oracle_contact* C, const double* u, double grad_factor,
                                         double* r, double* A, int mode) {
  const int dim = C->dim, nt = C->n_dof * dim;
  C->last_area = 0.0;
  compute_pressure(C, u, &C->last_area);
  C->last_pressure = 0.0;
  for (int i = 0; i < 3; ++i) C->last_force[i] = 0.0;
  double* K_e = (double*)malloc(sizeof(double) * nt * nt);
  for (int f = 0; f < C->n_faces; ++f) {
    double p_e[MAX_FDOF], x_e[MAX_FTDOF], R_e[MAX_FTDOF], fwd[MAX_FTDOF];
    if (!gather_pressure(C, f, p_e)) continue;
    current_x(C, f, u, x_e);
    element_residual(C, f, x_e, p_e, R_e, 1, C->last_force, &C->last_pressure);
    if (mode == TANGENT_FD) {
      double* g = K_e;
      double dummy_f[3], dummy_p;
      for (int i = 0; i < nt; ++i) {
        const double orig = x_e[i];
        const double step = (orig != 0.0) ? fabs(orig) * 1.0e-8 : 1.0e-10;
        const double step_inv = 1. / step;
        x_e[i] = orig + step;
        element_residual(C, f, x_e, p_e, fwd, 0, dummy_f, &dummy_p);
        for (int j = 0; j < nt; ++j) *g++ = (fwd[j] - R_e[j]) * step_inv;
        x_e[i] = orig;
      }
    } else {
      element_grad_exact(C, f, x_e, p_e, K_e);
    }
    const int* vd = C->v_dofs + (long)f * nt;
    for (int k = 0; k < nt; ++k) r[vd[k]] += R_e[k];
    const int* ids = C->a_ids + (long)f * nt * nt;
    for (int k = 0; k < nt * nt; ++k) A[ids[k]] += K_e[k] * grad_factor;
  }
  free(K_e);
  return 0;
}

/* mortar_contact.cpp:423-467 GapNorm */
double oracle_contact_gap_norm(const oracle_contact* C, const double* u) {
  const int dim = C->dim, nd = C->n_dof;
  double total = 0;
  for (int f = 0; f < C->n_faces; ++f) {
    double x_e[MAX_FTDOF];
    current_x(C, f, u, x_e);
    for (int q = 0; q < C->n_q; ++q) {
      const double* N = C->N + ((long)f * C->n_q + q) * nd;
      double xq[3];
      for (int i = 0; i < dim; ++i) {
        double s = 0;
        for (int a = 0; a < nd; ++a) s += x_e[a + i * nd] * N[a];
        xq[i] = s;
      }
      double g, dist;
      nearest(C, xq, &g, &dist);
      if (g < 0.0) total += g * g;
    }
  }
  return sqrt(total);
}
