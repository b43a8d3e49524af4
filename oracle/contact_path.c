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
} oracle_contact;

/* analytic stand-in for NearestDistance + Results::ComputeNormal<true> + NormalGap
 * (nearest_distance.hpp:139-193): rigid unit normal n_r, true gap
 * g = -n_r . (x_rigid - x_query), distance = |x_rigid - x_query| */
static void nearest(const oracle_contact* C, const double* xq, double* true_g, double* distance) {
  const int dim = C->dim;
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
