// C ABI + kernels of the contact integrator: integrators::MortarContact
// (reference: src/mimi/integrators/mortar_contact.{hpp,cpp}, MortarContactWorkData in
// integrators/integrator_utils.{hpp,cpp}, NearestDistanceBase::Results in
// coefficients/nearest_distance.hpp:54-194).
//
// The reference's closest-point query is splinepy's proximity search (an absent, un-pinned
// third-party library, nearest_distance.hpp:268-279); here the rigid body is analytic
// (sphere / half space).  Everything downstream of the query follows the reference:
//   pass 1  ElementGapAndArea + ComputePressure   mortar_contact.cpp:148-261
//           nodal area A_i += w |J| N_i, gap G_i += w |J| g N_i, p_i = eps G_i / A_i
//   pass 2  ElementResidual                        mortar_contact.hpp:99-134
//           R(a,i) += -(w |J| p_q) N_a n_i   for faces with any non-zero nodal pressure
//   tangent with the pressure frozen               mortar_contact.cpp:263-295 (FD in the reference;
//           analytic here, or the reference's FD rule in MIMI_HIP_TANGENT_REFERENCE_FD mode)
// Instead of per-thread copies + a mutex (mortar_contact.cpp:235-260,338-341,400-408) every contribution is stored
// densely -- per quadrature point the nodal area / gap shares, per face the residual vector and the tangent block --
// and summed afterwards in a fixed order through the marked nodes' (face, local node) incidences: no atomics, results
// bitwise reproducible like the domain paths (round 3; rounds 1-2 used fp64 atomics here).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <memory>
#include <vector>

#include "common.hpp"
#include "materials.hpp"
#include "spline_body.hpp"

namespace mimi_hip {

struct ContactArgs {
  SplineBodyDev spline;      // body_kind == MIMI_HIP_BODY_SPLINE
  int dim, n_faces, n_dof, n_q;
  const int32_t* dofs;       // [n_faces][n_dof] global node ids
  const int32_t* local;      // [n_faces][n_dof] index into the nodal arrays
  const double* N;           // [n_faces][n_q][n_dof]
  const double* dN;          // [n_faces][n_q][dim-1][n_dof]
  const double* weight;      // [n_faces][n_q]
  const double* x_ref;       // [n_nodes][dim]
  const int64_t* rowptr;
  const int32_t* pair_pos;   // [n_faces][n_dof][n_dof]
  int body_kind;
  double body[8];
  double penalty;
  const double* u;
  double* r;
  double* A;
  double grad_factor;
  double* area;              // [n_marked]
  double* gap;
  double* pressure;
  double* scalars;           // [0] area, [1] pressure integral, [2..4] force, [5] gap norm^2
  int mode;
  // No atomics (round 3): every contribution is stored densely and summed afterwards in a fixed order --
  // bitwise reproducible like the domain paths.
  double* pt_area;           // [n_faces][n_q][n_dof]  w |J| N_a        of a quadrature point
  double* pt_gap;            // [n_faces][n_q][n_dof]  w |J| g N_a
  double* pt_scal;           // [n_faces][n_q]         w |J|  (GapNorm: min(g, 0)^2)
  double* face_r;            // [n_faces][dim][n_dof]  face residual vectors
  double* face_k;            // [n_faces][(a, i)][(j, b)]  face tangent blocks
  double* face_scal;         // [n_faces][1 + dim]     pressure integral, force
  unsigned char* face_active;   // IsPressureZero == false
};

constexpr int kMaxFaceDof = 16;

// analytic stand-in for NearestDistance + ComputeNormal<true> + NormalGap
// (nearest_distance.hpp:139-193): true gap and |x_rigid - x_query|
template<int DIM>
MH_DEV void nearest_body(const ContactArgs& p, const double* xq, double& true_g, double& distance) {
  if (p.body_kind == MIMI_HIP_BODY_SPHERE) {
    double d[DIM], nrm = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      d[i] = xq[i] - p.body[i];
      nrm += d[i] * d[i];
    }
    nrm = sqrt(nrm);
    const double R = p.body[3];
    double g = 0, dist = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double n = d[i] / nrm;
      const double pmq = (p.body[i] + R * n) - xq[i];
      g -= n * pmq;
      dist += pmq * pmq;
    }
    true_g = g;
    distance = sqrt(dist);
  } else if (p.body_kind == MIMI_HIP_BODY_SPLINE) {
    sb_nearest(p.spline, xq, true_g, distance);
  } else {
    double s = 0, dist = 0, g = 0;
#pragma unroll
    for (int i = 0; i < DIM; ++i) s += (xq[i] - p.body[i]) * p.body[3 + i];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double pmq = -s * p.body[3 + i];
      g -= p.body[3 + i] * pmq;
      dist += pmq * pmq;
    }
    true_g = g;
    distance = sqrt(dist);
  }
}

// J = x_e^T dN_dxi (integrator_utils.cpp:101-105); returns |J| (DenseMatrix::Weight) and the
// NON-normalised normal m (ComputeUnitNormal before the division, integrator_utils.hpp:216-251)
template<int DIM>
MH_DEV double surface_normal(int n_dof, const double* x_e /*[DIM][n_dof]*/, const double* dN /*[DIM-1][n_dof]*/,
                             double* m, double* t /*[DIM-1][DIM]*/) {
#pragma unroll
  for (int k = 0; k < DIM - 1; ++k)
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      double s = 0;
      for (int a = 0; a < n_dof; ++a) s += x_e[i * n_dof + a] * dN[k * n_dof + a];
      t[k * DIM + i] = s;
    }
  if constexpr (DIM == 2) {
    m[0] = t[1];
    m[1] = -t[0];
    return sqrt(m[0] * m[0] + m[1] * m[1]);
  } else {
    m[0] = t[1] * t[5] - t[2] * t[4];
    m[1] = t[2] * t[3] - t[0] * t[5];
    m[2] = t[0] * t[4] - t[1] * t[3];
    return sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
  }
}

template<int DIM>
MH_DEV void gather_x(const ContactArgs& p, int f, double* x_e) {
  // integrator_utils.cpp:80-89: x_e = u[v_dofs] + X_ref
  for (int a = 0; a < p.n_dof; ++a) {
    const int64_t node = p.dofs[(int64_t)f * p.n_dof + a];
#pragma unroll
    for (int i = 0; i < DIM; ++i) x_e[i * p.n_dof + a] = p.u[node * DIM + i] + p.x_ref[node * DIM + i];
  }
}

// pass 1: one thread per (face, quadrature point)
template<int DIM>
__global__ void contact_gap_area_kernel(ContactArgs p, int gap_norm_only) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)p.n_faces * p.n_q) return;
  const int f = idx / p.n_q;
  const int64_t pt = idx;
  // x_e = u[v_dofs] + X_ref (integrator_utils.cpp:80-89) is used node by node as it arrives: the point's position and the
  // two tangents J = x_e^T dN_dxi (integrator_utils.cpp:101-105) in one pass, no private array
  const double* N = p.N + pt * p.n_dof;
  const double* dN = p.dN + pt * p.n_dof * (DIM - 1);
  double xq[DIM], t[(DIM - 1) * DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) xq[i] = 0.0;
#pragma unroll
  for (int k = 0; k < (DIM - 1) * DIM; ++k) t[k] = 0.0;
  for (int a = 0; a < p.n_dof; ++a) {
    const int64_t node = p.dofs[(int64_t)f * p.n_dof + a];
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double x = p.u[node * DIM + i] + p.x_ref[node * DIM + i];
      xq[i] = __builtin_fma(x, N[a], xq[i]);
#pragma unroll
      for (int k = 0; k < DIM - 1; ++k) t[k * DIM + i] = __builtin_fma(x, dN[k * p.n_dof + a], t[k * DIM + i]);
    }
  }
  double true_g, distance;
  nearest_body<DIM>(p, xq, true_g, distance);
  if (gap_norm_only) {
    // GapNorm (mortar_contact.cpp:423-467)
    p.pt_scal[pt] = true_g < 0.0 ? true_g * true_g : 0.0;
    return;
  }
  double g = true_g < 0. ? true_g : 0.;
  // |J| (DenseMatrix::Weight) from the non-normalised normal (integrator_utils.hpp:216-251)
  double detJ;
  if constexpr (DIM == 2) {
    detJ = sqrt(t[1] * t[1] + t[0] * t[0]);
  } else {
    const double m0 = t[1] * t[5] - t[2] * t[4], m1 = t[2] * t[3] - t[0] * t[5], m2 = t[0] * t[4] - t[1] * t[3];
    detJ = sqrt(m0 * m0 + m1 * m1 + m2 * m2);
  }
  const double fac = p.weight[pt] * detJ;
  p.pt_scal[pt] = fac;
  const double ratio = fabs(true_g) / distance;
  if (acos(ratio < 1. ? ratio : 1.) > 1.e-5) g = 0.0;  // angle tolerance, mortar_contact.cpp:172-181
  const double fac_g = fac * g;
  for (int a = 0; a < p.n_dof; ++a) {
    p.pt_area[pt * p.n_dof + a] = fac * N[a];
    p.pt_gap[pt * p.n_dof + a] = fac_g * N[a];
  }
}

// nodal area / gap (mortar_contact.cpp:182-190, the sums the reference forms under a mutex): one WAVE per marked node;
// entry k = (incidence, quadrature point) of the node goes to lane k % 64 in order, then a fixed-shape tree over the
// lanes -- the same bits every run
__global__ __launch_bounds__(256) void contact_nodal_kernel(int n_marked, int n_q, int n_dof, const int32_t* __restrict__ adj_ptr,
                                                            const int32_t* __restrict__ adj, const double* __restrict__ pt_area,
                                                            const double* __restrict__ pt_gap, double* __restrict__ area,
                                                            double* __restrict__ gap) {
  const int lane = threadIdx.x & 63;
  const int l = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (l >= n_marked) return;
  const int t0 = adj_ptr[l], n = (adj_ptr[l + 1] - t0) * n_q;
  double sa = 0.0, sg = 0.0;
  for (int k = lane; k < n; k += 64) {
    const int ea = adj[t0 + k / n_q];
    const int64_t pt = (int64_t)(ea >> 6) * n_q + k % n_q;
    sa += pt_area[pt * n_dof + (ea & 63)];
    sg += pt_gap[pt * n_dof + (ea & 63)];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    sa += __shfl_down(sa, off, 64);
    sg += __shfl_down(sg, off, 64);
  }
  if (lane == 0) {
    area[l] = sa;
    gap[l] = sg;
  }
}

// out[k] = sum_i in[i * stride + k] (k < n_out), optionally over the rows with flag[i] != 0: ONE workgroup, every thread a
// fixed subset of the rows, then a fixed-shape tree -- the same bits every run
__global__ __launch_bounds__(1024) void contact_sum_kernel(int64_t n, int stride, int n_out, const double* __restrict__ in,
                                                           const unsigned char* __restrict__ flag, double* __restrict__ out) {
  __shared__ double part[1024];
  for (int k = 0; k < n_out; ++k) {
    double s = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024)
      if (!flag || flag[i]) s += in[i * stride + k];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
      if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) out[k] = part[0];
    __syncthreads();
  }
}

__global__ void contact_pressure_kernel(int n, const double* area, const double* gap, double penalty, double* pressure) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) pressure[i] = gap[i] / area[i] * penalty;  // mortar_contact.cpp:253-257
}

MH_DEV double contact_lane_read(double v, int l) {   // the value lane l holds, in every lane (l wave-uniform)
  const unsigned long long u = __double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)u, l), hi = __builtin_amdgcn_readlane((unsigned)(u >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// element residual at given x_e (used by the FD mode too)
template<int DIM>
MH_DEV void face_residual(const ContactArgs& p, int f, const double* x_e, const double* p_e, double* R_e,
                          double* force, double* pint) {
  for (int k = 0; k < p.n_dof * DIM; ++k) R_e[k] = 0.0;
  for (int q = 0; q < p.n_q; ++q) {
    const int64_t pt = (int64_t)f * p.n_q + q;
    const double* N = p.N + pt * p.n_dof;
    double pq = 0;
    for (int a = 0; a < p.n_dof; ++a) pq += N[a] * p_e[a];
    double m[DIM], t[(DIM - 1) * DIM];
    const double detJ = surface_normal<DIM>(p.n_dof, x_e, p.dN + pt * p.n_dof * (DIM - 1), m, t);
    const double fac = p.weight[pt] * detJ * pq;
#pragma unroll
    for (int i = 0; i < DIM; ++i) {
      const double aw = (m[i] / detJ) * (-fac);
      for (int a = 0; a < p.n_dof; ++a) R_e[i * p.n_dof + a] += aw * N[a];
      if (force) force[i] += fac * (m[i] / detJ);
    }
    if (pint) *pint += fac;
  }
}

// reference-FD mode of pass 2: one thread per face, the face tangent block by forward differences of the face residual
// (stored densely: contact_gather_kernel)
template<int DIM>
__global__ void contact_residual_kernel(ContactArgs p, int with_grad) {
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= p.n_faces) return;
  double p_e[kMaxFaceDof];
  bool any = false;
  for (int a = 0; a < p.n_dof; ++a) {
    p_e[a] = p.pressure[p.local[(int64_t)f * p.n_dof + a]];
    any = any || (p_e[a] != 0.0);
  }
  if (!any) return;  // IsPressureZero (integrator_utils.cpp:112-119)
  double x_e[DIM * kMaxFaceDof], R_e[DIM * kMaxFaceDof];
  gather_x<DIM>(p, f, x_e);
  face_residual<DIM>(p, f, x_e, p_e, R_e, nullptr, nullptr);
  const int NT = p.n_dof * DIM;
  // (the face residual, pressure integral and force are stored by contact_residual_wave_kernel in every mode, so that
  // the history scalars do not depend on the tangent mode to the last bit; this kernel adds the difference tangent)
  if (!with_grad) return;
  double* Kf = p.face_k + (int64_t)f * NT * NT;     // row (a, i): [j][b]
  // mortar_contact.cpp:263-295: forward FD on the current POSITION, pressure frozen
  double fwd[DIM * kMaxFaceDof];
  for (int c = 0; c < NT; ++c) {
    const double orig = x_e[c];
    const double step = (orig != 0.0) ? fabs(orig) * 1.0e-8 : 1.0e-10;
    const double step_inv = 1. / step;
    x_e[c] = orig + step;
    face_residual<DIM>(p, f, x_e, p_e, fwd, nullptr, nullptr);
    x_e[c] = orig;
    const int b = c % p.n_dof, j = c / p.n_dof;
    for (int a = 0; a < p.n_dof; ++a)
#pragma unroll
      for (int i = 0; i < DIM; ++i)
        Kf[(a * DIM + i) * NT + j * p.n_dof + b] = (fwd[i * p.n_dof + a] - R_e[i * p.n_dof + a]) * step_inv;
  }
}

// pass 2 without the reference-FD tangent, one WAVE per face: lane = quadrature point for pressure, normal and weights,
// then lane = (i, a) for the residual entries -- summed over the points in the order of the one-thread form above, so
// the two give the same bits (that form left 9 216 faces to 144 waves: 0.19 ms of latency at configuration 4)
template<int DIM>
__global__ __launch_bounds__(256) void contact_residual_wave_kernel(ContactArgs p) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= p.n_faces) return;
  const int n_dof = p.n_dof, n_q = p.n_q, NT = n_dof * DIM;
  double pc = 0.0, xc[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) xc[i] = 0.0;
  if (lane < n_dof) {
    pc = p.pressure[p.local[(int64_t)f * n_dof + lane]];
    const int64_t node = p.dofs[(int64_t)f * n_dof + lane];
#pragma unroll
    for (int i = 0; i < DIM; ++i) xc[i] = p.u[node * DIM + i] + p.x_ref[node * DIM + i];
  }
  const bool any = __ballot(pc != 0.0) != 0;
  if (lane == 0) p.face_active[f] = any ? 1 : 0;
  if (!any) return;  // IsPressureZero (integrator_utils.cpp:112-119)
  // lane q: fac = w detJ p(q), n = m / detJ
  double fac = 0.0, nq[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) nq[i] = 0.0;
  {
    const int q = lane < n_q ? lane : 0;
    const int64_t pt = (int64_t)f * n_q + q;
    const double* N = p.N + pt * n_dof;
    const double* dN = p.dN + pt * n_dof * (DIM - 1);
    double pq = 0.0, t[(DIM - 1) * DIM];
#pragma unroll
    for (int k = 0; k < (DIM - 1) * DIM; ++k) t[k] = 0.0;
    for (int c = 0; c < n_dof; ++c) {
      pq = __builtin_fma(N[c], contact_lane_read(pc, c), pq);
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double x = contact_lane_read(xc[i], c);
#pragma unroll
        for (int k = 0; k < DIM - 1; ++k) t[k * DIM + i] = __builtin_fma(x, dN[k * n_dof + c], t[k * DIM + i]);
      }
    }
    double m[DIM], detJ;
    if constexpr (DIM == 2) {
      m[0] = t[1];
      m[1] = -t[0];
      detJ = sqrt(m[0] * m[0] + m[1] * m[1]);
    } else {
      m[0] = t[1] * t[5] - t[2] * t[4];
      m[1] = t[2] * t[3] - t[0] * t[5];
      m[2] = t[0] * t[4] - t[1] * t[3];
      detJ = sqrt(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]);
    }
    fac = p.weight[pt] * detJ * pq;
#pragma unroll
    for (int i = 0; i < DIM; ++i) nq[i] = m[i] / detJ;
  }
  // lane k = i n_dof + a: R(a, i) = sum_q (n_i (-fac))_q N_q[a]
  {
    const int k = lane < NT ? lane : 0, i = k / n_dof, a = k % n_dof;
    double R = 0.0;
    for (int q = 0; q < n_q; ++q) {
      const double mf = -contact_lane_read(fac, q);
      double aw = contact_lane_read(nq[0], q) * mf;
#pragma unroll
      for (int ii = 1; ii < DIM; ++ii) {
        const double v = contact_lane_read(nq[ii], q) * mf;
        aw = i == ii ? v : aw;
      }
      R += aw * p.N[((int64_t)f * n_q + q) * n_dof + a];
    }
    if (lane < NT) p.face_r[(int64_t)f * NT + lane] = R;      // [i][a]
  }
  // pressure integral and force of the face: lane 0, in the order of the points
  {
    double pint = 0.0, force[DIM];
#pragma unroll
    for (int i = 0; i < DIM; ++i) force[i] = 0.0;
    for (int q = 0; q < n_q; ++q) {
      const double fq = contact_lane_read(fac, q);
#pragma unroll
      for (int i = 0; i < DIM; ++i) force[i] += fq * contact_lane_read(nq[i], q);
      pint += fq;
    }
    if (lane == 0) {
      p.face_scal[(int64_t)f * (1 + DIM)] = pint;
#pragma unroll
      for (int i = 0; i < DIM; ++i) p.face_scal[(int64_t)f * (1 + DIM) + 1 + i] = force[i];
    }
  }
}

// analytic frozen-pressure tangent, one WAVE per face (R(a,i) = -w p N_a m_i, d m / d x_bj with p frozen).  Stage 1, lane =
// quadrature point: the pressure and the two surface tangents at the point, from the face's nodal values handed round by
// lane reads (lane c holds node c).  Stage 2, lane = node pair (a, b): its DIM x DIM block accumulated over the points in
// registers, the point's seven numbers read from the lane that computed them.  (Before: every pair lane recomputed
// pressure and tangents of every point from private arrays -- 0.44 ms for the 9 216 faces of configuration 4.)
template<int DIM>
__global__ __launch_bounds__(256) void contact_tangent_kernel(ContactArgs p) {
  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (f >= p.n_faces) return;
  const int n_dof = p.n_dof, n_q = p.n_q;   // (<= 16 nodes; <= 64 points: checked at create)
  // lane c < n_dof: node c of the face
  double pc = 0.0, xc[DIM];
#pragma unroll
  for (int i = 0; i < DIM; ++i) xc[i] = 0.0;
  if (lane < n_dof) {
    pc = p.pressure[p.local[(int64_t)f * n_dof + lane]];
    const int64_t node = p.dofs[(int64_t)f * n_dof + lane];
#pragma unroll
    for (int i = 0; i < DIM; ++i) xc[i] = p.u[node * DIM + i] + p.x_ref[node * DIM + i];   // integrator_utils.cpp:80-89
  }
  if (__ballot(pc != 0.0) == 0) return;  // IsPressureZero (integrator_utils.cpp:112-119)
  // stage 1: lane q < n_q
  double wq = 0.0, tq[(DIM - 1) * DIM];
#pragma unroll
  for (int k = 0; k < (DIM - 1) * DIM; ++k) tq[k] = 0.0;
  {
    const int q = lane < n_q ? lane : 0;
    const int64_t pt = (int64_t)f * n_q + q;
    const double* N = p.N + pt * n_dof;
    const double* dN = p.dN + pt * n_dof * (DIM - 1);
    double pq = 0.0;
    for (int c = 0; c < n_dof; ++c) {
      pq = __builtin_fma(N[c], contact_lane_read(pc, c), pq);
#pragma unroll
      for (int i = 0; i < DIM; ++i) {
        const double x = contact_lane_read(xc[i], c);
#pragma unroll
        for (int k = 0; k < DIM - 1; ++k) tq[k * DIM + i] = __builtin_fma(x, dN[k * n_dof + c], tq[k * DIM + i]);
      }
    }
    wq = -p.weight[pt] * pq;
  }
  const int n_pairs = n_dof * n_dof, NT = n_dof * DIM;
  double* Kf = p.face_k + (int64_t)f * NT * NT;
  for (int pair0 = 0; pair0 < n_pairs; pair0 += 64) {
    const int pair = pair0 + lane;
    const bool on = pair < n_pairs;
    const int a = on ? pair / n_dof : 0, b = on ? pair % n_dof : 0;
    double acc[DIM * DIM];
#pragma unroll
    for (int k = 0; k < DIM * DIM; ++k) acc[k] = 0.0;
    for (int q = 0; q < n_q; ++q) {
      const int64_t pt = (int64_t)f * n_q + q;
      const double wpn = contact_lane_read(wq, q) * p.N[pt * n_dof + a];
      const double* dN = p.dN + pt * n_dof * (DIM - 1);
      double t[(DIM - 1) * DIM];
#pragma unroll
      for (int k = 0; k < (DIM - 1) * DIM; ++k) t[k] = contact_lane_read(tq[k], q);
#pragma unroll
      for (int j = 0; j < DIM; ++j) {
        double dm[DIM];
        if constexpr (DIM == 2) {
          dm[0] = (j == 1) ? dN[b] : 0.0;
          dm[1] = (j == 0) ? -dN[b] : 0.0;
        } else {
          double e[3] = {0, 0, 0};
          e[j] = 1.0;
          const double* t1 = t;
          const double* t2 = t + 3;
          const double d1 = dN[b], d2 = dN[n_dof + b];
          dm[0] = d1 * (e[1] * t2[2] - e[2] * t2[1]) + d2 * (t1[1] * e[2] - t1[2] * e[1]);
          dm[1] = d1 * (e[2] * t2[0] - e[0] * t2[2]) + d2 * (t1[2] * e[0] - t1[0] * e[2]);
          dm[2] = d1 * (e[0] * t2[1] - e[1] * t2[0]) + d2 * (t1[0] * e[1] - t1[1] * e[0]);
        }
#pragma unroll
        for (int i = 0; i < DIM; ++i) acc[i * DIM + j] += wpn * dm[i];
      }
    }
    if (on) {
#pragma unroll
      for (int i = 0; i < DIM; ++i)
#pragma unroll
        for (int j = 0; j < DIM; ++j) Kf[(a * DIM + i) * NT + j * n_dof + b] = acc[i * DIM + j];
    }
  }
}

// The sums the reference forms under a mutex (mortar_contact.cpp:338-341,400-408), without atomics: one wave per CSR row
// (marked node, i); the wave walks the node's (face, local node) incidences in face order, adds row (a, i) of every ACTIVE
// face block into an LDS image of the CSR row through the pair positions (lane = column node b: distinct positions within
// an instruction), then adds the image to the caller's values in one coalesced pass; the residual entry likewise.
// Rows none of whose faces is active are left untouched.
constexpr int CG_WAVES = 4;
constexpr int CG_MAX_ROW = 1056;   // (2 p + 1)^3 neighbours x 3 at p = 3 is 1029
template<int DIM, int WITH_K>
__global__ __launch_bounds__(64 * CG_WAVES) void contact_gather_kernel(ContactArgs p, int n_marked, const int32_t* __restrict__ marked,
                                                                       const int32_t* __restrict__ adj_ptr, const int32_t* __restrict__ adj) {
  __shared__ double img_all[WITH_K ? CG_WAVES : 1][CG_MAX_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t R = (int64_t)blockIdx.x * CG_WAVES + wave;
  if (R >= (int64_t)n_marked * DIM) return;
  const int l = (int)(R / DIM), i = (int)(R % DIM);
  const int a_beg = adj_ptr[l], a_end = adj_ptr[l + 1];
  bool any = false;
  for (int t = a_beg; t < a_end; ++t) any = any || p.face_active[adj[t] >> 6];
  if (!any) return;
  const int64_t row = (int64_t)marked[l] * DIM + i;
  const int NT = p.n_dof * DIM;
  if constexpr (WITH_K) {
    double* img = img_all[wave];
    const int64_t beg = p.rowptr[row];
    const int len = (int)(p.rowptr[row + 1] - beg);
    for (int k = lane; k < len; k += 64) img[k] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int t = a_beg; t < a_end; ++t) {
      const int64_t f = adj[t] >> 6;
      const int a = adj[t] & 63;
      if (!p.face_active[f]) continue;
      const double* Kr = p.face_k + (f * NT + (a * DIM + i)) * (int64_t)NT;   // row (a, i): [j][b]
      for (int b = lane; b < p.n_dof; b += 64) {
        const int32_t off = p.pair_pos[(f * p.n_dof + a) * p.n_dof + b];
#pragma unroll
        for (int j = 0; j < DIM; ++j) img[off + j] += Kr[j * p.n_dof + b];
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < len; k += 64) p.A[beg + k] += p.grad_factor * img[k];
  }
  if (lane == 0) {
    double rs = 0.0;
    for (int t = a_beg; t < a_end; ++t) {
      const int64_t f = adj[t] >> 6;
      if (p.face_active[f]) rs += p.face_r[f * NT + i * p.n_dof + (adj[t] & 63)];
    }
    p.r[row] += rs;
  }
}

__global__ void contact_pair_pos_kernel(int n_faces, int n_dof, int dim, const int32_t* dofs, const int64_t* rowptr,
                                        const int32_t* col, int32_t* pair_pos, int* status) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (int64_t)n_faces * n_dof * n_dof) return;
  const int b = idx % n_dof, a = (idx / n_dof) % n_dof;
  const int64_t f = idx / ((int64_t)n_dof * n_dof);
  const int64_t row = (int64_t)dofs[f * n_dof + a] * dim;
  const int32_t target = dofs[f * n_dof + b] * dim;
  int64_t lo = rowptr[row], hi = rowptr[row + 1];
  const int64_t base = lo;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (col[mid] < target) lo = mid + 1; else hi = mid;
  }
  if (lo >= rowptr[row + 1] || col[lo] != target) {
    atomicOr(status, 4);
    pair_pos[idx] = 0;
    return;
  }
  pair_pos[idx] = (int32_t)(lo - base);
}

}  // namespace mimi_hip

using namespace mimi_hip;

struct mimi_hip_contact_s {
  int device = 0, dim = 0, n_faces = 0, n_dof = 0, n_q = 0, n_marked = 0;
  int64_t n_nodes = 0, n_vdofs = 0, nnz = 0;
  hipStream_t own_stream = nullptr, stream = nullptr;
  int mode = MIMI_HIP_TANGENT_ANALYTIC;
  int body_kind = 0;
  double body[8] = {0};
  double penalty = 1e4;
  DeviceBuffer<int32_t> dofs, local, pair_pos;
  DeviceBuffer<double> N, dN, weight, x_ref, area, gap, pressure, scalars;
  DeviceBuffer<int64_t> rowptr_own;
  const int64_t* rowptr = nullptr;
  DeviceBuffer<double> stage_u, stage_r, stage_A;
  DeviceBuffer<int> status;
  // dense stores of the atomic-free assembly and the node -> (face, local node) incidences of the marked nodes
  DeviceBuffer<double> pt_area, pt_gap, pt_scal, face_r, face_k, face_scal;
  DeviceBuffer<unsigned char> face_active;
  DeviceBuffer<int32_t> madj_ptr, madj, marked_dev;
  std::vector<int32_t> marked_nodes;   // sorted global node ids of the marked dofs (local index -> node)
  SplineBodyDev spline{};
  DeviceBuffer<double> sb_knots[2], sb_ctrl, sb_sample_xi, sb_sample_x;
  double last[6] = {0};
  ~mimi_hip_contact_s() {
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};

// NearestDistanceToSplines::AddSpline / PlantKdTree (nearest_distance.hpp:223-255): (re)upload the rigid spline and the
// sampled initial guesses.  Called at create time and whenever the caller moved the body and re-planted its tree.
static void upload_spline_body(mimi_hip_contact_s* h, const mimi_hip_spline_body* sp, int dim) {
  // NearestDistanceToSplines::AddSpline / PlantKdTree (nearest_distance.hpp:223-255)
  if (!sp) fail("body_kind spline without a spline description");
  if (sp->para_dim + 1 != dim) fail("boundary para_dim should be one smaller than dim.");   // :121-124
  SplineBodyDev host{};
  host.para_dim = sp->para_dim;
  host.dim = dim;
  size_t n_ctrl = 1;
  for (int k = 0; k < 2; ++k) {
    host.p[k] = k < sp->para_dim ? sp->degree[k] : 0;
    host.n_knots[k] = k < sp->para_dim ? sp->n_knots[k] : 2;
    host.n_ctrl[k] = host.n_knots[k] - host.p[k] - 1;
    if (host.p[k] < 0 || host.p[k] > kMaxBodyDegree) fail("spline body degree %d unsupported (<= %d)", host.p[k], kMaxBodyDegree);
    if (host.n_ctrl[k] < host.p[k] + 1) fail("spline body knot vector too short");
    n_ctrl *= host.n_ctrl[k];
  }
  const int hd = dim + 1;
  std::vector<double> ctrl_h(n_ctrl * hd);
  for (size_t a = 0; a < n_ctrl; ++a) {
    const double w = sp->weights ? sp->weights[a] : 1.0;
    if (!(w > 0.0)) fail("spline body weights must be positive");
    for (int i = 0; i < dim; ++i) ctrl_h[a * hd + i] = w * sp->control_points[a * dim + i];
    ctrl_h[a * hd + dim] = w;
  }
  static const double unit_knots[2] = {0.0, 1.0};
  host.knots[0] = sp->knots[0];
  host.knots[1] = sp->para_dim == 2 ? sp->knots[1] : unit_knots;
  host.ctrl_h = ctrl_h.data();
  int res = sp->kdtree_resolution > 1 ? sp->kdtree_resolution : 100;
  if (sp->para_dim == 2 && res > 1000) res = 1000;   // res^2 samples: the reference's kd-tree takes what it is given
  if (res > 1000000) res = 1000000;
  const int n_s = sp->para_dim == 2 ? res * res : res;
  std::vector<double> sxi((size_t)n_s * sp->para_dim), sx((size_t)n_s * dim);
  for (int s_ = 0; s_ < n_s; ++s_) {
    const int idx[2] = {s_ % res, s_ / res};
    double xi[2] = {0, 0}, S[3], S1[6], S2[12];
    for (int k = 0; k < sp->para_dim; ++k) {
      const double lo = host.knots[k][host.p[k]], hi = host.knots[k][host.n_knots[k] - host.p[k] - 1];
      xi[k] = lo + (hi - lo) * idx[k] / (res - 1);
      sxi[(size_t)s_ * sp->para_dim + k] = xi[k];
    }
    sb_evaluate(host, xi, S, S1, S2);
    for (int i = 0; i < dim; ++i) sx[(size_t)s_ * dim + i] = S[i];
  }
  h->spline = host;
  for (int k = 0; k < sp->para_dim; ++k) {
    h->sb_knots[k].assign(sp->knots[k], (size_t)sp->n_knots[k], h->stream);
    h->spline.knots[k] = h->sb_knots[k].ptr;
  }
  if (sp->para_dim == 1) {
    h->sb_knots[1].assign(unit_knots, 2, h->stream);
    h->spline.knots[1] = h->sb_knots[1].ptr;
  }
  h->sb_ctrl.assign(ctrl_h.data(), ctrl_h.size(), h->stream);
  h->sb_sample_xi.assign(sxi.data(), sxi.size(), h->stream);
  h->sb_sample_x.assign(sx.data(), sx.size(), h->stream);
  h->spline.ctrl_h = h->sb_ctrl.ptr;
  h->spline.sample_xi = h->sb_sample_xi.ptr;
  h->spline.sample_x = h->sb_sample_x.ptr;
  h->spline.n_samples = n_s;
  h->spline.max_iterations = sp->max_iterations;
}

template<typename F>
static int guarded_c(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  }
}

static ContactArgs contact_args(mimi_hip_contact_s* h, const double* u, double* r, double* A, double gf) {
  ContactArgs a{};
  a.dim = h->dim;
  a.n_faces = h->n_faces;
  a.n_dof = h->n_dof;
  a.n_q = h->n_q;
  a.dofs = h->dofs.ptr;
  a.local = h->local.ptr;
  a.N = h->N.ptr;
  a.dN = h->dN.ptr;
  a.weight = h->weight.ptr;
  a.x_ref = h->x_ref.ptr;
  a.rowptr = h->rowptr;
  a.pair_pos = h->pair_pos.ptr;
  a.body_kind = h->body_kind;
  a.spline = h->spline;
  for (int i = 0; i < 8; ++i) a.body[i] = h->body[i];
  a.penalty = h->penalty;
  a.u = u;
  a.r = r;
  a.A = A;
  a.grad_factor = gf;
  a.area = h->area.ptr;
  a.gap = h->gap.ptr;
  a.pressure = h->pressure.ptr;
  a.scalars = h->scalars.ptr;
  a.mode = h->mode;
  a.pt_area = h->pt_area.ptr;
  a.pt_gap = h->pt_gap.ptr;
  a.pt_scal = h->pt_scal.ptr;
  a.face_r = h->face_r.ptr;
  a.face_k = h->face_k.ptr;
  a.face_scal = h->face_scal.ptr;
  a.face_active = h->face_active.ptr;
  return a;
}

// pass 1 (mortar_contact.cpp:148-193): nodal area / gap of this handle's faces; pressure not formed yet
static void contact_pass1(mimi_hip_contact_s* h, const double* u_dev) {
  ContactArgs a = contact_args(h, u_dev, nullptr, nullptr, 0.0);
  // InitializeGapAreaPressure + last_* reset (mortar_contact.cpp:135-146,302-306)
  MH_HIP(hipMemsetAsync(h->scalars.ptr, 0, 6 * sizeof(double), h->stream));
  const int threads = 128;
  const int64_t npts = (int64_t)h->n_faces * h->n_q;
  const unsigned b1 = (unsigned)((npts + threads - 1) / threads);
  if (h->dim == 2) hipLaunchKernelGGL(contact_gap_area_kernel<2>, dim3(b1), dim3(threads), 0, h->stream, a, 0);
  else hipLaunchKernelGGL(contact_gap_area_kernel<3>, dim3(b1), dim3(threads), 0, h->stream, a, 0);
  hipLaunchKernelGGL(contact_nodal_kernel, dim3((unsigned)((h->n_marked + 3) / 4)), dim3(256), 0, h->stream,
                     h->n_marked, h->n_q, h->n_dof, h->madj_ptr.ptr, h->madj.ptr, h->pt_area.ptr, h->pt_gap.ptr, h->area.ptr, h->gap.ptr);
  hipLaunchKernelGGL(contact_sum_kernel, dim3(1), dim3(1024), 0, h->stream, npts, 1, 1, h->pt_scal.ptr,
                     (const unsigned char*)nullptr, h->scalars.ptr);
  MH_HIP(hipGetLastError());
}

// pressure from the nodal area / gap (mortar_contact.cpp:195-261), then pass 2: face residual vectors (+ tangent blocks)
// and their row gather
static void contact_pass2(mimi_hip_contact_s* h, const double* u_dev, double* r_dev, double* A_dev, double gf, bool with_grad) {
  ContactArgs a = contact_args(h, u_dev, r_dev, A_dev, gf);
  const int threads = 128;
  const unsigned b2 = (unsigned)((h->n_marked + threads - 1) / threads);
  const unsigned b3 = (unsigned)((h->n_faces + 63) / 64);
  hipLaunchKernelGGL(contact_pressure_kernel, dim3(b2), dim3(threads), 0, h->stream, h->n_marked, h->area.ptr, h->gap.ptr, h->penalty, h->pressure.ptr);
  if (with_grad && !h->face_k.ptr) {
    const size_t nt = (size_t)h->n_dof * h->dim;
    h->face_k.resize((size_t)h->n_faces * nt * nt);
    a.face_k = h->face_k.ptr;
  }
  // analytic tangent: the residual by the face-per-thread kernel, the tangent by one wave per face
  const bool wave_tangent = with_grad && h->mode != MIMI_HIP_TANGENT_REFERENCE_FD;
  const int grad_flag = (with_grad && !wave_tangent) ? 1 : 0;
  const unsigned b3w = (unsigned)((h->n_faces + 3) / 4);
  if (h->dim == 2) hipLaunchKernelGGL(contact_residual_wave_kernel<2>, dim3(b3w), dim3(256), 0, h->stream, a);
  else hipLaunchKernelGGL(contact_residual_wave_kernel<3>, dim3(b3w), dim3(256), 0, h->stream, a);
  if (grad_flag) {   // reference-FD tangent
    if (h->dim == 2) hipLaunchKernelGGL(contact_residual_kernel<2>, dim3(b3), dim3(64), 0, h->stream, a, grad_flag);
    else hipLaunchKernelGGL(contact_residual_kernel<3>, dim3(b3), dim3(64), 0, h->stream, a, grad_flag);
  }
  if (wave_tangent) {
    const unsigned b4 = (unsigned)((h->n_faces + 3) / 4);
    if (h->dim == 2) hipLaunchKernelGGL(contact_tangent_kernel<2>, dim3(b4), dim3(256), 0, h->stream, a);
    else hipLaunchKernelGGL(contact_tangent_kernel<3>, dim3(b4), dim3(256), 0, h->stream, a);
  }
  // pressure integral and force of the active faces (last_pressure_ / last_force_), in a fixed order
  hipLaunchKernelGGL(contact_sum_kernel, dim3(1), dim3(1024), 0, h->stream, (int64_t)h->n_faces, 1 + h->dim, 1 + h->dim,
                     h->face_scal.ptr, h->face_active.ptr, h->scalars.ptr + 1);
  const unsigned b5 = (unsigned)(((int64_t)h->n_marked * h->dim + CG_WAVES - 1) / CG_WAVES);
  if (h->dim == 2) {
    if (with_grad) hipLaunchKernelGGL((contact_gather_kernel<2, 1>), dim3(b5), dim3(64 * CG_WAVES), 0, h->stream, a, h->n_marked, h->marked_dev.ptr, h->madj_ptr.ptr, h->madj.ptr);
    else hipLaunchKernelGGL((contact_gather_kernel<2, 0>), dim3(b5), dim3(64 * CG_WAVES), 0, h->stream, a, h->n_marked, h->marked_dev.ptr, h->madj_ptr.ptr, h->madj.ptr);
  } else {
    if (with_grad) hipLaunchKernelGGL((contact_gather_kernel<3, 1>), dim3(b5), dim3(64 * CG_WAVES), 0, h->stream, a, h->n_marked, h->marked_dev.ptr, h->madj_ptr.ptr, h->madj.ptr);
    else hipLaunchKernelGGL((contact_gather_kernel<3, 0>), dim3(b5), dim3(64 * CG_WAVES), 0, h->stream, a, h->n_marked, h->marked_dev.ptr, h->madj_ptr.ptr, h->madj.ptr);
  }
  MH_HIP(hipGetLastError());
}

// which: 1 = pass 1 only (u), 2 = pass 2 only (u, r, A), 3 = both (the single-process call)
static void run_contact(mimi_hip_contact_s* h, const double* u, double* r, double* A, double gf, bool with_grad, int which = 3) {
  MH_HIP(hipSetDevice(h->device));
  if (!u || ((which & 2) && (!r || (with_grad && !A)))) fail("null vector argument");
  Mirror<double> mu = Mirror<double>::in(u, h->n_vdofs, h->stage_u, h->stream);
  Mirror<double> mr, mA;
  if (which & 2) mr = Mirror<double>::inout(r, h->n_vdofs, h->stage_r, h->stream);
  if ((which & 2) && with_grad) mA = Mirror<double>::inout(A, h->nnz, h->stage_A, h->stream);
  if (which & 1) contact_pass1(h, mu.dev);
  if (which & 2) {
    contact_pass2(h, mu.dev, mr.dev, mA.dev, gf, with_grad);
    mr.finish(h->stream);
    if (with_grad) mA.finish(h->stream);
  }
  if (mu.host || mr.host || mA.host) MH_HIP(hipStreamSynchronize(h->stream));
}

extern "C" {

int mimi_hip_contact_create(const mimi_hip_contact_tables* t, int device, mimi_hip_contact_t* out) {
  return guarded_c([&] {
    if (!t || !out) fail("null argument");
    if (t->dim != 2 && t->dim != 3) fail("Unsupported Dim: %d", t->dim);
    if (t->n_dof < 1 || t->n_dof > kMaxFaceDof) fail("face n_dof %d out of range [1,%d]", t->n_dof, kMaxFaceDof);
    if (t->n_faces < 1) fail("no marked boundary faces");
    if (t->n_quad < 1 || t->n_quad > 64) fail("face quadrature points %d out of range [1,64]", t->n_quad);   // (a lane per point)
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || count == 0)
      fail("libmimi_hip: no HIP device visible -- this library has no CPU fallback");
    auto h = std::make_unique<mimi_hip_contact_s>();
    h->device = device;
    MH_HIP(hipSetDevice(device));
    MH_HIP(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
    h->stream = h->own_stream;
    h->dim = t->dim;
    h->n_faces = t->n_faces;
    h->n_dof = t->n_dof;
    h->n_q = t->n_quad;
    h->n_nodes = t->n_nodes;
    h->n_vdofs = t->n_nodes * t->dim;
    h->body_kind = t->body_kind;
    for (int i = 0; i < 8; ++i) h->body[i] = t->body[i];
    h->penalty = t->penalty;
    if (t->body_kind == MIMI_HIP_BODY_SPLINE) {
      upload_spline_body(h.get(), t->spline, t->dim);
    } else if (t->body_kind != MIMI_HIP_BODY_SPHERE && t->body_kind != MIMI_HIP_BODY_PLANE) {
      fail("unknown rigid body kind %d", t->body_kind);
    }
    const size_t nfd = (size_t)t->n_faces * t->n_dof;
    // host copy of the connectivity for the dense local numbering of marked dofs
    // (mortar_contact.cpp:41-76: sorted unique marked dofs -> 0..n_marked-1)
    std::vector<int32_t> dofs(nfd);
    if (is_device_pointer(t->dofs))
      MH_HIP(hipMemcpy(dofs.data(), t->dofs, nfd * sizeof(int32_t), hipMemcpyDeviceToHost));
    else
      std::copy(t->dofs, t->dofs + nfd, dofs.begin());
    std::vector<int32_t> marked(dofs);
    std::sort(marked.begin(), marked.end());
    marked.erase(std::unique(marked.begin(), marked.end()), marked.end());
    h->n_marked = (int)marked.size();
    h->marked_nodes = marked;
    std::vector<int32_t> local(nfd);
    for (size_t k = 0; k < nfd; ++k)
      local[k] = (int32_t)(std::lower_bound(marked.begin(), marked.end(), dofs[k]) - marked.begin());
    h->dofs.assign(dofs.data(), nfd, h->stream);
    h->local.assign(local.data(), nfd, h->stream);
    const size_t npts = (size_t)t->n_faces * t->n_quad;
    h->N.assign(t->N, npts * t->n_dof, h->stream);
    h->dN.assign(t->dN_dxi, npts * t->n_dof * (t->dim - 1), h->stream);
    h->weight.assign(t->weight, npts, h->stream);
    h->x_ref.assign(t->x_ref, (size_t)t->n_nodes * t->dim, h->stream);
    h->area.resize(h->n_marked);
    h->gap.resize(h->n_marked);
    h->pressure.resize(h->n_marked);
    h->scalars.resize(6);
    MH_HIP(hipMemsetAsync(h->pressure.ptr, 0, h->n_marked * sizeof(double), h->stream));
    {
      // marked node -> its (face, local node) incidences, faces in ascending order: the summation order of every gather
      if (t->n_faces >= (1 << 25)) fail("too many boundary faces for the incidence encoding");
      std::vector<int32_t> ptr(marked.size() + 1, 0), adj(nfd);
      for (size_t k = 0; k < nfd; ++k) ++ptr[local[k] + 1];
      for (size_t l = 0; l < marked.size(); ++l) ptr[l + 1] += ptr[l];
      std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
      for (size_t k = 0; k < nfd; ++k) adj[fill[local[k]]++] = (int32_t)(((k / t->n_dof) << 6) | (k % t->n_dof));
      h->madj_ptr.assign(ptr.data(), ptr.size(), h->stream);
      h->madj.assign(adj.data(), adj.size(), h->stream);
      h->marked_dev.assign(marked.data(), marked.size(), h->stream);
      const size_t npts_ = (size_t)t->n_faces * t->n_quad;
      h->pt_area.resize(npts_ * t->n_dof);
      h->pt_gap.resize(npts_ * t->n_dof);
      h->pt_scal.resize(npts_);
      h->face_r.resize((size_t)t->n_faces * t->n_dof * t->dim);
      h->face_scal.resize((size_t)t->n_faces * (1 + t->dim));
      h->face_active.resize((size_t)t->n_faces);
      MH_HIP(hipMemsetAsync(h->face_active.ptr, 0, (size_t)t->n_faces, h->stream));
    }
    h->status.resize(1);
    MH_HIP(hipMemsetAsync(h->status.ptr, 0, sizeof(int), h->stream));
    if (!t->csr_rowptr || !t->csr_col) fail("csr_rowptr / csr_col must be given");
    if (is_device_pointer(t->csr_rowptr)) {
      h->rowptr = t->csr_rowptr;
    } else {
      h->rowptr_own.assign(t->csr_rowptr, h->n_vdofs + 1, h->stream);
      h->rowptr = h->rowptr_own.ptr;
    }
    MH_HIP(hipMemcpy(&h->nnz, h->rowptr + h->n_vdofs, sizeof(int64_t), hipMemcpyDeviceToHost));
    {
      // contact_gather_kernel keeps the CSR row of a marked dof in LDS (CG_MAX_ROW doubles): a caller's pattern with a
      // longer marked row (multi-patch, degree >= 4) is refused here instead of overflowing the image at assembly time
      std::vector<int64_t> rp_host;
      const int64_t* rp = t->csr_rowptr;
      if (is_device_pointer(t->csr_rowptr)) {
        rp_host.resize((size_t)h->n_vdofs + 1);
        MH_HIP(hipMemcpy(rp_host.data(), t->csr_rowptr, rp_host.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
        rp = rp_host.data();
      }
      int64_t longest = 0;
      for (int32_t node : marked)
        for (int i = 0; i < t->dim; ++i) {
          const int64_t row = (int64_t)node * t->dim + i;
          longest = std::max(longest, rp[row + 1] - rp[row]);
        }
      if (longest > CG_MAX_ROW)
        fail("a CSR row of a marked contact dof holds %lld entries; the contact gather supports at most %d", (long long)longest, CG_MAX_ROW);
    }
    DeviceBuffer<int32_t> col_tmp;
    const int32_t* col_dev = t->csr_col;
    if (!is_device_pointer(t->csr_col)) {
      col_tmp.assign(t->csr_col, h->nnz, h->stream);
      col_dev = col_tmp.ptr;
    }
    const int64_t total = (int64_t)nfd * t->n_dof;
    h->pair_pos.resize(total);
    hipLaunchKernelGGL(contact_pair_pos_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, h->stream, t->n_faces,
                       t->n_dof, t->dim, h->dofs.ptr, h->rowptr, col_dev, h->pair_pos.ptr, h->status.ptr);
    MH_HIP(hipGetLastError());
    int st = 0;
    MH_HIP(hipMemcpyAsync(&st, h->status.ptr, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
    if (st) fail("CSR pattern does not contain a boundary element's dof block");
    *out = h.release();
  });
}

int mimi_hip_contact_destroy(mimi_hip_contact_t h) {
  return guarded_c([&] {
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    delete h;
  });
}

int mimi_hip_contact_set_tangent_mode(mimi_hip_contact_t h, int mode) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    if (mode != MIMI_HIP_TANGENT_ANALYTIC && mode != MIMI_HIP_TANGENT_REFERENCE_FD) fail("bad tangent mode %d", mode);
    h->mode = mode;
  });
}

int mimi_hip_contact_set_stream(mimi_hip_contact_t h, void* stream) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    h->stream = stream == MIMI_HIP_STREAM_NULL ? nullptr : (stream ? reinterpret_cast<hipStream_t>(stream) : h->own_stream);
  });
}

int mimi_hip_contact_synchronize(mimi_hip_contact_t h) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipStreamSynchronize(h->stream));
  });
}

int mimi_hip_contact_add_residual(mimi_hip_contact_t h, const double* u, double* r) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    run_contact(h, u, r, nullptr, 0.0, false);
  });
}

int mimi_hip_contact_add_residual_and_grad(mimi_hip_contact_t h, const double* u, double grad_factor, double* r,
                                           double* A_values) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    run_contact(h, u, r, A_values, grad_factor, true);
  });
}

int mimi_hip_contact_gap_norm(mimi_hip_contact_t h, const double* u, double* out) {
  return guarded_c([&] {
    if (!h || !out) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    Mirror<double> mu = Mirror<double>::in(u, h->n_vdofs, h->stage_u, h->stream);
    ContactArgs a = contact_args(h, mu.dev, nullptr, nullptr, 0.0);
    const int threads = 128;
    const int64_t npts = (int64_t)h->n_faces * h->n_q;
    const unsigned b1 = (unsigned)((npts + threads - 1) / threads);
    if (h->dim == 2)
      hipLaunchKernelGGL(contact_gap_area_kernel<2>, dim3(b1), dim3(threads), 0, h->stream, a, 1);
    else
      hipLaunchKernelGGL(contact_gap_area_kernel<3>, dim3(b1), dim3(threads), 0, h->stream, a, 1);
    hipLaunchKernelGGL(contact_sum_kernel, dim3(1), dim3(1024), 0, h->stream, npts, 1, 1, h->pt_scal.ptr,
                       (const unsigned char*)nullptr, h->scalars.ptr + 5);
    MH_HIP(hipGetLastError());
    double g2 = 0;
    MH_HIP(hipMemcpyAsync(&g2, h->scalars.ptr + 5, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
    *out = std::sqrt(g2);
  });
}

int mimi_hip_contact_last_history(mimi_hip_contact_t h, double* out5) {
  return guarded_c([&] {
    if (!h || !out5) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipMemcpyAsync(out5, h->scalars.ptr, 5 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
  });
}

int mimi_hip_contact_update_body(mimi_hip_contact_t h, const mimi_hip_spline_body* spline, double penalty) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipStreamSynchronize(h->stream));
    if (spline) {
      if (h->body_kind != MIMI_HIP_BODY_SPLINE) fail("this contact handle has no spline body");
      upload_spline_body(h, spline, h->dim);
      MH_HIP(hipStreamSynchronize(h->stream));
    }
    if (penalty > 0.0) h->penalty = penalty;
  });
}

int mimi_hip_contact_gap_area(mimi_hip_contact_t h, const double* u) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    run_contact(h, u, nullptr, nullptr, 0.0, false, 1);
  });
}

int mimi_hip_contact_marked_nodes(mimi_hip_contact_t h, int32_t* out, int64_t capacity, int64_t* n) {
  return guarded_c([&] {
    if (!h || !n) fail("null argument");
    *n = h->n_marked;
    if (!out) return;
    if (capacity < h->n_marked) fail("node buffer too small");
    std::copy(h->marked_nodes.begin(), h->marked_nodes.end(), out);
  });
}

int mimi_hip_contact_nodal(mimi_hip_contact_t h, int set, double* area, double* gap) {
  return guarded_c([&] {
    if (!h || !area || !gap) fail("null argument");
    MH_HIP(hipSetDevice(h->device));
    const size_t bytes = (size_t)h->n_marked * sizeof(double);
    if (set) {
      MH_HIP(hipMemcpyAsync(h->area.ptr, area, bytes, hipMemcpyDefault, h->stream));
      MH_HIP(hipMemcpyAsync(h->gap.ptr, gap, bytes, hipMemcpyDefault, h->stream));
    } else {
      MH_HIP(hipMemcpyAsync(area, h->area.ptr, bytes, hipMemcpyDefault, h->stream));
      MH_HIP(hipMemcpyAsync(gap, h->gap.ptr, bytes, hipMemcpyDefault, h->stream));
    }
    if (!is_device_pointer(area) || !is_device_pointer(gap)) MH_HIP(hipStreamSynchronize(h->stream));
  });
}

int mimi_hip_contact_add_residual_from_nodal(mimi_hip_contact_t h, const double* u, double grad_factor, double* r,
                                             double* A_values) {
  return guarded_c([&] {
    if (!h) fail("null handle");
    run_contact(h, u, r, A_values, grad_factor, A_values != nullptr, 2);
  });
}

int mimi_hip_contact_get_pressure(mimi_hip_contact_t h, double* out, int64_t capacity, int64_t* n) {
  return guarded_c([&] {
    if (!h || !n) fail("null argument");
    *n = h->n_marked;
    if (!out) return;
    if (capacity < h->n_marked) fail("pressure buffer too small");
    MH_HIP(hipSetDevice(h->device));
    MH_HIP(hipMemcpyAsync(out, h->pressure.ptr, h->n_marked * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    MH_HIP(hipStreamSynchronize(h->stream));
  });
}

}  // extern "C"
