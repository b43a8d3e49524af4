// General-table domain kernels: one workgroup per element, any node numbering, any
// (rational or not) basis -- the tables are the caller's QuadData flattened
// (utils/precomputed.hpp:58-71).  Replaces
//   NonlinearSolid::ElementResidual / ElementResidualAndGrad / ThreadLocalResidual[AndGrad]
//   (integrators/nonlinear_solid.hpp:65-87, nonlinear_solid.cpp:48-149)
//   and the AddThreadLocal* reductions (integrators/nonlinear_base.hpp:90-151):
// instead of per-thread global copies + a reduction pass, element contributions go out as dense element blocks / element
// residual vectors, and general_gather_kernel (one wave per CSR row, node -> element adjacency built at setup, fixed
// summation order) adds them into r / the CSR values: no atomics, bitwise reproducible.  (Only when the blocks do not fit
// in device memory, or a CSR row is longer than the gather kernel's LDS image, do the contributions go straight into
// r / A with fp64 hardware atomics.)
//
// Work split inside a workgroup (256 threads = 4 waves):
//   phase 0  gather u_e = u[dofs] into LDS (integrator_utils.cpp:44-51)
//   phase 1  one lane per quadrature point: F = u_e^T dN_dX + I, P(F) [, dP/dF]
//            scaled by w*det -> LDS
//   phase 2  one lane per element dof (a,i): R_e(a,i) = sum_q dN_dX(a,:) . P(i,:)  -> r
//   phase 3  one lane per node pair (a,b): the dim x dim block
//            K(ai,bj) = sum_q dN_aJ A_iJjL dN_bL                                   -> A
#pragma once

#include <hip/hip_runtime.h>
#include <hip/amd_detail/amd_hip_unsafe_atomics.h>

#include "materials.hpp"
#include "materials_other.hpp"

namespace mimi_hip {

struct GeneralArgs {
  int n_el, n_dof, n_q;
  const int32_t* dofs;      // [n_el][n_dof]
  const double* dN_dX;      // [n_el][n_q][DIM][n_dof]
  const double* wdet;       // [n_el][n_q]
  const int64_t* rowptr;    // [n_vdofs+1]
  const int32_t* pair_pos;  // [n_el][n_dof][n_dof]: offset of column (B*dim) inside row (A*dim)
  const double* u;
  double* r;
  double* A;
  double grad_factor;
  double dt;
  MaterialDev mat;
  StateView state;
  int* status;
  int lds_per_element;      // WPE kernels: bytes of LDS per element (general_lds_bytes rounded up to 16)
  double* scratch_k;        // dense element blocks [n_el][(a, i)][(j, b)] instead of atomics (then gathered); nullptr: atomics
  double* scratch_r;        // element residual vectors [n_el][i][a] instead of atomics (then gathered); nullptr: atomics
  double* mat_rec;          // FAMILY 1 (the other materials' tangent assemblies): [n_el][n_q][DIM^2 + DIM^4] = w det P, w det dP/dF
};

MH_DEV void atomic_add_f64(double* p, double v) { unsafeAtomicAdd(p, v); }

// setup: pair_pos[e][a][b] by binary search of column dofs[b]*dim in row dofs[a]*dim
__global__ void pair_pos_kernel(int n_el, int n_dof, int dim, const int32_t* __restrict__ dofs,
                                const int64_t* __restrict__ rowptr, const int32_t* __restrict__ col,
                                int32_t* __restrict__ pair_pos, int* __restrict__ status) {
  const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t total = (int64_t)n_el * n_dof * n_dof;
  if (idx >= total) return;
  const int b = idx % n_dof;
  const int a = (idx / n_dof) % n_dof;
  const int64_t e = idx / ((int64_t)n_dof * n_dof);
  const int64_t row = (int64_t)dofs[e * n_dof + a] * dim;
  const int32_t target = dofs[e * n_dof + b] * dim;
  int64_t lo = rowptr[row], hi = rowptr[row + 1];
  const int64_t base = lo;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if (col[mid] < target) lo = mid + 1; else hi = mid;
  }
  if (lo >= rowptr[row + 1] || col[lo] != target) {
    atomicOr(status, 4);  // pattern does not contain the element block
    pair_pos[idx] = 0;
    return;
  }
  pair_pos[idx] = (int32_t)(lo - base);
}

template<int DIM>
MH_DEV void compute_F_general(int n_dof, const double* __restrict__ g /* [DIM][n_dof] */,
                              const double* u_e /* LDS [DIM][n_dof] */, double* F) {
#pragma unroll
  for (int k = 0; k < DIM * DIM; ++k) F[k] = 0.0;
  for (int a = 0; a < n_dof; ++a) {
    double ga[DIM], ua[DIM];
#pragma unroll
    for (int J = 0; J < DIM; ++J) ga[J] = g[J * n_dof + a];
#pragma unroll
    for (int i = 0; i < DIM; ++i) ua[i] = u_e[i * n_dof + a];
#pragma unroll
    for (int i = 0; i < DIM; ++i)
#pragma unroll
      for (int J = 0; J < DIM; ++J) MH_M(F, i, J) += ua[i] * ga[J];
  }
#pragma unroll
  for (int i = 0; i < DIM; ++i) MH_M(F, i, i) += 1.0;
}

#ifndef GEN_WAVES
#define GEN_WAVES 2
#endif
// GRAD: 0 residual only, 1 analytic tangent, 2 reference forward difference
// PP: node pairs per lane and pass of the node-pair phase (3 covers up to 768 pairs = p <= 2 in 3-D; 8 halves the
// passes of larger elements)
// THREADS: workgroup size (256; 512 for elements with more than 768 node pairs: one pass of the node-pair phase for p = 3)
// FAMILY: 0 neo-Hookean / J2 (closed-form tangents, materials.hpp); 2..5 that one of the other materials
//         (materials_other.hpp) as a compile-time constant -- all four behind a switch in one kernel spilled 170 - 640
//         registers (round 4: one instantiation per material)
// MF: 1 = node-pair phase on the fp64 matrix instruction (3-D, 64 nodes per element = p 3, 512 threads): per quadrature
//     point K[(a), (b, i, j)] += sum_J g[J][a] * (sum_L A_q[iJ, jL] g[L][b]) is a 64 x 576 x 3 product; wave w owns the
//     16 column nodes b of tile w & 3 and two of the four 16-row tiles, 18 accumulator tiles (2 x 9 (i, j)) in registers
typedef double mhg_d4 __attribute__((ext_vector_type(4)));
#ifndef GEN_MF_QC
#define GEN_MF_QC 8   // quadrature points staged per barrier pair of the matrix-instruction node-pair phase
#endif
#define GEN_SYNC() do { if constexpr (WPE) __builtin_amdgcn_wave_barrier(); else __syncthreads(); } while (0)
// Material pre-pass of the general path's TANGENT assemblies for the materials without a closed-form tangent (StVenant-
// Kirchhoff, J2Linear, J2Simo, J2Log; round 5): one 64-lane workgroup per element, lane = quadrature point, one instantiation
// per material, the whole register file to itself -- the dual-number stress routines of J2Simo / J2Log need ~ 470 registers
// and spilled 480 - 520 of them inside the element kernel, whose occupancy bound leaves 256.
// Leaves w det P and w det dP/dF per point (mat_rec); the element kernel then runs as FAMILY 1 and only loads them.
template<int DIM, int FK>
__global__ __launch_bounds__(64) void general_material_kernel(GeneralArgs p) {
  constexpr int DD = DIM * DIM, D4 = DD * DD;
  extern __shared__ __align__(16) unsigned char smem_mat[];
  double* u_e = reinterpret_cast<double*>(smem_mat);   // [DIM][n_dof]
  const int e = blockIdx.x, tid = threadIdx.x;
  const int n_dof = p.n_dof, n_q = p.n_q, n_tdof = n_dof * DIM;
  for (int t = tid; t < n_tdof; t += 64) {
    const int a = t % n_dof, i = t / n_dof;
    u_e[t] = p.u[(int64_t)p.dofs[(int64_t)e * n_dof + a] * DIM + i];
  }
  __syncthreads();
  const double* gE = p.dN_dX + (int64_t)e * n_q * n_tdof;
  int status = 0;
  for (int q = tid; q < n_q; q += 64) {
    double F[DD], P[DD];
    compute_F_general<DIM>(n_dof, gE + (int64_t)q * n_tdof, u_e, F);
    const double wd = p.wdet[(int64_t)e * n_q + q];
    double* rec = p.mat_rec + ((int64_t)e * n_q + q) * (DD + D4);
    // (the tangent one direction (j, L) at a time, as in the tensor pre-pass kernels: nothing but the direction's DIM^2
    // derivatives is live beside the material's own working set)
    OtherTangent<DIM> ot;
    status |= other_tangent_begin<DIM, FK>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F, P, ot);
#pragma unroll
    for (int k = 0; k < DD; ++k) rec[k] = wd * P[k];
#pragma unroll 1
    for (int j = 0; j < DIM; ++j)
#pragma unroll 1
      for (int L = 0; L < DIM; ++L) {
        double dP[DD], Fl[DD];
        // (F passes through an empty asm per direction: otherwise the compiler hoists everything of the dual-number stress
        // routine that depends on the value parts alone out of the loop and keeps it live across the nine directions --
        // ~ 200 values parked in the accumulation file, counted as spilled registers, for J2Log)
#pragma unroll
        for (int k = 0; k < DD; ++k) {
          Fl[k] = F[k];
          asm volatile("" : "+v"(Fl[k]));
        }
        other_tangent_dir<DIM, FK>(p.mat, p.dt, Fl, ot, j, L, dP);
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int J = 0; J < DIM; ++J) rec[DD + ((i * DIM + J) * DIM + j) * DIM + L] = wd * MH_M(dP, i, J);
      }
  }
  if (status) atomicOr(p.status, status);
}

// WPE: 1 = one WAVE per element (small elements: 2-D, p = 1): THREADS / 64 elements per workgroup, every barrier a wave
//      barrier, the LDS block of the element at wave * general_lds_bytes
template<int DIM, int GRAD, int PP = 3, int THREADS = 256, int FAMILY = 0, int MF = 0, int WPE = 0>
// (FAMILY: 0 closed-form materials; 1 tangent assembly of another material: P and dP/dF come from general_material_kernel;
//  2..5 that one of the other materials evaluated here -- residual-only and reference-FD assemblies.  The 3-D one-wave-per-
//  element tangent kernel gets the whole register file: at two waves per SIMD it spilled 67 - 76 registers)
__global__ __launch_bounds__(THREADS, THREADS == 256 ? ((WPE && DIM == 3 && GRAD == 1) ? 1 : GEN_WAVES) : 1) void domain_general_kernel(GeneralArgs p) {
  constexpr int DD = DIM * DIM;
  constexpr int D4 = DD * DD;
  extern __shared__ __align__(16) unsigned char smem_all[];
  const int e = WPE ? blockIdx.x * (THREADS / 64) + (threadIdx.x >> 6) : blockIdx.x;
  if (WPE && e >= p.n_el) return;   // the whole wave leaves together; the others only use wave barriers
  const int tid = WPE ? (threadIdx.x & 63) : threadIdx.x;
  const int n_threads = WPE ? 64 : (int)blockDim.x;
  const int n_dof = p.n_dof, n_q = p.n_q, n_tdof = n_dof * DIM;
  unsigned char* smem_raw = smem_all + (WPE ? (threadIdx.x >> 6) * p.lds_per_element : 0);

  double* u_e = reinterpret_cast<double*>(smem_raw);  // [DIM][n_dof]
  double* Pw = u_e + n_tdof;                          // [n_q][DD]   w*det*P
  double* Aw = Pw + n_q * DD;                         // [n_q][D4]   w*det*dP/dF   (GRAD==1)
  double* R_e = Aw + (GRAD == 1 ? n_q * D4 + n_dof * DD * DIM + (GEN_MF_QC > 8 ? GEN_MF_QC : 8) * n_tdof : 0);  // [n_tdof] (GRAD==2); GRAD==1: T[n_dof][DIM^3], gC[8][n_tdof] sit before it
  int32_t* node = reinterpret_cast<int32_t*>(R_e + (GRAD == 2 ? n_tdof : 0));  // [n_dof]

  const double* gE = p.dN_dX + (int64_t)e * n_q * n_tdof;
  const double* wE = p.wdet + (int64_t)e * n_q;

  for (int t = tid; t < n_dof; t += n_threads) node[t] = p.dofs[(int64_t)e * n_dof + t];
  GEN_SYNC();
  for (int t = tid; t < n_tdof; t += n_threads) {
    const int a = t % n_dof, i = t / n_dof;
    u_e[t] = p.u[(int64_t)node[a] * DIM + i];
  }
  GEN_SYNC();

  // phase 1: constitutive update per quadrature point
  int status = 0;
  if constexpr (FAMILY == 1) {
    static_assert(GRAD == 1, "FAMILY 1 is the tangent assembly behind general_material_kernel");
    const double* rec = p.mat_rec + (int64_t)e * n_q * (DD + D4);
    for (int t = tid; t < n_q * (DD + D4); t += n_threads) {
      const int q = t / (DD + D4), k = t % (DD + D4);
      if (k < DD) Pw[q * DD + k] = rec[t]; else Aw[q * D4 + (k - DD)] = rec[t];
    }
  }
  for (int q = tid; FAMILY != 1 && q < n_q; q += n_threads) {
    double F[DD];
    compute_F_general<DIM>(n_dof, gE + (int64_t)q * n_tdof, u_e, F);
    const double wd = wE[q];
    if constexpr (FAMILY >= 2) {
      double P[DD];
      status |= evaluate_other<DIM, (FAMILY >= 2 ? FAMILY : -1)>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F, P, GRAD == 1 ? Aw + q * D4 : nullptr, wd);
#pragma unroll
      for (int k = 0; k < DD; ++k) Pw[q * DD + k] = wd * P[k];
      continue;
    }
    PointResult<DIM> w;
    status |= evaluate_pk1<DIM>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F, w);
#pragma unroll
    for (int k = 0; k < DD; ++k) Pw[q * DD + k] = wd * w.P[k];
    if constexpr (GRAD == 1) {
      // one row dP_i./dF at a time (DIM^3 values live instead of DIM^4)
      constexpr int D3r = DD * DIM;
      {
        double Ar[D3r];
        tangent_row_of<DIM, 0>(p.mat.m, w, Ar);
#pragma unroll
        for (int k = 0; k < D3r; ++k) Aw[q * D4 + k] = wd * Ar[k];
      }
      {
        double Ar[D3r];
        tangent_row_of<DIM, 1>(p.mat.m, w, Ar);
#pragma unroll
        for (int k = 0; k < D3r; ++k) Aw[q * D4 + D3r + k] = wd * Ar[k];
      }
      if constexpr (DIM == 3) {
        double Ar[D3r];
        tangent_row_of<DIM, 2>(p.mat.m, w, Ar);
#pragma unroll
        for (int k = 0; k < D3r; ++k) Aw[q * D4 + 2 * D3r + k] = wd * Ar[k];
      }
    }
  }
  GEN_SYNC();

  // phase 2: residual (AddMult_a_ABt, nonlinear_solid.hpp:79-82)
  for (int t = tid; t < n_tdof; t += n_threads) {
    const int a = t % n_dof, i = t / n_dof;
    double s = 0.0;
    for (int q = 0; q < n_q; ++q) {
      const double* g = gE + (int64_t)q * n_tdof;
#pragma unroll
      for (int J = 0; J < DIM; ++J) s += g[J * n_dof + a] * Pw[q * DD + i + J * DIM];
    }
    if constexpr (GRAD == 2) R_e[t] = s;
    if (p.scratch_r) p.scratch_r[(int64_t)e * n_tdof + t] = s;
    else atomic_add_f64(&p.r[(int64_t)node[a] * DIM + i], s);
  }

  if constexpr (GRAD == 1 && MF == 1) {
    static_assert(DIM == 3 && THREADS == 512, "matrix-instruction node-pair phase: 3-D, 512 threads");
    // phase 3 on v_mfma_f64_16x16x4: A operand [row = lane % 16][k = lane / 16] = g[J = k][a], B operand
    // [k = lane / 16][col = lane % 16] = sum_L A_q[iJ, jL] g[L][b], D register r of lane l:
    // row (l / 16) + 4 r, column l % 16
    constexpr int QC = GEN_MF_QC;
    double* gC = Aw + n_q * D4 + n_dof * (DD * DIM);   // [QC][DIM][n_dof] (the T buffer of the other route is unused)
    const int wave = tid >> 6, lane = tid & 63;
    const int nt = wave & 3, mh = wave >> 2;
    const int l16 = lane & 15, kk = lane >> 4;
    mhg_d4 acc[2][DD];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int c = 0; c < DD; ++c) acc[mt][c] = mhg_d4{0.0, 0.0, 0.0, 0.0};
    for (int q0 = 0; q0 < n_q; q0 += QC) {
      const int nqc = n_q - q0 < QC ? n_q - q0 : QC;
      GEN_SYNC();
      for (int t = tid; t < nqc * n_tdof; t += n_threads) gC[t] = gE[(int64_t)q0 * n_tdof + t];
      GEN_SYNC();
      // the K index of the product is the flattened (point, J): four consecutive values per matrix instruction, no padding
      // (lane group kk takes value 4 s + kk of the chunk: its own point and J)
      const int n_k = nqc * DIM;
      for (int s4 = 0; s4 < n_k; s4 += 4) {
        const int kidx = s4 + kk;
        const bool valid = kidx < n_k;
        const int qq = valid ? kidx / DIM : 0, J = valid ? kidx % DIM : 0;
        const double* g = gC + qq * n_tdof;
        const double* Aq = Aw + (q0 + qq) * D4;
        const double a0 = valid ? g[J * n_dof + 16 * (2 * mh) + l16] : 0.0;
        const double a1 = valid ? g[J * n_dof + 16 * (2 * mh + 1) + l16] : 0.0;
        double gb[DIM];
#pragma unroll
        for (int L = 0; L < DIM; ++L) gb[L] = g[L * n_dof + 16 * nt + l16];
#pragma unroll
        for (int i = 0; i < DIM; ++i)
#pragma unroll
          for (int j = 0; j < DIM; ++j) {
            const double* Ar = Aq + ((i * DIM + J) * DIM + j) * DIM;
            double bv = 0.0;
#pragma unroll
            for (int L = 0; L < DIM; ++L) bv += Ar[L] * gb[L];
            bv = valid ? bv : 0.0;
            acc[0][i * DIM + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc[0][i * DIM + j], 0, 0, 0);
            acc[1][i * DIM + j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc[1][i * DIM + j], 0, 0, 0);
          }
      }
    }
    const int32_t* pp_tab = p.pair_pos + (int64_t)e * n_dof * n_dof;
    const int b = 16 * nt + l16;
    if (p.scratch_k) {
      // two-phase: the element block goes out densely, row (a, i) = 192 contiguous doubles [j][b] (16 lanes of an
      // accumulator register cover 128 contiguous bytes); general_gather_kernel
      // sums the rows of every CSR row afterwards (the scattered fp64 atomics below run memory-side and bound this path)
      double* Ke = p.scratch_k + (int64_t)e * (n_tdof * n_tdof);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int a = 16 * (2 * mh + mt) + kk + 4 * r;
#pragma unroll
          for (int i = 0; i < DIM; ++i)
#pragma unroll
            for (int j = 0; j < DIM; ++j) Ke[(a * DIM + i) * n_tdof + j * n_dof + b] = acc[mt][i * DIM + j][r];
        }
    } else
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int a = 16 * (2 * mh + mt) + kk + 4 * r;
        const int64_t rowA = (int64_t)node[a] * DIM;
        const int32_t off = pp_tab[a * n_dof + b];
#pragma unroll
        for (int i = 0; i < DIM; ++i) {
          double* dst = p.A + p.rowptr[rowA + i] + off;
#pragma unroll
          for (int j = 0; j < DIM; ++j) atomic_add_f64(dst + j, p.grad_factor * acc[mt][i * DIM + j][r]);
        }
      }
  } else if constexpr (GRAD == 1) {
    // phase 3: node-pair blocks K(ai,bj) = sum_q sum_J ga[J] T_b[iJ][j],  T_b[iJ][j] = sum_L A_q[iJ,jL] gb[L].
    // Per quadrature point T is built once for all nodes b (n_dof DIM^3 values in LDS), then every lane adds its
    // node pairs: DIM^3 multiply-adds per (pair, point) instead of DIM^4 + DIM^3.
    constexpr int D3 = DD * DIM;
    const int n_pairs = n_dof * n_dof;
    const int32_t* pp_tab = p.pair_pos + (int64_t)e * n_pairs;
    constexpr int QC = 8;
    double* T = Aw + n_q * D4;   // [n_dof][D3]
    double* gC = T + n_dof * D3;  // [QC][DIM][n_dof]
    // a lane owns one column node b and PP row nodes a = ag PP .. ag PP + PP - 1: T_b is read once per point for
    // all of them (the LDS traffic of this phase is what bounds large elements)
    const int n_groups = (n_dof + PP - 1) / PP;
    const int n_slots = n_dof * n_groups;
    for (int slot0 = 0; slot0 < n_slots; slot0 += (int)n_threads) {
      const int slot = slot0 + tid;
      const bool active = slot < n_slots;
      const int b = active ? slot % n_dof : 0, ag = active ? slot / n_dof : 0;
      double acc[PP][DD];  // acc[.][i*DIM + j]
#pragma unroll
      for (int k = 0; k < PP; ++k)
#pragma unroll
        for (int c = 0; c < DD; ++c) acc[k][c] = 0.0;
      for (int q0 = 0; q0 < n_q; q0 += QC) {
        // the gradients of QC quadrature points -> LDS (one global read per value instead of one per use)
        const int nqc = n_q - q0 < QC ? n_q - q0 : QC;
        for (int t = tid; t < nqc * n_tdof; t += n_threads) gC[t] = gE[(int64_t)q0 * n_tdof + t];
        GEN_SYNC();
        for (int qq = 0; qq < nqc; ++qq) {
          const double* g = gC + qq * n_tdof;
          const double* Aq = Aw + (q0 + qq) * D4;
          for (int t = tid; t < n_dof * D3; t += n_threads) {
            const int bb = t / D3, c = t % D3;   // c = (i*DIM + J)*DIM + j
            double tv = 0.0;
#pragma unroll
            for (int L = 0; L < DIM; ++L) tv += Aq[c * DIM + L] * g[L * n_dof + bb];
            T[t] = tv;
          }
          GEN_SYNC();
          if (active) {
            double Tb[D3];
#pragma unroll
            for (int c = 0; c < D3; ++c) Tb[c] = T[b * D3 + c];
#pragma unroll
            for (int k = 0; k < PP; ++k) {
              const int a = ag * PP + k;
              if (a < n_dof) {
#pragma unroll
                for (int J = 0; J < DIM; ++J) {
                  const double gaJ = g[J * n_dof + a];
#pragma unroll
                  for (int i = 0; i < DIM; ++i)
#pragma unroll
                    for (int j = 0; j < DIM; ++j) acc[k][i * DIM + j] += gaJ * Tb[(i * DIM + J) * DIM + j];
                }
              }
            }
          }
          GEN_SYNC();
        }
      }
      if (active) {
#pragma unroll
        for (int k = 0; k < PP; ++k) {
          const int a = ag * PP + k;
          if (a < n_dof && p.scratch_k) {
            double* Ke = p.scratch_k + (int64_t)e * (n_tdof * n_tdof);
#pragma unroll
            for (int i = 0; i < DIM; ++i)
#pragma unroll
              for (int j = 0; j < DIM; ++j) Ke[(a * DIM + i) * n_tdof + j * n_dof + b] = acc[k][i * DIM + j];
          } else if (a < n_dof) {
            const int64_t rowA = (int64_t)node[a] * DIM;
            const int32_t off = pp_tab[a * n_dof + b];
#pragma unroll
            for (int i = 0; i < DIM; ++i) {
              double* dst = p.A + p.rowptr[rowA + i] + off;
#pragma unroll
              for (int j = 0; j < DIM; ++j) atomic_add_f64(dst + j, p.grad_factor * acc[k][i * DIM + j]);
            }
          }
        }
      }
    }
  }

  if constexpr (GRAD == 2) {
    // reference rule (nonlinear_solid.cpp:48-76): column c = (R_e(u_e + h e_c) - R_e(u_e)) / h,
    // h = |u_c|*1e-8 or 1e-10; one column at a time, all quadrature points re-evaluated.
    const int n_pairs = n_dof * n_dof;
    const int32_t* pp = p.pair_pos + (int64_t)e * n_pairs;
    for (int c = 0; c < n_tdof; ++c) {
      GEN_SYNC();
      const double orig = u_e[c];
      const double step = (orig != 0.0) ? fabs(orig) * 1.0e-8 : 1.0e-10;
      const double step_inv = 1. / step;
      GEN_SYNC();
      if (tid == 0) u_e[c] = orig + step;
      GEN_SYNC();
      for (int q = tid; q < n_q; q += n_threads) {
        double F[DD];
        compute_F_general<DIM>(n_dof, gE + (int64_t)q * n_tdof, u_e, F);
        const double wd = wE[q];
        if constexpr (FAMILY != 0) {
          double P[DD];
          status |= evaluate_other<DIM, (FAMILY >= 2 ? FAMILY : -1)>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F, P, nullptr, wd);
#pragma unroll
          for (int k = 0; k < DD; ++k) Pw[q * DD + k] = wd * P[k];
        } else {
          PointResult<DIM> w;
          status |= evaluate_pk1<DIM>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F, w);
#pragma unroll
          for (int k = 0; k < DD; ++k) Pw[q * DD + k] = wd * w.P[k];
        }
      }
      GEN_SYNC();
      if (tid == 0) u_e[c] = orig;
      const int b = c % n_dof, j = c / n_dof;
      for (int t = tid; t < n_tdof; t += n_threads) {
        const int a = t % n_dof, i = t / n_dof;
        double s = 0.0;
        for (int q = 0; q < n_q; ++q) {
          const double* g = gE + (int64_t)q * n_tdof;
#pragma unroll
          for (int J = 0; J < DIM; ++J) s += g[J * n_dof + a] * Pw[q * DD + i + J * DIM];
        }
        const double k_entry = (s - R_e[t]) * step_inv;
        const int64_t rowA = (int64_t)node[a] * DIM + i;
        if (p.scratch_k) p.scratch_k[(int64_t)e * (n_tdof * n_tdof) + (a * DIM + i) * n_tdof + j * n_dof + b] = k_entry;
        else atomic_add_f64(p.A + p.rowptr[rowA] + pp[a * n_dof + b] + j, p.grad_factor * k_entry);
      }
    }
  }

  if (status) atomicOr(p.status, status);
}

#undef GEN_SYNC

// DomainPostTimeAdvance (nonlinear_solid.cpp:179-199): one lane per quadrature point
template<int DIM, int FAMILY = 0>
__global__ __launch_bounds__(256) void post_time_advance_general_kernel(GeneralArgs p) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int e = blockIdx.x;
  const int tid = threadIdx.x;
  const int n_dof = p.n_dof, n_q = p.n_q, n_tdof = n_dof * DIM;
  double* u_e = reinterpret_cast<double*>(smem_raw);
  for (int t = tid; t < n_tdof; t += blockDim.x) {
    const int a = t % n_dof, i = t / n_dof;
    u_e[t] = p.u[(int64_t)p.dofs[(int64_t)e * n_dof + a] * DIM + i];
  }
  __syncthreads();
  int status = 0;
  for (int q = tid; q < n_q; q += blockDim.x) {
    double F[DIM * DIM];
    compute_F_general<DIM>(n_dof, p.dN_dX + ((int64_t)e * n_q + q) * n_tdof, u_e, F);
    if constexpr (FAMILY != 0) status |= accumulate_other<DIM, (FAMILY >= 2 ? FAMILY : -1)>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F);
    else status |= accumulate_state<DIM>(p.mat, p.dt, p.state, (int64_t)e * n_q + q, F);
  }
  if (status) atomicOr(p.status, status);
}

// phase 2 of the two-phase general path (64-node elements): one wave per CSR row (node, i).  The wave walks the elements
// that contain the node (adjacency built at setup), adds row (a, i) of each element block into an LDS image of the CSR
// row through the pair positions (lane = column node b: distinct positions within an instruction), then adds the image
// to the caller's values in one coalesced pass: no atomics, a fixed summation order.
constexpr int GG_WAVES = 4;
constexpr int GG_MAX_ROW = 1056;   // (2 p + 1)^3 neighbours x 3 at p = 3 is 1029
// adjacency entries are (element << 6) | local node (n_dof <= 64).  WITH_K 0: residual rows only.
template<int DIM, int WITH_K>
__global__ __launch_bounds__(64 * GG_WAVES) void general_gather_kernel(int64_t n_rows, int n_dof, const int64_t* __restrict__ rowptr,
                                                                       const int64_t* __restrict__ adj_ptr, const int32_t* __restrict__ adj,
                                                                       const int32_t* __restrict__ pair_pos, const double* __restrict__ scratch_k,
                                                                       const double* __restrict__ scratch_r, double grad_factor,
                                                                       const double* A_base, double* A, double* __restrict__ r) {
  __shared__ double img_all[WITH_K ? GG_WAVES : 1][GG_MAX_ROW];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * GG_WAVES + wave;
  if (row >= n_rows) return;
  const int64_t node = row / DIM;
  const int i = (int)(row % DIM);
  const int64_t a_beg = adj_ptr[node], a_end = adj_ptr[node + 1];
  if (a_beg == a_end) return;   // no element of this handle touches the node (element slabs)
  const int NT = n_dof * DIM;
  if constexpr (WITH_K) {
    double* img = img_all[wave];
    const int64_t beg = rowptr[row];
    const int len = (int)(rowptr[row + 1] - beg);
    for (int k = lane; k < len; k += 64) img[k] = 0.0;
    __builtin_amdgcn_wave_barrier();
    for (int64_t t = a_beg; t < a_end; ++t) {
      const int32_t ea = adj[t];
      const int64_t e = ea >> 6;
      const int a = ea & 63;
      const double* Kr = scratch_k + (e * NT + (a * DIM + i)) * (int64_t)NT;   // row (a, i): [j][b]
      for (int b = lane; b < n_dof; b += 64) {    // (distinct b: distinct positions within an instruction)
        const int32_t off = pair_pos[(e * n_dof + a) * n_dof + b];
#pragma unroll
        for (int j = 0; j < DIM; ++j) img[off + j] += Kr[j * n_dof + b];
      }
      __builtin_amdgcn_wave_barrier();
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < len; k += 64) A[beg + k] = A_base[beg + k] + grad_factor * img[k];
  }
  // residual entry: the incident elements in adjacency order over the lanes, then a fixed-shape tree
  double rs = 0.0;
  for (int64_t t = a_beg + lane; t < a_end; t += 64) {
    const int32_t ea = adj[t];
    rs += scratch_r[(int64_t)(ea >> 6) * NT + i * n_dof + (ea & 63)];
  }
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) rs += __shfl_down(rs, off, 64);
  if (lane == 0) r[row] += rs;
}

inline size_t general_lds_bytes(int dim, int n_dof, int n_q, int grad) {
  const int dd = dim * dim, n_tdof = n_dof * dim;
  size_t doubles = n_tdof + (size_t)n_q * dd;
  if (grad == 1) doubles += (size_t)n_q * dd * dd + (size_t)n_dof * dd * dim + (size_t)(GEN_MF_QC > 8 ? GEN_MF_QC : 8) * n_tdof;
  if (grad == 2) doubles += n_tdof;
  return doubles * sizeof(double) + (size_t)n_dof * sizeof(int32_t);
}

}  // namespace mimi_hip
