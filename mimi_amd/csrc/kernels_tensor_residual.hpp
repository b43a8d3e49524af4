// Residual-only assembly (AddDomainResidual; the line search of the reference's Newton calls it several times
// per iteration) in the two-phase style: no colouring, no atomics, bitwise reproducible.
//
//   phase 1  tensor_residual_kernel: one wave per element, lane = quadrature point: gather u, F, P(F),
//            then per component i the element residual piece by sum factorisation (through LDS)
//            -> scratch_r[element][a][i].
//   phase 2  tensor_residual_gather_kernel: one wave per node: lane = element of the 3 x 3 x 3
//            neighbourhood, fixed-shape tree sum, r[node, i] += sum.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor_wgs.hpp"

namespace mimi_hip {

struct ResidualLds {
  static constexpr int ND = 27, NB = 3, NQ = 4, NQ3 = 64, NB2 = 9;
  static constexpr int off_ue = 0;                      // [3][27] (+1)
  static constexpr int off_tab = off_ue + 3 * ND + 1;   // [3][2][3][4]
  static constexpr int off_r = off_tab + 6 * NB * NQ;   // stage-R scratch
  static constexpr int r_size = 3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ;   // stage R: 444 doubles
  static constexpr int g_size = 2 * 3 * NQ * NB2 + 3 * 3 * NQ * NQ * NB;     // grad u stages (before stage R): 648 doubles
  static constexpr int per_wave = off_r + (g_size > r_size ? g_size : r_size);
};

template<int KIND>
__global__ __launch_bounds__(256) void tensor_residual_kernel(TensorArgs p, int n_el) {
  using L = ResidualLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64;
  __shared__ double lds_all[4][L::per_wave];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t e = (int64_t)blockIdx.x * 4 + wave;
  if (e >= n_el) return;
  double* ue = lds_all[wave] + L::off_ue;
  double* tab = lds_all[wave] + L::off_tab;
  double* RS = lds_all[wave] + L::off_r;
  int el[3];
  el[0] = e % p.box_n[0];
  el[1] = (e / p.box_n[0]) % p.box_n[1];
  el[2] = e / ((int64_t)p.box_n[0] * p.box_n[1]);
  if (lane < ND) {
    const int64_t node = p.dofs[e * ND + lane];
#pragma unroll
    for (int c = 0; c < 3; ++c) ue[c * ND + lane] = p.u[node * 3 + c];
  }
  for (int t = lane; t < 6 * NB * NQ; t += 64) {
    const int dir = t / (2 * NB * NQ), rem = t % (2 * NB * NQ), isD = rem / (NB * NQ), k = rem % (NB * NQ);
    const int span = p.box_begin[dir] + el[dir];
    tab[t] = ((isD ? p.tabD[dir] : p.tabB[dir]) + (int64_t)span * NB * NQ)[k];
  }
  double Ji[9];
  const double* g = p.geo + e * 10 * NQ3 + lane;
#pragma unroll
  for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NQ3];
  const double wd = g[(int64_t)9 * NQ3];
  __builtin_amdgcn_s_waitcnt(0);
  __builtin_amdgcn_wave_barrier();
  // F at the quadrature point of this lane, q = q0 + 4 q1 + 16 q2.  grad_xi u by sum factorisation, one direction per
  // stage through LDS (round 4: the direct sum over the 27 nodes was 27 x 12 multiply-adds per lane, 40 % of this kernel's
  // vector instructions; now 3 per output of the two stages -- 216 + 432 outputs over 64 lanes -- and 27 per point)
  double F[9];
  {
    const int q0 = lane & 3, q1 = (lane >> 2) & 3, q2 = lane >> 4;
    double* SA = RS;                       // [variant of direction 0: B, D][i][q0][a1 + 3 a2]
    double* SB = RS + 2 * 3 * NQ * NB2;    // [D0 B1, B0 D1, B0 B1][i][q0 + 4 q1][a2]
    for (int t = lane; t < 2 * 3 * NQ * NB2; t += 64) {
      const int a12 = t % NB2, r = t / NB2, qq = r % NQ, vi = r / NQ;
      const double* T = tab_ptr<P>(tab, 0, vi / 3) + qq;
      const double* U = ue + (vi % 3) * ND + NB * a12;
      SA[t] = T[0] * U[0] + T[NQ] * U[1] + T[2 * NQ] * U[2];
    }
    __builtin_amdgcn_wave_barrier();
    for (int t = lane; t < 3 * 3 * NQ * NQ * NB; t += 64) {
      const int a2 = t % NB, r = t / NB, q01 = r % (NQ * NQ), wi = r / (NQ * NQ), w = wi / 3, i = wi % 3;
      const double* T = tab_ptr<P>(tab, 1, w == 1 ? 1 : 0) + q01 / NQ;
      const double* S = SA + (((w == 0 ? 1 : 0) * 3 + i) * NQ + q01 % NQ) * NB2 + NB * a2;
      SB[t] = T[0] * S[0] + T[NQ] * S[1] + T[2 * NQ] * S[2];
    }
    __builtin_amdgcn_wave_barrier();
    double H[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double* T = tab_ptr<P>(tab, 2, k == 2 ? 1 : 0) + q2;
        const double* S = SB + ((k * 3 + i) * (NQ * NQ) + (q0 + NQ * q1)) * NB;
        H[i * 3 + k] = T[0] * S[0] + T[NQ] * S[1] + T[2 * NQ] * S[2];
      }
    __builtin_amdgcn_wave_barrier();     // (RS is reused by the residual stages below)
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int J = 0; J < 3; ++J) {
        double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
        for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * Ji[m * 3 + J];
        F[i + J * 3] = sf;
      }
  }
  PointResult<3> w;
  int status;
  if constexpr (KIND == WGS_KIND_RECORD) {
    status = evaluate_other<3>(p.mat, p.dt, p.state, e * NQ3 + lane, F, w.P, nullptr, 1.0);
  } else {
    MaterialDev mat = p.mat;
    mat.m.kind = KIND;
    status = evaluate_pk1<3>(mat, p.dt, p.state, e * NQ3 + lane, F, w);
  }
  if (status) atomicOr(p.status, status);
  // element residual piece of every component by sum factorisation (as kernels_tensor_2phase.hpp, stage R)
  double* PH = RS;                   // [3 m][64]
  double* V = PH + 3 * NQ3;          // [3 m][3 a2][16]
  double* W = V + 3 * NB * NQ * NQ;  // [3 m][9 a1a2][4]
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double sp = 0.0;
#pragma unroll
      for (int J = 0; J < 3; ++J) sp += w.P[I + J * 3] * Ji[m * 3 + J];
      PH[m * NQ3 + lane] = wd * sp;
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NB * NQ * NQ) {
      const int q01 = lane % (NQ * NQ), a2 = lane / (NQ * NQ);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T2 = tab_ptr<P>(tab, 2, m == 2 ? 1 : 0) + a2 * NQ;
        double sv = 0.0;
#pragma unroll
        for (int q2 = 0; q2 < NQ; ++q2) sv += T2[q2] * PH[m * NQ3 + q01 + NQ * NQ * q2];
        V[(m * NB + a2) * NQ * NQ + q01] = sv;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < NB2 * NQ) {
      const int q0 = lane % NQ, a12 = lane / NQ, a1 = a12 % NB, a2 = a12 / NB;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T1 = tab_ptr<P>(tab, 1, m == 1 ? 1 : 0) + a1 * NQ;
        double sw = 0.0;
#pragma unroll
        for (int q1 = 0; q1 < NQ; ++q1) sw += T1[q1] * V[(m * NB + a2) * NQ * NQ + q0 + NQ * q1];
        W[(m * NB2 + a12) * NQ + q0] = sw;
      }
    }
    __builtin_amdgcn_wave_barrier();
    if (lane < ND) {
      const int a0 = lane % NB, a12 = lane / NB;
      double sr = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* T0 = tab_ptr<P>(tab, 0, m == 0 ? 1 : 0) + a0 * NQ;
#pragma unroll
        for (int q0 = 0; q0 < NQ; ++q0) sr += T0[q0] * W[(m * NB2 + a12) * NQ + q0];
      }
      p.scratch_r[(e * ND + lane) * 3 + I] = sr;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// one wave per node: lane = element (dz, dy, dx) of the 3 x 3 x 3 neighbourhood, fixed-shape tree sum
__global__ __launch_bounds__(256) void tensor_residual_gather_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, NB = 3, ND = 27;
  const int64_t Al = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);   // node index inside the shard's node box
  const int lane = threadIdx.x & 63;
  if (Al >= n_nodes) return;
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1];
  const int m0 = p.box_n[0] + P, m1 = p.box_n[1] + P;
  const int A0 = p.box_begin[0] + (int)(Al % m0), A1 = p.box_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.box_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  const int bx0 = p.box_begin[0], bx1 = p.box_begin[1], bx2 = p.box_begin[2];
  const int ex_lo = max(A0 - P, bx0), ex_hi = min(A0, bx0 + p.box_n[0] - 1);
  const int ey_lo = max(A1 - P, bx1), ey_hi = min(A1, bx1 + p.box_n[1] - 1);
  const int ez_lo = max(A2 - P, bx2), ez_hi = min(A2, bx2 + p.box_n[2] - 1);
  if (ex_lo > ex_hi || ey_lo > ey_hi || ez_lo > ez_hi) return;
  const int dz = lane / 9, dy = (lane / 3) % 3, dx = lane % 3;
  const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
  const bool in = lane < ND && ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
  const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
  const int64_t e = in ? (ex - bx0) + (int64_t)p.box_n[0] * ((ey - bx1) + (int64_t)p.box_n[1] * (ez - bx2)) : 0;
  double rs[3];
#pragma unroll
  for (int I = 0; I < 3; ++I) rs[I] = in ? p.scratch_r[(e * ND + a) * 3 + I] : 0.0;
#pragma unroll
  for (int I = 0; I < 3; ++I) {
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) rs[I] += __shfl_down(rs[I], off, 32);
  }
  if (lane == 0) {
    const int64_t gA = p.perm ? p.perm[A] : A;
#pragma unroll
    for (int I = 0; I < 3; ++I) p.r[gA * 3 + I] += rs[I];
  }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the same assembly with one wave per ELEMENT COLUMN (the elements along the third direction), as the tangent
// kernels walk it.  Why: one wave per element is bound by the CU's LDS pipe, not by memory or arithmetic -- ~270 LDS
// instructions per element at 4 cycles each (512 bytes against 128 bytes per clock, ONE pipe for the four SIMDs of a CU)
// are 0.46 ms at this mesh, of 0.61 measured; sum-factorising grad u (first attempt of the round) changed nothing.  Here:
//   * the three node planes of element es are planes 1, 2 of element es - 1 and one new plane: u and the direction-0 /
//     direction-1 stages of grad u live in a ring of three LDS slots, one plane gathered and contracted per element;
//   * every stage is blocked so that a lane computes ALL outputs along the contracted direction from the few values it
//     reads (4 reads -> 3 outputs x 4 multiply-adds instead of 4 reads per output), the three row components together;
//   * no table read from LDS at all: a register pair holds the 12 entries [a][q] of a 1-D table in the first lanes of every
//     row of 16 lanes, and `v_fmac_f64_dpp row_newbcast:N` takes entry N as the multiplicand of the multiply-add itself
//     (as S2 of tensor_p3.hip; the B / D variant of a stage is uniform per row by the choice of the work items);
//     ~115 LDS instructions per element;
//   * the residual entries of a node plane are summed over the (up to) three elements of the column that touch it in a
//     ring of accumulators and leave the wave once per (column, plane): 27 doubles per element instead of 81, and the node
//     gather adds 9 column pieces instead of 27 element pieces;
//   * geometry, the direction-2 tables, u and node ids of the next element are requested while this one is computed;
//   * no branch and no execution mask in the loop: lanes without a work item compute on a valid dummy item and store to a
//     dump slot (a DPP operand read from a masked-off lane would be undefined).
// Same fixed summation orders in every run: bitwise reproducible.  Closed-form hyperelastic law only (the return mapping of
// the J2 family is latency-bound and wants many waves: those stay with one wave per element, above).
struct ResidualColLds {
  // strides padded against bank conflicts (64 banks of 4 bytes for 8-byte reads within a half wave, 32 for writes within
  // 16 lanes): the rows of a half wave, whose work items differ in im / w, must land on different banks
  static constexpr int SBI = 20, SBS = 9 * SBI;       // SB: [slot][w 3 + i][20: q01 in the first 16]
  static constexpr int PHS = 80;                      // PH: [im][80: point in the first 64]
  static constexpr int VA = 20, VS = 3 * VA;          // V: [im][a2][20: q01 in the first 16]
  static constexpr int off_ue = 0;                    // [3 slots][3 i][9 a01]
  static constexpr int off_sa = off_ue + 81;          // [3 slots][2 v][3 i][4 q0][3 a1]
  static constexpr int off_sb = off_sa + 3 * 72;      // [3 slots][3 w][3 i][q01]
  static constexpr int off_acc = off_sb + 3 * SBS;    // [3 slots][3 i][9 a01]
  static constexpr int off_ph = off_acc + 81;         // PH [3 m][point] of one row component
  static constexpr int off_v = off_ph + 3 * PHS;      // V [3 m][3 a2][q01] of one row component
  static constexpr int off_w = off_v + 3 * VS;        // W [9 im][9 a12][4 q0]
  static constexpr int off_dump = off_w + 324;        // where lanes without a work item store
  static constexpr int per_wave = off_dump + 2;       // 1 664 doubles = 13.3 KB: three workgroups of four waves per CU
};

template<int N>
MH_DEV void rc_fmac(double& acc, double table, double w) {   // acc += table[lane N of this row of 16] * w
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(table), "v"(w), "n"(N));
}
template<int N>
MH_DEV double rc_mul(double table, double w) {               // table[lane N of this row] * w
  double c;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(table), "n"(N));
  return c * w;
}
// A DPP operand must not be read within two wait states of the vector instruction that wrote it, and nothing pads inline
// asm: the table registers pass through this statement after they are written.  tests/test_isa_lint_cpu.py lints THIS kernel
// (tensor_residual_col_kernel: its 188 asm DPP instructions, no finding, no spill).  Round 5: with this compiler the fences
// are a second line -- compiled away (or replaced by a vector write of the register) the nearest write -> DPP read of the
// kernel is still more than 8 wait states, so the lint's negative test on this kernel mutates the compiled instruction
// stream instead (a vector write of the table register placed directly in front of a DPP read of it must be flagged).
#define RC_DPP_FENCE1(a) asm volatile("s_nop 1" : "+v"(a))
#define RC_DPP_FENCE3(a, b, c) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c))
#define RC_DPP_FENCE4(a, b, c, d) asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))

// out[q] = sum_a T[a][q] in[a]  (3 -> 4: the forward stages) with T entry a * 4 + q
MH_DEV void rc_3to4(double T, const double (&in)[3], double (&out)[4]) {
  out[0] = rc_mul<0>(T, in[0]); out[1] = rc_mul<1>(T, in[0]); out[2] = rc_mul<2>(T, in[0]); out[3] = rc_mul<3>(T, in[0]);
  rc_fmac<4>(out[0], T, in[1]); rc_fmac<5>(out[1], T, in[1]); rc_fmac<6>(out[2], T, in[1]); rc_fmac<7>(out[3], T, in[1]);
  rc_fmac<8>(out[0], T, in[2]); rc_fmac<9>(out[1], T, in[2]); rc_fmac<10>(out[2], T, in[2]); rc_fmac<11>(out[3], T, in[2]);
}
// out[a] = sum_q T[a][q] in[q]  (4 -> 3: the residual stages)
MH_DEV void rc_4to3(double T, const double (&in)[4], double (&out)[3]) {
  out[0] = rc_mul<0>(T, in[0]); out[1] = rc_mul<4>(T, in[0]); out[2] = rc_mul<8>(T, in[0]);
  rc_fmac<1>(out[0], T, in[1]); rc_fmac<5>(out[1], T, in[1]); rc_fmac<9>(out[2], T, in[1]);
  rc_fmac<2>(out[0], T, in[2]); rc_fmac<6>(out[1], T, in[2]); rc_fmac<10>(out[2], T, in[2]);
  rc_fmac<3>(out[0], T, in[3]); rc_fmac<7>(out[1], T, in[3]); rc_fmac<11>(out[2], T, in[3]);
}
MH_DEV void rc_4to3_add(double T, const double (&in)[4], double (&out)[3]) {
  rc_fmac<0>(out[0], T, in[0]); rc_fmac<4>(out[1], T, in[0]); rc_fmac<8>(out[2], T, in[0]);
  rc_fmac<1>(out[0], T, in[1]); rc_fmac<5>(out[1], T, in[1]); rc_fmac<9>(out[2], T, in[1]);
  rc_fmac<2>(out[0], T, in[2]); rc_fmac<6>(out[1], T, in[2]); rc_fmac<10>(out[2], T, in[2]);
  rc_fmac<3>(out[0], T, in[3]); rc_fmac<7>(out[1], T, in[3]); rc_fmac<11>(out[2], T, in[3]);
}

template<int KIND>
__global__ __launch_bounds__(256) void tensor_residual_col_kernel(TensorArgs p, int n_cols) {
  using L = ResidualColLds;
  constexpr int NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64;
  extern __shared__ __align__(16) double lds_col[];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  // (a wave beyond the last column walks the last column again and stores nothing but what that column's wave stores too:
  // the same values to the same places -- no early exit, no mask)
  const int col = min((int)blockIdx.x * 4 + wave, n_cols - 1);
  double* base = lds_col + (size_t)wave * L::per_wave;
  double* ue = base + L::off_ue;
  double* SA = base + L::off_sa;
  double* SB = base + L::off_sb;
  double* acc = base + L::off_acc;
  double* PH = base + L::off_ph;
  double* W = base + L::off_w;
  double* V = base + L::off_v;
  double* dump = base + L::off_dump;
  const int ex = col % p.box_n[0], ey = col / p.box_n[0], nz = p.box_n[2];
  const int64_t e_step = (int64_t)p.box_n[0] * p.box_n[1];
  const int64_t e0 = ex + (int64_t)p.box_n[0] * ey;
  const int row = lane >> 4, c16 = lane & 15;

  // 1-D tables as DPP operands: lane n of every row holds entry n = a * 4 + q (n < 12)
  const int tn = c16 < 12 ? c16 : 11;
  double TB0 = (p.tabB[0] + (int64_t)(p.box_begin[0] + ex) * NB * NQ)[tn], TD0 = (p.tabD[0] + (int64_t)(p.box_begin[0] + ex) * NB * NQ)[tn];
  double TB1 = (p.tabB[1] + (int64_t)(p.box_begin[1] + ey) * NB * NQ)[tn], TD1 = (p.tabD[1] + (int64_t)(p.box_begin[1] + ey) * NB * NQ)[tn];
  // direction 2, per element, in two forms: entry n as above, and for the points (row = q2) lane n = [a = n][q = row]
  const int tr = (c16 < 3 ? c16 : 2) * NQ + row;
  for (int t = lane; t < 81; t += 64) acc[t] = 0.0;
  // the node plane gather: lane = c 9 + a01 (lanes >= 27 repeat lane 26 and store to the dump slot)
  const int lp = lane < ND ? lane : ND - 1;
  const int pc = lp / 9, pa = lp % 9;
  auto plane_dof = [&](int P) -> const int32_t* {
    const int es = P < nz ? P : nz - 1;     // plane P = the a2 = 0 nodes of element P; the last two: a2 = 1, 2 of the last element
    return p.dofs + (e0 + e_step * es) * ND + NB2 * (P - es) + pa;
  };
  // work items of the two forward stages (fixed per lane)
  //   stage A: lanes 0..31 variant B, 32..63 variant D; item (i, a1) = 9 of the 32
  const int av = lane >> 5, aidx = (lane & 31) < 9 ? (lane & 31) : 0, ai = aidx / 3, aa1 = aidx % 3;
  const bool a_ok = (lane & 31) < 9;
  double TA0 = av ? TD0 : TB0;
  //   stage B: row = w (0: D0 B1, 1: B0 D1, 2: B0 B1; row 3 idle), item (i, q0) = 12 of the 16
  const int bw = row < 3 ? row : 2, bidx = c16 < 12 ? c16 : 0, bi = bidx >> 2, bq0 = bidx & 3;
  const bool b_ok = row < 3 && c16 < 12;
  double TB1s = bw == 1 ? TD1 : TB1;
  const int vm = row < 3 ? row : 2;          // the residual stages: row = m
  double Ty = vm == 1 ? TD1 : TB1;           // contraction of q1: the D table for m = 1
  RC_DPP_FENCE4(TB0, TD0, TB1, TD1);
  RC_DPP_FENCE4(TA0, TB1s, Ty, TD0);
  // contract a0, then a1, of the plane in ring slot `slot`
  auto ingest = [&](int slot) {
    {
      const double* U = ue + slot * 27 + ai * 9 + 3 * aa1;
      const double in[3] = {U[0], U[1], U[2]};
      double out[4];
      rc_3to4(TA0, in, out);
      double* dst = a_ok ? SA + slot * 72 + ((av * 3 + ai) * 4) * 3 + aa1 : dump;
      const int st = a_ok ? 3 : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q * st] = out[q];
    }
    __builtin_amdgcn_wave_barrier();
    {
      const double* S = SA + slot * 72 + (((bw == 0 ? 1 : 0) * 3 + bi) * 4 + bq0) * 3;
      const double in[3] = {S[0], S[1], S[2]};
      double out[4];
      rc_3to4(TB1s, in, out);
      double* dst = b_ok ? SB + slot * L::SBS + (bw * 3 + bi) * L::SBI + bq0 : dump;
      const int st = b_ok ? 4 : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[q * st] = out[q];
    }
    __builtin_amdgcn_wave_barrier();
  };
  // prologue: planes 0, 1 (plane 2 arrives as the "next plane" of the first pass of the loop)
  {
    const int64_t n0 = *plane_dof(0), n1 = *plane_dof(1);
    const double v0 = p.u[n0 * 3 + pc], v1 = p.u[n1 * 3 + pc];
    double* d0 = lane < ND ? ue + lane : dump;
    double* d1 = lane < ND ? ue + 27 + lane : dump;
    *d0 = v0;
    *d1 = v1;
    __builtin_amdgcn_wave_barrier();
    ingest(0);
    ingest(1);
  }
  // what travels one element ahead: geometry (10 per lane), the direction-2 tables (4 per lane), u of the new plane
  double Jn[9], wdn, un, TB2n, TD2n, TB2rn, TD2rn;
  int64_t nid_next;
  {
    const double* g = p.geo + e0 * 10 * NQ3 + lane;
#pragma unroll
    for (int k = 0; k < 9; ++k) Jn[k] = g[(int64_t)k * NQ3];
    wdn = g[(int64_t)9 * NQ3];
    const double* B2 = p.tabB[2] + (int64_t)p.box_begin[2] * NB * NQ;
    const double* D2 = p.tabD[2] + (int64_t)p.box_begin[2] * NB * NQ;
    TB2n = B2[tn];
    TD2n = D2[tn];
    TB2rn = B2[tr];
    TD2rn = D2[tr];
    un = p.u[(int64_t)(*plane_dof(2)) * 3 + pc];
    nid_next = *plane_dof(nz > 1 ? 3 : 2);
  }
  int o0 = 0, o1 = 1, o2 = 2;      // ring slots of the planes a2 = 0, 1, 2 of the current element
  int status = 0;
  double* out = p.scratch_r + (int64_t)col * (nz + 2) * ND;
  const int q01 = c16;             // the point of this lane: q0 + 4 q1 = lane & 15, q2 = row
#pragma unroll 1
  for (int es = 0; es < nz; ++es) {
    double Ji[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ji[k] = Jn[k];
    const double wd = wdn;
    double TB2r = TB2rn, TD2r = TD2rn;
    double Tz = vm == 2 ? TD2n : TB2n;       // contraction of q2: row = m, the D table for m = 2
    RC_DPP_FENCE3(Tz, TB2r, TD2r);
    {
      double* d = lane < ND ? ue + o2 * 27 + lane : dump;
      *d = un;
    }
    __builtin_amdgcn_wave_barrier();
    // requests for the next element (the last one asks for itself again: no branch, nothing out of range)
    {
      const int en = es + 1 < nz ? es + 1 : es;
      const double* g = p.geo + (e0 + e_step * en) * 10 * NQ3 + lane;
#pragma unroll
      for (int k = 0; k < 9; ++k) Jn[k] = g[(int64_t)k * NQ3];
      wdn = g[(int64_t)9 * NQ3];
      const double* B2 = p.tabB[2] + (int64_t)(p.box_begin[2] + en) * NB * NQ;
      const double* D2 = p.tabD[2] + (int64_t)(p.box_begin[2] + en) * NB * NQ;
      TB2n = B2[tn];
      TD2n = D2[tn];
      TB2rn = B2[tr];
      TD2rn = D2[tr];
      un = p.u[nid_next * 3 + pc];
      const int Pn = es + 4 < nz + 2 ? es + 4 : nz + 1;
      nid_next = *plane_dof(Pn);
    }
    ingest(o2);
    // grad_xi u at the point of this lane (q01, q2 = row): contract a2 over the three ring slots
    double F[9];
    {
      double H[9];
      const double* S0 = SB + o0 * L::SBS + q01;
      const double* S1 = SB + o1 * L::SBS + q01;
      const double* S2 = SB + o2 * L::SBS + q01;
#pragma unroll
      for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const double Tk = k == 2 ? TD2r : TB2r;
          double h = rc_mul<0>(Tk, S0[(k * 3 + i) * L::SBI]);
          rc_fmac<1>(h, Tk, S1[(k * 3 + i) * L::SBI]);
          rc_fmac<2>(h, Tk, S2[(k * 3 + i) * L::SBI]);
          H[i * 3 + k] = h;
        }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int J = 0; J < 3; ++J) {
          double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * Ji[m * 3 + J];
          F[i + J * 3] = sf;
        }
    }
    const int64_t e = e0 + e_step * es;
    PointResult<3> w;
    {
      MaterialDev mat = p.mat;
      mat.m.kind = KIND;
      status |= evaluate_pk1<3>(mat, p.dt, p.state, e * NQ3 + lane, F, w);
    }
    // one row component I at a time (LDS for three waves per SIMD): Phat[I][m] of the point -> LDS [m][point]; contract q2
    // (row = m, lane of the row = q01; all a2 per lane); contract q1 (row = m, lane of the row = (a2, q0), 12 of 16; all a1
    // per lane) -> W[I 3 + m], which stays for the last stage
#pragma unroll
    for (int I = 0; I < 3; ++I) {
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double sp = 0.0;
#pragma unroll
        for (int J = 0; J < 3; ++J) sp += w.P[I + J * 3] * Ji[m * 3 + J];
        PH[m * L::PHS + lane] = wd * sp;
      }
      __builtin_amdgcn_wave_barrier();
      {
        const double* src = PH + vm * L::PHS + c16;
        const double in[4] = {src[0], src[16], src[32], src[48]};
        double o3[3];
        rc_4to3(Tz, in, o3);
        double* dst = row < 3 ? V + vm * L::VS + c16 : dump;
        const int st = row < 3 ? L::VA : 0;
#pragma unroll
        for (int a2 = 0; a2 < 3; ++a2) dst[a2 * st] = o3[a2];
      }
      __builtin_amdgcn_wave_barrier();
      {
        const bool ok = row < 3 && c16 < 12;
        const int cc = c16 < 12 ? c16 : 0, a2 = cc >> 2, r0 = cc & 3;
        const double* src = V + vm * L::VS + a2 * L::VA + r0;
        const double in[4] = {src[0], src[4], src[8], src[12]};
        double o3[3];
        rc_4to3(Ty, in, o3);
        double* dst = ok ? W + ((I * 3 + vm) * 9 + 3 * a2) * 4 + r0 : dump;
        const int st = ok ? 4 : 0;
#pragma unroll
        for (int a1 = 0; a1 < 3; ++a1) dst[a1 * st] = o3[a1];
      }
      __builtin_amdgcn_wave_barrier();
    }
    // contract q0 and sum over m: lane = (I, a12) (27 lanes); all a0 per lane; into the ring of plane accumulators
    {
      const int fl = lane < ND ? lane : 0, I = fl / 9, a12 = fl % 9, a2 = a12 / 3, a1 = a12 % 3;
      double o3[3];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const double* src = W + ((I * 3 + m) * 9 + a12) * 4;
        const double in[4] = {src[0], src[1], src[2], src[3]};
        if (m == 0) rc_4to3(TD0, in, o3);
        else rc_4to3_add(TB0, in, o3);
      }
      const int slot = a2 == 0 ? o0 : (a2 == 1 ? o1 : o2);
      double* dst = lane < ND ? acc + slot * 27 + I * 9 + 3 * a1 : dump;
      const int st = lane < ND ? 1 : 0;
#pragma unroll
      for (int a0 = 0; a0 < 3; ++a0) dst[a0 * st] += o3[a0];
    }
    __builtin_amdgcn_wave_barrier();
    // plane es is complete: out, and its slot starts plane es + 3 from zero
    {
      const int fl = lane < ND ? lane : 0;
      const double v = acc[o0 * 27 + fl];
      double* dz = lane < ND ? acc + o0 * 27 + lane : dump;
      *dz = 0.0;
      if (lane < ND) out[(int64_t)es * ND + lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    const int t = o0;
    o0 = o1;
    o1 = o2;
    o2 = t;
  }
  // the two planes above the last element
  if (lane < ND) {
    out[(int64_t)nz * ND + lane] = acc[o0 * 27 + lane];
    out[(int64_t)(nz + 1) * ND + lane] = acc[o1 * 27 + lane];
  }
  if (status) atomicOr(p.status, status);
}

// one thread per node of the shard: the 3 x 3 columns that contain it, in a fixed order
__global__ __launch_bounds__(256) void tensor_residual_col_gather_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, ND = 27;
  const int64_t Al = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (Al >= n_nodes) return;
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1];
  const int m0 = p.box_n[0] + P, m1 = p.box_n[1] + P, nz = p.box_n[2];
  const int l0 = (int)(Al % m0), l1 = (int)((Al / m0) % m1), l2 = (int)(Al / ((int64_t)m0 * m1));   // node inside the shard's box
  double rs[3] = {0.0, 0.0, 0.0};
#pragma unroll
  for (int dy = 0; dy < 3; ++dy)
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      const int cx = l0 - dx, cy = l1 - dy;          // column whose local node (a0, a1) = (dx, dy) this is
      if (cx < 0 || cx >= p.box_n[0] || cy < 0 || cy >= p.box_n[1]) continue;
      const double* src = p.scratch_r + (((int64_t)cx + (int64_t)p.box_n[0] * cy) * (nz + 2) + l2) * ND + dx + 3 * dy;
#pragma unroll
      for (int I = 0; I < 3; ++I) rs[I] += src[I * 9];
    }
  const int64_t A = (p.box_begin[0] + l0) + (int64_t)n0 * ((p.box_begin[1] + l1) + (int64_t)n1 * (p.box_begin[2] + l2));
  const int64_t gA = p.perm ? p.perm[A] : A;
#pragma unroll
  for (int I = 0; I < 3; ++I) p.r[gA * 3 + I] += rs[I];
}

inline void launch_tensor_residual(mimi_hip_domain_s* h, TensorArgs a) {
  // MIMI_HIP_RESIDUAL_VARIANT=element: one wave per element (rounds 1-3; kept for A/B timing)
  static const bool per_element = getenv("MIMI_HIP_RESIDUAL_VARIANT") && getenv("MIMI_HIP_RESIDUAL_VARIANT")[0] == 'e';
  if (!per_element && h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN) {
    const int n_cols = a.box_n[0] * a.box_n[1];
    h->scratch_r.resize(std::max((size_t)h->n_el * 3 * 27, (size_t)n_cols * (a.box_n[2] + 2) * 27));
    a.scratch_r = h->scratch_r.ptr;
    const unsigned blocks = (unsigned)((n_cols + 3) / 4);
    const size_t lds = (size_t)4 * ResidualColLds::per_wave * sizeof(double);
    if (lds > 64 * 1024) ensure_dynamic_lds(reinterpret_cast<const void*>(tensor_residual_col_kernel<MIMI_HIP_MAT_NEOHOOKEAN>), (int)lds);
    hipLaunchKernelGGL(tensor_residual_col_kernel<MIMI_HIP_MAT_NEOHOOKEAN>, dim3(blocks), dim3(256), lds, h->stream, a, n_cols);
    MH_HIP(hipGetLastError());
    const int64_t n_nodes = (int64_t)(a.box_n[0] + 2) * (a.box_n[1] + 2) * (a.box_n[2] + 2);   // nodes of the shard
    hipLaunchKernelGGL(tensor_residual_col_gather_kernel, dim3((unsigned)((n_nodes + 255) / 256)), dim3(256), 0, h->stream, a, n_nodes);
    MH_HIP(hipGetLastError());
    return;
  }
  h->scratch_r.resize((size_t)h->n_el * 3 * 27);
  a.scratch_r = h->scratch_r.ptr;
  const unsigned blocks = (unsigned)((h->n_el + 3) / 4);
  if (h->mat.m.kind != MIMI_HIP_MAT_NEOHOOKEAN && h->mat.m.kind != MIMI_HIP_MAT_J2)
    hipLaunchKernelGGL(tensor_residual_kernel<WGS_KIND_RECORD>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  else if (h->mat.m.kind == MIMI_HIP_MAT_NEOHOOKEAN)
    hipLaunchKernelGGL(tensor_residual_kernel<MIMI_HIP_MAT_NEOHOOKEAN>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  else
    hipLaunchKernelGGL(tensor_residual_kernel<MIMI_HIP_MAT_J2>, dim3(blocks), dim3(256), 0, h->stream, a, (int)h->n_el);
  MH_HIP(hipGetLastError());
  const int64_t n_nodes = (int64_t)(a.box_n[0] + 2) * (a.box_n[1] + 2) * (a.box_n[2] + 2);   // nodes of the shard
  hipLaunchKernelGGL(tensor_residual_gather_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, a, n_nodes);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
