// Phase 1 for materials with a major-symmetric tangent (hyperelastic: dP_iJ/dF_jL = dP_jL/dF_iJ), p = 2, 3-D.
//
// K_(a,i),(b,j) = K_(b,j),(a,i): of the nine (i, j) blocks of an element only the six with i >= j
// are contracted; an off-diagonal block is stored twice, as computed into piece (element, i) and
// transposed (a <-> b) into piece (element, j).  The scratch layout and phase 2 are those of
// kernels_tensor_2phase.hpp; the stored matrix is exactly symmetric.
//
// Workgroup = 4 waves as in kernels_tensor_wgs.hpp (same roles, same register carry, same LDS
// hand-off), but TWO steps per element instead of three:
//
//   step A(e)  [global step 2e-1]   Y0: block (2,1) of element e-1   X: rows 0, 1 of element e
//                                   Y1: block (2,0)      "
//                                   Y2: block (2,2)      "
//   step B(e)  [global step 2e]     Y0: block (0,0) of element e     X: row 2 of element e, then the
//                                   Y1: block (1,1)      "               quadrature-point stage of e+1
//                                   Y2: block (1,0)      "
//   a step = [Y: flush / read operands from LDS] barrier [X, Y: compute, write LDS] barrier
//
// Rows 0, 1 of Ahat are read (into registers) in the read window of B(e) and rewritten by X in A(e+1);
// row 2 is read in A(e+1) and rewritten in B(e+1): one LDS buffer.  The three store-transposition
// buffers (one per piece i) are written by all three contraction waves during B(e) and A(e+1) and
// flushed, piece w by wave Y_w, in the read window of B(e+1).  After the last element one more step
// flushes the carried rows.  Every wave executes 2 (2 n + 3) barriers.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor_wgs.hpp"

namespace mimi_hip {

#ifndef WGSYM_DIAG_MODE
#define WGSYM_DIAG_MODE 2   // 2: contract only the a1 >= b1 chains of a diagonal block; 0: all nine
#endif

// ------------------------------------------------------------------------------------------------
// wave X, two steps per element
// ------------------------------------------------------------------------------------------------
template<int KIND>
MH_DEV void wgsym_x_loop(const TensorArgs& p, double* lds, int eu, int ev, int& status) {
  using L = WgsLds;
  constexpr int P = 2, NB = 3, NQ = 4, ND = 27, NQ3 = 64;
  constexpr int TROUNDS = 2;
  const int lane = threadIdx.x & 63;
  double* ue = lds + L::off_ue;
  const int n_seq = p.box_n[2];
  auto element_at = [&](int es) -> int64_t { return eu + (int64_t)p.box_n[0] * (ev + (int64_t)p.box_n[1] * es); };
  auto table_src = [&](int es, int t) -> const double* {
    const int dir = t / (2 * NB * NQ);
    const int rem = t % (2 * NB * NQ);
    const int isD = rem / (NB * NQ);
    const int k = rem % (NB * NQ);
    const int span = (dir == 0 ? p.box_begin[0] + eu : dir == 1 ? p.box_begin[1] + ev : p.box_begin[2] + es);
    return (isD ? (dir == 0 ? p.tabD[0] : dir == 1 ? p.tabD[1] : p.tabD[2])
                : (dir == 0 ? p.tabB[0] : dir == 1 ? p.tabB[1] : p.tabB[2])) + (int64_t)span * NB * NQ + k;
  };
  int32_t node_n = lane < ND ? p.dofs[element_at(0) * ND + lane] : 0;
  double ue_r[3], tab_r[TROUNDS], geo_r[10];
  auto request = [&](int es) {
    const int64_t e_n = element_at(es);
#pragma unroll
    for (int c = 0; c < 3; ++c) ue_r[c] = p.u[(int64_t)node_n * 3 + c];
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      tab_r[rd] = *table_src(es, t < 6 * NB * NQ ? t : 0);
    }
    const double* g = p.geo + e_n * 10 * NQ3 + lane;
#pragma unroll
    for (int k = 0; k < 10; ++k) geo_r[k] = g[(int64_t)k * NQ3];
    if (es + 1 < n_seq) node_n = lane < ND ? p.dofs[element_at(es + 1) * ND + lane] : 0;
  };
  WgsPoint<KIND> s;
  // quadrature-point stage of element es from the requested operands (tables -> LDS parity es & 1)
  auto point_stage = [&](int es) {
    double* tab = lds + L::off_tab + (es & 1) * 6 * NB * NQ;
    if (lane < ND) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ue[c * ND + lane] = ue_r[c];
    }
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      if (t < 6 * NB * NQ) tab[t] = tab_r[rd];
    }
    double Ji[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) Ji[k] = geo_r[k];
    const double wd = geo_r[9];
    __builtin_amdgcn_wave_barrier();
    double F[9];
    {
      const int q0 = lane & 3, q1 = (lane >> 2) & 3, q2 = lane >> 4;
      double b0[NB], d0[NB], b1[NB], d1[NB], b2[NB], d2[NB];
#pragma unroll
      for (int a = 0; a < NB; ++a) {
        b0[a] = tab_ptr<P>(tab, 0, 0)[a * NQ + q0];
        d0[a] = tab_ptr<P>(tab, 0, 1)[a * NQ + q0];
        b1[a] = tab_ptr<P>(tab, 1, 0)[a * NQ + q1];
        d1[a] = tab_ptr<P>(tab, 1, 1)[a * NQ + q1];
        b2[a] = tab_ptr<P>(tab, 2, 0)[a * NQ + q2];
        d2[a] = tab_ptr<P>(tab, 2, 1)[a * NQ + q2];
      }
      double H[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) H[k] = 0.0;
#pragma unroll
      for (int a2 = 0; a2 < NB; ++a2)
#pragma unroll
        for (int a1 = 0; a1 < NB; ++a1) {
          const double tbb = b1[a1] * b2[a2], tdb = d1[a1] * b2[a2], tbd = b1[a1] * d2[a2];
#pragma unroll
          for (int a0 = 0; a0 < NB; ++a0) {
            const int a = a0 + NB * (a1 + NB * a2);
            const double dn0 = d0[a0] * tbb, dn1 = b0[a0] * tdb, dn2 = b0[a0] * tbd;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
              const double uu = ue[i * ND + a];
              H[i * 3 + 0] += uu * dn0;
              H[i * 3 + 1] += uu * dn1;
              H[i * 3 + 2] += uu * dn2;
            }
          }
          // keep the LDS reads of later (a1, a2) where they are (hoisted together they need 162 registers)
          #pragma unroll
          for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(H[k]) : : "memory");
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
    status |= wgs_x_point<KIND>(p, element_at(es) * NQ3 + lane, F, Ji, wd, s);
    __builtin_amdgcn_wave_barrier();
  };

  request(0);
  point_stage(0);
  if (1 < n_seq) request(1);
  for (int it = 0; it <= n_seq; ++it) {
    const bool valid = it < n_seq;
    // ---- step A(it): rows 0, 1 of element it ---------------------------------------------------------
    wgs_barrier();
    if (valid) {
      wgs_x_row<KIND, 0>(p, lds, lane, element_at(it), it & 1, s);
      wgs_x_row<KIND, 1>(p, lds, lane, element_at(it), it & 1, s);
    }
    wgs_barrier();
    // ---- step B(it): row 2 of element it, then the point stage of element it + 1 -----------------------
    wgs_barrier();
    if (valid) {
      wgs_x_row<KIND, 2>(p, lds, lane, element_at(it), it & 1, s);
      if (it + 1 < n_seq) {
        point_stage(it + 1);
        if (it + 2 < n_seq) request(it + 2);
      }
    }
    wgs_barrier();
  }
  // ---- final step: the contraction waves flush the carried rows ------------------------------------------
  wgs_barrier();
  wgs_barrier();
}

// ------------------------------------------------------------------------------------------------
// wave Y_W: blocks (I0, J0) in step B and (I1, J1) in step A; flushes piece W
// ------------------------------------------------------------------------------------------------
template<int W>
MH_DEV void wgsym_y_loop(const TensorArgs& p, double* lds, int eu, int ev) {
  using L = WgsLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81, NK = ND * NROW;
  // step B blocks: Y0 (0,0), Y1 (1,1), Y2 (1,0);  step A blocks: Y0 (2,1), Y1 (2,0), Y2 (2,2)
  constexpr int I0 = W == 0 ? 0 : 1, J0 = W == 0 ? 0 : W == 1 ? 1 : 0;
  constexpr int I1 = 2, J1 = W == 0 ? 1 : W == 1 ? 0 : 2;
  const WgsLane lc = wgs_lane_constants();
  const int lane = lc.lane;
  const int n_seq = p.box_n[2];
  const double* AH0 = lds + L::off_ah + I0 * ND * NQ3;
  const double* AH1 = lds + L::off_ah + I1 * ND * NQ3;
  auto st_of = [&](int piece) -> double* { return lds + L::off_st + piece * L::st_size; };
  auto piece_of = [&](int es) -> double* {
    return p.scratch_k + ((eu + (int64_t)p.box_n[0] * (ev + (int64_t)p.box_n[1] * es)) * 3 + W) * (int64_t)NK;
  };
  const int mrow = lane & 15, mk = lane >> 4;
  const bool mrow_ok = mrow < NB2;
  const int mra = mrow_ok ? mrow / NB : 0, mrb = mrow_ok ? mrow % NB : 0;

  double C0[NB2], C1[NB2];  // packed carries of the two blocks
#pragma unroll
  for (int k = 0; k < NB2; ++k) C0[k] = C1[k] = 0.0;
  double aS0[4], aS2[4];
  double uB1[NB][NQ], uD1[NB][NQ];
#pragma unroll
  for (int v = 0; v < 4; ++v) aS0[v] = aS2[v] = 0.0;
#pragma unroll
  for (int a = 0; a < NB; ++a)
#pragma unroll
    for (int q = 0; q < NQ; ++q) uB1[a][q] = uD1[a][q] = 0.0;
  {
    // step A(0) runs on zeros (element -1: its results are never stored): the slots this wave reads then
    double* AHw = lds + L::off_ah + I1 * ND * NQ3;
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) AHw[((m * 3 + J1) * 3 + n) * NQ3 + lane] = 0.0;
  }

  for (int it = 0; it <= n_seq; ++it) {
    // ---- step A(it): block (I1, J1) of element it - 1 -----------------------------------------------------
    {
      double ah[9];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) ah[m * 3 + n] = AH1[((m * 3 + J1) * 3 + n) * NQ3 + lane];
      wgs_barrier();
      wgs_contract_block<(I1 != J1 ? 1 : WGSYM_DIAG_MODE)>(lc, ah, aS0, aS2, uB1, uD1, C1, st_of(I1), J1, st_of(J1), I1);
      wgs_barrier();
    }
    // ---- step B(it): flush element it - 1, block (I0, J0) of element it --------------------------------------
    {
      if (it >= 1) wgs_flush_final(lane, st_of(W), piece_of(it - 1));
      if (it < n_seq) {
        const double* tab = lds + L::off_tab + (it & 1) * 6 * NB * NQ;
        {
          const double Ba = tab_ptr<P>(tab, 0, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 0, 1)[mra * NQ + mk];
          const double Bb = tab_ptr<P>(tab, 0, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 0, 1)[mrb * NQ + mk];
          aS0[0] = mrow_ok ? Ba * Bb : 0.0;
          aS0[1] = mrow_ok ? Da * Bb : 0.0;
          aS0[2] = mrow_ok ? Ba * Db : 0.0;
          aS0[3] = mrow_ok ? Da * Db : 0.0;
        }
        {
          const double Ba = tab_ptr<P>(tab, 2, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 2, 1)[mra * NQ + mk];
          const double Bb = tab_ptr<P>(tab, 2, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 2, 1)[mrb * NQ + mk];
          aS2[0] = mrow_ok ? Ba * Bb : 0.0;
          aS2[1] = mrow_ok ? Da * Bb : 0.0;
          aS2[2] = mrow_ok ? Ba * Db : 0.0;
          aS2[3] = mrow_ok ? Da * Db : 0.0;
        }
#pragma unroll
        for (int a = 0; a < NB; ++a)
#pragma unroll
          for (int q1 = 0; q1 < NQ; ++q1) {
            const unsigned long long vb = __double_as_longlong(tab_ptr<P>(tab, 1, 0)[a * NQ + q1]);
            const unsigned long long vd = __double_as_longlong(tab_ptr<P>(tab, 1, 1)[a * NQ + q1]);
            const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)vb), bhi = __builtin_amdgcn_readfirstlane((unsigned)(vb >> 32));
            const unsigned dlo = __builtin_amdgcn_readfirstlane((unsigned)vd), dhi = __builtin_amdgcn_readfirstlane((unsigned)(vd >> 32));
            uB1[a][q1] = __longlong_as_double(((unsigned long long)bhi << 32) | blo);
            uD1[a][q1] = __longlong_as_double(((unsigned long long)dhi << 32) | dlo);
          }
      }
      double ah[9];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) ah[m * 3 + n] = AH0[((m * 3 + J0) * 3 + n) * NQ3 + lane];
      wgs_barrier();
      if (it < n_seq) {
        wgs_contract_block<(I0 != J0 ? 1 : WGSYM_DIAG_MODE)>(lc, ah, aS0, aS2, uB1, uD1, C0, st_of(I0), J0, st_of(J0), I0);
      } else {
        // past the last element: the carried rows have no successor and are stored as well
        wgs_stage_carry<(I0 != J0 ? 1 : WGSYM_DIAG_MODE)>(lc, C0, st_of(I0), J0, st_of(J0), I0);
        wgs_stage_carry<(I1 != J1 ? 1 : WGSYM_DIAG_MODE)>(lc, C1, st_of(I1), J1, st_of(J1), I1);
      }
      wgs_barrier();
    }
  }
  // ---- final step: flush the carried rows of the last element ------------------------------------------------
  wgs_flush_carry(lane, st_of(W), piece_of(n_seq - 1));
  wgs_barrier();
  wgs_barrier();
}

template<int KIND>
__global__ __launch_bounds__(256, 2) void tensor_wgsym_kernel(TensorArgs p) {
  extern __shared__ __align__(16) double smem_wgsym[];
  const int role = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + WGS_ROT(blockIdx.x)) & 3);
  const int unit = blockIdx.x;
  const int eu = unit % p.box_n[0], ev = unit / p.box_n[0];
  // (WGSYM_EXP_SKIP_X / _Y: timing experiments only — the skipped role just keeps the barrier count)
  if (role == 0) {
    int status = 0;
#ifdef WGSYM_EXP_SKIP_X
    for (int k = 0; k < 2 * (2 * p.box_n[2] + 3); ++k) wgs_barrier();
#else
    wgsym_x_loop<KIND>(p, smem_wgsym, eu, ev, status);
#endif
    if (status) atomicOr(p.status, status);
  } else {
#ifdef WGSYM_EXP_SKIP_Y
    for (int k = 0; k < 2 * (2 * p.box_n[2] + 3); ++k) wgs_barrier();
#else
    if (role == 1) wgsym_y_loop<0>(p, smem_wgsym, eu, ev);
    else if (role == 2) wgsym_y_loop<1>(p, smem_wgsym, eu, ev);
    else wgsym_y_loop<2>(p, smem_wgsym, eu, ev);
#endif
  }
}

inline void launch_tensor_wgsym(mimi_hip_domain_s* h, TensorArgs a) {
  constexpr int NK = 27 * 81;
  h->scratch_k.resize((size_t)h->n_el * 3 * NK);
  h->scratch_r.resize((size_t)h->n_el * 3 * 27);
  a.scratch_k = h->scratch_k.ptr;
  a.scratch_r = h->scratch_r.ptr;
  a.n_units_u = a.box_n[0];
  a.n_units_v = a.box_n[1];
  const size_t lds = WgsLds::total * sizeof(double);
  auto kernel = tensor_wgsym_kernel<MIMI_HIP_MAT_NEOHOOKEAN>;
  MH_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(kernel, dim3(a.box_n[0] * a.box_n[1]), dim3(256), lds, h->stream, a);
  MH_HIP(hipGetLastError());
  launch_tensor_p2(h, a);
}

}  // namespace mimi_hip
