// Phase 1 for materials with a major-symmetric tangent (hyperelastic: dP_iJ/dF_jL = dP_jL/dF_iJ), p = 2, 3-D.
//
// K_(a,i),(b,j) = K_(b,j),(a,i): of the nine (i, j) blocks of an element only the six with i >= j
// are contracted; an off-diagonal block is stored twice, as computed into piece (element, i) and
// transposed (a <-> b) into piece (element, j).  The scratch layout and phase 2 are those of
// kernels_tensor_2phase.hpp; the stored matrix is exactly symmetric.
//
// Workgroup = 4 waves as in kernels_tensor_wgs.hpp (same roles, same register carry, same LDS
// hand-off), but TWO steps per element instead of three, and homogeneous ones -- the three
// contraction waves do equally expensive blocks in the same step (a diagonal block costs 0.73 of an
// off-diagonal one, and a lock step lasts as long as its slowest wave):
//
//   step D(e)   Y0: block (0,0) of element e   Y1: (1,1)   Y2: (2,2)     X: quadrature-point stage of e+1
//   step O(e)   Y0: block (1,0) of element e   Y1: (2,0)   Y2: (2,1)     X: rows 0, 1, 2 of element e+1
//   O(e) = [Y: read operands from LDS] barrier [X, Y: compute, write LDS] barrier;  D(e) runs free (no barriers):
//   what it reads was written before the previous barrier and nothing it writes is read before the next one
//
// The rows of Ahat of element e are written during O(e-1), read (into registers) in the read windows
// of D(e) and O(e) and rewritten during O(e): one LDS buffer, only the 54 entries (i, j <= i) kept.
// The three store-transposition buffers (one per piece i) are written by all three contraction waves
// during D(e) and O(e) and flushed, piece w by wave Y_w, in the read window of D(e+1).  A prologue
// step lets X write the rows of element 0; after the last element each contraction wave stores the carried
// rows (outside the lock steps).  Every wave executes 2 n + 1 barriers.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor_wgs.hpp"

namespace mimi_hip {

#ifndef WGSYM_SPLIT_BELOW
#define WGSYM_SPLIT_BELOW 512   // cut the columns into segments while the launch has at most this many workgroups
#endif
#ifndef WGSYM_MAX_COLS
#define WGSYM_MAX_COLS 4      // element columns per workgroup at most
#define WGSYM_MIN_WGS 2048    // ... while at least this many workgroups remain (4 per resident slot)
#endif

#ifndef WGSYM_DIAG_MODE
#define WGSYM_DIAG_MODE 2   // 2: contract only the a1 >= b1 chains of a diagonal block; 0: all nine
#endif

struct WgsymLds {
  static constexpr int NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81;
  static constexpr int off_ue = 0;                          // [3][27] (+1 pad)              X private
  static constexpr int off_tab = off_ue + 3 * ND + 1;       // [2 parity][3 dir][2][3][4]     X -> Y
  static constexpr int off_r = off_tab + 2 * 6 * NB * NQ;   // residual scratch, three rows    X private
  static constexpr int r_size = 3 * (3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ);
  static constexpr int off_ah = off_r + r_size;             // [6 blocks (i, j <= i)][9 (m,n)][64]  X -> Y
  static constexpr int off_st = off_ah + 6 * 9 * NQ3;       // [3 i][1216] store transposition      Y
  static constexpr int off_dump = off_st + 3 * WgsLds::st_size;   // where lanes without an entry store (wgs_contract_block)
  static constexpr int total = off_dump + 512;
  // first Ahat entry of block (i, j), j <= i
  MH_DEV static constexpr int ah_block(int i, int j) { return (i * (i + 1) / 2 + j) * 9; }
};

// rows 0, 1, 2 of one element in one go: Ahat blocks (i, j <= i) -> LDS, residual pieces -> scratch_r
// (the three residual rows share their four LDS passes)
template<int KIND>
MH_DEV void wgsym_x_rows(const TensorArgs& p, double* lds, int lane, int64_t e, int par, const WgsPoint<KIND>& s) {
  using L = WgsymLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64;
  static_assert(KIND == MIMI_HIP_MAT_NEOHOOKEAN, "symmetric-half kernel: hyperelastic materials only");
  const double* tab = lds + L::off_tab + par * 6 * NB * NQ;
  double* AH = lds + L::off_ah;
  double* PH = lds + L::off_r;                  // [3 I][3 m][64]
  double* V = PH + 9 * NQ3;                     // [3 I][3 m][3 a2][16]
  double* W = V + 9 * NB * NQ * NQ;             // [3 I][3 m][9 a1a2][4]
#pragma unroll
  for (int I = 0; I < 3; ++I)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      const double c2gm = s.c2_w * s.G[m * 3 + I];
#pragma unroll
      for (int j = 0; j <= I; ++j) {
        const double c1gm = s.c1_w * s.G[m * 3 + j];
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          double v = c2gm * s.G[n * 3 + j] - c1gm * s.G[n * 3 + I];
          if (I == j) {
            const int lo = m < n ? m : n, hi = m < n ? n : m;
            v += s.mu_w * s.M[lo * 3 - lo * (lo - 1) / 2 + (hi - lo)];
          }
          AH[(L::ah_block(I, j) + m * 3 + n) * NQ3 + lane] = v;
        }
      }
    }
#pragma unroll
  for (int k = 0; k < 9; ++k) PH[k * NQ3 + lane] = s.Phat[k];   // k = I * 3 + m
  __builtin_amdgcn_wave_barrier();
  // Lane -> output maps of the three stages (round 5, as in the degree-3 pre-pass): a lane owns ONE pair of the indices that are
  // neither summed nor a tensor component, reads its table values once and walks (I, m) with compile-time offsets -- 48 / 36 /
  // 54 of the 64 lanes busy, a third of the instructions of the loops over the flat output index they replace (20 - 30 integer
  // instructions of index arithmetic per four multiply-adds); the sums run in the same order: same bits.
  // V[I m][a2][q0 q1] = sum_q2 T2^m[a2][q2] PH[I m][q0 q1 q2]
  if (lane < NB * NQ * NQ) {
    const int q01 = lane & 15, a2 = lane >> 4;
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q2 = 0; q2 < NQ; ++q2) T[v][q2] = tab_ptr<P>(tab, 2, v)[a2 * NQ + q2];
#pragma unroll
    for (int im = 0; im < 9; ++im) {
      double sv = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < NQ; ++q2) sv = __builtin_fma(T[im % 3 == 2 ? 1 : 0][q2], PH[im * NQ3 + q01 + NQ * NQ * q2], sv);
      V[im * (NB * NQ * NQ) + lane] = sv;   // (im * NB + a2) * 16 + q01
    }
  }
  __builtin_amdgcn_wave_barrier();
  // W[I m][a1 a2][q0] = sum_q1 T1^m[a1][q1] V[I m][a2][q0 q1]
  if (lane < NB2 * NQ) {
    const int q0 = lane & 3, a12 = lane >> 2, a1 = a12 % NB, a2 = a12 / NB;
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q1 = 0; q1 < NQ; ++q1) T[v][q1] = tab_ptr<P>(tab, 1, v)[a1 * NQ + q1];
#pragma unroll
    for (int im = 0; im < 9; ++im) {
      double sw = 0.0;
#pragma unroll
      for (int q1 = 0; q1 < NQ; ++q1) sw = __builtin_fma(T[im % 3 == 1 ? 1 : 0][q1], V[(im * NB + a2) * NQ * NQ + q0 + NQ * q1], sw);
      W[im * (NB2 * NQ) + lane] = sw;   // (im * NB2 + a12) * 4 + q0
    }
  }
  __builtin_amdgcn_wave_barrier();
  // rows: lanes 0..26 take I = 0 and then I = 2, lanes 27..53 take I = 1
  if (lane < 2 * ND) {
    const int a = lane < ND ? lane : lane - ND, a0 = a % NB, a12 = a / NB;
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q0 = 0; q0 < NQ; ++q0) T[v][q0] = tab_ptr<P>(tab, 0, v)[a0 * NQ + q0];
    const int n_rows = lane < ND ? 2 : 1;
    for (int k = 0; k < n_rows; ++k) {
      const int I = lane < ND ? 2 * k : 1;
      double sr = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q0 = 0; q0 < NQ; ++q0) sr = __builtin_fma(T[m == 0 ? 1 : 0][q0], W[((I * 3 + m) * NB2 + a12) * NQ + q0], sr);
      p.scratch_r[(e * ND + a) * 3 + I] = sr;       // [element][a][i]: tensor_p2_kernel reads a node's three rows from one sector
    }
  }
  __builtin_amdgcn_wave_barrier();
}

// ------------------------------------------------------------------------------------------------
// wave X, two steps per element
// ------------------------------------------------------------------------------------------------
template<int KIND>
MH_DEV void wgsym_x_loop(const TensorArgs& p, double* lds, int col0, int n_cols, int& status) {
  using L = WgsymLds;
  constexpr int P = 2, NB = 3, NQ = 4, ND = 27, NQ3 = 64;
  constexpr int TROUNDS = 2;
  const int lane = threadIdx.x & 63;
  double* ue = lds + L::off_ue;
  // the workgroup walks n_cols units back to back, a unit = one segment of seg_len elements of an element column
  // (unit index = column * segments-per-column + segment): sequence index g = unit-in-workgroup * seg_len + position
  const int sl = p.seg_len, nseg = p.box_n[2] / sl;
  const int n_seq = n_cols * sl;
  // The element of sequence index g, its spans and its layer, kept by a cursor that moves one element at a time (round 5: the
  // point wave asked for them by division -- g / sl, % nseg, col % n0, col / n0 at every use: ~ 12 run-time integer divisions,
  // 300 - 400 scalar instructions per element; the contraction waves went incremental in round 4): the unit's column by
  // division when a unit begins, then one layer further per step.
  struct Cursor {
    int g, pos, unit, cx, cy, ez;
    int64_t e;
  };
  const int64_t e_step = (int64_t)p.box_n[0] * p.box_n[1];
  auto cursor_at_unit = [&](int g, int u) -> Cursor {
    Cursor c;
    c.g = g;
    c.pos = 0;
    c.unit = u;
    const int un = col0 + u, col = un / nseg;
    c.cx = col % p.box_n[0];
    c.cy = col / p.box_n[0];
    c.ez = (un % nseg) * sl;
    c.e = c.cx + (int64_t)p.box_n[0] * (c.cy + (int64_t)p.box_n[1] * c.ez);
    return c;
  };
  auto advance = [&](Cursor& c) {
    ++c.g;
    if (++c.pos == sl) {
      c = cursor_at_unit(c.g, c.unit + 1);
    } else {
      ++c.ez;
      c.e += e_step;
    }
  };
  auto table_src = [&](const Cursor& c, int t) -> const double* {
    const int dir = t / (2 * NB * NQ);
    const int rem = t % (2 * NB * NQ);
    const int isD = rem / (NB * NQ);
    const int k = rem % (NB * NQ);
    const int span = (dir == 0 ? p.box_begin[0] + c.cx : dir == 1 ? p.box_begin[1] + c.cy : p.box_begin[2] + c.ez);
    return (isD ? (dir == 0 ? p.tabD[0] : dir == 1 ? p.tabD[1] : p.tabD[2])
                : (dir == 0 ? p.tabB[0] : dir == 1 ? p.tabB[1] : p.tabB[2])) + (int64_t)span * NB * NQ + k;
  };
  Cursor cq = cursor_at_unit(0, 0);      // the element whose operands are requested next
  Cursor cs = cq;                        // the element of the next quadrature-point stage
  Cursor cr = cq;                        // the element whose rows are written next
  int32_t node_n = lane < ND ? p.dofs[cq.e * ND + lane] : 0;
  double ue_r[3], tab_r[TROUNDS], geo_r[10];
  // requests the operands of the element at cq (the cursor moves on), and the node ids of the one after it
  auto request = [&]() {
#pragma unroll
    for (int c = 0; c < 3; ++c) ue_r[c] = p.u[(int64_t)node_n * 3 + c];
#pragma unroll
    for (int rd = 0; rd < TROUNDS; ++rd) {
      const int t = rd * 64 + lane;
      tab_r[rd] = *table_src(cq, t < 6 * NB * NQ ? t : 0);
    }
    const double* g = p.geo + cq.e * 10 * NQ3 + lane;
#pragma unroll
    for (int k = 0; k < 10; ++k) geo_r[k] = g[(int64_t)k * NQ3];
    if (cq.g + 1 < n_seq) {
      advance(cq);
      node_n = lane < ND ? p.dofs[cq.e * ND + lane] : 0;
    }
  };
  WgsPoint<KIND> s;
  // quadrature-point stage of element es from the requested operands (tables -> LDS parity es & 1)
  auto point_stage = [&]() {
    const int es = cs.g;
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
    status |= wgs_x_point<KIND>(p, cs.e * NQ3 + lane, F, Ji, wd, s);
    __builtin_amdgcn_wave_barrier();
    if (cs.g + 1 < n_seq) advance(cs);
  };

  request();
  point_stage();
  if (1 < n_seq) request();
  // ---- prologue: rows of element 0 ---------------------------------------------------------------------------
  wgsym_x_rows<KIND>(p, lds, lane, cr.e, 0, s);
  wgs_barrier();
  for (int it = 0; it < n_seq; ++it) {
    // ---- D(it), free running: quadrature-point stage of element it + 1 (touches nothing the other waves read
    // before the next barrier: its own ue / point data and the table buffer of the OTHER parity) -----------------
    if (it + 1 < n_seq) {
      point_stage();
      if (it + 2 < n_seq) request();
    }
    // ---- O(it), lock step: rows of element it + 1 once every contraction wave holds its operands of element it --
    wgs_barrier();
    if (it + 1 < n_seq) {
      advance(cr);
      wgsym_x_rows<KIND>(p, lds, lane, cr.e, (it + 1) & 1, s);
    }
    wgs_barrier();
  }
}

// ------------------------------------------------------------------------------------------------
// wave Y_W: diagonal block (W, W) in step D, off-diagonal block (I1, J1) in step O; flushes piece W
// ------------------------------------------------------------------------------------------------
template<int W>
MH_DEV void wgsym_y_loop(const TensorArgs& p, double* lds, int col0, int n_cols) {
  using L = WgsymLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81, NK = ND * NROW;
  // step O blocks: Y0 (1,0), Y1 (2,0), Y2 (2,1)
  constexpr int I1 = W == 0 ? 1 : 2, J1 = W == 2 ? 1 : 0;
  const WgsLane lc = wgs_lane_constants(true);
  const int lane = lc.lane;
  const int sl = p.seg_len, nseg = p.box_n[2] / sl;
  const int n_seq = n_cols * sl;   // sequence index g = unit-in-workgroup * seg_len + position (see wgsym_x_loop)
  const double* AH0 = lds + L::off_ah + L::ah_block(W, W) * NQ3;
  const double* AH1 = lds + L::off_ah + L::ah_block(I1, J1) * NQ3;
  auto st_of = [&](int piece) -> double* { return lds + L::off_st + piece * WgsLds::st_size; };
  // the element of sequence index g, kept incrementally (round 4: four integer divisions by run-time numbers per element
  // and wave were ~80 scalar instructions of the loop): first element of a unit by division, then + one layer per step
  auto unit_first = [&](int u) -> int64_t {
    const int unit = col0 + u, col = unit / nseg;
    return col % p.box_n[0] + (int64_t)p.box_n[0] * (col / p.box_n[0] + (int64_t)p.box_n[1] * ((unit % nseg) * sl));
  };
  const int64_t e_step = (int64_t)p.box_n[0] * p.box_n[1];
  auto block_at = [&](int64_t e) -> double* { return p.scratch_k + e * (int64_t)P2Block::size; };
  int pos = 0, unit_i = 0;        // position inside the unit (= it % seg_len), unit inside the workgroup
  int64_t e_cur = unit_first(0), e_prev = e_cur;
  const int mrow = lane & 15, mk = lane >> 4;
  const bool mrow_ok = mrow < NB2;
  const int mra = mrow_ok ? mrow / NB : 0, mrb = mrow_ok ? mrow % NB : 0;

  double C0[NB2], C1[NB2];  // packed carries of the two blocks
#pragma unroll
  for (int k = 0; k < NB2; ++k) C0[k] = C1[k] = 0.0;
  double aS0[4], aS2[4];
  double uB1[NB][NQ], uD1[NB][NQ];

  // ---- prologue: X writes the rows of element 0 ---------------------------------------------------------------
  wgs_barrier();
  for (int it = 0; it < n_seq; ++it) {
    // ---- D(it), free running: flush element it - 1, tables of element it, diagonal block ----------------------
    {
      if (it >= 1) wgs_flush_final(lane, st_of(W), block_at(e_prev), W);
      const double* tab = lds + L::off_tab + (it & 1) * 6 * NB * NQ;
      // the tables of directions 0 and 1 belong to the element COLUMN: read (24 LDS reads, 48 v_readfirstlane, the pair
      // products of direction 0) at the first element of a unit only -- every vector instruction of a contraction wave
      // is on the kernel's critical path; direction 2 changes with every element
      if (pos == 0) {
        const double Ba = tab_ptr<P>(tab, 0, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 0, 1)[mra * NQ + mk];
        const double Bb = tab_ptr<P>(tab, 0, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 0, 1)[mrb * NQ + mk];
        aS0[0] = mrow_ok ? Ba * Bb : 0.0;
        aS0[1] = mrow_ok ? Da * Bb : 0.0;
        aS0[2] = mrow_ok ? Ba * Db : 0.0;
        aS0[3] = mrow_ok ? Da * Db : 0.0;
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
      {
        const double Ba = tab_ptr<P>(tab, 2, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, 2, 1)[mra * NQ + mk];
        const double Bb = tab_ptr<P>(tab, 2, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, 2, 1)[mrb * NQ + mk];
        aS2[0] = mrow_ok ? Ba * Bb : 0.0;
        aS2[1] = mrow_ok ? Da * Bb : 0.0;
        aS2[2] = mrow_ok ? Ba * Db : 0.0;
        aS2[3] = mrow_ok ? Da * Db : 0.0;
      }
      double ah[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) ah[k] = AH0[k * NQ3 + lane];
      // no barriers around the diagonal block: it reads operands nobody rewrites before O(it)'s first barrier and
      // writes only slots of this wave's own buffer that no other wave touches (the transposed entries other
      // waves add to it during O steps use the other two column components)
      wgs_contract_block<WGSYM_DIAG_MODE, true>(lc, ah, aS0, aS2, uB1, uD1, C0, st_of(W), W, st_of(W), W, lds + L::off_dump);
    }
    // ---- O(it), lock step: off-diagonal block, stored as computed and transposed ------------------------------
    {
      double ah[9];
#pragma unroll
      for (int k = 0; k < 9; ++k) ah[k] = AH1[k * NQ3 + lane];
      wgs_barrier();
      wgs_contract_block<1, true>(lc, ah, aS0, aS2, uB1, uD1, C1, st_of(I1), J1, st_of(J1), I1, lds + L::off_dump);
      wgs_barrier();
      if (pos == sl - 1) {
        // last element of a unit (column, or column segment): the carried rows have no successor -- straight from the registers into the third
        // part of the pieces (no LDS, no lock step) -- and the next column starts with an empty carry
        double* E = block_at(e_cur);
        wgs_stage_carry<WGSYM_DIAG_MODE>(lc, C0, P2Block::carry_of(E, W), W, P2Block::carry_of(E, W), W);
        wgs_stage_carry<1>(lc, C1, P2Block::carry_of(E, I1), J1, P2Block::carry_of(E, J1), I1);
#pragma unroll
        for (int k = 0; k < NB2; ++k) C0[k] = C1[k] = 0.0;
      }
    }
    e_prev = e_cur;
    if (++pos == sl) {
      pos = 0;
      ++unit_i;
      if (unit_i < n_cols) e_cur = unit_first(unit_i);
    } else {
      e_cur += e_step;
    }
  }
  // ---- after the last element (no more lock steps): its pieces from the buffers ------------------------------------
  wgs_flush_final(lane, st_of(W), block_at(e_prev), W);
}

template<int KIND>
__global__ __launch_bounds__(256, 2) void tensor_wgsym_kernel(TensorArgs p) {
  extern __shared__ __align__(16) double smem_wgsym[];
  const int role = __builtin_amdgcn_readfirstlane((int)((threadIdx.x >> 6) + WGS_ROT(blockIdx.x)) & 3);
  const int n_cols_all = p.box_n[0] * p.box_n[1] * (p.box_n[2] / p.seg_len);   // units
  const int col0 = blockIdx.x * p.cols_per_wg;
  const int n_cols = n_cols_all - col0 < p.cols_per_wg ? n_cols_all - col0 : p.cols_per_wg;
  if (role == 0) {
    int status = 0;
    wgsym_x_loop<KIND>(p, smem_wgsym, col0, n_cols, status);
    if (status) atomicOr(p.status, status);
  } else {
    if (role == 1) wgsym_y_loop<0>(p, smem_wgsym, col0, n_cols);
    else if (role == 2) wgsym_y_loop<1>(p, smem_wgsym, col0, n_cols);
    else wgsym_y_loop<2>(p, smem_wgsym, col0, n_cols);
  }
}

inline void launch_tensor_wgsym(mimi_hip_domain_s* h, TensorArgs a) {
  h->scratch_k.resize((size_t)h->n_el * P2Block::size);
  h->scratch_r.resize((size_t)h->n_el * 3 * 27);
  a.scratch_k = h->scratch_k.ptr;
  a.scratch_r = h->scratch_r.ptr;
  a.n_units_u = a.box_n[0];
  a.n_units_v = a.box_n[1];
  const size_t lds = WgsymLds::total * sizeof(double);
  auto kernel = tensor_wgsym_kernel<MIMI_HIP_MAT_NEOHOOKEAN>;
  ensure_dynamic_lds(reinterpret_cast<const void*>(kernel), (int)lds);
  // several columns per workgroup (the pipeline of the four waves then runs through the column boundaries: one
  // prologue per workgroup instead of one per column) as long as the grid still fills the chip several times over
  // few columns (the boundary layers of a multi-GPU slab): each column is cut into segments with their own workgroup, so
  // that the launch still fills the chip; a segment end stores its carried rows like a column end (phase 2 knows)
  int nseg = 1;
  while (a.box_n[0] * a.box_n[1] * nseg * 2 <= WGSYM_SPLIT_BELOW && a.box_n[2] % (nseg * 2) == 0 && a.box_n[2] / (nseg * 2) >= 4) nseg *= 2;
  a.seg_len = a.box_n[2] / nseg;
  const int n_cols_all = a.box_n[0] * a.box_n[1] * nseg;
  a.cols_per_wg = n_cols_all / WGSYM_MIN_WGS < 1 ? 1 : (n_cols_all / WGSYM_MIN_WGS > WGSYM_MAX_COLS ? WGSYM_MAX_COLS : n_cols_all / WGSYM_MIN_WGS);
  h->phase_has_prepass = false;
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[0], h->stream));
  if (h->phase_select != 2) {
    hipLaunchKernelGGL(kernel, dim3((n_cols_all + a.cols_per_wg - 1) / a.cols_per_wg), dim3(256), lds, h->stream, a);
    MH_HIP(hipGetLastError());
  }
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[1], h->stream));
  if (h->phase_select != 1) launch_tensor_p2(h, a);
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[2], h->stream));
}

}  // namespace mimi_hip
