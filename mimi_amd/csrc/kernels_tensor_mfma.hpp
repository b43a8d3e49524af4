// Tensor-product tangent kernel, p = 2, with the three sum-factorisation contractions on the
// fp64 matrix pipe (v_mfma_f64_16x16x4_f64) and every intermediate in registers.
//
// Same mathematics, decomposition ((unit, i) waves walking an element column, (p+1)^2 colour
// launches, plain read-modify-write) and stage A / stage R as kernels_tensor.hpp.  What changes
// is stage C, K_ai,bj = sum_q sum_mn dN_a/dxi_m Ahat_im,jn dN_b/dxi_n:
//
//   S1  X^mn[a2b2][q0q1]   = sum_q2 T2^m[a2][q2] T2^n[b2][q2] Ahat^mn[q0q1q2]
//       one MFMA per (j, mn): A = the pair table (row = a2b2 < 9 of 16, k = q2), B = Ahat itself:
//       lane l = q0 + 4 q1 + 16 q2 of the constitutive stage IS the B layout (k = l>>4, col = l&15).
//   T1  lane shuffle (ds_bpermute) of X so that q1 becomes the k index: col = (q0, a2b2 & 3)
//   S2  Z_g[a1b1][(q0, a2b2)] += sum_q1 T1^m[a1][q1] T1^n[b1][q1] X^mn        (g = (m==0) + 2 (n==0))
//   T2  lane shuffle so that q0 becomes the k index: col = (a2b2 & 3, a1b1 & 3)
//   S3  K[a0b0][(a2b2, a1b1)] = sum_g sum_q0 T0^g[a0b0][q0] Z_g
// 72 MFMAs and 126 bpermutes per 27x27 block (i, j) instead of ~1000 vector FMAs and ~25 KB of
// LDS staging.  (fp64 MFMA is exactly an fp64 FMA chain in k order: no precision change.)
//
// The block results are added into a wave-private LDS tile KS[a][(b2 b1)(b0 j)] = the 27 CSR
// row pieces of the element for component i.  KS is persistent along the column walk: entries
// shared with the next element (a2 >= 1 and b2 >= 1) are moved to their (a2-1, b2-1) slots instead
// of going through memory; all other entries are flushed with coalesced read-modify-writes
// (9 consecutive lanes = the 9 contiguous values (b0, j) of a CSR row segment).
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor.hpp"

namespace mimi_hip {

typedef double mh_d4 __attribute__((ext_vector_type(4)));

struct MfmaLds {
  static constexpr int NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81;
  static constexpr int off_ue = 0;                      // [3][27]
  static constexpr int off_tab = off_ue + 3 * ND;       // [3 dir][2][3][4]
  static constexpr int off_rs = off_tab + 6 * NB * NQ;  // [27] int64
  static constexpr int off_r = off_rs + ND + 1;         // stage-R scratch
  static constexpr int r_size = 3 * NQ3 + 3 * NB * NQ * NQ + 3 * NB2 * NQ;
  static constexpr int off_ks = off_r + r_size + (r_size & 1);  // [27][81]
  static constexpr int off_doff = off_ks + ND * NROW + 1;       // [35][64] uint32: CSR value index per KS slot
  static constexpr int total = off_doff + 35 * 32;
};

MH_DEV double bperm_f64(int byte_index, double v) {
  const unsigned long long u = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_ds_bpermute(byte_index, (int)(u & 0xffffffffull));
  const int hi = __builtin_amdgcn_ds_bpermute(byte_index, (int)(u >> 32));
  return __longlong_as_double(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}

template<int I>
MH_DEV void tensor_mfma_body(const TensorArgs& p, double* lds, int eu, int ev, int& status) {
  using L = MfmaLds;
  constexpr int P = 2, NB = 3, NQ = 4, NB2 = 9, ND = 27, NQ3 = 64, NROW = 81;
  constexpr int TROUNDS = 2;  // 72 table values
  const int lane = threadIdx.x & 63;
  double* ue = lds + L::off_ue;
  double* tab = lds + L::off_tab;
  int64_t* rs = reinterpret_cast<int64_t*>(lds + L::off_rs);
  double* RS = lds + L::off_r;
  double* KS = lds + L::off_ks;
  unsigned* DOFF = reinterpret_cast<unsigned*>(lds + L::off_doff);
  // flush lanes: 7 CSR row segments of 9 contiguous values (b0, j) per round; lane 63 idles
  const int fl_s7 = lane / 9, fl_e9 = lane % 9, fl_b0 = fl_e9 / 3, fl_j = fl_e9 % 3;
  const bool use_carry = p.seq_axis == 2;
#ifdef MH_PROFILE
  unsigned long long prof_acc[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long prof_last;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(prof_last)::"memory");
#endif

  const int n_seq = p.seq_axis == 0 ? p.box_n[0] : (p.seq_axis == 1 ? p.box_n[1] : p.box_n[2]);
  auto element_of = [&](int es, int* el) -> int64_t {
#pragma unroll
    for (int d = 0; d < 3; ++d) el[d] = (d == p.seq_axis) ? es : (d == p.u_axis ? eu : ev);
    return el[0] + (int64_t)p.box_n[0] * (el[1] + (int64_t)p.box_n[1] * el[2]);
  };
  auto table_src = [&](const int* el, int t) -> const double* {
    const int dir = t / (2 * NB * NQ);
    const int rem = t % (2 * NB * NQ);
    const int isD = rem / (NB * NQ);
    const int k = rem % (NB * NQ);
    const int span = (dir == 0 ? p.box_begin[0] + el[0] : dir == 1 ? p.box_begin[1] + el[1] : p.box_begin[2] + el[2]);
    return (isD ? (dir == 0 ? p.tabD[0] : dir == 1 ? p.tabD[1] : p.tabD[2])
                : (dir == 0 ? p.tabB[0] : dir == 1 ? p.tabB[1] : p.tabB[2])) + (int64_t)span * NB * NQ + k;
  };

  // lane constants of the matrix stages
  const int mrow = lane & 15, mk = lane >> 4;
  const bool mrow_ok = mrow < NB2;
  const int mra = mrow_ok ? mrow / NB : 0, mrb = mrow_ok ? mrow % NB : 0;
  const int idx1 = 4 * ((lane & 3) + 4 * (lane >> 4) + 16 * ((lane >> 2) & 3));
  const int idx2 = 4 * ((lane >> 4) + 4 * (lane & 3) + 16 * ((lane >> 2) & 3));

  for (int k = lane; k < ND * NROW; k += 64) KS[k] = 0.0;

  // ---- pipeline prologue ------------------------------------------------------------------------
  int el_c[3], el_n[3];
  int64_t e_cur = element_of(0, el_c);
  int32_t node_c = lane < ND ? p.dofs[e_cur * ND + lane] : 0, node_n = 0;
  double ue_r[3];
  int64_t rs_r;
  double tab_r[TROUNDS];
  double geo_r[10];
#pragma unroll
  for (int c = 0; c < 3; ++c) ue_r[c] = p.u[(int64_t)node_c * 3 + c];
  rs_r = p.rowptr[(int64_t)node_c * 3 + I];
#pragma unroll
  for (int rd = 0; rd < TROUNDS; ++rd) {
    const int t = rd * 64 + lane;
    tab_r[rd] = *table_src(el_c, t < 6 * NB * NQ ? t : 0);
  }
  {
    const double* g = p.geo + e_cur * 10 * NQ3 + lane;
#pragma unroll
    for (int k = 0; k < 10; ++k) geo_r[k] = g[(int64_t)k * NQ3];
  }
  if (n_seq > 1) {
    const int64_t e1 = element_of(1, el_n);
    node_n = lane < ND ? p.dofs[e1 * ND + lane] : 0;
  }

  for (int es = 0; es < n_seq; ++es) {
    MH_STAMP(0);
    // ---- stage 0: registers -> LDS, then issue the loads of the NEXT element -------------------
    if (lane < ND) {
#pragma unroll
      for (int c = 0; c < 3; ++c) ue[c * ND + lane] = ue_r[c];
      rs[lane] = rs_r;
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
    const int32_t node_w = node_c;
    const int64_t e = e_cur;
    int el_w[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) el_w[d] = el_c[d];
    if (es + 1 < n_seq) {
      e_cur = element_of(es + 1, el_c);
      node_c = node_n;
#pragma unroll
      for (int c = 0; c < 3; ++c) ue_r[c] = p.u[(int64_t)node_c * 3 + c];
      rs_r = p.rowptr[(int64_t)node_c * 3 + I];
#pragma unroll
      for (int rd = 0; rd < TROUNDS; ++rd) {
        const int t = rd * 64 + lane;
        tab_r[rd] = *table_src(el_c, t < 6 * NB * NQ ? t : 0);
      }
      const double* g = p.geo + e_cur * 10 * NQ3 + lane;
#pragma unroll
      for (int k = 0; k < 10; ++k) geo_r[k] = g[(int64_t)k * NQ3];
      if (es + 2 < n_seq) {
        const int64_t e2 = element_of(es + 2, el_n);
        node_n = lane < ND ? p.dofs[e2 * ND + lane] : 0;
      }
    }
    __builtin_amdgcn_wave_barrier();

    // ---- issue the loads of the CSR entries this element will flush; they are consumed after the
    // matrix stage.  KS slot k = 63 c + lane <-> (segment 7 c + lane/9 = (a, b1 b2), (b0 j) = lane % 9);
    // offsets fit 32 bits (nnz < 2^32).
    constexpr int NK = ND * NROW;   // 2187 slots
    constexpr int NSEG = ND * 9;    // 243 row segments
    constexpr int NROUND = (NSEG + 6) / 7;  // 35
    const bool last = es + 1 >= n_seq;
    double old[NROUND];
    {
      int f0 = 0, f1 = 0, f2 = 0;
      if (p.structured) {
        f0 = p.first[0][p.box_begin[0] + el_w[0]];
        f1 = p.first[1][p.box_begin[1] + el_w[1]];
        f2 = p.first[2][p.box_begin[2] + el_w[2]];
      }
      const int32_t* pp = p.pair_pos + e * ND * ND;
#pragma unroll 1
      for (int c = 0; c < NROUND; ++c) {
        const int sg = c * 7 + fl_s7;
        const bool act = fl_s7 < 7 && sg < NSEG;
        const int sgg = act ? sg : 0;
        const int a = sgg / 9, seg = sgg % 9;
        const int b1 = seg % NB, b2 = seg / NB;
        const int a0 = a % NB, a1 = (a / NB) % NB, a2 = a / NB2;
        const bool carried = use_carry && !last && a2 >= 1 && b2 >= 1;
        int off;
        if (p.structured) {
          const int A0 = f0 + a0, A1 = f1 + a1, A2 = f2 + a2;
          const int lo0 = A0 - P < 0 ? 0 : A0 - P, lo1 = A1 - P < 0 ? 0 : A1 - P, lo2 = A2 - P < 0 ? 0 : A2 - P;
          const int hi0 = A0 + P > p.n_ctrl[0] - 1 ? p.n_ctrl[0] - 1 : A0 + P;
          const int hi1 = A1 + P > p.n_ctrl[1] - 1 ? p.n_ctrl[1] - 1 : A1 + P;
          const int w0 = hi0 - lo0 + 1, w1 = hi1 - lo1 + 1;
          off = 3 * ((f0 + fl_b0 - lo0) + w0 * ((f1 + b1 - lo1) + w1 * (f2 + b2 - lo2))) + fl_j;
        } else {
          off = pp[a * ND + (fl_b0 + NB * (b1 + NB * b2))] + fl_j;
        }
        // 0xffffffff marks "nothing to flush" (idle lane or carried entry)
        DOFF[c * 64 + lane] = (act && !carried) ? (unsigned)(rs[a] + off) : 0xffffffffu;
      }
#pragma unroll
      for (int c = 0; c < NROUND; ++c) {
        const unsigned d = DOFF[c * 64 + lane];
        old[c] = d != 0xffffffffu ? p.A[d] : 0.0;
      }
    }

    MH_STAMP(1);
    // ---- stage A: constitutive update, lane = quadrature point q = q0 + 4 q1 + 16 q2 -------------
    double Ahat[27];  // [(m*3 + j)*3 + n] for row I
    double Phat[3];
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
        }
      double F[9];
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int J = 0; J < 3; ++J) {
          double sf = (i == J) ? 1.0 : 0.0;
#pragma unroll
          for (int m = 0; m < 3; ++m) sf += H[i * 3 + m] * Ji[m * 3 + J];
          F[i + J * 3] = sf;
        }
      PointResult<3> w;
      status |= evaluate_pk1<3>(p.mat, p.dt, p.state, e * NQ3 + lane, F, w);
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double sp = 0.0;
#pragma unroll
        for (int J = 0; J < 3; ++J) sp += w.P[I + J * 3] * Ji[m * 3 + J];
        Phat[m] = wd * sp;
      }
      double A[27];
      tangent_row_of<3, I>(p.mat.m, w, A);
      double T[27];
#pragma unroll
      for (int J = 0; J < 3; ++J)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int n = 0; n < 3; ++n) {
            double st = 0.0;
#pragma unroll
            for (int Lx = 0; Lx < 3; ++Lx) st += A[(J * 3 + j) * 3 + Lx] * Ji[n * 3 + Lx];
            T[(J * 3 + j) * 3 + n] = st;
          }
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int n = 0; n < 3; ++n) {
            double sa = 0.0;
#pragma unroll
            for (int J = 0; J < 3; ++J) sa += Ji[m * 3 + J] * T[(J * 3 + j) * 3 + n];
            Ahat[(m * 3 + j) * 3 + n] = wd * sa;
          }
    }

    MH_STAMP(2);
    // ---- stage R: residual row I by sum factorisation ---------------------------------------------
    {
      double* PH = RS;                   // [3 m][64]
      double* V = PH + 3 * NQ3;          // [3 m][3 a2][16]
      double* W = V + 3 * NB * NQ * NQ;  // [3 m][9 a1a2][4]
#pragma unroll
      for (int m = 0; m < 3; ++m) PH[m * NQ3 + lane] = Phat[m];
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
        double* dst = p.r + (int64_t)node_w * 3 + I;
        *dst += sr;
      }
    }

    MH_STAMP(3);
    // ---- stage C on the matrix pipe ----------------------------------------------------------------
    // A operands: pair tables of the three directions, lane = (row = pair index, k = quadrature index)
    double aS[3][4];  // [dir][variant]: 0 B.B, 1 D(a).B(b), 2 B(a).D(b), 3 D.D
#pragma unroll
    for (int dir = 0; dir < 3; ++dir) {
      const double Ba = tab_ptr<P>(tab, dir, 0)[mra * NQ + mk], Da = tab_ptr<P>(tab, dir, 1)[mra * NQ + mk];
      const double Bb = tab_ptr<P>(tab, dir, 0)[mrb * NQ + mk], Db = tab_ptr<P>(tab, dir, 1)[mrb * NQ + mk];
      aS[dir][0] = mrow_ok ? Ba * Bb : 0.0;
      aS[dir][1] = mrow_ok ? Da * Bb : 0.0;
      aS[dir][2] = mrow_ok ? Ba * Db : 0.0;
      aS[dir][3] = mrow_ok ? Da * Db : 0.0;
    }
    const mh_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      mh_d4 Z[4][3];  // [g][r]: rows a1b1 = (lane>>4) + 4 r2, cols (q0, a2b2 & 3), a2b2 = (a2b2 & 3) + 4 r
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int r = 0; r < 3; ++r) Z[g][r] = zero4;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const int v2 = (m == 2 ? 1 : 0) + (n == 2 ? 2 : 0);
          const int v1 = (m == 1 ? 1 : 0) + (n == 1 ? 2 : 0);
          const int g = (m == 0 ? 1 : 0) + (n == 0 ? 2 : 0);
          const mh_d4 X = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[2][v2], Ahat[(m * 3 + j) * 3 + n], zero4, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 3; ++r) {
            const double xt = bperm_f64(idx1, X[r]);
            Z[g][r] = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[1][v1], xt, Z[g][r], 0, 0, 0);
          }
        }
      // S3 per (r, r2) tile, then add the tile into KS
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int r2 = 0; r2 < 3; ++r2) {
          mh_d4 K = zero4;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const double zt = bperm_f64(idx2, Z[g][r][r2]);
            K = __builtin_amdgcn_mfma_f64_16x16x4f64(aS[0][g], zt, K, 0, 0, 0);
          }
          const int ab2 = (lane & 3) + 4 * r, a1b1 = ((lane >> 2) & 3) + 4 * r2;
          if (ab2 < NB2 && a1b1 < NB2) {
            const int a2 = ab2 / NB, b2 = ab2 % NB, a1 = a1b1 / NB, b1 = a1b1 % NB;
#pragma unroll
            for (int r3 = 0; r3 < 3; ++r3) {
              const int a0b0 = (lane >> 4) + 4 * r3;
              if (a0b0 < NB2) {
                const int a0 = a0b0 / NB, b0 = a0b0 % NB;
                const int idx = (a0 + NB * (a1 + NB * a2)) * NROW + (b2 * NB + b1) * 9 + b0 * 3 + j;
                KS[idx] += K[r3];
              }
            }
          }
        }
    }
    __builtin_amdgcn_wave_barrier();

    MH_STAMP(5);
    // ---- flush: coalesced CSR read-modify-write, entries shared with the next element stay in KS ----
    if (lane < 63) {
#pragma unroll
      for (int c = 0; c < NROUND; ++c) {
        const int k = c * 63 + lane;
        if (k < NK) {
          const unsigned d = DOFF[c * 64 + lane];
          const double v = KS[k];
          KS[k] = 0.0;
          if (d != 0xffffffffu) {
            p.A[d] = old[c] + p.grad_factor * v;
          } else if (use_carry && !last) {
            // (a2, b2) -> (a2-1, b2-1): 9*81 + 27 slots down, always a slot of an earlier round
            KS[k - (NB2 * NROW + NB * 9)] = v;
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    MH_STAMP(9);
  }
#ifdef MH_PROFILE
  if (lane == 0 && p.prof)
    for (int k = 0; k < 12; ++k) atomicAdd(&p.prof[k], prof_acc[k]);
#endif
}

__global__ __launch_bounds__(64) void tensor_mfma_kernel(TensorArgs p) {
  extern __shared__ __align__(16) double smem_m[];
  const int comp = blockIdx.x % 3, unit = blockIdx.x / 3;
  const int ku = unit % p.n_units_u, kv = unit / p.n_units_u;
  const int eu = p.colour_u + 3 * ku, ev = p.colour_v + 3 * kv;
  int status = 0;
  if (comp == 0) tensor_mfma_body<0>(p, smem_m, eu, ev, status);
  else if (comp == 1) tensor_mfma_body<1>(p, smem_m, eu, ev, status);
  else tensor_mfma_body<2>(p, smem_m, eu, ev, status);
  if (status) atomicOr(p.status, status);
}

inline void launch_tensor_mfma(mimi_hip_domain_s* h, int grad, TensorArgs a) {
  (void)grad;
  const size_t lds = MfmaLds::total * sizeof(double);
  const int nu = a.box_n[a.u_axis], nv = a.box_n[a.v_axis];
  for (int cv = 0; cv < 3; ++cv)
    for (int cu = 0; cu < 3; ++cu) {
      a.colour_u = cu;
      a.colour_v = cv;
      a.n_units_u = cu < nu ? (nu - cu + 2) / 3 : 0;
      a.n_units_v = cv < nv ? (nv - cv + 2) / 3 : 0;
      const int n_units = a.n_units_u * a.n_units_v;
      if (n_units == 0) continue;
      hipLaunchKernelGGL(tensor_mfma_kernel, dim3(n_units * 3), dim3(64), lds, h->stream, a);
      MH_HIP(hipGetLastError());
    }
}

}  // namespace mimi_hip
