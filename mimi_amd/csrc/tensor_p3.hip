// Sum-factorised two-phase tangent assembly for structured 3-D patches of degree p = 3 (64 nodes, 5 x 5 x 5 Gauss
// points per element): BASELINE configuration 3 (128 x 128 x 16, J2).  Replaces, for such patches, the reference's
// per-element loop integrators/nonlinear_solid.hpp:65-87 + nonlinear_solid.cpp:48-149 (see kernels_tensor.hpp for the
// quantities; same definitions, n_dof = 64, n_q = 125).
//
//   pre-pass   tp3_point_kernel     one lane per quadrature point: F, material, and -- pulled back to the reference
//                                   element and weighted -- Ahat_i[m][j][n], Phat_i[m] -> record [element][90][128];
//                                   the element residual pieces by sum factorisation -> scratch_r[element][64][i].
//   phase 1    tp3_contract_kernel  one WAVE per (element, i), no LDS, no barriers: per column component j the block
//                                   K[(a, i), (b, j)] = sum_q sum_mn dN_a/dxi_m Ahat_i[m][j][n] dN_b/dxi_n
//                                   one parametric direction at a time,
//                                     S1 (matrix pipe, contracts q2)  D1[mn][(q0 q1), (a2 b2)]
//                                     S2 (vector pipe, contracts q1)  E_g[(a1 b1)][q0][(a2 b2)], the nine (m, n) merged
//                                                                     into the four table variants g of direction 0
//                                     S3 (matrix pipe, contracts q0)  K[(a2 b2), (a1 b1), (a0 b0)]
//                                   with v_mfma_f64_16x16x4: the 16 node pairs (a_d, b_d) of a direction ARE the 16
//                                   rows / columns of the instruction.  Five points per direction = one full k-step
//                                   (or register group) of four plus one odd plane, which rides in the second tile.
//                                   The block leaves the wave as 128-byte runs: piece (element, i) = 64 rows a of
//                                   192 values [j][b1][b2][b0].
//   phase 2    tp3_gather_kernel    one wave per CSR row (node A, i): walks the <= 64 elements containing A, reads
//                                   row a = local index of A of each piece (1536 contiguous bytes), adds it into an
//                                   LDS image of the row, then ONE coalesced A[row] += grad_factor * image.  The
//                                   residual row is a fixed-shape tree sum over the 64 element pieces.
// Nothing is atomic: results are bitwise reproducible.
#include "tensor_p3.hpp"

#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

namespace mimi_hip {

namespace {

typedef double t3_d4 __attribute__((ext_vector_type(4)));
typedef double t3_d2 __attribute__((ext_vector_type(2)));

// the record is written once by the pre-pass and read once by the contraction, 21.6 GB at BASELINE configuration 3: stored
// with the non-temporal hint (pre-pass 7.48 -> 7.14 ms, 7.6 -> 7.3 ms on a second box; the same hint on the contraction's
// loads changes nothing, scratch/p3_variants.sh A/B in round 4)
#define T3_REC_STORE(ptr, v) __builtin_nontemporal_store((v), (ptr))
// (the same hint on the contraction's piece stores -- 128-byte runs -- costs 5 ms, 19.5 -> 24.9 ms, and on the gather's piece
// loads nothing: profiles/r04_cfg3_nt_variants.txt)

constexpr int T3_NB = 4, T3_NQ = 5, T3_ND = 64, T3_NPT = 125, T3_PS = 128, T3_NROW = 192;
constexpr int T3_REC = 90;                        // fields of 128 points per element: 81 used (Ahat), 9 spare
// Record layout (round 5): per (i, j) nine fields mn = 3 m + n, stored as four PAIR fields [point][mn = 2 q, 2 q + 1] and one
// single field [point] (mn = 8) -- a lane's two values of a pair are 16 adjacent bytes: the pre-pass writes a pair with
// one 16-byte store per lane (45 fully coalesced store instructions per element instead of 81), a contraction wave reads
// it with one 16-byte load (15 operand loads per block instead of 27; a global-memory instruction costs a wave of
// this kernel 50 - 80 cycles whatever its width: profiles/r05_cfg3_contract_ablations_v2.txt).
MH_DEV constexpr int t3_rec_block(int i, int j) { return (i * 3 + j) * 9 * T3_PS; }        // doubles, + pair q 2 PS + 2 point (+ 1) | 8 PS + point
constexpr int T3_PIECE = 16 * 192 + 48 * 48;       // doubles per (element, i): rows a2 = 0, then rows a2 >= 1 at b2 = 0
constexpr int T3_TAIL = 48 * 144;                  // per (column, i): rows a2 >= 1 at b2 >= 1 of the column's last element

// ------------------------------------------------------------------------------------------------
// pre-pass
// ------------------------------------------------------------------------------------------------
// Closed forms of the pulled-back, weighted tangent Ahat_i[m][j][n] = wd sum_JL Jinv[m][J] dP_iJ/dF_jL Jinv[n][L] for the
// two materials with a closed-form dP/dF (materials.hpp tangent_of; the same expressions as the p = 2 point wave,
// kernels_tensor_wgs.hpp), written straight into the record (layout: t3_rec_block).
//   with G = Jinv F^-1 (G[m][i] = sum_J Jinv[m][J] Finv[J][i]):
//   neo-Hookean   wd (mu d_ij M[m][n] - c1 G[n][i] G[m][j] + c2 G[n][j] G[m][i]),  M = Jinv Jinv^T
//   J2            wd J (G[n][j] S_i[m] - G[m][j] S_i[n] + (K - beta 2G/3) G[m][i] Jinv[n][j]
//                       + beta G (d_ij N[m][n] + G[m][j] Jinv[n][i]) - 2G gamma Ts_i[m] Q[n][j]),
//                 N = G Jinv^T, Q[n][j] = sum_L s_jL Jinv[n][L], S_i[m] = sum_k sigma_ik G[m][k], Ts_i[m] = sum_k s_ik G[m][k]
// the nine values mn of one (i, j) at point `pt`; `blk` = the (i, j) block of the element's record
MH_DEV void t3_store_block_at(double* blk, int pt, const double (&v)[9]) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    t3_d2 two = {v[2 * q], v[2 * q + 1]};
    T3_REC_STORE(reinterpret_cast<t3_d2*>(blk + q * 2 * T3_PS + 2 * pt), two);
  }
  T3_REC_STORE(blk + 8 * T3_PS + pt, v[8]);
}
MH_DEV void t3_closed_form_record(const mimi_hip_material& mm, const PointResult<3>& w, const double* Ji, double wd, double* rec, int pt) {
  constexpr int PS = T3_PS;
  double G[9];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      double g = 0.0;
#pragma unroll
      for (int Jx = 0; Jx < 3; ++Jx) g += Ji[m * 3 + Jx] * w.Finv[Jx + c * 3];
      G[m * 3 + c] = g;
    }
  if (mm.kind == MIMI_HIP_MAT_NEOHOOKEAN) {
    const double J = w.detF;
    const double mu_w = wd * mm.mu, c1_w = wd * (mm.lambda * J * (J - 1.) - mm.mu), c2_w = wd * (mm.lambda * (2. * J - 1.) * J);
    double M[9];
#pragma unroll
    for (int m = 0; m < 3; ++m)
#pragma unroll
      for (int n = 0; n < 3; ++n) {
        double v = 0.0;
#pragma unroll
        for (int Jx = 0; Jx < 3; ++Jx) v += Ji[m * 3 + Jx] * Ji[n * 3 + Jx];
        M[m * 3 + n] = v;
      }
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        double blk[9];
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
          for (int n = 0; n < 3; ++n) {
            double v = c2_w * G[m * 3 + i] * G[n * 3 + j] - c1_w * G[m * 3 + j] * G[n * 3 + i];
            if (i == j) v += mu_w * M[m * 3 + n];
            blk[m * 3 + n] = v;
          }
        t3_store_block_at(rec + t3_rec_block(i, j), pt, blk);
      }
    return;
  }
  double beta = 1.0, gamma = 0.0;
  if (w.plastic) {
    const double q = w.q, Gm = mm.G;
    beta = 1.0 - 3.0 * Gm * w.delta / q;
    gamma = 3.0 * Gm * (1.5 / q) * (1.0 / ((3.0 * Gm + w.hprime) * q) - w.delta / (q * q));
  }
  const double G2 = 2.0 * mm.G;
  const double wdJ = wd * w.detF, Kc = mm.K - beta * G2 / 3.0, hb = 0.5 * beta * G2, gg = G2 * gamma;
  double N[9], Q[9];
#pragma unroll
  for (int m = 0; m < 3; ++m)
#pragma unroll
    for (int n = 0; n < 3; ++n) {
      double v = 0.0, t = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        v += G[m * 3 + k] * Ji[n * 3 + k];
        t += w.s_trial[n + k * 3] * Ji[m * 3 + k];   // Q[m][n] = sum_L s_nL Jinv[m][L]
      }
      N[m * 3 + n] = v;
      Q[m * 3 + n] = t;
    }
  // every entry as five fused multiply-adds (six on the diagonal blocks) of factors scaled once: w det J into S and the
  // volumetric / deviatoric / plastic coefficients into G, N and Ts
  const double hbw = wdJ * hb, Kw = wdJ * Kc, ggw = wdJ * gg;
  double hG[9], hN[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    hG[k] = hbw * G[k];
    hN[k] = hbw * N[k];
  }
  // (S and Ts of all three i first: the stress and the deviator are dead before the 81 entries are formed)
  double S[9], Ts[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int m = 0; m < 3; ++m) {
      double a = 0.0, b = 0.0;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        a += w.sigma[i + k * 3] * G[m * 3 + k];
        b += w.s_trial[i + k * 3] * G[m * 3 + k];
      }
      S[i * 3 + m] = wdJ * a;
      Ts[i * 3 + m] = ggw * b;
    }
#pragma unroll
  for (int k = 0; k < 9; ++k) asm volatile("" : "+v"(S[k]), "+v"(Ts[k]));
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    double KG[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) KG[m] = Kw * G[m * 3 + i];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      double blk[9];
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          double v = G[n * 3 + j] * S[i * 3 + m];
          v = __builtin_fma(-G[m * 3 + j], S[i * 3 + n], v);
          v = __builtin_fma(KG[m], Ji[n * 3 + j], v);
          v = __builtin_fma(hG[m * 3 + j], Ji[n * 3 + i], v);
          if (i == j) v += hN[m * 3 + n];
          v = __builtin_fma(-Ts[i * 3 + m], Q[n * 3 + j], v);
          blk[m * 3 + n] = v;
        }
      t3_store_block_at(rec + t3_rec_block(i, j), pt, blk);
    }
  }
}

// What the J2 return mapping does not need while it iterates, handed to LDS for its duration (materials.hpp j2_stress, Park):
// a 3 x 3 tensor (9) and the trial deviator (6: it is symmetric to the bit -- 0.5 (F_ij + F_ji) - ep_ij with ep built from
// it), [k][128] in the pool behind PH.  Round 4, the state commit (GRAD 2: plastic strain + deviator): 158 registers instead
// of 213, a third wave per SIMD, 5.67 -> 4.96 ms at BASELINE configuration 3 (the kernel is latency-bound: ONE wave per
// SIMD costs 1.75 x, measured); the residual-only mode parked still needed 179 registers -- two waves -- and lost 0.3 ms to
// the detour (profiles/r04_cfg3_prepass_variants.txt).  Round 5: 48 of those registers held the coefficients of
// pow_positive across the Newton loop; with them in scalar registers (scalar_coefficients, materials.hpp horner_step) and the
// geometry factors read again behind the solve the residual-only mode needs 143 -- three waves per SIMD, 6.42 -> 4.96 ms --
// and the commit 123 -- four, the LDS pool cut to 18.9 KB (W in PH's place): 4.68 -> 3.73 ms (profiles/r05_cfg3_horner_ab.txt).
// With the library's pow gone from the hardening laws (materials.hpp pow_any: its inlined copies had set the register count)
// the residual-only mode needs 123 -- four waves too, 4.33 -> 3.81 ms -- and the residual+Jacobian mode 177 parked: held to
// 168 (three waves; ten doubles go to scratch memory once per point) it runs 6.33 -> 5.9 - 6.0 ms, parked at two waves 6.7.
struct T3Park {
  static constexpr bool scalar_coefficients = true;     // (materials.hpp horner_step: what makes the parked kernels fit)
  double* slot;
  MH_DEV void save(const double (&Finv)[9], const double (&s)[9]) const {
#pragma unroll
    for (int k = 0; k < 9; ++k) slot[k * 128] = Finv[k];
    slot[9 * 128] = s[0];
    slot[10 * 128] = s[4];
    slot[11 * 128] = s[8];
    slot[12 * 128] = s[1];
    slot[13 * 128] = s[2];
    slot[14 * 128] = s[5];
    asm volatile("" ::: "memory");
  }
  MH_DEV void restore(double (&Finv)[9], double (&s)[9]) const {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int k = 0; k < 9; ++k) Finv[k] = slot[k * 128];
    s[0] = slot[9 * 128];
    s[4] = slot[10 * 128];
    s[8] = slot[11 * 128];
    s[1] = s[3] = slot[12 * 128];
    s[2] = s[6] = slot[13 * 128];
    s[5] = s[7] = slot[14 * 128];
  }
};

// FAMILY 0: closed-form materials (materials.hpp); 2..5: that one of the other materials (materials_other.hpp) as a
// compile-time constant -- with all four in one kernel the register allocation spilled 142 .. 360 registers, one at a time
// none.  GRAD 0: residual pieces only;
// 2: DomainPostTimeAdvance (nonlinear_solid.cpp:179-199) -- F at the points as for an assembly, then the material's state
// commit, nothing else (the degree-2 commit kernel's direct 64-node sum per point spilled 821 registers at degree 3)
// (closed-form family, residual+Jacobian mode: held to three waves per SIMD -- 168 registers; parked it needs 177, and the ten
// doubles that go to scratch memory do so once per point, around the record: 6.33 -> 5.9 - 6.0 ms, profiles/r05_cfg3_horner_ab.txt)
template<int FAMILY, int GRAD>
__global__ __launch_bounds__(128)
__attribute__((amdgpu_waves_per_eu(FAMILY == 0 && GRAD == 1 ? 3 : 1, FAMILY == 0 && GRAD == 1 ? 3 : 8)))
void tp3_point_kernel(TensorArgs p) {
  constexpr int NB = T3_NB, NQ = T3_NQ, ND = T3_ND, NPT = T3_NPT, PS = T3_PS;
  constexpr int FK = FAMILY >= 2 ? FAMILY : -1;
  __shared__ double ue[3 * ND];
  __shared__ double tab[6 * NB * NQ];       // [dir][B, D][a][q]
  // one pool: PH = W | V.  W holds the first grad u stage and the second residual stage; PH lives between them (written
  // behind the point evaluation, dead once the first residual stage has read it), with one row of 128 per (i, m) so that a
  // lane's entries are its own first nine T3Park slots.  V and the PH rows are free while the points are evaluated: the J2
  // return mapping parks F^-1 and the trial deviator there (T3Park: 15 rows of 128).  18.9 KB in all: eight workgroups per CU.
  constexpr int PHS = 128;                  // row stride of PH
  __shared__ double pool[9 * PHS + 9 * NB * NQ * NQ];
  static_assert(9 * PHS >= 9 * NB * NB * NQ && 9 * PHS + 9 * NB * NQ * NQ >= 15 * 128, "W and the park rows fit");
  double* PH = pool;                        // Phat [i*3 + m][point]
  double* W = pool;                         // [i*3 + m][a1 + 4 a2][q0]
  double* V = pool + 9 * PHS;               // [i*3 + m][a2][q0 + 5 q1]
  const int tid = threadIdx.x;
  const int64_t e = blockIdx.x;
  int el[3];
  el[0] = (int)(e % p.box_n[0]);
  el[1] = (int)((e / p.box_n[0]) % p.box_n[1]);
  el[2] = (int)(e / ((int64_t)p.box_n[0] * p.box_n[1]));
  if (tid < ND) {
    const int64_t node = p.dofs[e * ND + tid];
#pragma unroll
    for (int c = 0; c < 3; ++c) ue[c * ND + tid] = p.u[node * 3 + c];
  }
  if (tid < 6 * NB * NQ) {
    const int dir = tid / (2 * NB * NQ), rem = tid % (2 * NB * NQ), isD = rem / (NB * NQ), k = rem % (NB * NQ);
    const int span = p.box_begin[dir] + el[dir];
    tab[tid] = ((isD ? p.tabD[dir] : p.tabB[dir]) + (int64_t)span * NB * NQ)[k];
  }
  // the geometry factors of the lane's point, requested now in the residual-only mode: they are first used three barriers
  // further down, and a load is not moved up across a barrier -- their round trip to memory would start only there
  // (4.375 -> 4.30 ms; the commit would pay with its fourth wave per SIMD, 123 -> 143 registers, 3.43 -> 3.85 ms, and the
  // residual+Jacobian mode gains nothing: profiles/r05_cfg3_horner_ab.txt)
  const double* g = p.geo + e * 10 * NPT + (tid < NPT ? tid : 0);
  double Ji[9], wd;
  constexpr bool EARLY_GEO = GRAD == 0;
  if constexpr (EARLY_GEO) {
#pragma unroll
    for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NPT];
    wd = g[(int64_t)9 * NPT];
  }
  __syncthreads();
  // grad_xi u at the 125 points by sum factorisation, one direction per stage through LDS (V and W are free until the
  // residual rows below): 9 x 4 multiply-adds per point and ~45 per lane in the two stages before, instead of 64 nodes
  // x 12 per point (the direct sum was a sixth of this kernel's vector instructions).
  // Lane -> output maps of the stages (here and in the residual stages below): a lane owns ONE combination of the two
  // indices that are not summed and not a tensor component, reads its table values once and walks the components with
  // compile-time offsets.  (Round 5; before, the stages were loops t = tid, tid + 128, .. over the flat output index: 20 -
  // 30 integer instructions of index arithmetic around the 4 - 5 multiply-adds of every output, a fifth of the kernel's
  // vector instructions.  80 - 100 of the 128 lanes are busy now instead of all, at a third of the instructions; the sums
  // run in the same order: same bits.)
  {
    double* SA = W;   // [variant of direction 0: B, D][i][q0][a1 + 4 a2]
    double* SB = V;   // [D0 B1, B0 D1, B0 B1][i][q0 + 5 q1][a2]
    if (tid < NQ * NB * NB) {
      const int a12 = tid & 15, q0 = tid >> 4;
      double T[2][NB];
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0) T[v][a0] = tab_ptr<3>(tab, 0, v)[a0 * NQ + q0];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        const double* U = ue + i * ND + NB * a12;
        double u4[NB];
#pragma unroll
        for (int a0 = 0; a0 < NB; ++a0) u4[a0] = U[a0];
#pragma unroll
        for (int v = 0; v < 2; ++v) {
          double sv = 0.0;
#pragma unroll
          for (int a0 = 0; a0 < NB; ++a0) sv = __builtin_fma(T[v][a0], u4[a0], sv);
          SA[(v * 3 + i) * (NQ * NB * NB) + tid] = sv;
        }
      }
    }
    __syncthreads();
    if (tid < NQ * NQ * NB) {
      const int a2 = tid & 3, q01 = tid >> 2, q0 = q01 % NQ, q1 = q01 / NQ;
      double T[2][NB];
#pragma unroll
      for (int v = 0; v < 2; ++v)
#pragma unroll
        for (int a1 = 0; a1 < NB; ++a1) T[v][a1] = tab_ptr<3>(tab, 1, v)[a1 * NQ + q1];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        double sB[NB], sD[NB];   // the first stage's sums with the direction-0 values / derivatives
#pragma unroll
        for (int a1 = 0; a1 < NB; ++a1) {
          sB[a1] = SA[((0 * 3 + i) * NQ + q0) * (NB * NB) + NB * a2 + a1];
          sD[a1] = SA[((1 * 3 + i) * NQ + q0) * (NB * NB) + NB * a2 + a1];
        }
#pragma unroll
        for (int w = 0; w < 3; ++w) {
          double sv = 0.0;
#pragma unroll
          for (int a1 = 0; a1 < NB; ++a1) sv = __builtin_fma(T[w == 1 ? 1 : 0][a1], w == 0 ? sD[a1] : sB[a1], sv);
          SB[(w * 3 + i) * (NQ * NQ * NB) + tid] = sv;
        }
      }
    }
    __syncthreads();
  }
  double H[9];
  if (tid < NPT) {
    const int q2 = tid / (NQ * NQ), q01 = tid % (NQ * NQ);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const double* T = tab_ptr<3>(tab, 2, k == 2 ? 1 : 0) + q2;
        const double* S = V + ((k * 3 + i) * (NQ * NQ) + q01) * NB;
        double sv = 0.0;
#pragma unroll
        for (int a2 = 0; a2 < NB; ++a2) sv += T[a2 * NQ] * S[a2];
        H[i * 3 + k] = sv;
      }
  }
  constexpr bool PARK = FAMILY == 0;
  if constexpr (PARK) __syncthreads();           // (V and W are free from here on: T3Park)
  if (tid < NPT) {
    if constexpr (!EARLY_GEO) {
#pragma unroll
      for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NPT];
      wd = g[(int64_t)9 * NPT];
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
    if constexpr (GRAD == 2) {
      int st;
      if constexpr (FAMILY != 0) st = accumulate_other<3, FK>(p.mat, p.dt, p.state, e * NPT + tid, F);
      else st = accumulate_state<3>(p.mat, p.dt, p.state, e * NPT + tid, F, T3Park{pool + tid});
      if (st) atomicOr(p.status, st);
      return;
    }
    double Pk[9];
    int status;
    if constexpr (FAMILY != 0 && GRAD == 1) {
      // the other materials' tangent one direction (j, L) at a time, pulled back and accumulated over L as it arrives:
      // Ahat_i[m][j][n] = wd sum_L (sum_J Jinv[m][J] dP_iJ/dF_jL) Jinv[n][L] -- 27 accumulators per j instead of the 81
      // entries of dP/dF (which, with the material's own working set, spilled 360 registers)
      OtherTangent<3> ot;
      status = other_tangent_begin<3, FK>(p.mat, p.dt, p.state, e * NPT + tid, F, Pk, ot);
      double* rec = p.scratch_pt + e * (int64_t)(T3_REC * PS);
#pragma unroll 1
      for (int j = 0; j < 3; ++j) {
        double acc[27];
#pragma unroll
        for (int k = 0; k < 27; ++k) acc[k] = 0.0;
#pragma unroll 1
        for (int L = 0; L < 3; ++L) {
          double dP[9];
          other_tangent_dir<3, FK>(p.mat, p.dt, F, ot, j, L, dP);
          double jl[3];     // Jinv[n][L] (selects: a register array indexed by a loop variable would go to scratch memory)
#pragma unroll
          for (int n = 0; n < 3; ++n) jl[n] = L == 0 ? Ji[n * 3] : (L == 1 ? Ji[n * 3 + 1] : Ji[n * 3 + 2]);
#pragma unroll
          for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int m = 0; m < 3; ++m) {
              double b = 0.0;
#pragma unroll
              for (int J = 0; J < 3; ++J) b += Ji[m * 3 + J] * dP[i + J * 3];
#pragma unroll
              for (int n = 0; n < 3; ++n) acc[(i * 3 + m) * 3 + n] += b * jl[n];
            }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          double blk[9];
#pragma unroll
          for (int mn = 0; mn < 9; ++mn) blk[mn] = wd * acc[i * 9 + mn];
          t3_store_block_at(rec + (i * 3 + j) * 9 * T3_PS, tid, blk);     // (j is a loop variable here: t3_rec_block written out)
        }
      }
    } else if constexpr (FAMILY != 0) {
      status = evaluate_other<3, FK>(p.mat, p.dt, p.state, e * NPT + tid, F, Pk, nullptr, 1.0);
    } else {
      PointResult<3> w;
      if constexpr (PARK) {
        status = evaluate_pk1<3>(p.mat, p.dt, p.state, e * NPT + tid, F, w, T3Park{pool + tid});
        // (the geometry factors are not held across the return mapping either: read again, from the cache)
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 9; ++k) Ji[k] = g[(int64_t)k * NPT];
        wd = g[(int64_t)9 * NPT];
      } else {
        status = evaluate_pk1<3>(p.mat, p.dt, p.state, e * NPT + tid, F, w);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) Pk[k] = w.P[k];
      if constexpr (GRAD == 1) t3_closed_form_record(p.mat.m, w, Ji, wd, p.scratch_pt + e * (int64_t)(T3_REC * PS), tid);
    }
    if (status) atomicOr(p.status, status);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        double t = 0.0;
#pragma unroll
        for (int J = 0; J < 3; ++J) t += Pk[i + J * 3] * Ji[m * 3 + J];
        PH[(i * 3 + m) * PHS + tid] = wd * t;
      }
  }
  if constexpr (GRAD == 2) return;
  __syncthreads();
  // element residual pieces R_i[a] = sum_q sum_m dN_a/dxi_m Phat_i[m], one direction at a time (lane -> output maps as in
  // the grad u stages above)
  if (tid < NB * NQ * NQ) {
    const int q01 = tid % (NQ * NQ), a2 = tid / (NQ * NQ);
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q2 = 0; q2 < NQ; ++q2) T[v][q2] = tab_ptr<3>(tab, 2, v)[a2 * NQ + q2];
#pragma unroll
    for (int im = 0; im < 9; ++im) {
      double sv = 0.0;
#pragma unroll
      for (int q2 = 0; q2 < NQ; ++q2) sv = __builtin_fma(T[im % 3 == 2 ? 1 : 0][q2], PH[im * PHS + q01 + NQ * NQ * q2], sv);
      V[im * (NB * NQ * NQ) + tid] = sv;
    }
  }
  __syncthreads();
  if (tid < NB * NB * NQ) {
    const int q0 = tid % NQ, a12 = tid / NQ, a1 = a12 % NB, a2 = a12 / NB;
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q1 = 0; q1 < NQ; ++q1) T[v][q1] = tab_ptr<3>(tab, 1, v)[a1 * NQ + q1];
#pragma unroll
    for (int im = 0; im < 9; ++im) {
      double sw = 0.0;
#pragma unroll
      for (int q1 = 0; q1 < NQ; ++q1) sw = __builtin_fma(T[im % 3 == 1 ? 1 : 0][q1], V[(im * NB + a2) * NQ * NQ + q0 + NQ * q1], sw);
      W[im * (NB * NB * NQ) + tid] = sw;
    }
  }
  __syncthreads();
  {
    // (the first wave takes components 0 and 1, the second component 2)
    const int a = tid & (ND - 1), a0 = a % NB, a12 = a / NB;
    double T[2][NQ];
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int q0 = 0; q0 < NQ; ++q0) T[v][q0] = tab_ptr<3>(tab, 0, v)[a0 * NQ + q0];
    const int i_begin = tid < ND ? 0 : 2, i_end = tid < ND ? 2 : 3;
    for (int i = i_begin; i < i_end; ++i) {
      double sr = 0.0;
#pragma unroll
      for (int m = 0; m < 3; ++m)
#pragma unroll
        for (int q0 = 0; q0 < NQ; ++q0) sr = __builtin_fma(T[m == 0 ? 1 : 0][q0], W[((i * 3 + m) * NB * NB + a12) * NQ + q0], sr);
      p.scratch_r[(e * ND + a) * 3 + i] = sr;       // [element][a][i]: see tp3_gather_kernel
    }
  }
}

// ------------------------------------------------------------------------------------------------
// phase 1
// ------------------------------------------------------------------------------------------------
MH_DEV double t3_readlane(double x, int l) {
  const unsigned long long v = __double_as_longlong(x);
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)v, l), hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), l);
  return __longlong_as_double(((unsigned long long)hi << 32) | lo);
}

// lanes 0..31: a of the same lane; lanes 32..63: b of lane - 32   (v_permlane32_swap: checked on gfx950, scratch/pl)
MH_DEV double t3_low_halves(double a, double b) {
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  return __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
}

// rows of 16 lanes r0..r3: a = [a0, a1, a2, a3], b = [b0, b1, b2, b3]  ->  [a0 + a1, b0 + b1, a2 + a3, b2 + b3]
// (v_permlane16_swap: rows 1, 3 of the first operand <-> rows 0, 2 of the second -- tests/test_tensor_p3_gpu.py pins it
// through the oracle parity of the whole assembly)
MH_DEV double t3_row_pair_sums(double a, double b) {
  const unsigned long long ua = __double_as_longlong(a), ub = __double_as_longlong(b);
  const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)ua, (unsigned)ub, false, false);
  const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)(ua >> 32), (unsigned)(ub >> 32), false, false);
  const double x = __longlong_as_double(((unsigned long long)hi[0] << 32) | lo[0]);
  const double y = __longlong_as_double(((unsigned long long)hi[1] << 32) | lo[1]);
  return x + y;
}

// compile-time loop: the index arrives as a type, so that it can be an immediate of an asm statement
template<int... Is, class F>
MH_DEV void t3_for(std::integer_sequence<int, Is...>, F&& f) {
  (f(std::integral_constant<int, Is>{}), ...);
}

// The four accumulator tiles K[(a1, b1)], a1 = 0..3, of the pair column b1 in flight live in FIXED registers of the
// accumulation file -- tile a1 = a[8 a1 : 8 a1 + 7], register r of it = a[8 a1 + 2 r : 8 a1 + 2 r + 1] -- from the LDS read
// of their carry to the stores: the compiler moves LDS data and asm operands through the architectural registers
// (8 v_accvgpr_write per tile going in, an AGPR -> VGPR -> AGPR round trip per value going out, 7.7 cycles per half
// each -- scratch/issue_bench.hip); the hardware does not need to.  To the compiler a tile is four doubles bound to
// those registers (a 256-bit value bound to a physical tuple crashes its copy lowering).  Carry in LDS:
// [pair a1 4 + b1][slot 0..3][lane], slot 3 stays zero.  EVERY lane writes its registers 1..3 to the slots 0..2 -- also
// the lane group whose register is final (b2 = 0) and must not be carried on: the READ of slot s = r - 1 by that lane
// group goes to slot 3 instead (addr[s] = this lane's slot-0 address of pair 0, plus (3 - s) slots for lane group 3 - s),
// so that neither side needs an execution mask (with masks: 3 branches and 12 single writes per pair column, 1.0 of
// the 19.4 ms a variant of this kernel without its memory traffic took).
template<int B1>
MH_DEV void t3_carry_in_0(const unsigned (&addr)[4], double (&K)[4]) {   // requests the carry of (0, B1) as the tile's start value (no wait)
  asm volatile("ds_read_b64 a[0:1], %4 offset:%8\n\tds_read_b64 a[2:3], %5 offset:%9\n\t"
               "ds_read_b64 a[4:5], %6 offset:%10\n\tds_read_b64 a[6:7], %7 offset:%11"
               : "={a[0:1]}"(K[0]), "={a[2:3]}"(K[1]), "={a[4:5]}"(K[2]), "={a[6:7]}"(K[3])
               : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "n"((0 + B1 * 4) * 512), "n"((0 + B1 * 4 + 1) * 512),
                 "n"((0 + B1 * 4 + 2) * 512), "n"((0 + B1 * 4 + 3) * 512));
}
MH_DEV void t3_s3_main_0(double (&K)[4], double e0, double e1, double e2, double e3, const double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t"
               "v_mfma_f64_16x16x4_f64 a[0:7], %4, %8, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[0:7], %5, %9, a[0:7]\n\t"
               "v_mfma_f64_16x16x4_f64 a[0:7], %6, %10, a[0:7]\n\tv_mfma_f64_16x16x4_f64 a[0:7], %7, %11, a[0:7]"
               : "+{a[0:1]}"(K[0]), "+{a[2:3]}"(K[1]), "+{a[4:5]}"(K[2]), "+{a[6:7]}"(K[3])
               : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
MH_DEV void t3_s3_plane_0(double (&K)[4], double e, double b) {
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 a[0:7], %4, %5, a[0:7]"
               : "+{a[0:1]}"(K[0]), "+{a[2:3]}"(K[1]), "+{a[4:5]}"(K[2]), "+{a[6:7]}"(K[3])
               : "v"(e), "v"(b));
}
template<int B1>
MH_DEV void t3_carry_in_1(const unsigned (&addr)[4], double (&K)[4]) {   // requests the carry of (1, B1) as the tile's start value (no wait)
  asm volatile("ds_read_b64 a[8:9], %4 offset:%8\n\tds_read_b64 a[10:11], %5 offset:%9\n\t"
               "ds_read_b64 a[12:13], %6 offset:%10\n\tds_read_b64 a[14:15], %7 offset:%11"
               : "={a[8:9]}"(K[0]), "={a[10:11]}"(K[1]), "={a[12:13]}"(K[2]), "={a[14:15]}"(K[3])
               : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "n"((16 + B1 * 4) * 512), "n"((16 + B1 * 4 + 1) * 512),
                 "n"((16 + B1 * 4 + 2) * 512), "n"((16 + B1 * 4 + 3) * 512));
}
MH_DEV void t3_s3_main_1(double (&K)[4], double e0, double e1, double e2, double e3, const double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t"
               "v_mfma_f64_16x16x4_f64 a[8:15], %4, %8, a[8:15]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %5, %9, a[8:15]\n\t"
               "v_mfma_f64_16x16x4_f64 a[8:15], %6, %10, a[8:15]\n\tv_mfma_f64_16x16x4_f64 a[8:15], %7, %11, a[8:15]"
               : "+{a[8:9]}"(K[0]), "+{a[10:11]}"(K[1]), "+{a[12:13]}"(K[2]), "+{a[14:15]}"(K[3])
               : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
MH_DEV void t3_s3_plane_1(double (&K)[4], double e, double b) {
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 a[8:15], %4, %5, a[8:15]"
               : "+{a[8:9]}"(K[0]), "+{a[10:11]}"(K[1]), "+{a[12:13]}"(K[2]), "+{a[14:15]}"(K[3])
               : "v"(e), "v"(b));
}
template<int B1>
MH_DEV void t3_carry_in_2(const unsigned (&addr)[4], double (&K)[4]) {   // requests the carry of (2, B1) as the tile's start value (no wait)
  asm volatile("ds_read_b64 a[16:17], %4 offset:%8\n\tds_read_b64 a[18:19], %5 offset:%9\n\t"
               "ds_read_b64 a[20:21], %6 offset:%10\n\tds_read_b64 a[22:23], %7 offset:%11"
               : "={a[16:17]}"(K[0]), "={a[18:19]}"(K[1]), "={a[20:21]}"(K[2]), "={a[22:23]}"(K[3])
               : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "n"((32 + B1 * 4) * 512), "n"((32 + B1 * 4 + 1) * 512),
                 "n"((32 + B1 * 4 + 2) * 512), "n"((32 + B1 * 4 + 3) * 512));
}
MH_DEV void t3_s3_main_2(double (&K)[4], double e0, double e1, double e2, double e3, const double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t"
               "v_mfma_f64_16x16x4_f64 a[16:23], %4, %8, a[16:23]\n\tv_mfma_f64_16x16x4_f64 a[16:23], %5, %9, a[16:23]\n\t"
               "v_mfma_f64_16x16x4_f64 a[16:23], %6, %10, a[16:23]\n\tv_mfma_f64_16x16x4_f64 a[16:23], %7, %11, a[16:23]"
               : "+{a[16:17]}"(K[0]), "+{a[18:19]}"(K[1]), "+{a[20:21]}"(K[2]), "+{a[22:23]}"(K[3])
               : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
MH_DEV void t3_s3_plane_2(double (&K)[4], double e, double b) {
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 a[16:23], %4, %5, a[16:23]"
               : "+{a[16:17]}"(K[0]), "+{a[18:19]}"(K[1]), "+{a[20:21]}"(K[2]), "+{a[22:23]}"(K[3])
               : "v"(e), "v"(b));
}
template<int B1>
MH_DEV void t3_carry_in_3(const unsigned (&addr)[4], double (&K)[4]) {   // requests the carry of (3, B1) as the tile's start value (no wait)
  asm volatile("ds_read_b64 a[24:25], %4 offset:%8\n\tds_read_b64 a[26:27], %5 offset:%9\n\t"
               "ds_read_b64 a[28:29], %6 offset:%10\n\tds_read_b64 a[30:31], %7 offset:%11"
               : "={a[24:25]}"(K[0]), "={a[26:27]}"(K[1]), "={a[28:29]}"(K[2]), "={a[30:31]}"(K[3])
               : "v"(addr[0]), "v"(addr[1]), "v"(addr[2]), "v"(addr[3]), "n"((48 + B1 * 4) * 512), "n"((48 + B1 * 4 + 1) * 512),
                 "n"((48 + B1 * 4 + 2) * 512), "n"((48 + B1 * 4 + 3) * 512));
}
MH_DEV void t3_s3_main_3(double (&K)[4], double e0, double e1, double e2, double e3, const double (&b)[4]) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 1\n\t"
               "v_mfma_f64_16x16x4_f64 a[24:31], %4, %8, a[24:31]\n\tv_mfma_f64_16x16x4_f64 a[24:31], %5, %9, a[24:31]\n\t"
               "v_mfma_f64_16x16x4_f64 a[24:31], %6, %10, a[24:31]\n\tv_mfma_f64_16x16x4_f64 a[24:31], %7, %11, a[24:31]"
               : "+{a[24:25]}"(K[0]), "+{a[26:27]}"(K[1]), "+{a[28:29]}"(K[2]), "+{a[30:31]}"(K[3])
               : "v"(e0), "v"(e1), "v"(e2), "v"(e3), "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]));
}
MH_DEV void t3_s3_plane_3(double (&K)[4], double e, double b) {
  asm volatile("s_nop 1\n\tv_mfma_f64_16x16x4_f64 a[24:31], %4, %5, a[24:31]"
               : "+{a[24:25]}"(K[0]), "+{a[26:27]}"(K[1]), "+{a[28:29]}"(K[2]), "+{a[30:31]}"(K[3])
               : "v"(e), "v"(b));
}
// register R (1..3) of the four tiles -> slot R - 1 of their carry (every lane)
template<int B1, int R>
MH_DEV void t3_carry_out(unsigned addr, double k0, double k1, double k2, double k3) {
  static_assert(R >= 1 && R <= 3, "register 0 is always final");
  if constexpr (R == 1)
    asm volatile("ds_write_b64 %4, a[2:3] offset:%5\n\tds_write_b64 %4, a[10:11] offset:%6\n\t"
                 "ds_write_b64 %4, a[18:19] offset:%7\n\tds_write_b64 %4, a[26:27] offset:%8"
                 ::"{a[2:3]}"(k0), "{a[10:11]}"(k1), "{a[18:19]}"(k2), "{a[26:27]}"(k3), "v"(addr), "n"((B1 * 4 + R - 1) * 512), "n"((16 + B1 * 4 + R - 1) * 512),
                 "n"((32 + B1 * 4 + R - 1) * 512), "n"((48 + B1 * 4 + R - 1) * 512) : "memory");
  else if constexpr (R == 2)
    asm volatile("ds_write_b64 %4, a[4:5] offset:%5\n\tds_write_b64 %4, a[12:13] offset:%6\n\t"
                 "ds_write_b64 %4, a[20:21] offset:%7\n\tds_write_b64 %4, a[28:29] offset:%8"
                 ::"{a[4:5]}"(k0), "{a[12:13]}"(k1), "{a[20:21]}"(k2), "{a[28:29]}"(k3), "v"(addr), "n"((B1 * 4 + R - 1) * 512), "n"((16 + B1 * 4 + R - 1) * 512),
                 "n"((32 + B1 * 4 + R - 1) * 512), "n"((48 + B1 * 4 + R - 1) * 512) : "memory");
  else
    asm volatile("ds_write_b64 %4, a[6:7] offset:%5\n\tds_write_b64 %4, a[14:15] offset:%6\n\t"
                 "ds_write_b64 %4, a[22:23] offset:%7\n\tds_write_b64 %4, a[30:31] offset:%8"
                 ::"{a[6:7]}"(k0), "{a[14:15]}"(k1), "{a[22:23]}"(k2), "{a[30:31]}"(k3), "v"(addr), "n"((B1 * 4 + R - 1) * 512), "n"((16 + B1 * 4 + R - 1) * 512),
                 "n"((32 + B1 * 4 + R - 1) * 512), "n"((48 + B1 * 4 + R - 1) * 512) : "memory");
}
// The final entries with b2 = 0 -- register 4 - kk of lane groups kk = 1..3 -- of the pair column B1 of the element BEFORE:
// they lie in the carry (every lane writes its registers 1..3 there) until this element's t3_carry_out<B1> overwrites
// them.  One read per tile at the lane group's own slot (addr = this lane's address of slot 3 - kk) and then ONE store
// per tile instead of three under three execution masks: a store costs the CU's address unit the same whatever its
// mask, and the 48 masked stores per element were 1.3 of this kernel's 21.2 ms (scratch/p3_variants.sh, round 3).
template<int B1>
MH_DEV void t3_finals_in(unsigned addr, double (&f)[4]) {
  asm volatile("ds_read_b64 %0, %4 offset:%5\n\tds_read_b64 %1, %4 offset:%6\n\tds_read_b64 %2, %4 offset:%7\n\t"
               "ds_read_b64 %3, %4 offset:%8\n\ts_waitcnt lgkmcnt(0)"
               : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3])
               : "v"(addr), "n"((B1 * 4) * 512), "n"((16 + B1 * 4) * 512), "n"((32 + B1 * 4) * 512), "n"((48 + B1 * 4) * 512)
               : "memory");
}
// S2 coefficients by DPP: a register pair holds sixteen coefficients per row of 16 lanes (lane n of the row = coefficient
// n); `row_newbcast:N` hands every lane of a row the value of the row's lane N as the operand of the multiply-add itself --
// at the cost of the plain instruction (4.9 cycles, scratch/issue_bench.hip; what it does: scratch/dpp_bcast.hip).  The
// rows of 16 lanes are the lane groups kk, so the plane q0 = 4, whose coefficients differ per lane group, needs nothing
// special.  Before: 32 LDS reads of 16 bytes per lane and pair column to fetch the same eight doubles again and again.
template<int N>
MH_DEV void t3_fmac_bc(double& acc, double table, double w) {   // acc += table[lane N of the row] * w
  asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(table), "v"(w), "n"(N));
}
template<int N>
MH_DEV double t3_bc(double table) {   // table[lane N of the row] in every lane of the row
  double c;
  asm("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(c) : "v"(table), "n"(N));
  return c;
}
// the wait states between a matrix write and a store / LDS read of its result (nothing is padded inside asm)
// (round 4: sized by the disassembly lint, tests/test_isa_lint_cpu.py -- without it the first stores of the plane tiles come 3
// wait states behind their matrix instructions, 18 are required: 16 more.  It was `s_nop 15; s_nop 15`, 32 wait states,
// before the lint could say how many are needed.  T3_LINT_DROP_RESULTS_GUARD: the lint's own negative test compiles it away.)
#ifdef T3_LINT_DROP_RESULTS_GUARD
MH_DEV void t3_results_guard() { asm volatile("" ::: "memory"); }
#else
MH_DEV void t3_results_guard() { asm volatile("s_nop 15" ::: "memory"); }
#endif

// Lane layout of v_mfma_f64_16x16x4 (gfx950): A[row = lane & 15][k = lane >> 4], B[k = lane >> 4][col = lane & 15],
// D register r: [row = (lane >> 4) + 4 r][col = lane & 15].
//
// One wave per (element column along the third direction, row component i, column component j); it walks the column.
// Node pairs of directions 0 and 1 are indexed 4 a + b; those of direction 2 -- the rows of the result --
// DIAGONALLY, (b2 - a2 mod 4) + 4 a2, so that result register r is a2 and the lane group is b2 - a2: the entry
// (a2, b2) of this element and the entry (a2 - 1, b2 - 1) of the next one -- the same pair of nodes -- are registers
// r and r - 1 of the SAME lane.  The sums over the elements of a column therefore run inside the wave (the carry of the
// 48 values per lane waits in LDS, private to the wave, and enters the next element as the accumulator input of S3), and
// every (node pair, column) leaves the wave once, from the highest element that contains both nodes.
//
// S1, per (m, n): two tiles of 16 point rows rho = kk + 4 r (kk = lane >> 4 of the RESULT, r = result register)
//   tile U: rho -> (q0, q1) = (rho & 3, rho >> 2)                        i.e. result lane group kk = q0, register r = q1
//   tile V: rho < 4 -> (rho, 4);  rho = 4, 8, 12 -> (4, rho / 4 - 1);  rho = 5, 9 -> (4, 3), (4, 4);  others unused
//           i.e. register 0 = (q0 = kk, q1 = 4), registers 1..3 = the plane q0 = 4: lane group 0 holds q1 = 0, 1, 2,
//           lane group 1 holds q1 = 3, 4
//   two k-steps each: q2 = kk, then q2 = 4 (operand lane group 0 only).
// S2 runs lane-local on the result registers: the points q0 < 4 with wave-uniform coefficients, the plane q0 = 4
// with per-lane ones (two partial sums, lane groups 0 and 1).
// S3, per (a1, b1): one k-step q0 = kk per table variant g, and for the plane q0 = 4 ONE k-step whose k index is the variant
// (round 5: the two partial sums of a variant are added and variant g's total moved to lane group g with a v_permlane16_swap
// and a v_permlane32_swap pair -- 116 matrix instructions per block; rounds 2-4: two k-steps with the partial sums of two
// variants each, 132).
//
// Piece of (element, i), 5376 doubles -- what phase 2 reads contiguously (round 5: the pair columns b1 = 2 h, 2 h + 1 ADJACENT, so
// that a lane of the contraction stores its two values of a pair with one 16-byte store):
//   [0, 3072)     rows a2 = 0:  (a0 + 4 a1) 192 + j 64 + (b1 / 2) 32 + b2 8 + b0 2 + b1 % 2
//   [3072, 5376)  rows a2 >= 1, b2 = 0:  3072 + (a0 + 4 a1 + 16 (a2 - 1)) 48 + j 16 + (b1 / 2) 8 + b0 2 + b1 % 2
// and per (column, i) the tail of its last element, rows a2 >= 1, b2 >= 1:
//   (a0 + 4 a1 + 16 (a2 - 1)) 144 + j 48 + b1 12 + (b2 - 1) 4 + b0
__global__ __launch_bounds__(64) void tp3_contract_kernel(TensorArgs p) {
  constexpr int NB = T3_NB, NQ = T3_NQ, PS = T3_PS;
  extern __shared__ __align__(16) double carry[];   // [16 (a1, b1)][4 slots, the last one zero][64 lanes], then the direction-1 table [6][2][4]
  const int lane = threadIdx.x;
  const int J = (int)(blockIdx.x % 3), I = (int)((blockIdx.x / 3) % 3);
  const int64_t col = blockIdx.x / 9;
  const int eu = (int)(col % p.box_n[0]), ev = (int)(col / p.box_n[0]);
  const int n_seq = p.box_n[2];
  const int64_t e_step = (int64_t)p.box_n[0] * p.box_n[1];
  const int c16 = lane & 15, kk = lane >> 4, pa = c16 >> 2, pb = c16 & 3;
  const int pa2 = c16 >> 2, pb2 = (pa2 + (c16 & 3)) & 3;   // direction 2: diagonal pair index

  // direction 0 (S3 B operands): variants (a: B / D) + 2 (b: B / D); k-step 0: q0 = kk; the plane q0 = 4: variants 0 | 1 and 2 | 3
  double bS0[4], bS0y;
  {
    const double* B0 = p.tabB[0] + (int64_t)(p.box_begin[0] + eu) * NB * NQ;
    const double* D0 = p.tabD[0] + (int64_t)(p.box_begin[0] + eu) * NB * NQ;
    const double Ba = B0[pa * NQ + kk], Da = D0[pa * NQ + kk], Bb = B0[pb * NQ + kk], Db = D0[pb * NQ + kk];
    const double Bax = B0[pa * NQ + 4], Dax = D0[pa * NQ + 4], Bbx = B0[pb * NQ + 4], Dbx = D0[pb * NQ + 4];
#pragma unroll
    for (int v = 0; v < 4; ++v) bS0[v] = ((v & 1) ? Da : Ba) * ((v & 2) ? Db : Bb);
    bS0y = ((kk & 1) ? Dax : Bax) * ((kk & 2) ? Dbx : Bbx);     // the plane q0 = 4: k = lane group = variant
  }
  // direction 1 (S2 coefficients): table rows [q1][B, D][a] in LDS, q1 = 0..4 and a row of zeros.  The points q0 < 4 read
  // row q1 = slot (the same for every lane); for the plane q0 = 4 register r = 1..3 of tile V holds q1 = r - 1 in lane
  // group 0 and q1 = r + 2 in lane group 1 (nothing elsewhere): per-lane rows
  double* tl = carry + 16 * 4 * 64;
  if (lane < 2 * NB * NQ) {
    const int v = lane / (NB * NQ), a = (lane % (NB * NQ)) / NQ, q = lane % NQ;
    tl[q * 8 + v * 4 + a] = ((v ? p.tabD[1] : p.tabB[1]) + (int64_t)(p.box_begin[1] + ev) * NB * NQ)[a * NQ + q];
  } else if (lane < 2 * NB * NQ + 8) {
    tl[lane] = 0.0;
  }
  // the first element of the column starts from a zero carry
#pragma unroll
  for (int k = 0; k < 16 * 4; ++k) carry[k * 64 + lane] = 0.0;
  __syncthreads();
  // ... and from there into four coefficient registers (t3_fmac_bc): lane n of a row holds coefficient n & 7 = v 4 + a of
  // slot n >> 3.  TA: slots q1 = 0, 1 of the points q0 < 4; TB: q1 = 2, 3; TC: q1 = 4 | slot 0 of the plane q0 = 4; TD: slots
  // 1, 2 of the plane -- whose slot r holds q1 = r in lane group 0, q1 = r + 3 in lane group 1 (r < 2), zeros elsewhere
  double TA, TB, TC, TD;
  {
    const int ti = lane & 15, th = ti >> 3, tk = ti & 7;
    auto plane_row = [&](int r) -> int { return kk == 0 ? r : ((kk == 1 && r < 2) ? r + 3 : NQ); };
    TA = tl[8 * th + tk];
    TB = tl[8 * (2 + th) + tk];
    TC = tl[8 * (th == 0 ? 4 : plane_row(0)) + tk];
    TD = tl[8 * plane_row(1 + th) + tk];
  }

  // S1 A operands: the points of this lane
  const int ptU = (c16 & 3) + NQ * (c16 >> 2);
  const bool vrow = c16 < 4 || c16 == 4 || c16 == 8 || c16 == 12 || c16 == 5 || c16 == 9;
  const int vq0 = c16 < 4 ? c16 : 4;
  const int vq1 = c16 < 4 ? 4 : (c16 == 5 ? 3 : (c16 == 9 ? 4 : c16 / 4 - 1));
  const int ptV = vrow ? vq0 + NQ * vq1 : 0;
  const int offU0 = ptU + NQ * NQ * kk, offV0 = ptV + NQ * NQ * kk;
  // second k-step (q2 = 4, one live k): ONE operand register serves both tiles -- lane group 0 holds tile U's points,
  // lane group 1 tile V's, and the B operands of the two instructions are zero outside lane group 0 / 1 (27 record
  // loads per element instead of 36: a load costs the CU's address unit its 16 cycles whatever it fetches)
  const int offX1 = (kk == 1 ? ptV : ptU) + NQ * NQ * (NQ - 1);

  const int64_t e0 = eu + (int64_t)p.box_n[0] * ev;
  const t3_d4 zero4 = {0.0, 0.0, 0.0, 0.0};
  // operands of the element about to be contracted: Ahat values [mn][tile U first k-step, second k-step of BOTH tiles, tile V
  // first k-step], raw direction-2 tables
  double aop[9][3], t2[2][4];
  auto request = [&](int es) {
    const int64_t e = e0 + e_step * es;
    const double* rec = p.scratch_pt + e * (int64_t)(T3_REC * PS) + (I * 3 + J) * 9 * PS;      // (t3_rec_block)
#pragma unroll
    for (int mn = 0; mn < 9; ++mn) {
      // every lane loads a real (finite) value: the lanes that carry no point meet a zero B operand (second k-step)
      // or feed rows of the result that nothing reads
      const double* f = rec + (mn < 8 ? (mn >> 1) * 2 * PS + (mn & 1) : 8 * PS);
      const int st = mn < 8 ? 2 : 1;
      aop[mn][0] = f[st * offU0];
      aop[mn][1] = f[st * offX1];
      aop[mn][2] = f[st * offV0];
    }
    const double* B2 = p.tabB[2] + (int64_t)(p.box_begin[2] + es) * NB * NQ;
    const double* D2 = p.tabD[2] + (int64_t)(p.box_begin[2] + es) * NB * NQ;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int q = s == 0 ? kk : NQ - 1;
      t2[s][0] = B2[pa2 * NQ + q];
      t2[s][1] = D2[pa2 * NQ + q];
      t2[s][2] = B2[pb2 * NQ + q];
      t2[s][3] = D2[pb2 * NQ + q];
    }
  };
  request(0);
  double* cl = carry + lane;
  const unsigned cl_addr = (unsigned)(uintptr_t)cl;   // (the low half of a shared-aperture address is the LDS offset)
  // where this lane reads slot s of a pair's carry: the lane group that stored register s + 1 as final reads the zero slot
  const unsigned cl_in[4] = {cl_addr + (kk == 3 ? 3u * 512u : 0u), cl_addr + (kk == 2 ? 2u * 512u : 0u),
                             cl_addr + (kk == 1 ? 1u * 512u : 0u), cl_addr};
  const unsigned fin_addr = cl_addr + (kk == 0 ? 3u : (unsigned)(3 - kk)) * 512u;   // (lane group 0 has no such entry: the zero slot)
  double* out1_prev = nullptr;
#pragma unroll 1
  for (int es = 0; es < n_seq; ++es) {
    // S1
    double bS2[4], bS2U[4], bS2V[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      bS2[v] = ((v & 1) ? t2[0][1] : t2[0][0]) * ((v & 2) ? t2[0][3] : t2[0][2]);
      const double x = ((v & 1) ? t2[1][1] : t2[1][0]) * ((v & 2) ? t2[1][3] : t2[1][2]);
      bS2U[v] = kk == 0 ? x : 0.0;
      bS2V[v] = kk == 1 ? x : 0.0;
    }
    // (written as asm to pin the register files: the results, which the vector pipe reads four times each, in the
    // architectural registers; the prefetched record values, which only these instructions read, in the accumulation
    // file -- a value on the wrong side costs a v_accvgpr move of 7.7 cycles per half, scratch/issue_bench.hip)
    t3_d4 DU[9], DV[9];
#pragma unroll
    for (int mn = 0; mn < 9; ++mn) {
      const int m = mn / 3, n = mn % 3;
      const int v2 = (m == 2 ? 1 : 0) + (n == 2 ? 2 : 0);
      asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(DU[mn]) : "a"(aop[mn][0]), "v"(bS2[v2]));
      asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, 0" : "=&v"(DV[mn]) : "a"(aop[mn][2]), "v"(bS2[v2]));
    }
#pragma unroll
    for (int mn = 0; mn < 9; ++mn) {
      const int m = mn / 3, n = mn % 3;
      const int v2 = (m == 2 ? 1 : 0) + (n == 2 ? 2 : 0);
      asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(DU[mn]) : "a"(aop[mn][1]), "v"(bS2U[v2]));
      asm("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(DV[mn]) : "a"(aop[mn][1]), "v"(bS2V[v2]));
    }
    // the matrix results are read by the vector pipe from here on.  Nothing pads wait states behind asm matrix
    // instructions, and the S1 statements above are not volatile: what holds the schedule to the 19 / 18 wait states
    // gfx950 needs is the disassembly lint of tests/test_isa_lint_cpu.py (mimi_amd/isa_lint.py: the nearest vector read of an
    // S1 result is 42 wait states away -- the loads and the address arithmetic of the next element sit in between).  Rounds
    // 2-3 had `s_nop 15; s_nop 15` here; the lint showed that pair protects nothing (the compiler moved most of the S1
    // instructions behind it anyway), so only the scheduling fence is left.
    asm volatile("" ::: "memory");
    // the operands of the next element travel while this one is contracted
    request(es + 1 < n_seq ? es + 1 : es);   // (the last element once more: no branch in the loop)
    const int64_t e = e0 + e_step * es;
    double* piece = p.scratch_k + (e * 3 + I) * (int64_t)T3_PIECE;
    double* out0 = piece + pa * 192 + J * 64 + kk * 8 + pb * 2;                   // + 4 a1 192 + (b1 / 2) 32 + b1 % 2
    // lane groups 1..3 hold one final entry with b2 = 0 (register 4 - kk); lane group 0 none
    double* out1 = piece + 3072 + (pa + 16 * (3 - (kk ? kk : 1))) * 48 + J * 16 + pb * 2;   // + 4 a1 48 + (b1 / 2) 8 + b1 % 2
    t3_for(std::make_integer_sequence<int, 4>{}, [&](auto b1_c) {
      constexpr int b1 = decltype(b1_c)::value;
      __builtin_amdgcn_sched_barrier(0);   // one b1 at a time: interleaved they need more registers than there are
      // (m, n) -> variant of direction 1 (a: m == 1, b: n == 1) and group g of direction 0 (a: m == 0, b: n == 0):
      //   g = 3: (0,0)   g = 1: (0,1) (0,2)   g = 2: (1,0) (2,0)   g = 0: (1,1) (1,2) (2,1) (2,2)
      // EXTRA 0: the points q0 < 4 (slots q1 = 0..4, wave-uniform coefficients); 1: the plane q0 = 4 (registers 1..3
      // of tile V, per-lane coefficients; two partial sums in lane groups 0 and 1)
      auto s2 = [&](auto extra_tag, double (&E)[4][NB]) {
        constexpr bool ex = decltype(extra_tag)::value;
        t3_for(std::make_integer_sequence<int, (ex ? 3 : NQ)>{}, [&](auto s_c) {
          constexpr int s = decltype(s_c)::value;
          double X[9];
#pragma unroll
          for (int mn = 0; mn < 9; ++mn) X[mn] = ex ? DV[mn][s + 1] : (s < 4 ? DU[mn][s < 4 ? s : 0] : DV[mn][0]);
          // this slot's eight coefficients [B a = 0..3, D a = 0..3]: table register and first lane
          constexpr int slot = ex ? NQ + s : s;
          const double T = slot < 2 ? TA : (slot < 4 ? TB : (slot < 6 ? TC : TD));
          constexpr int cB = (slot & 1) * 8, cD = cB + 4;
          // every accumulation is ONE fused multiply-add (written out: a sum of two products added to E would cost a
          // multiply, a multiply-add and an add -- on this chip every vector instruction of the wave, fp64 or not, waits
          // for the matrix pipe and the matrix instructions for it, scratch/issue_bench.hip, so the count is the time)
          const double cbB = t3_bc<cB + b1>(T);   // (a plain multiplication takes no DPP operand)
          const double W3 = cbB * X[0];
          double W1 = cbB * X[2];
          t3_fmac_bc<cD + b1>(W1, T, X[1]);
          const double W2a = cbB * X[3], W2b = cbB * X[6];
          double W0a = cbB * X[5], W0b = cbB * X[8];
          t3_fmac_bc<cD + b1>(W0a, T, X[4]);
          t3_fmac_bc<cD + b1>(W0b, T, X[7]);
          t3_for(std::make_integer_sequence<int, NB>{}, [&](auto a1_c) {
            constexpr int a1 = decltype(a1_c)::value;
            if constexpr (s == 0) {   // (no accumulator starts from 0.0: x + 0.0 is an instruction)
              const double caB = t3_bc<cB + a1>(T);
              E[3][a1] = caB * W3;
              E[1][a1] = caB * W1;
              E[2][a1] = caB * W2b;
              E[0][a1] = caB * W0b;
            } else {
              t3_fmac_bc<cB + a1>(E[3][a1], T, W3);
              t3_fmac_bc<cB + a1>(E[1][a1], T, W1);
              t3_fmac_bc<cB + a1>(E[2][a1], T, W2b);
              t3_fmac_bc<cB + a1>(E[0][a1], T, W0b);
            }
            t3_fmac_bc<cD + a1>(E[2][a1], T, W2a);
            t3_fmac_bc<cD + a1>(E[0][a1], T, W0a);
          });
        });
      };
      // the carry of the four pairs (a1, b1): requested before the vector work, waited for after it
      double K0[4], K1[4], K2[4], K3[4];
      t3_carry_in_0<b1>(cl_in, K0);
      t3_carry_in_1<b1>(cl_in, K1);
      t3_carry_in_2<b1>(cl_in, K2);
      t3_carry_in_3<b1>(cl_in, K3);
      {
        double Em[4][NB];
        s2(std::false_type{}, Em);
        __builtin_amdgcn_sched_barrier(0);
        t3_s3_main_0(K0, Em[0][0], Em[1][0], Em[2][0], Em[3][0], bS0);
        t3_s3_main_1(K1, Em[0][1], Em[1][1], Em[2][1], Em[3][1], bS0);
        t3_s3_main_2(K2, Em[0][2], Em[1][2], Em[2][2], Em[3][2], bS0);
        t3_s3_main_3(K3, Em[0][3], Em[1][3], Em[2][3], Em[3][3], bS0);
        // behind the sixteen matrix instructions just issued: the b2 = 0 entries of the element before, LDS -> scratch
        double fin[4];
        t3_finals_in<b1>(fin_addr, fin);
        if (es > 0 && kk > 0) {
          out1_prev[0 * 4 * 48 + (b1 / 2) * 8 + b1 % 2] = fin[0];
          out1_prev[1 * 4 * 48 + (b1 / 2) * 8 + b1 % 2] = fin[1];
          out1_prev[2 * 4 * 48 + (b1 / 2) * 8 + b1 % 2] = fin[2];
          out1_prev[3 * 4 * 48 + (b1 / 2) * 8 + b1 % 2] = fin[3];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        double Ex[4][NB];
        s2(std::true_type{}, Ex);
        // the plane's two partial sums per variant (lane groups 0 and 1) added, variant g's total moved to lane group g:
        // ONE matrix instruction per tile (k = variant) instead of two (round 5; see gen_tp3_contract.py plane_totals)
        double Ey[NB];
#pragma unroll
        for (int a1 = 0; a1 < NB; ++a1)
          Ey[a1] = t3_low_halves(t3_row_pair_sums(Ex[0][a1], Ex[1][a1]), t3_row_pair_sums(Ex[2][a1], Ex[3][a1]));
        __builtin_amdgcn_sched_barrier(0);
        t3_s3_plane_0(K0, Ey[0], bS0y);
        t3_s3_plane_1(K1, Ey[1], bS0y);
        t3_s3_plane_2(K2, Ey[2], bS0y);
        t3_s3_plane_3(K3, Ey[3], bS0y);
        t3_results_guard();
      }
      // register r of this lane: (a2 = r, b2 = (r + kk) & 3), column (a0 = pa, b0 = pb).  Final: a2 = 0 (register 0, every
      // lane) or b2 = 0 (register 4 - kk of lane groups 1..3); the pairs (a2 >= 1, b2 >= 1) go on to the next element as its
      // (a2 - 1, b2 - 1): register r -> slot r - 1 of the carry.  Register r >= 1 is stored by the lane group with b2 = 0
      // (under its execution mask: no selects) and written to the carry by every lane; the next element's read skips
      // what the storing lane group wrote (t3_carry_in).
      out0[0 * 4 * 192 + (b1 / 2) * 32 + b1 % 2] = K0[0];
      out0[1 * 4 * 192 + (b1 / 2) * 32 + b1 % 2] = K1[0];
      out0[2 * 4 * 192 + (b1 / 2) * 32 + b1 % 2] = K2[0];
      out0[3 * 4 * 192 + (b1 / 2) * 32 + b1 % 2] = K3[0];
      t3_for(std::make_integer_sequence<int, 3>{}, [&](auto rr_c) {
        constexpr int r = decltype(rr_c)::value + 1;
        t3_carry_out<b1, r>(cl_addr, K0[r], K1[r], K2[r], K3[r]);   // (every lane: see t3_carry_in, t3_finals_in)
      });
    });
    out1_prev = out1;
  }
  // (the carry of the last element is read below through plain loads: its asm stores must have landed)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // the b2 = 0 entries of the last element (t3_finals_in found those of every other one)
  {
    const double* fl = cl + (kk == 0 ? 3 : 3 - kk) * 64;
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      const double v = fl[ab * 4 * 64];
      if (kk > 0) out1_prev[(ab / NB) * 4 * 48 + ((ab % NB) / 2) * 8 + (ab % NB) % 2] = v;
    }
  }
  // what the last element of the column would have passed on: rows a2 >= 1, b2 >= 1 -> the tail of (column, i)
  double* tail = p.scratch_tail + ((e0 * 3 + I) * (int64_t)T3_TAIL) + pa * 144 + J * 48 + pb;
#pragma unroll
  for (int ab = 0; ab < 16; ++ab)
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const int b2 = (r + kk) & 3;
      const double v = cl[(ab * 4 + r - 1) * 64];
      if (b2) tail[(4 * (ab / NB) + 16 * (r - 1)) * 144 + (ab % NB) * 12 + (b2 - 1) * 4] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// phase 1, hand-scheduled (round 5): the same arithmetic as tp3_contract_kernel above -- per value the same operations in the
// same order, the sums are bitwise equal -- with the whole column loop as ONE asm statement generated by
// gen_tp3_contract.py (fixed register map; every LDS / memory / scalar instruction in the shadow of a matrix instruction,
// double-buffered accumulator tiles, scalar-base addressing, one `s_waitcnt vmcnt(0)` per element: see the generator).
// This function computes what the loop needs per lane and per column, hands it over through LDS (the carry area, which
// the loop zeroes once it has read it: 33 slots of [64 lanes] x 8 bytes) and finishes the column as the C++ form does.
// ------------------------------------------------------------------------------------------------
// The eight direction-2 table values a lane of the contraction needs per element -- t2[s][k]: k = 0 B[a2], 1 D[a2], 2 B[b2], 3 D[b2]
// of the lane's DIAGONAL pair (a2, b2) at q2 = lane >> 4 (s = 0) and q2 = 4 (s = 1) -- depend on (span, lane) only: built
// once per handle as [span][lane][8], so that the loop fetches them with four 16-byte loads (were eight 8-byte ones).
__global__ __launch_bounds__(64) void tp3_t2pack_kernel(const double* __restrict__ B2, const double* __restrict__ D2, double* __restrict__ pack) {
  constexpr int NB = T3_NB, NQ = T3_NQ;
  const int lane = threadIdx.x, c16 = lane & 15, kk = lane >> 4;
  const int pa2 = c16 >> 2, pb2 = (pa2 + (c16 & 3)) & 3;
  const double* B = B2 + (int64_t)blockIdx.x * NB * NQ;
  const double* D = D2 + (int64_t)blockIdx.x * NB * NQ;
  double* out = pack + ((int64_t)blockIdx.x * 64 + lane) * 8;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int q = s == 0 ? kk : NQ - 1;
    out[4 * s + 0] = B[pa2 * NQ + q];
    out[4 * s + 1] = D[pa2 * NQ + q];
    out[4 * s + 2] = B[pb2 * NQ + q];
    out[4 * s + 3] = D[pb2 * NQ + q];
  }
}

#ifndef T3_LOOP_INC            // (scratch/p3_asm_variants.sh: a loop generated with other options, for timing experiments)
#define T3_LOOP_INC "tp3_contract_loop.inc"
#endif
#include T3_LOOP_INC

__global__ __launch_bounds__(64) void tp3_contract_asm_kernel(TensorArgs p) {
  constexpr int NB = T3_NB, NQ = T3_NQ, PS = T3_PS;
  extern __shared__ __align__(16) double carry[];   // [16 (a1, b1)][4 slots, the last one zero][64 lanes], then the direction-1 table [6][2][4]
  const int lane = threadIdx.x;
  const int J = (int)(blockIdx.x % 3), I = (int)((blockIdx.x / 3) % 3);
  const int64_t col = blockIdx.x / 9;
  const int eu = (int)(col % p.box_n[0]), ev = (int)(col / p.box_n[0]);
  const int n_seq = p.box_n[2];
  const int64_t e_step = (int64_t)p.box_n[0] * p.box_n[1];
  const int c16 = lane & 15, kk = lane >> 4, pa = c16 >> 2, pb = c16 & 3;
  double* par = carry + lane;                              // slot k of this lane: par[k * 64]
  auto put_int = [&](int slot, unsigned v) { *reinterpret_cast<unsigned*>(par + slot * 64) = v; };
  auto put_ptr = [&](int slot, const void* q) { *reinterpret_cast<uint64_t*>(par + slot * 64) = (uint64_t)(uintptr_t)q; };
  auto put_i64 = [&](int slot, int64_t v) { *reinterpret_cast<int64_t*>(par + slot * 64) = v; };

  // direction 0 (S3 B operands), as in tp3_contract_kernel
  {
    const double* B0 = p.tabB[0] + (int64_t)(p.box_begin[0] + eu) * NB * NQ;
    const double* D0 = p.tabD[0] + (int64_t)(p.box_begin[0] + eu) * NB * NQ;
    const double Ba = B0[pa * NQ + kk], Da = D0[pa * NQ + kk], Bb = B0[pb * NQ + kk], Db = D0[pb * NQ + kk];
    const double Bax = B0[pa * NQ + 4], Dax = D0[pa * NQ + 4], Bbx = B0[pb * NQ + 4], Dbx = D0[pb * NQ + 4];
#pragma unroll
    for (int v = 0; v < 4; ++v) par[(T3A_P_BS0 + v) * 64] = ((v & 1) ? Da : Ba) * ((v & 2) ? Db : Bb);
    par[(T3A_P_BS0X + 0) * 64] = ((kk & 1) ? Dax : Bax) * ((kk & 2) ? Dbx : Bbx);     // the plane q0 = 4: k = lane group = variant
    par[(T3A_P_BS0X + 1) * 64] = 0.0;
  }
  // direction 1 (S2 coefficients) through the LDS table behind the carry, as in tp3_contract_kernel
  double* tl = carry + 16 * 4 * 64;
  if (lane < 2 * NB * NQ) {
    const int v = lane / (NB * NQ), a = (lane % (NB * NQ)) / NQ, q = lane % NQ;
    tl[q * 8 + v * 4 + a] = ((v ? p.tabD[1] : p.tabB[1]) + (int64_t)(p.box_begin[1] + ev) * NB * NQ)[a * NQ + q];
  } else if (lane < 2 * NB * NQ + 8) {
    tl[lane] = 0.0;
  }
  __syncthreads();
  {
    const int ti = lane & 15, th = ti >> 3, tk = ti & 7;
    auto plane_row = [&](int r) -> int { return kk == 0 ? r : ((kk == 1 && r < 2) ? r + 3 : NQ); };
    par[(T3A_P_TA + 0) * 64] = tl[8 * th + tk];
    par[(T3A_P_TA + 1) * 64] = tl[8 * (2 + th) + tk];
    par[(T3A_P_TA + 2) * 64] = tl[8 * (th == 0 ? 4 : plane_row(0)) + tk];
    par[(T3A_P_TA + 3) * 64] = tl[8 * plane_row(1 + th) + tk];
  }
  // S1 A operands: the points of this lane (byte offsets inside a record field)
  const int ptU = (c16 & 3) + NQ * (c16 >> 2);
  const bool vrow = c16 < 4 || c16 == 4 || c16 == 8 || c16 == 12 || c16 == 5 || c16 == 9;
  const int vq0 = c16 < 4 ? c16 : 4;
  const int vq1 = c16 < 4 ? 4 : (c16 == 5 ? 3 : (c16 == 9 ? 4 : c16 / 4 - 1));
  const int ptV = vrow ? vq0 + NQ * vq1 : 0;
  const unsigned cl_addr = (unsigned)(uintptr_t)par;   // (the low half of a shared-aperture address is the LDS offset)
  put_int(T3A_P_INT + 0, 16u * (unsigned)(ptU + NQ * NQ * kk));            // (a pair field holds 16 bytes per point)
  put_int(T3A_P_INT + 1, 16u * (unsigned)((kk == 1 ? ptV : ptU) + NQ * NQ * (NQ - 1)));
  put_int(T3A_P_INT + 2, 16u * (unsigned)(ptV + NQ * NQ * kk));
  put_int(T3A_P_INT + 3, 8u * (unsigned)(ptU + NQ * NQ * kk));             // (the single field mn = 8: 8 bytes per point)
  put_int(T3A_P_INT + 4, 8u * (unsigned)((kk == 1 ? ptV : ptU) + NQ * NQ * (NQ - 1)));
  put_int(T3A_P_INT + 5, 8u * (unsigned)(ptV + NQ * NQ * kk));
  put_int(T3A_P_INT + 6, 64u * (unsigned)lane);                            // this lane's eight table values in tp3_t2pack_kernel's array
  // where this lane reads slot s of a pair's carry: the lane group that stored register s + 1 as final reads the zero slot
  put_int(T3A_P_INT + 7, cl_addr + (kk == 3 ? 3u * 512u : 0u));
  put_int(T3A_P_INT + 8, cl_addr + (kk == 2 ? 2u * 512u : 0u));
  put_int(T3A_P_INT + 9, cl_addr + (kk == 1 ? 1u * 512u : 0u));
  put_int(T3A_P_INT + 10, cl_addr);
  // the final entry with b2 = 0 of lane groups 1..3 (register 4 - kk -> slot 3 - kk); lane group 0 has none and doubles
  // lane group 1's (same slot of lane + 16, same destination: the same value stored twice, no execution mask)
  put_int(T3A_P_INT + 11, kk == 0 ? cl_addr + 16u * 8u + 2u * 512u : cl_addr + (unsigned)(3 - kk) * 512u);
  put_int(T3A_P_INT + 12, 8u * (unsigned)(pa * 192 + J * 64 + kk * 8 + pb * 2));                        // rows a2 = 0: + 4 a1 192 + (b1 / 2) 32 (+ b1 % 2)
  put_int(T3A_P_INT + 13, 8u * (unsigned)((pa + 16 * (3 - (kk ? kk : 1))) * 48 + J * 16 + pb * 2));    // rows a2 >= 1, b2 = 0: + 4 a1 48 + (b1 / 2) 8 (+ b1 % 2)
  put_int(T3A_P_INT + 14, kk == 0 ? 0xffffffffu : 0u);
  put_int(T3A_P_INT + 15, kk == 1 ? 0xffffffffu : 0u);
  const int64_t e0 = eu + (int64_t)p.box_n[0] * ev;
  put_ptr(T3A_P_REC, p.scratch_pt + e0 * (int64_t)(T3_REC * PS) + (int64_t)(I * 3 + J) * 9 * PS);      // (t3_rec_block)
  put_i64(T3A_P_RSTRIDE, e_step * (int64_t)(T3_REC * PS) * 8);
  put_ptr(T3A_P_PIECE, p.scratch_k + (e0 * 3 + I) * (int64_t)T3_PIECE);
  put_i64(T3A_P_PSTRIDE, e_step * 3 * (int64_t)T3_PIECE * 8);
  put_ptr(T3A_P_B2, p.t2pack + (int64_t)p.box_begin[2] * 64 * 8);
  put_i64(T3A_P_NSEQ, n_seq);
  __syncthreads();
  T3_ASM_LOOP(cl_addr);
  // (the loop ends on s_waitcnt vmcnt(0) lgkmcnt(0): the carry of the last element lies in LDS)
  const double* cl = par;
  double* out1_prev = p.scratch_k + ((e0 + e_step * (n_seq - 1)) * 3 + I) * (int64_t)T3_PIECE + 3072 +
                      (pa + 16 * (3 - (kk ? kk : 1))) * 48 + J * 16 + pb * 2;
  // the b2 = 0 entries of the last element
  {
    const double* fl = cl + (kk == 0 ? 3 : 3 - kk) * 64;
#pragma unroll
    for (int ab = 0; ab < 16; ++ab) {
      const double v = fl[ab * 4 * 64];
      if (kk > 0) out1_prev[(ab / NB) * 4 * 48 + ((ab % NB) / 2) * 8 + (ab % NB) % 2] = v;
    }
  }
  // what the last element of the column would have passed on: rows a2 >= 1, b2 >= 1 -> the tail of (column, i)
  double* tail = p.scratch_tail + ((e0 * 3 + I) * (int64_t)T3_TAIL) + pa * 144 + J * 48 + pb;
#pragma unroll
  for (int ab = 0; ab < 16; ++ab)
#pragma unroll
    for (int r = 1; r < 4; ++r) {
      const int b2 = (r + kk) & 3;
      const double v = cl[(ab * 4 + r - 1) * 64];
      if (b2) tail[(4 * (ab / NB) + 16 * (r - 1)) * 144 + (ab % NB) * 12 + (b2 - 1) * 4] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// phase 2
// ------------------------------------------------------------------------------------------------
// One wave per (node A of the shard's node box, component i).  WITH_K 0: residual rows only.
// Four waves per SIMD (round 4): 133 registers left room for three; capped at 128 (two spilled) the gather of BASELINE
// configuration 3 takes 9.61 instead of 10.15 ms on the same box, same bits (it waits on memory three quarters of its cycles;
// its 33 KB of LDS per workgroup admit four workgroups per CU)
template<int WITH_K>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void tp3_gather_kernel(TensorArgs p, int64_t n_rows) {
  constexpr int P = 3, NB = T3_NB, ND = T3_ND;
  constexpr int LMAX = 3 * 343;
  __shared__ double img_all[WITH_K ? 4 : 1][LMAX + 3];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t R = (int64_t)blockIdx.x * 4 + wave;
  if (R >= n_rows) return;
  const int64_t Al = R / 3;
  const int I = (int)(R % 3);
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1], n2 = p.n_ctrl[2];
  const int m0 = p.win_n[0], m1 = p.win_n[1];
  const int A0 = p.win_begin[0] + (int)(Al % m0), A1 = p.win_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.win_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  const int bx0 = p.box_begin[0], bx1 = p.box_begin[1], bx2 = p.box_begin[2];
  const int ex_lo = max(A0 - P, bx0), ex_hi = min(A0, bx0 + p.box_n[0] - 1);
  const int ey_lo = max(A1 - P, bx1), ey_hi = min(A1, bx1 + p.box_n[1] - 1);
  const int ez_lo = max(A2 - P, bx2), ez_hi = min(A2, bx2 + p.box_n[2] - 1);
  if (ex_lo > ex_hi || ey_lo > ey_hi || ez_lo > ez_hi) return;
  auto elem = [&](int ex, int ey, int ez) -> int64_t {
    return (ex - bx0) + (int64_t)p.box_n[0] * ((ey - bx1) + (int64_t)p.box_n[1] * (ez - bx2));
  };
  if constexpr (WITH_K) {
    double* img = img_all[wave];
    const int lo0 = max(A0 - P, 0), lo1 = max(A1 - P, 0), lo2 = max(A2 - P, 0);
    const int w0 = min(A0 + P, n0 - 1) - lo0 + 1, w1 = min(A1 + P, n1 - 1) - lo1 + 1, w2 = min(A2 + P, n2 - 1) - lo2 + 1;
    const int L = 3 * w0 * w1 * w2;
    for (int t = lane; t < L; t += 64) img[t] = 0.0;
    // lane = position in a row of a piece.  Rows a2 = 0: [b1][b2][b0] per load, the three loads are j = 0, 1, 2;
    // rows a2 >= 1: [j][b1][b0] (b2 = 0) in one load; tail rows: [b1][b2 - 1][b0] per load, 48 lanes
    // (b1 = 2 (lane >> 5) + (lane & 1), b2 = (lane >> 3) & 3, b0 = (lane >> 1) & 3  |  j = lane >> 4, b1 = 2 ((lane >> 3) & 1) + (lane & 1))
    const int toff_full = 3 * (((lane >> 1) & 3) + w0 * ((2 * (lane >> 5) + (lane & 1)) + w1 * ((lane >> 3) & 3)));
    const int toff_b20 = 3 * (((lane >> 1) & 3) + w0 * (2 * ((lane >> 3) & 1) + (lane & 1))) + (lane >> 4);
    const int toff_tail = 3 * ((lane & 3) + w0 * (lane / 12 + w1 * ((lane >> 2) % 3 + 1)));
    const int last_ez = bx2 + p.box_n[2] - 1;
    __builtin_amdgcn_wave_barrier();
    for (int ez = ez_lo; ez <= ez_hi; ++ez) {
      const int a2 = A2 - ez;
      const bool tail = a2 > 0 && ez == last_ez;
      if (a2 == 0) {
        // whole rows (192 values = three loads per element): the four elements along x of one ey in flight together
        // (more in flight costs registers, i.e. resident waves: measured 11.1 / 12.2 / 14.8 ms for 1 / 2 / 4 element rows)
        for (int ey = ey_lo; ey <= ey_hi; ++ey) {
          double v[NB][3];
#pragma unroll
          for (int c = 0; c < NB; ++c) {
            const int ex = ex_lo + c;
            const bool in = ex <= ex_hi;
            const int exx = in ? ex : ex_lo;
            const double* row = p.scratch_k + (elem(exx, ey, ez) * 3 + I) * (int64_t)T3_PIECE + ((A0 - exx) + NB * (A1 - ey)) * 192 + lane;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[c][j] = in ? row[j * 64] : 0.0;
          }
#pragma unroll
          for (int c = 0; c < NB; ++c) {
            const int ex = ex_lo + c;
            if (ex <= ex_hi) {
              const int tb = 3 * ((ex - lo0) + w0 * ((ey - lo1) + w1 * (ez - lo2)));
#pragma unroll
              for (int j = 0; j < 3; ++j) img[tb + toff_full + j] += v[c][j];
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      } else {
        // the 48 values at b2 = 0 (one load per element): all 16 elements of this ez in flight together
        double v[NB * NB];
#pragma unroll
        for (int c = 0; c < NB * NB; ++c) {
          const int ex = ex_lo + c % NB, ey = ey_lo + c / NB;
          const bool in = ex <= ex_hi && ey <= ey_hi;
          const int exx = in ? ex : ex_lo, eyy = in ? ey : ey_lo;
          const double* piece = p.scratch_k + (elem(exx, eyy, ez) * 3 + I) * (int64_t)T3_PIECE;
          const int ar = (A0 - exx) + NB * (A1 - eyy) + 16 * (a2 - 1);
          v[c] = (in && lane < 48) ? piece[3072 + ar * 48 + lane] : 0.0;
        }
#pragma unroll
        for (int c = 0; c < NB * NB; ++c) {
          const int ex = ex_lo + c % NB, ey = ey_lo + c / NB;
          if (ex <= ex_hi && ey <= ey_hi && lane < 48) {
            const int tb = 3 * ((ex - lo0) + w0 * ((ey - lo1) + w1 * (ez - lo2)));
            img[tb + toff_b20] += v[c];
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      constexpr int EYB = 1;
      for (int ey0 = ey_lo; tail && ey0 <= ey_hi; ey0 += EYB) {
        double v[EYB * NB][3];
        {
          // the column ends inside the support of A: the pairs (a2 >= 1, b2 >= 1) were kept by its last element
#pragma unroll
          for (int c = 0; c < EYB * NB; ++c) {
            const int ex = ex_lo + c % NB, ey = ey0 + c / NB;
            const bool in = ex <= ex_hi && ey <= ey_hi && lane < 48;
            const int exx = (ex <= ex_hi && ey <= ey_hi) ? ex : ex_lo, eyy = (ex <= ex_hi && ey <= ey_hi) ? ey : ey_lo;
            const int64_t colm = (exx - bx0) + (int64_t)p.box_n[0] * (eyy - bx1);
            const double* row = p.scratch_tail + (colm * 3 + I) * (int64_t)T3_TAIL
                                + ((A0 - exx) + NB * (A1 - eyy) + 16 * (a2 - 1)) * 144 + lane;
#pragma unroll
            for (int j = 0; j < 3; ++j) v[c][j] = in ? row[j * 48] : 0.0;
          }
#pragma unroll
          for (int c = 0; c < EYB * NB; ++c) {
            const int ex = ex_lo + c % NB, ey = ey0 + c / NB;
            if (ex <= ex_hi && ey <= ey_hi && lane < 48) {
              const int tb = 3 * ((ex - lo0) + w0 * ((ey - lo1) + w1 * (ez - lo2)));
#pragma unroll
              for (int j = 0; j < 3; ++j) img[tb + toff_tail + j] += v[c][j];
            }
            __builtin_amdgcn_wave_barrier();
          }
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    // u, r and the CSR may live in the caller's node numbering (perm: lexicographic -> caller's id, e.g. MFEM's NURBS dof
    // map): row perm[A], and entry t of the lexicographic window at the rank of its permuted column inside that row
    const int64_t beg = p.rowptr[(p.perm ? p.perm[A] : A) * 3 + I];
    double* dst = p.A + beg;
    const double* base = p.A_base + beg;     // (the row's old values: A itself, or the caller's base array)
    if (p.perm) {
      const uint16_t* pos = p.nbr_pos16 + A * 343;
      for (int t = lane; t < L; t += 64) {
        const int k = 3 * (int)pos[t / 3] + t % 3;
        dst[k] = base[k] + p.grad_factor * img[t];
      }
    } else {
      for (int t = lane; t < L; t += 64) dst[t] = base[t] + p.grad_factor * img[t];
    }
  }
  // residual row: lane = element (dz, dy, dx) of the 4 x 4 x 4 neighbourhood, fixed-shape tree sum
  // (a tangent assembly keeps its residual rows here, hidden behind the value rows: as a launch of their own -- the kernel below --
  // they cost 0.4 ms more, profiles/r05_cfg3_horner_ab.txt)
  {
    const int dz = lane >> 4, dy = (lane >> 2) & 3, dx = lane & 3;
    const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
    const bool in = ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
    const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
    const int64_t el = in ? elem(ex, ey, ez) : 0;
    // [element][a][i], i fastest: the three rows of a node -- three waves of this workgroup -- read the same 64 sectors (one
    // 8-byte read per element and row is a 32-byte sector fetched; with [element][i][a] the residual rows cost 3.2 GB of
    // fetch per assembly for 0.4 GB of pieces)
    double rs = in ? p.scratch_r[(el * ND + a) * 3 + I] : 0.0;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) rs += __shfl_down(rs, off, 64);
    if (lane == 0) p.r[(p.perm ? p.perm[A] : A) * 3 + I] += rs;
  }
}

// residual-only assemblies: one wave per NODE (its three rows), lane = element (dz, dy, dx) of the 4 x 4 x 4 neighbourhood --
// the three values a lane needs are 24 adjacent bytes of scratch_r[element][a][i]; the same fixed-shape tree sums as the
// per-row form above (tp3_gather_kernel<0>: three waves per node, each with one 8-byte read per lane), i.e. the same bits
__global__ __launch_bounds__(256) void tp3_residual_gather_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 3, NB = T3_NB, ND = T3_ND;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int64_t Al = (int64_t)blockIdx.x * 4 + wave;
  if (Al >= n_nodes) return;
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1];
  const int m0 = p.win_n[0], m1 = p.win_n[1];
  const int A0 = p.win_begin[0] + (int)(Al % m0), A1 = p.win_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.win_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  const int bx0 = p.box_begin[0], bx1 = p.box_begin[1], bx2 = p.box_begin[2];
  const int ex_lo = max(A0 - P, bx0), ex_hi = min(A0, bx0 + p.box_n[0] - 1);
  const int ey_lo = max(A1 - P, bx1), ey_hi = min(A1, bx1 + p.box_n[1] - 1);
  const int ez_lo = max(A2 - P, bx2), ez_hi = min(A2, bx2 + p.box_n[2] - 1);
  if (ex_lo > ex_hi || ey_lo > ey_hi || ez_lo > ez_hi) return;
  const int dz = lane >> 4, dy = (lane >> 2) & 3, dx = lane & 3;
  const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
  const bool in = ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
  const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
  const int64_t el = in ? (ex - bx0) + (int64_t)p.box_n[0] * ((ey - bx1) + (int64_t)p.box_n[1] * (ez - bx2)) : 0;
  const double* q = p.scratch_r + (el * ND + a) * 3;
  double rs[3];
#pragma unroll
  for (int I = 0; I < 3; ++I) rs[I] = in ? q[I] : 0.0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1)
#pragma unroll
    for (int I = 0; I < 3; ++I) rs[I] += __shfl_down(rs[I], off, 64);
  if (lane == 0) {
    double* r = p.r + (p.perm ? p.perm[A] : A) * 3;
#pragma unroll
    for (int I = 0; I < 3; ++I) r[I] += rs[I];
  }
}

}  // namespace

bool tensor_p3_ready(const mimi_hip_domain_s* h) {
  // the structured pattern (lexicographic numbering, or a permuted one with its window ranks), no repeated interior knots
  return h->dim == 3 && h->degree[0] == 3 && h->degree[1] == 3 && h->degree[2] == 3 && h->nq1[0] == 5 &&
         (h->structured_csr || h->structured_perm) && h->first_is_identity;
}

// the pre-pass kernel of a material kind (one instantiation per kind: see tp3_point_kernel)
template<int GRAD>
static auto t3_point_kernel_of(int kind) -> void (*)(TensorArgs) {
  switch (kind) {
  case MIMI_HIP_MAT_NEOHOOKEAN:
  case MIMI_HIP_MAT_J2: return tp3_point_kernel<0, GRAD>;
  case MIMI_HIP_MAT_STVK: return tp3_point_kernel<MIMI_HIP_MAT_STVK, GRAD>;
  case MIMI_HIP_MAT_J2LINEAR: return tp3_point_kernel<MIMI_HIP_MAT_J2LINEAR, GRAD>;
  case MIMI_HIP_MAT_J2SIMO: return tp3_point_kernel<MIMI_HIP_MAT_J2SIMO, GRAD>;
  default: return tp3_point_kernel<MIMI_HIP_MAT_J2LOG, GRAD>;
  }
}

void launch_tensor_p3(mimi_hip_domain_s* h, int grad, TensorArgs a) {
  const int kind = h->mat.m.kind;
  h->scratch_r.resize((size_t)h->n_el * 3 * T3_ND);
  a.scratch_r = h->scratch_r.ptr;
  const int64_t n_cols = (int64_t)a.box_n[0] * a.box_n[1];
  if (grad) {
    h->scratch_pt.resize((size_t)h->n_el * T3_REC * T3_PS);
    h->scratch_k.resize((size_t)h->n_el * 3 * T3_PIECE);
    h->scratch_tail.resize((size_t)n_cols * 3 * T3_TAIL);
    a.scratch_pt = h->scratch_pt.ptr;
    a.scratch_k = h->scratch_k.ptr;
    a.scratch_tail = h->scratch_tail.ptr;
    if (!h->t3_t2pack.ptr) {
      h->t3_t2pack.resize((size_t)h->el_total[2] * 64 * 8);
      hipLaunchKernelGGL(tp3_t2pack_kernel, dim3((unsigned)h->el_total[2]), dim3(64), 0, h->stream, a.tabB[2], a.tabD[2], h->t3_t2pack.ptr);
      MH_HIP(hipGetLastError());
    }
    a.t2pack = h->t3_t2pack.ptr;
  }
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[0], h->stream));
  if (h->phase_select != 2) {
    hipLaunchKernelGGL(grad ? t3_point_kernel_of<1>(kind) : t3_point_kernel_of<0>(kind), dim3((unsigned)h->n_el), dim3(128), 0, h->stream, a);
    MH_HIP(hipGetLastError());
  }
  h->phase_has_prepass = true;
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[3], h->stream));
  if (grad && h->phase_select != 2) {
    // MIMI_HIP_P3_CONTRACT=cxx: the compiler-scheduled form of the same arithmetic (the A/B reference of the bitwise test)
    // (read per launch, so that one process can run both: tests/test_tensor_p3_gpu.py)
    const char* variant = getenv("MIMI_HIP_P3_CONTRACT");
    const bool cxx = variant && !strcmp(variant, "cxx");
    hipLaunchKernelGGL(cxx ? tp3_contract_kernel : tp3_contract_asm_kernel, dim3((unsigned)(n_cols * 9)), dim3(64),
                       (16 * 4 * 64 + 48) * sizeof(double), h->stream, a);
    MH_HIP(hipGetLastError());
  }
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[1], h->stream));
  const int64_t n_rows = (int64_t)a.win_n[0] * a.win_n[1] * a.win_n[2] * 3;   // (node window: the shard's nodes unless a gather asks for a part)
  if (h->phase_select == 1) {
    // integrate only
  } else if (grad)
    hipLaunchKernelGGL(tp3_gather_kernel<1>, dim3((unsigned)((n_rows + 3) / 4)), dim3(256), 0, h->stream, a, n_rows);
  else
    hipLaunchKernelGGL(tp3_residual_gather_kernel, dim3((unsigned)((n_rows / 3 + 3) / 4)), dim3(256), 0, h->stream, a, n_rows / 3);
  MH_HIP(hipGetLastError());
  if (h->phase_timing) MH_HIP(hipEventRecord(h->phase_ev[2], h->stream));
}

void launch_tensor_p3_post(mimi_hip_domain_s* h, TensorArgs a) {
  const int kind = h->mat.m.kind;
  hipLaunchKernelGGL(t3_point_kernel_of<2>(kind), dim3((unsigned)h->n_el), dim3(128), 0, h->stream, a);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
