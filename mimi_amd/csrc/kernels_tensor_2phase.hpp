// Two-phase tangent assembly for structured p = 2 patches (the benchmarked path): what both phase-1
// kernels (kernels_tensor_wgsym.hpp, kernels_tensor_wgs.hpp) share.
//
// Why two phases: with colour-partitioned read-modify-write inside the integration kernel the
// wave is bound by its CU's outstanding-request capacity -- every (element, i) touches 243 CSR row
// segments of 72 B, i.e. partial cache lines, both ways (profiles/r01_colour_rmw_*).  So the
// integration kernel only STORES: its row pieces go, coalesced, to a dense scratch, and a second
// kernel that owns CSR rows gathers them and does ONE coalesced read-modify-write per row.
//
//   phase 1  stores, for every (element, i), the 27 x 81 row piece scratch_k[element][i][a][b2][b1][b0][j]
//            and the residual piece scratch_r[element][i][a].  Entries shared with the next element of
//            the walked (third) axis are carried inside the kernel, so each (node pair, element column)
//            is stored exactly once, by the highest element of the column that contains both nodes.
//   phase 2  tensor_p2_kernel below.
// Results are bitwise reproducible; nothing is atomic.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor.hpp"

namespace mimi_hip {

typedef double mh_d4 __attribute__((ext_vector_type(4)));

// phase 2: gather.  Requires: lexicographic numbering, structured CSR, first[e] == e (no repeated
// interior knots), walk axis == 2.
//
// One wave per node A (its three CSR rows).  The wave walks the <= 27 elements that contain A; of
// each piece (element, i) it needs row a = local index of A: 81 contiguous doubles [b2][b1][b0][j]
// (or the first 27, b2 = 0, when the element is not the one that stored the (a2 >= 1) entries).
// lane = position in that row, so the reads are contiguous runs; the position of (b, j) in A's CSR
// row is  t = t_base(element) + t_off(lane)  with a per-lane constant t_off.  Row sums are built
// in LDS (one wave adds piece after piece: fixed order, no conflicts inside an instruction), then
// A[row] += grad_factor * sum is one coalesced read-modify-write per row.
#ifndef P2_WAVES
#define P2_WAVES 5
#endif
__global__ __launch_bounds__(256, P2_WAVES) void tensor_p2_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, NB = 3, ND = 27, NROW = 81, NK = ND * NROW;
  constexpr int LMAX = 3 * 125;
  __shared__ double sums_all[4][LMAX + 1];
  const int wave = threadIdx.x >> 6;
  const int64_t gw = (int64_t)blockIdx.x * 4 + wave;   // one wave per CSR row (node A, component I)
  const int64_t Al = gw / 3;                            // node index inside the shard's node box
  const int I = (int)(gw % 3);
  const int lane = threadIdx.x & 63;
  if (Al >= n_nodes) return;
  double* sums = sums_all[wave];
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1], n2 = p.n_ctrl[2];
  // the nodes this shard's elements touch: box_begin[d] .. box_begin[d] + box_n[d] + P - 1 per direction
  const int m0 = p.box_n[0] + P, m1 = p.box_n[1] + P;
  const int A0 = p.box_begin[0] + (int)(Al % m0), A1 = p.box_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.box_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  // elements of THIS shard containing node A: e_d in [A_d - P, A_d] clipped to the box
  const int bx0 = p.box_begin[0], bx1 = p.box_begin[1], bx2 = p.box_begin[2];
  const int ex_lo = max(A0 - P, bx0), ex_hi = min(A0, bx0 + p.box_n[0] - 1);
  const int ey_lo = max(A1 - P, bx1), ey_hi = min(A1, bx1 + p.box_n[1] - 1);
  const int ez_lo = max(A2 - P, bx2), ez_hi = min(A2, bx2 + p.box_n[2] - 1);
  if (ex_lo > ex_hi || ey_lo > ey_hi || ez_lo > ez_hi) return;
  const int last_ez = bx2 + p.box_n[2] - 1;
  const int lo0 = max(A0 - P, 0), lo1 = max(A1 - P, 0), lo2 = max(A2 - P, 0);
  const int w0 = min(A0 + P, n0 - 1) - lo0 + 1, w1 = min(A1 + P, n1 - 1) - lo1 + 1, w2 = min(A2 + P, n2 - 1) - lo2 + 1;
  const int L = 3 * w0 * w1 * w2;
  auto elem = [&](int ex, int ey, int ez) -> int64_t {
    return (ex - bx0) + (int64_t)p.box_n[0] * ((ey - bx1) + (int64_t)p.box_n[1] * (ez - bx2));
  };
  for (int t = lane; t < L; t += 64) sums[t] = 0.0;
  // lane constants: row positions k0 = lane and k1 = lane + 64 (< 81), k = ((b2 3 + b1) 3 + b0) 3 + j
  const int k1 = lane + 64;
  const int toff0 = 3 * ((lane / 3) % 3 + w0 * ((lane / 9) % 3 + w1 * (lane / 27))) + lane % 3;
  const int toff1 = 3 * ((k1 / 3) % 3 + w0 * ((k1 / 9) % 3 + w1 * (k1 / 27))) + k1 % 3;
  __builtin_amdgcn_wave_barrier();
  for (int ez = ez_lo; ez <= ez_hi; ++ez) {
    const int a2 = A2 - ez;
    const int nb = (a2 == 0 || ez == last_ez) ? NROW : ND;
    const bool act0 = lane < nb, act1 = k1 < nb;
    // all nine pieces of this element layer in flight
    double v0[9], v1[9];
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int ey = ey_lo + c / 3, ex = ex_lo + c % 3;
      const bool in = ey <= ey_hi && ex <= ex_hi;
      const int a = (A0 - ex) + NB * ((A1 - ey) + NB * a2);
      const double* src = p.scratch_k + (elem(in ? ex : ex_lo, in ? ey : ey_lo, ez) * 3 + I) * (int64_t)NK + (in ? a : 0) * NROW;
      v0[c] = (in && act0) ? src[lane] : 0.0;
      v1[c] = (in && act1) ? src[k1] : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const int ey = ey_lo + c / 3, ex = ex_lo + c % 3;
      const bool in = ey <= ey_hi && ex <= ex_hi;
      const int tbase = 3 * ((ex - lo0) + w0 * ((ey - lo1) + w1 * (ez - lo2)));
      if (in) {
        if (act0) sums[tbase + toff0] += v0[c];
        if (act1) sums[tbase + toff1] += v1[c];
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
  __builtin_amdgcn_wave_barrier();
  const int64_t gA = p.perm ? p.perm[A] : A;   // the caller's id of node A
  {
    double* row = p.A + p.rowptr[gA * 3 + I];
    if (p.perm) {
      // permuted numbering: window neighbour t / 3 sits at rank nbr_pos inside the row
      const unsigned char* pos = p.nbr_pos + A * 125;
      for (int t = lane; t < L; t += 64) row[3 * (int)pos[t / 3] + t % 3] += p.grad_factor * sums[t];
    } else {
      for (int t = lane; t < L; t += 64) row[t] += p.grad_factor * sums[t];
    }
  }
  // residual row: lane = element (dz, dy, dx) of the 3 x 3 x 3 neighbourhood, fixed-shape tree sum
  {
    const int dz = lane / 9, dy = (lane / 3) % 3, dx = lane % 3;
    const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
    const bool in = lane < ND && ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
    const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
    const int64_t e = in ? elem(ex, ey, ez) : 0;
    double rs = in ? p.scratch_r[(e * 3 + I) * ND + a] : 0.0;
#pragma unroll
    for (int off = 16; off >= 1; off >>= 1) rs += __shfl_down(rs, off, 32);
    if (lane == 0) p.r[gA * 3 + I] += rs;
  }
}

inline bool two_phase_supported(const mimi_hip_domain_s* h) {
  // (the phase-1 kernels always walk the third direction, whatever the shape of the element box)
  return (h->structured_csr || h->structured_perm) && h->first_is_identity;
}

}  // namespace mimi_hip
