// Two-phase tangent assembly for structured p = 2 patches (the benchmarked path): what both phase-1
// kernels (kernels_tensor_wgsym.hpp, kernels_tensor_wgs.hpp) share.
//
// Why two phases: with colour-partitioned read-modify-write inside the integration kernel the
// wave is bound by its CU's outstanding-request capacity -- every (element, i) touches 243 CSR row
// segments of 72 B, i.e. partial cache lines, both ways (profiles/r01_colour_rmw_*).  So the
// integration kernel only STORES: its row pieces go, coalesced, to a dense scratch, and a second
// kernel that owns CSR rows gathers them and does ONE coalesced read-modify-write per row.
//
//   phase 1  stores, for every (element, i), a piece of 2187 slots in scratch_k and the residual piece
//            scratch_r[element][i][a].  Entries shared with the next element of the walked (third) axis are
//            carried inside the kernel, so each (node pair, element column) is stored exactly once, by the
//            highest element of the column that contains both nodes.  Piece layout (a = a0 + 3 a1 + 9 a2 local
//            row node, (b2,b1,b0) local column node, j column component), what phase 2 reads contiguously:
//              [0, 729)      rows a < 9 (a2 = 0):            a 81 + b2 27 + b1 9 + b0 3 + j
//              [729, 1215)   rows a >= 9, b2 = 0:            729 + (a - 9) 27 + b1 9 + b0 3 + j
//              [1215, 2187)  rows a >= 9, b2 = 1, 2 (written by the last element of a column only):
//                                                            1215 + (a - 9) 54 + (b2 - 1) 27 + b1 9 + b0 3 + j
//   phase 2  tensor_p2_kernel below.
// Results are bitwise reproducible; nothing is atomic.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor.hpp"

namespace mimi_hip {

typedef double mh_d4 __attribute__((ext_vector_type(4)));

// phase 2: gather.  Requires: the structured CSR pattern (lexicographic numbering, or a permuted one with
// the window ranks of permuted_window_kernel) and first[e] == e (no repeated interior knots).
//
// One wave per node A (its three CSR rows).  The wave walks the <= 27 elements that contain A; of
// each piece (element, i) it needs row a = local index of A: 81 contiguous doubles [b2][b1][b0][j]
// (or the first 27, b2 = 0, when the element is not the one that stored the (a2 >= 1) entries).
// lane = position in that row, so the reads are contiguous runs; the position of (b, j) in A's CSR
// row is  t = t_base(element) + t_off(lane)  with a per-lane constant t_off.  Row sums are built
// in LDS (one wave adds piece after piece: fixed order, no conflicts inside an instruction), then
// A[row] += grad_factor * sum is one coalesced read-modify-write per row.
__global__ __launch_bounds__(256) void tensor_p2_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, NB = 3, ND = 27, NROW = 81, NK = ND * NROW;
  constexpr int LMAX = 3 * 125;
  __shared__ double sums_all[4][3][LMAX + 1];
  const int wave = threadIdx.x >> 6;
  const int64_t Al = (int64_t)blockIdx.x * 4 + wave;   // node index inside the shard's node box
  const int lane = threadIdx.x & 63;
  if (Al >= n_nodes) return;
  double (*sums)[LMAX + 1] = sums_all[wave];
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1], n2 = p.n_ctrl[2];
  const int m0 = p.box_n[0] + P, m1 = p.box_n[1] + P;
  const int A0 = p.box_begin[0] + (int)(Al % m0), A1 = p.box_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.box_begin[2] + (int)(Al / ((int64_t)m0 * m1));
  const int64_t A = A0 + (int64_t)n0 * (A1 + (int64_t)n1 * A2);
  const int64_t gA = p.perm ? p.perm[A] : A;
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
  for (int t = lane; t < L; t += 64) {
    sums[0][t] = 0.0;
    sums[1][t] = 0.0;
    sums[2][t] = 0.0;
  }
  // lane constants: row positions k0 = lane and k1 = lane + 64 (< 81), k = ((b2 3 + b1) 3 + b0) 3 + j
  const int k1 = lane + 64;
  const int toff0 = 3 * ((lane / 3) % 3 + w0 * ((lane / 9) % 3 + w1 * (lane / 27))) + lane % 3;
  const int toff1 = 3 * ((k1 / 3) % 3 + w0 * ((k1 / 9) % 3 + w1 * (k1 / 27))) + k1 % 3;
  __builtin_amdgcn_wave_barrier();
  for (int ez = ez_lo; ez <= ez_hi; ++ez) {
    const int a2 = A2 - ez;
    const bool unit_end = ez == last_ez || (ez - bx2) % p.seg_len == p.seg_len - 1;
    const int nb = (a2 == 0 || unit_end) ? NROW : ND;
    const bool act0 = lane < nb, act1 = k1 < nb;
    // P2_BATCH pieces in flight per wave (registers against occupancy)
#ifndef P2_BATCH
#define P2_BATCH 3
#endif
    for (int c0 = 0; c0 < 9; c0 += P2_BATCH) {
      double v0[P2_BATCH][3], v1[P2_BATCH][3];
#pragma unroll
      for (int cc = 0; cc < P2_BATCH; ++cc) {
        const int c = c0 + cc;
        const int ey = ey_lo + c / 3, ex = ex_lo + c % 3;
        const bool in = c < 9 && ey <= ey_hi && ex <= ex_hi;
        const int a = (A0 - ex) + NB * ((A1 - ey) + NB * a2);
        const double* piece = p.scratch_k + (elem(in ? ex : ex_lo, in ? ey : ey_lo, ez) * 3) * (int64_t)NK;
        // row a of the piece: 81 contiguous values (a2 = 0), or 27 (b2 = 0) [+ 54 (b2 = 1, 2) from the last element]
        const int a9 = in ? a - 9 : 0;
        const int o0 = !in ? 0 : (a2 == 0 ? a * NROW + lane : (lane < ND ? 9 * NROW + a9 * ND + lane : 15 * NROW + a9 * 54 + lane - ND));
        const int o1 = !in ? 0 : (a2 == 0 ? a * NROW + k1 : 15 * NROW + a9 * 54 + k1 - ND);
#pragma unroll
        for (int I = 0; I < 3; ++I) {
          v0[cc][I] = (in && act0) ? piece[I * NK + o0] : 0.0;
          v1[cc][I] = (in && act1) ? piece[I * NK + o1] : 0.0;
        }
      }
#pragma unroll
      for (int cc = 0; cc < P2_BATCH; ++cc) {
        const int c = c0 + cc;
        const int ey = ey_lo + c / 3, ex = ex_lo + c % 3;
        const bool in = c < 9 && ey <= ey_hi && ex <= ex_hi;
        const int tbase = 3 * ((ex - lo0) + w0 * ((ey - lo1) + w1 * (ez - lo2)));
        if (in) {
#pragma unroll
          for (int I = 0; I < 3; ++I) {
            if (act0) sums[I][tbase + toff0] += v0[cc][I];
            if (act1) sums[I][tbase + toff1] += v1[cc][I];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int I = 0; I < 3; ++I) {
    double* row = p.A + p.rowptr[gA * 3 + I];
    if (p.perm) {
      const unsigned char* pos = p.nbr_pos + A * 125;
      for (int t = lane; t < L; t += 64) row[3 * (int)pos[t / 3] + t % 3] += p.grad_factor * sums[I][t];
    } else {
      for (int t = lane; t < L; t += 64) row[t] += p.grad_factor * sums[I][t];
    }
  }
  // residual rows: lane = element (dz, dy, dx) of the 3 x 3 x 3 neighbourhood, fixed-shape tree sum
  {
    const int dz = lane / 9, dy = (lane / 3) % 3, dx = lane % 3;
    const int ez = ez_lo + dz, ey = ey_lo + dy, ex = ex_lo + dx;
    const bool in = lane < ND && ez <= ez_hi && ey <= ey_hi && ex <= ex_hi;
    const int a = in ? (A0 - ex) + NB * ((A1 - ey) + NB * (A2 - ez)) : 0;
    const int64_t e = in ? elem(ex, ey, ez) : 0;
    double rs[3];
#pragma unroll
    for (int I = 0; I < 3; ++I) rs[I] = in ? p.scratch_r[(e * 3 + I) * ND + a] : 0.0;
#pragma unroll
    for (int I = 0; I < 3; ++I) {
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) rs[I] += __shfl_down(rs[I], off, 32);
    }
    if (lane == 0) {
#pragma unroll
      for (int I = 0; I < 3; ++I) p.r[gA * 3 + I] += rs[I];
    }
  }
}


inline bool two_phase_supported(const mimi_hip_domain_s* h) {
  // (the phase-1 kernels always walk the third direction, whatever the shape of the element box)
  return (h->structured_csr || h->structured_perm) && h->first_is_identity;
}

inline void launch_tensor_p2(mimi_hip_domain_s* h, const TensorArgs& a) {
  const int64_t n_nodes = (int64_t)(a.box_n[0] + 2) * (a.box_n[1] + 2) * (a.box_n[2] + 2);   // nodes of the shard
  hipLaunchKernelGGL(tensor_p2_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, a, n_nodes);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
