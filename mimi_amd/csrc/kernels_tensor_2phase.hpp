// Two-phase tangent assembly for structured p = 2 patches (the benchmarked path): what both phase-1
// kernels (kernels_tensor_wgsym.hpp, kernels_tensor_wgs.hpp) share.
//
// Why two phases: with colour-partitioned read-modify-write inside the integration kernel the
// wave is bound by its CU's outstanding-request capacity -- every (element, i) touches 243 CSR row
// segments of 72 B, i.e. partial cache lines, both ways (profiles/r01_colour_rmw_*).  So the
// integration kernel only STORES: its row pieces go, coalesced, to a dense scratch, and a second
// kernel that owns CSR rows gathers them and does ONE coalesced read-modify-write per row.
//
//   phase 1  stores, for every element, a block of 3 x 2187 slots in scratch_k -- the pieces of the three row components I,
//            interleaved ROW BY ROW -- and the residual piece scratch_r[element][a][i].  Entries shared with the next
//            element of the walked (third) axis are carried inside the kernel, so each (node pair, element column) is
//            stored exactly once, by the highest element of the column that contains both nodes.  Block layout (P2Block;
//            a = a0 + 3 a1 + 9 a2 local row node, (b2,b1,b0) local column node, j column component):
//              [0, 2187)      rows a < 9 (a2 = 0):            a 243 + I 81 + b2 27 + b1 9 + b0 3 + j
//              [2187, 3645)   rows a >= 9, b2 = 0:            2187 + (a - 9) 81 + I 27 + b1 9 + b0 3 + j
//              [3645, 6561)   rows a >= 9, b2 = 1, 2 (written by the last element of a column / segment only):
//                                                             3645 + (a - 9) 162 + I 54 + (b2 - 1) 27 + b1 9 + b0 3 + j
//            so that everything phase 2 needs of one element for one node A -- the three rows (a, I) -- is ONE run of
//            243 (or 81, or 81 + 162) doubles: scattered reads of 1 944 B reach 6.0 TB/s on this chip, of 648 B 4.5, of
//            216 B 2.1 (scratch/run_bench.hip), and the rows used to be separate runs of 648 / 216 B.
//   phase 2  tensor_p2_kernel below.
// Results are bitwise reproducible; nothing is atomic.
#pragma once

#include <hip/hip_runtime.h>

#include "kernels_tensor.hpp"

namespace mimi_hip {

typedef double mh_d4 __attribute__((ext_vector_type(4)));

struct P2Block {
  static constexpr int size = 3 * 27 * 81;          // 6561 doubles per element
  static constexpr int off_b20 = 9 * 243;           // 2187
  static constexpr int off_tail = off_b20 + 18 * 81;  // 3645
  // where piece I keeps its carried rows: + (a - 9) 162 + (b2 - 1) 27 + b1 9 + b0 3 + j
  MH_DEV static double* carry_of(double* E, int I) { return E + off_tail + I * 54; }
};

// phase 2: gather.  Requires: the structured CSR pattern (lexicographic numbering, or a permuted one with
// the window ranks of permuted_window_kernel) and first[e] == e (no repeated interior knots).
//
// One wave per node A (its three CSR rows).  The wave walks the <= 27 elements that contain A; of each element block it
// needs the rows (a, I = 0..2) of a = local index of A: one run of 243 doubles [I][b2][b1][b0][j] when a2 = 0, else one of
// 81 ([I][b1][b0][j], b2 = 0) plus, when the element is the one that stored the (a2 >= 1, b2 >= 1) entries, one of 162.
// lane + 64 t = position in the run, so the reads are contiguous; the position of (I, b, j) in the wave's LDS image of
// A's three rows is  I (LMAX + 1) + t_base(element) + t_off(b, j)  with per-lane constants for the run shape at hand.
// Row sums are built in LDS (one wave adds run after run: fixed order, no conflicts inside an instruction), then
// A[row] += grad_factor * sum is one coalesced read-modify-write per row.
// Four waves per SIMD (round 4): the kernel waits on memory two thirds of its cycles and compiled to 129 registers -- one
// over the 128 that admit a fourth wave (its 36 KB of LDS per workgroup admit exactly four workgroups per CU).  With the
// attribute: 127 registers, no spills, phase 2 2.78 -> 2.66 ms on the same box, same bits.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void tensor_p2_kernel(TensorArgs p, int64_t n_nodes) {
  constexpr int P = 2, NB = 3, ND = 27, NROW = 81;
  constexpr int LMAX = 3 * 125, SROW = LMAX + 1;
  __shared__ double sums_all[4][3 * SROW];
  const int wave = threadIdx.x >> 6;
  const int64_t Al = (int64_t)blockIdx.x * 4 + wave;   // node index inside the shard's node box
  const int lane = threadIdx.x & 63;
  if (Al >= n_nodes) return;
  double* sums = sums_all[wave];
  const int n0 = p.n_ctrl[0], n1 = p.n_ctrl[1], n2 = p.n_ctrl[2];
  const int m0 = p.win_n[0], m1 = p.win_n[1];
  const int A0 = p.win_begin[0] + (int)(Al % m0), A1 = p.win_begin[1] + (int)((Al / m0) % m1);
  const int A2 = p.win_begin[2] + (int)(Al / ((int64_t)m0 * m1));
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
    sums[t] = 0.0;
    sums[SROW + t] = 0.0;
    sums[2 * SROW + t] = 0.0;
  }
  // position of column (b, j), k = ((b2 3 + b1) 3 + b0) 3 + j, relative to the element's first column in A's row
  auto toff = [&](int k) -> int { return 3 * ((k / 3) % 3 + w0 * ((k / 9) % 3 + w1 * (k / 27))) + k % 3; };
  __builtin_amdgcn_wave_barrier();
  for (int ez = ez_lo; ez <= ez_hi; ++ez) {
    const int a2 = A2 - ez;
    const bool full = a2 == 0;                                                                  // (wave-uniform)
    const bool tail = !full && (ez == last_ez || (ez - bx2) % p.seg_len == p.seg_len - 1);     // (wave-uniform)
    // lane constants of the run shape: value lane + 64 t of the first run (243 = [I][81] or 81 = [I][27]) ...
    const int per = full ? NROW : ND, len0 = 3 * per;
    int loff0[4];
    bool act0[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int v = lane + 64 * t;
      act0[t] = v < len0;
      const int vv = act0[t] ? v : 0;
      loff0[t] = (vv / per) * SROW + toff(vv % per);
    }
    // ... and of the run of carried rows (162 = [I][54], b2 = 1, 2)
    int loff1[3];
    bool act1[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int v = lane + 64 * t;
      act1[t] = v < 162;
      const int vv = act1[t] ? v : 0;
      loff1[t] = (vv / 54) * SROW + toff(ND + vv % 54);
    }
    // P2_BATCH elements in flight per wave (registers against occupancy)
#ifndef P2_BATCH
#define P2_BATCH 3
#endif
    for (int c0 = 0; c0 < 9; c0 += P2_BATCH) {
      double v0[P2_BATCH][4], v1[P2_BATCH][3];
#pragma unroll
      for (int cc = 0; cc < P2_BATCH; ++cc) {
        const int c = c0 + cc;
        const int ey = ey_lo + c / 3, ex = ex_lo + c % 3;
        const bool in = c < 9 && ey <= ey_hi && ex <= ex_hi;
        const int a = (A0 - ex) + NB * ((A1 - ey) + NB * a2);
        const double* E = p.scratch_k + elem(in ? ex : ex_lo, in ? ey : ey_lo, ez) * (int64_t)P2Block::size;
        const int a9 = in && !full ? a - 9 : 0;
        const double* run0 = E + (full ? (in ? a : 0) * 3 * NROW : P2Block::off_b20 + a9 * NROW) + lane;
#pragma unroll
        for (int t = 0; t < 2; ++t) v0[cc][t] = (in && act0[t]) ? run0[64 * t] : 0.0;
        if (full) {
#pragma unroll
          for (int t = 2; t < 4; ++t) v0[cc][t] = (in && act0[t]) ? run0[64 * t] : 0.0;
        }
        if (tail) {
          const double* run1 = E + P2Block::off_tail + a9 * 162 + lane;
#pragma unroll
          for (int t = 0; t < 3; ++t) v1[cc][t] = (in && act1[t]) ? run1[64 * t] : 0.0;
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
          for (int t = 0; t < 2; ++t)
            if (act0[t]) sums[tbase + loff0[t]] += v0[cc][t];
          if (full) {
#pragma unroll
            for (int t = 2; t < 4; ++t)
              if (act0[t]) sums[tbase + loff0[t]] += v0[cc][t];
          }
          if (tail) {
#pragma unroll
            for (int t = 0; t < 3; ++t)
              if (act1[t]) sums[tbase + loff1[t]] += v1[cc][t];
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    }
  }
  __builtin_amdgcn_wave_barrier();
  if (p.perm) {
#pragma unroll
    for (int I = 0; I < 3; ++I) {
      const int64_t beg = p.rowptr[gA * 3 + I];
      double* row = p.A + beg;
      const double* base = p.A_base + beg;
      const unsigned char* pos = p.nbr_pos + A * 125;
      for (int t = lane; t < L; t += 64) {
        const int k = 3 * (int)pos[t / 3] + t % 3;
        row[k] = base[k] + p.grad_factor * sums[I * SROW + t];
      }
    }
  } else {
    // all 18 loads of the three rows in flight together (L <= 375 = 6 x 64 - 9), then the adds and the stores: one
    // round trip to memory per wave instead of one per 64 values
    constexpr int NT = (LMAX + 63) / 64;
    double* row[3];
    double old[3][NT];
#pragma unroll
    for (int I = 0; I < 3; ++I) {
      const int64_t beg = p.rowptr[gA * 3 + I] + lane;
      row[I] = p.A + beg;
      const double* base = p.A_base + beg;
#pragma unroll
      for (int q = 0; q < NT; ++q) old[I][q] = lane + 64 * q < L ? base[64 * q] : 0.0;
    }
#pragma unroll
    for (int I = 0; I < 3; ++I)
#pragma unroll
      for (int q = 0; q < NT; ++q)
        if (lane + 64 * q < L) row[I][64 * q] = old[I][q] + p.grad_factor * sums[I * SROW + lane + 64 * q];
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
    for (int I = 0; I < 3; ++I) rs[I] = in ? p.scratch_r[(e * ND + a) * 3 + I] : 0.0;     // (three adjacent doubles: one sector per element instead of three)
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
  const int64_t n_nodes = (int64_t)a.win_n[0] * a.win_n[1] * a.win_n[2];   // nodes of the shard (or of the gather window)
  hipLaunchKernelGGL(tensor_p2_kernel, dim3((unsigned)((n_nodes + 3) / 4)), dim3(256), 0, h->stream, a, n_nodes);
  MH_HIP(hipGetLastError());
}

}  // namespace mimi_hip
