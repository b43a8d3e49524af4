// Issue costs on one SIMD of gfx950, in shader cycles per instruction (s_memtime around an unrolled loop), with one
// and with two waves per SIMD: fp64 vector instructions, the fp64 matrix instruction, and what can be issued in the
// shadow of a matrix instruction (non-fp64 vector moves, accumulator-file reads, LDS reads, fp64 FMAs).
// Build: hipcc --offload-arch=gfx950 -O3 -o issue_bench scratch/issue_bench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1);} } while (0)
typedef double d4 __attribute__((ext_vector_type(4)));

__device__ inline unsigned long long now() {
  unsigned long long t;
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  return t;
}

// MODE: what the body holds.  NM matrix instructions, each followed by NF fillers of kind MODE.
//  0: fp64 FMA (independent chains)   1: v_mov_b32 (non-fp64)   2: v_accvgpr_read   3: ds_read_b64   4: v_mul_f64   5: v_add_f64
//  6: fp64 FMA, all fillers AFTER all matrix instructions (not interleaved)
//  7: global_store_dwordx2 (4 x 128-byte segments per wave instruction, from the accumulation file)
//  8: global_load_dwordx2 into the accumulation file     9: all matrix instructions on ONE accumulator (dependent chain)
// 10: ds_write_b64 from the accumulation file    11: global_store_dwordx2 with 16 of 64 lanes active (4 x 32-byte segments)
// 14: v_fmac_f64 with a DPP operand (row_newbcast: lane N of the row of 16 to all lanes of the row)   15: v_mov_b64 with the same
// 16: global_store_dwordx4 (64 lanes x 16 B contiguous = eight 128-byte lines)   17: global_store_dwordx2, one line per lane
// 12: v_mfma_f64_4x4x4_4b_f64 (four independent 4x4x4 blocks, one double of the result per lane) as the filler, NM = 0
template<int MODE, int NM, int NF>
__global__ void bench(unsigned long long* out, double* sink, int iters, double* buf) {
  __shared__ double lds[1024];
  lds[threadIdx.x & 1023] = threadIdx.x;
  __syncthreads();
  d4 acc[4];
  for (int k = 0; k < 4; ++k) acc[k] = d4{0, 0, 0, 0};
  double f[16];
  for (int k = 0; k < 16; ++k) f[k] = 1.0 + k * 1e-3 + threadIdx.x * 1e-6;
  unsigned m[16];
  for (int k = 0; k < 16; ++k) m[k] = k + threadIdx.x;
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 2e-3;
  const double c0 = 0.999, c1 = 1e-9;
  const unsigned ldsaddr = (threadIdx.x & 63) * 8;
  // the store pattern of tp3_contract_kernel: 16 lanes contiguous (128 B), 4 such segments 1536 B apart; a wave's own 64 KB
  const int l = threadIdx.x & 63;
  double* gp = buf + ((size_t)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)) * 8192 + (l & 15) + (l >> 4) * 192;
  // the masked stores of tp3_contract_kernel: lanes (pa, pb) of one lane group: 4 doubles contiguous, 4 such 384 B apart
  double* gp2 = buf + ((size_t)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)) * 8192 + (l & 3) + ((l >> 4) & 3) * 48;
  // 16 bytes per lane, contiguous over the wave (1 KB); and 8 bytes per lane, every lane its own 128-byte line
  double* gp4 = buf + ((size_t)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)) * 8192 + l * 2;
  double* gp1 = buf + ((size_t)(blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64)) * 8192 + l * 16;
  unsigned long long t0 = now();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int rep = 0; rep < 4; ++rep) {
      if constexpr (MODE == 6) {
#pragma unroll
        for (int k = 0; k < NM; ++k)
          asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[k & 3]) : "v"(a), "v"(b));
#pragma unroll
        for (int j = 0; j < NF * NM; ++j) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[j & 15]) : "v"(c0), "v"(c1));
      } else {
#pragma unroll
        for (int k = 0; k < (NM ? NM : 1); ++k) {
          if constexpr (NM > 0) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[MODE == 9 ? 0 : (k & 3)]) : "v"(a), "v"(b));
#pragma unroll
          for (int j = 0; j < NF; ++j) {
            const int q = (k * NF + j) & 15;
            if constexpr (MODE == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(f[q]) : "v"(c0), "v"(c1));
            if constexpr (MODE == 1) asm volatile("v_mov_b32 %0, %1" : "=v"(m[q]) : "v"(m[(q + 1) & 15]));
            if constexpr (MODE == 2) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(m[q]));
            if constexpr (MODE == 3) asm volatile("ds_read_b64 %0, %1" : "=v"(f[q]) : "v"(ldsaddr));
            if constexpr (MODE == 4) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(f[q]) : "v"(c0));
            if constexpr (MODE == 5) asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[q]) : "v"(c1));
            if constexpr (MODE == 7) asm volatile("global_store_dwordx2 %0, a[2:3], off offset:%1" ::"v"(gp), "n"(((0) & 3) * 8) : "memory", "a2", "a3");
            if constexpr (MODE == 8) asm volatile("global_load_dwordx2 a[4:5], %0, off" ::"v"(gp) : "memory", "a4", "a5");
            if constexpr (MODE == 11) asm volatile("s_mov_b64 exec, %1\n\tglobal_store_dwordx2 %0, a[2:3], off\n\ts_mov_b64 exec, -1" ::"v"(gp2), "s"(0x000f000f000f000full) : "memory", "a2", "a3");
            if constexpr (MODE == 10) asm volatile("ds_write_b64 %0, a[2:3]" ::"v"(ldsaddr) : "memory", "a2", "a3");
            if constexpr (MODE == 14) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(f[q]) : "v"(a), "v"(b));
            if constexpr (MODE == 15) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "=v"(f[q]) : "v"(a));
            if constexpr (MODE == 16) asm volatile("global_store_dwordx4 %0, a[4:7], off" ::"v"(gp4) : "memory", "a4", "a5", "a6", "a7");
            if constexpr (MODE == 17) asm volatile("global_store_dwordx2 %0, a[2:3], off" ::"v"(gp1) : "memory", "a2", "a3");
            if constexpr (MODE == 12) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(f[q]) : "v"(a), "v"(b));
            if constexpr (MODE == 13) asm volatile("v_mfma_f64_4x4x4_4b_f64 %0, %1, %2, %0" : "+v"(f[0]) : "v"(a), "v"(b));
          }
        }
      }
    }
    if constexpr (MODE == 3 || MODE == 10) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (MODE == 7 || MODE == 8 || MODE == 11 || MODE == 16 || MODE == 17) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  unsigned long long t1 = now();
  double s = 0;
  for (int k = 0; k < 4; ++k) s += acc[k][0] + acc[k][1] + acc[k][2] + acc[k][3];
  for (int k = 0; k < 16; ++k) s += f[k] + m[k];
  if (s == 12345.678) sink[0] = s;
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

static double* g_buf = nullptr;   // 256 blocks x 8 waves x 8192 doubles: every wave stores into / loads from its own 64 KB

template<int MODE, int NM, int NF>
static void run(const char* what, unsigned long long* d_out, double* d_sink) {
  double* buf = g_buf;
  if (!buf) { printf("no buffer\n"); exit(1); }
  const int iters = 2000;
  for (int threads : {256, 512}) {
    const int blocks = 256, waves = blocks * threads / 64;
    hipLaunchKernelGGL((bench<MODE, NM, NF>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, 10, buf);
    hipLaunchKernelGGL((bench<MODE, NM, NF>), dim3(blocks), dim3(threads), 0, 0, d_out, d_sink, iters, buf);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(waves);
    CK(hipMemcpy(h.data(), d_out, waves * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    const double per_body = (double)h[waves / 2] / (iters * 4.0);
    const int nm = NM ? NM : 1;
    printf("%-44s %d wave(s)/SIMD: %8.1f cycles per body (%d mfma + %d fillers) = %6.2f per mfma-group, %6.2f per filler beyond 64/mfma\n",
           what, threads / 256, per_body, NM, NF * nm, per_body / nm,
           NF ? (per_body - 64.0 * NM) / (NF * nm) : 0.0);
  }
}

int main() {
  unsigned long long* d_out;
  double* d_sink;
  CK(hipMalloc(&d_out, 1 << 20));
  CK(hipMalloc(&d_sink, 64));
  double* d_buf;
  CK(hipMalloc(&d_buf, (size_t)256 * 8 * 8192 * sizeof(double)));
  CK(hipMemset(d_buf, 0, (size_t)256 * 8 * 8192 * sizeof(double)));
  g_buf = d_buf;
  run<0, 0, 16>("16 independent v_fma_f64", d_out, d_sink);
  run<4, 0, 16>("16 independent v_mul_f64", d_out, d_sink);
  run<5, 0, 16>("16 independent v_add_f64", d_out, d_sink);
  run<1, 0, 16>("16 v_mov_b32", d_out, d_sink);
  run<2, 0, 16>("16 v_accvgpr_read_b32", d_out, d_sink);
  run<3, 0, 16>("16 ds_read_b64", d_out, d_sink);
  run<0, 4, 0>("4 mfma_f64_16x16x4 alone", d_out, d_sink);
  run<1, 4, 4>("mfma + 4 v_mov_b32 each", d_out, d_sink);
  run<1, 4, 8>("mfma + 8 v_mov_b32 each", d_out, d_sink);
  run<1, 4, 12>("mfma + 12 v_mov_b32 each", d_out, d_sink);
  run<1, 4, 16>("mfma + 16 v_mov_b32 each", d_out, d_sink);
  run<2, 4, 8>("mfma + 8 v_accvgpr_read each", d_out, d_sink);
  run<2, 4, 12>("mfma + 12 v_accvgpr_read each", d_out, d_sink);
  run<3, 4, 4>("mfma + 4 ds_read_b64 each", d_out, d_sink);
  run<0, 4, 1>("mfma + 1 v_fma_f64 each", d_out, d_sink);
  run<0, 4, 2>("mfma + 2 v_fma_f64 each", d_out, d_sink);
  run<0, 4, 4>("mfma + 4 v_fma_f64 each", d_out, d_sink);
  run<0, 4, 8>("mfma + 8 v_fma_f64 each", d_out, d_sink);
  run<6, 4, 8>("4 mfma then 32 v_fma_f64 (not interleaved)", d_out, d_sink);
  run<9, 4, 0>("4 mfma on ONE accumulator (dependent)", d_out, d_sink);
  run<7, 0, 16>("16 global_store_dwordx2 (burst)", d_out, d_sink);
  run<8, 0, 16>("16 global_load_dwordx2 (burst)", d_out, d_sink);
  run<7, 4, 1>("mfma + 1 global_store each", d_out, d_sink);
  run<7, 4, 2>("mfma + 2 global_store each", d_out, d_sink);
  run<8, 4, 1>("mfma + 1 global_load each", d_out, d_sink);
  run<8, 4, 2>("mfma + 2 global_load each", d_out, d_sink);
  run<16, 0, 16>("16 global_store_dwordx4 (1 KB contiguous)", d_out, d_sink);
  run<16, 4, 1>("mfma + 1 global_store_dwordx4 each", d_out, d_sink);
  run<17, 0, 16>("16 global_store_dwordx2, one line per lane", d_out, d_sink);
  run<11, 0, 16>("16 masked global_store (16 lanes)", d_out, d_sink);
  run<11, 4, 1>("mfma + 1 masked global_store each", d_out, d_sink);
  run<11, 4, 2>("mfma + 2 masked global_store each", d_out, d_sink);
  run<14, 0, 16>("16 independent v_fmac_f64_dpp row_newbcast", d_out, d_sink);
  run<15, 0, 16>("16 v_mov_b64_dpp row_newbcast", d_out, d_sink);
  run<14, 4, 4>("mfma + 4 v_fmac_f64_dpp each", d_out, d_sink);
  run<12, 0, 16>("16 independent v_mfma_f64_4x4x4_4b", d_out, d_sink);
  run<13, 0, 16>("16 v_mfma_f64_4x4x4_4b on ONE accumulator", d_out, d_sink);
  run<10, 4, 2>("mfma + 2 ds_write_b64 each", d_out, d_sink);
  run<10, 4, 4>("mfma + 4 ds_write_b64 each", d_out, d_sink);
  return 0;
}
