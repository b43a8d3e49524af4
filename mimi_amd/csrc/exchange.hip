// Packing of CSR rows for the one exchange step of a sharded assembly: the rows of the node planes a slab
// shares with its neighbour leave as ONE message -- [n_rows residual entries][the rows' values, row after row] -- and
// what arrives is added into the rows it belongs to.  The reference's counterpart is the in-process reduction of the
// per-thread global arrays (integrators/nonlinear_base.hpp:90-151); there is no distributed path to cite.
//
// One wave per row: the row's start comes from rowptr, the lanes stride over its values, so the value array is read and
// written in whole 512-byte runs and no per-value index array is read (an index_select / index_add_ over precomputed
// positions reads 8 bytes of index per 8 bytes of payload).  Nothing is atomic: the rows of a message are distinct.
#include <hip/hip_runtime.h>

#include <exception>

#include "../../include/mimi_hip.h"
#include "common.hpp"

namespace mimi_hip {
namespace {

constexpr int XW = 4;   // waves (rows) per workgroup

// MODE 0: zero the rows;  1: message <- rows;  2: rows += message
template<int MODE>
__global__ __launch_bounds__(64 * XW) void rows_kernel(const int64_t* __restrict__ rowptr, const int64_t* __restrict__ rows,
                                                       const int64_t* __restrict__ offsets, int64_t n_rows,
                                                       double* __restrict__ r, double* __restrict__ A,
                                                       double* __restrict__ msg) {
  const int64_t k = (int64_t)blockIdx.x * XW + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (k >= n_rows) return;
  const int64_t row = rows[k];
  if (lane == 0) {
    if (MODE == 0) r[row] = 0.0;
    if (MODE == 1) msg[k] = r[row];
    if (MODE == 2) r[row] += msg[k];
  }
  if (!A) return;
  const int64_t start = rowptr[row];
  const int len = (int)(rowptr[row + 1] - start);
  double* a = A + start;
  double* m = MODE == 0 ? nullptr : msg + offsets[k];
  for (int t = lane; t < len; t += 64) {
    if (MODE == 0) a[t] = 0.0;
    if (MODE == 1) m[t] = a[t];
    if (MODE == 2) a[t] += m[t];
  }
}

template<typename F>
int guarded_x(F&& f) {
  try {
    f();
    return 0;
  } catch (const std::exception& e) {
    set_last_error(e.what());
    return 1;
  } catch (...) {
    set_last_error("unknown error");
    return 1;
  }
}

template<int MODE>
void launch_rows(void* stream, const int64_t* rowptr, const int64_t* rows, const int64_t* offsets, int64_t n_rows,
                 double* r, double* A, double* msg) {
  if (n_rows == 0) return;
  if (n_rows < 0 || !rowptr || !rows || !r) fail("rows exchange: null argument");
  if (MODE != 0 && !msg) fail("rows exchange: no message buffer");
  if (MODE != 0 && A && !offsets) fail("rows exchange: offsets must be given with A");
  hipStream_t s = stream == MIMI_HIP_STREAM_NULL ? nullptr : (hipStream_t)stream;
  const int64_t blocks = (n_rows + XW - 1) / XW;
  hipLaunchKernelGGL(rows_kernel<MODE>, dim3((unsigned)blocks), dim3(64 * XW), 0, s, rowptr, rows, offsets, n_rows, r, A, msg);
  MH_HIP(hipGetLastError());
}

}  // namespace
}  // namespace mimi_hip

using namespace mimi_hip;

extern "C" {

int mimi_hip_rows_zero(void* stream, const int64_t* rowptr, const int64_t* rows, int64_t n_rows, double* r, double* A_values) {
  return guarded_x([&] { launch_rows<0>(stream, rowptr, rows, nullptr, n_rows, r, A_values, nullptr); });
}

int mimi_hip_rows_pack(void* stream, const int64_t* rowptr, const int64_t* rows, const int64_t* offsets, int64_t n_rows,
                       const double* r, const double* A_values, double* message) {
  return guarded_x([&] {
    launch_rows<1>(stream, rowptr, rows, offsets, n_rows, const_cast<double*>(r), const_cast<double*>(A_values), message);
  });
}

int mimi_hip_rows_unpack_add(void* stream, const int64_t* rowptr, const int64_t* rows, const int64_t* offsets, int64_t n_rows,
                             const double* message, double* r, double* A_values) {
  return guarded_x([&] { launch_rows<2>(stream, rowptr, rows, offsets, n_rows, r, A_values, const_cast<double*>(message)); });
}

}  // extern "C"
