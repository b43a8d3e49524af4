// Packing of CSR rows for the one exchange step of a sharded assembly: the rows of the node planes a slab
// shares with its neighbour leave as ONE message -- [n_rows residual entries][the rows' values, row after row] -- and
// what arrives is added into the rows it belongs to.  The reference's counterpart is the in-process reduction of the
// per-thread global arrays (integrators/nonlinear_base.hpp:90-151); there is no distributed path to cite.
//
// One wave per row: the row's start comes from rowptr, the lanes stride over its values, so the value array is read and
// written in whole 512-byte runs and no per-value index array is read (an index_select / index_add_ over precomputed
// positions reads 8 bytes of index per 8 bytes of payload).  Nothing is atomic: the rows of a message are distinct.
#include <hip/hip_runtime.h>

#include <algorithm>
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

// The same message with the rows TRIMMED to the entries the sender can have written (round 5): a slab's elements touch only
// the columns of their own node planes, so of a shared row only the part inside the sender's planes is not zero by
// construction -- 3 of the 5 column planes of a degree-2 row, 40 % less on the wire.  The entries are then no longer whole
// rows: positions[i] = place of message value n_rows + i in the value array (lane = entry; the positions are runs of
// w0 * dim * selected planes, read and written in whole lines; 8 bytes of index per 8 bytes of payload is what the wire saves
// five times over at xGMI's rate).  MODE 1: message <- entries;  2: entries += message
template<int MODE>
__global__ __launch_bounds__(256) void entries_kernel(const int64_t* __restrict__ rows, int64_t n_rows,
                                                      const int64_t* __restrict__ positions, int64_t n_positions,
                                                      double* __restrict__ r, double* __restrict__ A, double* __restrict__ msg) {
  const int64_t n = n_rows + (A ? n_positions : 0);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    if (i < n_rows) {
      if (MODE == 1) msg[i] = r[rows[i]];
      else r[rows[i]] += msg[i];
    } else {
      const int64_t at = positions[i - n_rows];
      if (MODE == 1) msg[i] = A[at];
      else A[at] += msg[i];
    }
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

template<int MODE>
void launch_entries(void* stream, const int64_t* rows, int64_t n_rows, const int64_t* positions, int64_t n_positions,
                    double* r, double* A, double* msg) {
  if (n_rows < 0 || n_positions < 0) fail("entries exchange: negative count");
  const int64_t n = n_rows + (A ? n_positions : 0);
  if (n == 0) return;
  if (!msg || (n_rows && (!rows || !r))) fail("entries exchange: null argument");
  if (A && n_positions && !positions) fail("entries exchange: positions must be given with A");
  hipStream_t s = stream == MIMI_HIP_STREAM_NULL ? nullptr : (hipStream_t)stream;
  const int64_t blocks = std::min<int64_t>((n + 255) / 256, 1 << 16);
  hipLaunchKernelGGL(entries_kernel<MODE>, dim3((unsigned)blocks), dim3(256), 0, s, rows, n_rows, positions, n_positions, r, A, msg);
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

int mimi_hip_entries_pack(void* stream, const int64_t* rows, int64_t n_rows, const int64_t* positions, int64_t n_positions,
                          const double* r, const double* A_values, double* message) {
  return guarded_x([&] {
    launch_entries<1>(stream, rows, n_rows, positions, n_positions, const_cast<double*>(r), const_cast<double*>(A_values), message);
  });
}

int mimi_hip_entries_unpack_add(void* stream, const int64_t* rows, int64_t n_rows, const int64_t* positions, int64_t n_positions,
                                const double* message, double* r, double* A_values) {
  return guarded_x([&] {
    launch_entries<2>(stream, rows, n_rows, positions, n_positions, r, A_values, const_cast<double*>(message));
  });
}

}  // extern "C"
