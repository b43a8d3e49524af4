// Shared host-side plumbing of libmimi_hip: error reporting, device buffers, pointer
// classification.  No kernels here.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/mimi_hip.h"

namespace mimi_hip {

// utils/print.hpp:47-56 PrintAndThrowError -> std::runtime_error; the C ABI catches it,
// stores the text for mimi_hip_last_error() and returns non-zero.
struct Error : std::runtime_error {
  using std::runtime_error::runtime_error;
};

[[noreturn]] inline void fail(const char* fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof(buf), fmt, ap);
  va_end(ap);
  throw Error(buf);
}

#define MH_HIP(expr)                                                                       \
  do {                                                                                     \
    hipError_t e_ = (expr);                                                                \
    if (e_ != hipSuccess)                                                                  \
      ::mimi_hip::fail("HIP error %s at %s:%d: %s", hipGetErrorName(e_), __FILE__, __LINE__, \
                       hipGetErrorString(e_));                                             \
  } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize once per (kernel, device) instead of once per launch (it costs microseconds,
// which the small meshes of the reference's examples notice)
inline void ensure_dynamic_lds(const void* kernel, int bytes) {
  static std::mutex guard;
  static std::map<std::pair<const void*, int>, int> granted;
  int device = 0;
  MH_HIP(hipGetDevice(&device));
  std::lock_guard<std::mutex> lock(guard);
  int& have = granted[std::make_pair(kernel, device)];
  if (bytes > have) {
    MH_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
    have = bytes;
  }
}


void set_last_error(const std::string& s);

// true when p points to device memory of any kind
inline bool is_device_pointer(const void* p) {
  if (!p) return false;
  hipPointerAttribute_t attr;
  hipError_t e = hipPointerGetAttributes(&attr, p);
  if (e != hipSuccess) {
    (void)hipGetLastError();  // plain malloc'd host memory: not an error for us
    return false;
  }
  return attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged;
}

template<typename T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t count = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    count = 0;
  }
  void resize(size_t n) {
    if (n <= count) return;
    release();
    MH_HIP(hipMalloc(reinterpret_cast<void**>(&ptr), n * sizeof(T)));
    count = n;
  }
  // copy n elements from a host or device source
  void assign(const T* src, size_t n, hipStream_t s) {
    resize(n);
    if (n == 0) return;
    MH_HIP(hipMemcpyAsync(ptr, src, n * sizeof(T),
                          is_device_pointer(src) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s));
    if (!is_device_pointer(src)) MH_HIP(hipStreamSynchronize(s));
  }
};

// A caller-provided array that may live on the host (then mirrored in `stage`) or on the
// device (used in place).
template<typename T>
struct Mirror {
  T* dev = nullptr;      // what kernels use
  T* host = nullptr;     // non-null when the caller's buffer is on the host
  size_t count = 0;
  DeviceBuffer<T>* stage = nullptr;

  static Mirror in(const T* p, size_t n, DeviceBuffer<T>& stage, hipStream_t s) {
    Mirror m;
    m.count = n;
    if (is_device_pointer(p)) {
      m.dev = const_cast<T*>(p);
    } else {
      stage.resize(n);
      MH_HIP(hipMemcpyAsync(stage.ptr, p, n * sizeof(T), hipMemcpyHostToDevice, s));
      m.dev = stage.ptr;
      m.host = const_cast<T*>(p);
      m.stage = &stage;
    }
    return m;
  }
  // for += outputs: host contents are uploaded first, downloaded by finish()
  static Mirror inout(T* p, size_t n, DeviceBuffer<T>& stage, hipStream_t s) { return in(p, n, stage, s); }
  void finish(hipStream_t s) {
    if (host) MH_HIP(hipMemcpyAsync(host, dev, count * sizeof(T), hipMemcpyDeviceToHost, s));
  }
};

struct MaterialDev {
  mimi_hip_material m;
  double const_temperature_contribution;  // material_hardening.hpp:310-318
  double sigma_y_ref;                     // HardeningBase::SigmaY()
};

MaterialDev make_material_dev(const mimi_hip_material& m);

}  // namespace mimi_hip
