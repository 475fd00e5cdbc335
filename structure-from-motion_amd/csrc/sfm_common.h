// sfm_common.h — library context, error plumbing and small device utilities shared by the
// translation units of libsfm_hip.so.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "../../include/sfm_hip.h"
#include "sfm_math.h"

namespace sfm {

struct Context {
  bool inited = false;
  int device = -1;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int num_cus = 256;
};

Context& ctx();
int ensure_init();
void set_error(const char* fmt, ...);
int hip_fail(hipError_t e, const char* what, int line);

#define SFM_HIP(call)                                                   \
  do {                                                                  \
    hipError_t _e = (call);                                             \
    if (_e != hipSuccess) return ::sfm::hip_fail(_e, #call, __LINE__);  \
  } while (0)

#define SFM_TRY(call)            \
  do {                           \
    int _s = (call);             \
    if (_s != SFM_OK) return _s; \
  } while (0)

// RAII device buffer for the host-pointer convenience entry points.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t count) {
    n = count;
    if (count == 0) count = 1;
    hipError_t e = hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __LINE__);
    return SFM_OK;
  }
  int upload(const T* host, size_t count, hipStream_t s) {
    SFM_TRY(alloc(count));
    if (count) SFM_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    return SFM_OK;
  }
  int download(T* host, size_t count, hipStream_t s) const {
    if (count) SFM_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    return SFM_OK;
  }
};

// First-failure status word written by kernels: status[0] = code (0 = ok), status[1] = index.
__device__ __forceinline__ void report_status(int* status, int code, int index) {
  if (code != SFM_OK && atomicCAS(&status[0], 0, code) == 0) status[1] = index;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
  return v;
}

}  // namespace sfm
