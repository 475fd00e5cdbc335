// Does hipExtAnyOrderLaunch let a kernel start before its predecessor ON THE SAME STREAM has finished (gfx950, ROCm 7.2)?
// hip_ext.h says the flag "is not supported on AMD GFX9xx boards"; this measures it.  A = 64 workgroups spinning 100 us,
// B = 64 workgroups spinning 60 us (no LDS, 320 free CU slots either way).  Serial: ~165 us; overlapped: ~105 us.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_anyorder.hip -o tools/bin/microbench_anyorder
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void spin(int us, int* sink) {
  const unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 100ull * (unsigned long long)us) __builtin_amdgcn_s_sleep(2);
  if (us < 0) sink[0] = 1;
}

int main() {
  CHECK(hipSetDevice(0));
  int* sink = nullptr;
  CHECK(hipMalloc(&sink, 64));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  auto median_of = [&](auto&& body) -> float {
    std::vector<float> ms;
    for (int rep = 0; rep < 15; ++rep) {
      (void)hipEventRecord(e0, s);
      body();
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      float t = 0;
      (void)hipEventElapsedTime(&t, e0, e1);
      ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1000.0f;
  };
  for (int i = 0; i < 3; ++i) spin<<<64, 256, 0, s>>>(10, sink);
  CHECK(hipDeviceSynchronize());
  printf("A then B, plain launches:                 %.1f us\n", median_of([&] { spin<<<64, 256, 0, s>>>(100, sink); spin<<<64, 256, 0, s>>>(60, sink); }));
  printf("A then B with hipExtAnyOrderLaunch:       %.1f us\n", median_of([&] {
           spin<<<64, 256, 0, s>>>(100, sink);
           hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 60, sink);
         }));
  printf("A, B(any order), C plain (must wait both): %.1f us (overlap: ~125; serial ~185)\n", median_of([&] {
           spin<<<64, 256, 0, s>>>(100, sink);
           hipExtLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, 60, sink);
           spin<<<64, 256, 0, s>>>(20, sink);
         }));
  hipError_t err = hipGetLastError();
  printf("last error: %s\n", hipGetErrorString(err));
  return 0;
}
