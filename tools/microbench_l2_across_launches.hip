// Does an XCD's L2 keep the lines a kernel wrote (plain stores) for the NEXT launch on the same stream (gfx950, ROCm 7.2)?
// Kernel W: workgroup b writes chunk b (32 KB).  Kernel R (next launch): workgroup b reads chunk (b + shift) % n and reduces it.
// Workgroups are dealt round-robin to the 8 XCDs, so shift = 0 (and 8) reads what the same XCD wrote, shift = 1 what another
// XCD wrote.  16 MB in all (half of the aggregate L2).  If L2 keeps the lines across the boundary, shift 0 / 8 read faster.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_l2_across_launches.hip -o tools/bin/microbench_l2_across_launches
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int kChunkDoubles = 4096;      // 32 KB
__global__ __launch_bounds__(256) void write_chunks(double* buf, double v) {
  double* c = buf + (size_t)blockIdx.x * kChunkDoubles;
  for (int i = threadIdx.x; i < kChunkDoubles; i += 256) c[i] = v + i;
}
__global__ __launch_bounds__(256) void rmw_chunks(double* buf, int n, int shift, double* sink) {
  double* c = buf + (size_t)((blockIdx.x + shift) % n) * kChunkDoubles;
  double s = 0;
  for (int i = threadIdx.x; i < kChunkDoubles; i += 256) { const double v = c[i]; s += v; c[i] = v * 0.5 + 1.0; }
  if (s == 123.456) sink[0] = s;
}
int main() {
  CHECK(hipSetDevice(0));
  const int n = 512;
  double *buf = nullptr, *sink = nullptr;
  CHECK(hipMalloc(&buf, sizeof(double) * (size_t)n * kChunkDoubles));
  CHECK(hipMalloc(&sink, 64));
  hipStream_t s; CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int shift : {0, 1, 8, 3, 0, 1, 16, 4}) {
    std::vector<float> us;
    for (int rep = 0; rep < 21; ++rep) {
      write_chunks<<<n, 256, 0, s>>>(buf, 1.0);
      rmw_chunks<<<n, 256, 0, s>>>(buf, n, 0, sink);          // a first read-modify-write pass, same mapping as the writer
      (void)hipEventRecord(e0, s);
      rmw_chunks<<<n, 256, 0, s>>>(buf, n, shift, sink);      // the measured pass: same XCD (shift % 8 == 0) or another
      (void)hipEventRecord(e1, s);
      (void)hipEventSynchronize(e1);
      float t = 0; (void)hipEventElapsedTime(&t, e0, e1); us.push_back(t * 1000.0f);
    }
    std::sort(us.begin(), us.end());
    printf("shift %2d (%s XCD): read-modify-write of 16 MB in %.1f us (median of 21, incl. ~7 us of event bracket)\n", shift, shift % 8 == 0 ? "same   " : "another", us[10]);
  }
  return 0;
}
