// Can a latency chain of small launches run UNDER a launch that fills the chip?  (gfx950; sizes the pipelined
// reduced solve of DESIGN.md section 4; not part of the product.)
//
//   big(n_wgs, us)      stands in for ba_schur_mfma: 512 threads, 144 KB of dynamic LDS per workgroup (so a CU hosts ONE
//                       of them), spinning on the wall clock for `us` microseconds
//   small(n_wgs, us)    stands in for a column step of the reduced solve: 256 threads, 33 KB of static LDS (or 0 / 16 KB)
//
// Cases (each: median of 21 repetitions, wall time between two events on stream 1, stream 2 joined back before the
// second event):
//   A      big alone
//   B      the chain of `nchain` dependent small launches alone
//   A|B    big on stream 1, the chain on stream 2 behind an event recorded BEFORE big -- with big on all 256 CUs, on
//          256 - 8 (one CU free per XCD: workgroups are dealt round-robin over the 8 XCDs) and on 256 - 16
//   A0,A1|B  two phases of big back to back on stream 1; the chain waits for phase 0 and runs under phase 1
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_overlap.hip -o tools/bin/microbench_overlap
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void spin_us(int us) {
  const unsigned long long t0 = wall_clock64();        // 100 MHz
  const unsigned long long ticks = 100ull * (unsigned long long)us;
  while (wall_clock64() - t0 < ticks && wall_clock64() - t0 < 100000000ull) __builtin_amdgcn_s_sleep(2);
}

__global__ __launch_bounds__(512) void big(int us, double* sink) {
  extern __shared__ double img[];
  img[threadIdx.x] = threadIdx.x;
  __syncthreads();
  spin_us(us);
  if (img[(threadIdx.x + 1) & 511] < 0) sink[0] = 1;
}

template <int LDS_DOUBLES>
__global__ __launch_bounds__(256) void small(int us, double* sink) {
  __shared__ double arena[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  arena[threadIdx.x % (LDS_DOUBLES > 0 ? LDS_DOUBLES : 1)] = threadIdx.x;
  __syncthreads();
  spin_us(us);
  if (arena[0] < 0) sink[0] = 1;
}

int main() {
  int dev = 0;
  CHECK(hipSetDevice(dev));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, dev));
  const int cus = prop.multiProcessorCount;
  printf("device: %s, %d CUs, LDS per block max %zu\n", prop.name, cus, (size_t)prop.sharedMemPerBlock);
  const size_t big_lds = 147456;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)big_lds));
  double* sink = nullptr;
  CHECK(hipMalloc(&sink, 64));
  hipStream_t s1, s2, s2hi;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  int lo_pri = 0, hi_pri = 0;
  CHECK(hipDeviceGetStreamPriorityRange(&lo_pri, &hi_pri));
  CHECK(hipStreamCreateWithPriority(&s2hi, hipStreamNonBlocking, hi_pri));
  printf("stream priority range: least %d .. greatest %d\n", lo_pri, hi_pri);
  hipEvent_t e0, e1, fork, join, mid;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&mid, hipEventDisableTiming));

  const int big_us = 100, small_us = 5, nchain = 10;
  auto chain = [&](hipStream_t s, int wgs, int lds_kind) {
    for (int i = 0; i < nchain; ++i) {
      if (lds_kind == 0) small<0><<<wgs, 256, 0, s>>>(small_us, sink);
      else if (lds_kind == 1) small<2048><<<wgs, 256, 0, s>>>(small_us, sink);      // 16 KB
      else small<4226><<<wgs, 256, 0, s>>>(small_us, sink);                          // 33 KB: the column step's arena
    }
  };
  auto median_of = [&](auto&& body) -> float {
    std::vector<float> ms;
    for (int rep = 0; rep < 21; ++rep) {
      (void)hipEventRecord(e0, s1);
      body();
      (void)hipEventRecord(e1, s1);
      (void)hipEventSynchronize(e1);
      float t = 0;
      (void)hipEventElapsedTime(&t, e0, e1);
      ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2] * 1000.0f;
  };
  // warm-up
  for (int i = 0; i < 5; ++i) { big<<<cus, 512, big_lds, s1>>>(20, sink); chain(s1, 8, 2); }
  CHECK(hipDeviceSynchronize());

  printf("A      big(%d WGs, %d us) alone:                        %.1f us\n", cus, big_us, median_of([&] { big<<<cus, 512, big_lds, s1>>>(big_us, sink); }));
  for (int wgs : {8, 32}) {
    printf("B      chain of %d x small(%d WGs, %d us, 33 KB) alone:     %.1f us\n", nchain, wgs, small_us, median_of([&] { chain(s1, wgs, 2); }));
  }
  for (int hi = 0; hi < 2; ++hi) {
    hipStream_t sb = hi ? s2hi : s2;
    for (int free_cus : {0, 8, 16}) {
      for (int lds_kind : {2, 1, 0}) {
        for (int wgs : {8, 32}) {
          const float t = median_of([&] {
            (void)hipEventRecord(fork, s1);
            (void)hipStreamWaitEvent(sb, fork, 0);
            big<<<cus - free_cus, 512, big_lds, s1>>>(big_us, sink);
            chain(sb, wgs, lds_kind);
            (void)hipEventRecord(join, sb);
            (void)hipStreamWaitEvent(s1, join, 0);
          });
          printf("A|B    big on %3d CUs | chain small(%2d WGs, LDS %s)%s: %.1f us\n", cus - free_cus, wgs,
                 lds_kind == 2 ? "33 KB" : (lds_kind == 1 ? "16 KB" : " 0 KB"), hi ? " [high-priority stream]" : "", t);
        }
      }
    }
  }
  // two phases: A0 (60 us, all CUs) then A1 (40 us, 248 CUs); the chain waits for A0 and runs under A1
  for (int hi = 0; hi < 2; ++hi) {
    hipStream_t sb = hi ? s2hi : s2;
    for (int free_cus : {0, 8}) {
      const float t = median_of([&] {
        big<<<cus, 512, big_lds, s1>>>(60, sink);
        (void)hipEventRecord(mid, s1);
        (void)hipStreamWaitEvent(sb, mid, 0);
        big<<<cus - free_cus, 512, big_lds, s1>>>(40, sink);
        chain(sb, 24, 2);
        (void)hipEventRecord(join, sb);
        (void)hipStreamWaitEvent(s1, join, 0);
      });
      printf("A0,A1|B  phase 0 (60 us, %d CUs), phase 1 (40 us, %d CUs) | chain(24 WGs, 33 KB) behind phase 0%s: %.1f us (serial: ~%d)\n",
             cus, cus - free_cus, hi ? " [high-priority stream]" : "", t, 100 + nchain * (small_us + 2));
    }
  }
  // cost of the fork / join events themselves on an otherwise serial stream
  printf("fork/join overhead: chain(8) on stream 2 bracketed by events on stream 1: %.1f us (chain alone above)\n", median_of([&] {
           (void)hipEventRecord(fork, s1);
           (void)hipStreamWaitEvent(s2, fork, 0);
           chain(s2, 8, 2);
           (void)hipEventRecord(join, s2);
           (void)hipStreamWaitEvent(s1, join, 0);
         }));
  CHECK(hipDeviceSynchronize());
  return 0;
}
