// Second overlap microbenchmark (gfx950; sizes the pipelined reduced solve of DESIGN.md section 4; not part of the product).
// tools/microbench_overlap.hip showed: (1) a cross-stream event dependency costs 12-15 us on this stack, (2) workgroups that
// carry LDS do not start next to a launch whose workgroups take a CU's LDS each, and with more than one of them per XCD even
// free CUs do not help.  This one measures what a pipeline could be built from instead:
//   T  placement threshold: big on 256 - F CUs | chain of small(W workgroups, 33 KB), F in {8, 16, 24, 32}, W in {8, 12, 16, 24}
//   C  co-residency: big with a 3-stage ring (110 KB) on all 256 CUs | chain of small(W, 33 KB)
//   G  gate kernels instead of events: a one-wave kernel polls a device counter that a kernel of the other stream bumps;
//      steady state of  stream 1: [phase0 60 us | bump f0 | phase1 40 us on 248 CUs | gate(f1 >= k)]
//                       stream 2: [gate(f0 >= k) | chain of 8 x small(8 WGs, 5 us) | bump f1]     for k = 1 .. 20, no host sync
//      against the serial form of the same work on one stream.
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_overlap2.hip -o tools/bin/microbench_overlap2
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ void spin_us(int us) {
  const unsigned long long t0 = wall_clock64();        // 100 MHz
  const unsigned long long ticks = 100ull * (unsigned long long)us;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

__global__ __launch_bounds__(512) void big(int us, double* sink) {
  extern __shared__ double img[];
  img[threadIdx.x] = threadIdx.x;
  __syncthreads();
  spin_us(us);
  if (img[(threadIdx.x + 1) & 511] < 0) sink[0] = 1;
}

__global__ __launch_bounds__(256) void small33(int us, double* sink) {
  __shared__ double arena[4226];
  arena[threadIdx.x] = threadIdx.x;
  __syncthreads();
  spin_us(us);
  if (arena[0] < 0) sink[0] = 1;
}

__global__ void bump(int* ctr, int value) {
  if (threadIdx.x == 0) __hip_atomic_store(ctr, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

// one wave: wait until *ctr >= value (bounded: ~50 ms, then report and leave)
__global__ void gate(const int* ctr, int value, int* err) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < value) {
    if (wall_clock64() - t0 > 5000000ull) { atomicAdd(err, 1); return; }
    __builtin_amdgcn_s_sleep(8);
  }
}

int main() {
  CHECK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const size_t lds4 = 147456, lds3 = 110592;
  CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(big), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
  double* sink = nullptr;
  int* flags = nullptr;
  CHECK(hipMalloc(&sink, 64));
  CHECK(hipMalloc(&flags, 64));
  CHECK(hipMemset(flags, 0, 64));
  hipStream_t s1, s2;
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1, fork, join;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  CHECK(hipEventCreateWithFlags(&fork, hipEventDisableTiming));
  CHECK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
  const int big_us = 100, small_us = 5, nchain = 10;
  auto chain = [&](hipStream_t s, int wgs, int n) { for (int i = 0; i < n; ++i) small33<<<wgs, 256, 0, s>>>(small_us, sink); };
  auto median_of = [&](auto&& body) -> float {
    std::vector<float> ms;
    for (int rep = 0; rep < 15; ++rep) {
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
  for (int i = 0; i < 5; ++i) { big<<<cus, 512, lds4, s1>>>(20, sink); chain(s1, 8, 3); }
  CHECK(hipDeviceSynchronize());
  printf("T  big(100 us, 144 KB) on 256 - F CUs | chain of %d x small(W WGs, 33 KB, %d us) behind a fork event; serial ~195, overlapped ~116\n", nchain, small_us);
  for (int F : {8, 16, 24, 32}) {
    for (int W : {8, 12, 16, 24}) {
      const float t = median_of([&] {
        (void)hipEventRecord(fork, s1);
        (void)hipStreamWaitEvent(s2, fork, 0);
        big<<<cus - F, 512, lds4, s1>>>(big_us, sink);
        chain(s2, W, nchain);
        (void)hipEventRecord(join, s2);
        (void)hipStreamWaitEvent(s1, join, 0);
      });
      printf("T  F = %2d free CUs, W = %2d workgroups: %.1f us\n", F, W, t);
    }
  }
  printf("C  big(100 us, 110 KB = 3-stage ring) on all 256 CUs | chain of small(W, 33 KB)\n");
  for (int W : {8, 16, 32, 64}) {
    const float t = median_of([&] {
      (void)hipEventRecord(fork, s1);
      (void)hipStreamWaitEvent(s2, fork, 0);
      big<<<cus, 512, lds3, s1>>>(big_us, sink);
      chain(s2, W, nchain);
      (void)hipEventRecord(join, s2);
      (void)hipStreamWaitEvent(s1, join, 0);
    });
    printf("C  W = %2d workgroups: %.1f us\n", W, t);
  }
  // G: gates instead of events, steady state over 20 iterations, wall clock around the whole (host sync at the end only)
  int* f0 = flags; int* f1 = flags + 1; int* err = flags + 2;
  const int iters = 20, nch = 8;
  auto wall_us = [&](auto&& body) -> double {
    std::vector<double> v;
    for (int rep = 0; rep < 7; ++rep) {
      (void)hipMemset(flags, 0, 64);
      (void)hipDeviceSynchronize();
      const auto t0 = std::chrono::steady_clock::now();
      body();
      (void)hipStreamSynchronize(s1);
      (void)hipStreamSynchronize(s2);
      v.push_back(std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters);
    }
    std::sort(v.begin(), v.end());
    return v[v.size() / 2];
  };
  const double serial = wall_us([&] {
    for (int k = 1; k <= iters; ++k) {
      big<<<cus, 512, lds4, s1>>>(60, sink);
      big<<<cus, 512, lds4, s1>>>(40, sink);
      chain(s1, 8, nch);
    }
  });
  printf("G  serial on one stream: phase0 60 us, phase1 40 us, chain of %d x small(8 WGs, 5 us): %.1f us per iteration\n", nch, serial);
  for (int F : {0, 8}) {
    const double piped = wall_us([&] {
      for (int k = 1; k <= iters; ++k) {
        big<<<cus, 512, lds4, s1>>>(60, sink);
        bump<<<1, 64, 0, s1>>>(f0, k);
        big<<<cus - F, 512, lds4, s1>>>(40, sink);
        gate<<<1, 64, 0, s1>>>(f1, k, err);
        gate<<<1, 64, 0, s2>>>(f0, k, err);
        chain(s2, 8, nch);
        bump<<<1, 64, 0, s2>>>(f1, k);
      }
    });
    int h_err = 0;
    (void)hipMemcpy(&h_err, err, sizeof(int), hipMemcpyDeviceToHost);
    printf("G  gates, phase1 on %d CUs: %.1f us per iteration (ideal ~%d; gate time-outs: %d)\n", cus - F, piped, 60 + std::max(40, nch * 7), h_err);
  }
  // the same with events (what the first microbenchmark priced at 12-15 us per dependency)
  {
    hipEvent_t ev0[32], ev1[32];
    for (int i = 0; i < 32; ++i) { (void)hipEventCreateWithFlags(&ev0[i], hipEventDisableTiming); (void)hipEventCreateWithFlags(&ev1[i], hipEventDisableTiming); }
    const double piped = wall_us([&] {
      for (int k = 1; k <= iters; ++k) {
        big<<<cus, 512, lds4, s1>>>(60, sink);
        (void)hipEventRecord(ev0[k], s1);
        big<<<cus - 8, 512, lds4, s1>>>(40, sink);
        (void)hipStreamWaitEvent(s2, ev0[k], 0);
        chain(s2, 8, nch);
        (void)hipEventRecord(ev1[k], s2);
        (void)hipStreamWaitEvent(s1, ev1[k], 0);
      }
    });
    printf("G  events, phase1 on %d CUs: %.1f us per iteration\n", cus - 8, piped);
  }
  CHECK(hipDeviceSynchronize());
  return 0;
}
