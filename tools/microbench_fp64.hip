// FP64 microbenchmarks for gfx950 design decisions (not part of the product):
//   1. v_fma_f64 VALU rate          2. v_mfma_f64_16x16x4_f64 rate (1 / 4 accumulators)
//   3. ds_add_f64 rate (conflict-free / pseudo-random addresses)
//   4. global_atomic_add_f64 rate (spread / contended)
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_fp64.hip -o gpurun_out/microbench_fp64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_ __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fma(double* out, int iters) {
  double a0 = threadIdx.x * 1e-9, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
  double m = 1.0000001, c = 1e-7;
  for (int i = 0; i < iters; ++i) {
    a0 = a0 * m + c; a1 = a1 * m + c; a2 = a2 * m + c; a3 = a3 * m + c;
    a4 = a4 * m + c; a5 = a5 * m + c; a6 = a6 * m + c; a7 = a7 * m + c;
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int NACC>
__global__ void k_mfma(double* out, int iters) {
  double4_ acc[NACC];
  for (int j = 0; j < NACC; ++j) acc[j] = double4_{0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - threadIdx.x * 1e-3;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[j], 0, 0, 0);
  }
  double s = 0;
  for (int j = 0; j < NACC; ++j) s += acc[j][0] + acc[j][1] + acc[j][2] + acc[j][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
__global__ void k_lds_atomic(double* out, int iters) {
  __shared__ double s[8192];
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) s[i] = 0;
  __syncthreads();
  unsigned idx = threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    unsigned a;
    if (MODE == 0) a = (idx + i * 64) & 8191;                       // conflict-free, consecutive doubles
    else if (MODE == 1) a = ((idx * 2654435761u) >> 7 ^ (i * 97u)) & 8191;     // pseudo-random
    else a = ((idx & 63) * 7 * 112 + (i * 7) ) & 8191;              // stride like a 7x7 block scatter
    atomicAdd(&s[a], 1.0);
  }
  __syncthreads();
  double t = 0;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) t += s[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = t;
}

template <int MODE>
__global__ void k_global_atomic(double* buf, size_t n, int iters) {
  size_t gid = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    size_t a;
    if (MODE == 0) a = (gid + (size_t)i * 1048573) % n;             // spread, coalesced per wave
    else a = ((threadIdx.x & 63) * 352 + (blockIdx.x % 49) + (size_t)i * 7) % (352 * 352);   // S-like contended
    atomicAdd(&buf[a], 1.0);
  }
}

template <typename F>
float time_ms(F launch, int reps = 5) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < reps; ++r) {
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  return best;
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs %d clock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  double* out; CHECK(hipMalloc(&out, 1 << 26));
  const int CUS = prop.multiProcessorCount;

  {
    int iters = 20000, blocks = CUS * 8, threads = 256;
    float ms = time_ms([&] { k_fma<<<blocks, threads>>>(out, iters); });
    double flops = 2.0 * 8 * iters * (double)blocks * threads;
    printf("fma_f64 VALU: %.3f ms  %.2f TFLOP/s\n", ms, flops / ms * 1e-9);
  }
  {
    int iters = 4000, blocks = CUS * 4, threads = 256;
    float ms1 = time_ms([&] { k_mfma<1><<<blocks, threads>>>(out, iters); });
    float ms4 = time_ms([&] { k_mfma<4><<<blocks, threads>>>(out, iters * 4 / 4); });
    double f1 = 2048.0 * iters * (double)blocks * (threads / 64);
    printf("mfma_f64_16x16x4 1acc: %.3f ms %.2f TFLOP/s   4acc: %.3f ms %.2f TFLOP/s\n", ms1, f1 / ms1 * 1e-9, ms4, 4 * f1 / ms4 * 1e-9);
    blocks = CUS; threads = 256;   // one wave per SIMD
    float msa = time_ms([&] { k_mfma<4><<<blocks, threads>>>(out, iters); });
    double fa = 4 * 2048.0 * iters * (double)blocks * (threads / 64);
    printf("mfma_f64 1 wave/SIMD 4acc: %.3f ms %.2f TFLOP/s -> %.1f cycles/MFMA at 2.4GHz\n", msa, fa / msa * 1e-9,
           msa * 1e-3 * 2.4e9 / (4.0 * iters));
  }
  {
    int iters = 4000, blocks = CUS * 2, threads = 512;
    float m0 = time_ms([&] { k_lds_atomic<0><<<blocks, threads>>>(out, iters); });
    float m1 = time_ms([&] { k_lds_atomic<1><<<blocks, threads>>>(out, iters); });
    float m2 = time_ms([&] { k_lds_atomic<2><<<blocks, threads>>>(out, iters); });
    double n = (double)iters * blocks * threads;
    printf("ds_add_f64 consecutive: %.3f ms %.1f Gadd/s (%.2f adds/clk/CU) | random: %.3f ms %.1f Gadd/s | 7x7-stride: %.3f ms %.1f Gadd/s\n",
           m0, n / m0 * 1e-6, n / (m0 * 1e-3 * 2.4e9 * CUS), m1, n / m1 * 1e-6, m2, n / m2 * 1e-6);
  }
  {
    size_t n = 1 << 23;   // 64 MB of doubles
    CHECK(hipMemset(out, 0, n * 8));
    int iters = 64, blocks = CUS * 16, threads = 256;
    float m0 = time_ms([&] { k_global_atomic<0><<<blocks, threads>>>(out, n, iters); });
    float m1 = time_ms([&] { k_global_atomic<1><<<blocks, threads>>>(out, n, iters); });
    double cnt = (double)iters * blocks * threads;
    printf("global_atomic_add_f64 spread: %.3f ms %.1f Gadd/s (%.2f TB/s) | S-like 352x352 contended: %.3f ms %.1f Gadd/s\n",
           m0, cnt / m0 * 1e-6, cnt * 8 / m0 * 1e-9, m1, cnt / m1 * 1e-6);
  }
  hipFree(out);
  return 0;
}
