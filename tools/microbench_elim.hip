// What bounds the 32x32 elimination of ba_chol_step (gfx950; not part of the product)?  One wave, lanes 0-31 = rows
// of an SPD block, lanes 32-63 = rows of T; s_memtime around sixteen 2x2-pivot steps of:
//   0  the pivot chain alone (broadcasts, two rsqrt, x / y, next pair's two columns)
//   1  + publishing (x, y) to LDS
//   2  + the deferred update of all other columns (the product kernel's chol_trsm_rows)
//   3  chain alone with ONE Newton step per rsqrt
//   4  chain alone, next pair's columns through LDS broadcast reads instead of v_readlane
//   5  one column per step (32 steps), chain alone
// Build: hipcc --offload-arch=gfx950 -O3 tools/microbench_elim.hip -o tools/bin/microbench_elim
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
constexpr int NB = 32;
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
template <int NR>
__device__ __forceinline__ double rsqrt_n(double d) {
  double r = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  r = r * (1.5 - h * r * r);
  if (NR > 1) r = r * (1.5 - h * r * r);
  return r;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_elim(const double* in, double* out, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(16))) f64x2 xy[16][64];
  const int lane = threadIdx.x;
  double a[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) a[k] = in[lane * NB + k];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 5) {
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const double inv = rsqrt_n<2>(lane_bcast(a[j], j));
      a[j] *= inv;
      if (j + 1 < NB) a[j + 1] -= a[j] * lane_bcast(a[j], j + 1);
      if (j + 2 < NB) a[j + 2] -= a[j] * lane_bcast(a[j], j + 2);
    }
  } else {
    f64x2 prev[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) prev[k] = f64x2{0.0, 0.0};
#pragma unroll
    for (int s = 0; s < NB / 2; ++s) {
      const int j = 2 * s;
      const double pa = lane_bcast(a[j], j), pb = lane_bcast(a[j], j + 1), pc = lane_bcast(a[j + 1], j + 1);
      const double det = __builtin_fma(pa, pc, -(pb * pb));
      const double r1 = MODE == 3 ? rsqrt_n<1>(pa) : rsqrt_n<2>(pa), r2 = MODE == 3 ? rsqrt_n<1>(det) : rsqrt_n<2>(det);
      const double l11 = pa * r1, l21 = pb * r1, i22 = r2 * l11;
      if (MODE == 2 && s > 0) {
#pragma unroll
        for (int k = j + 2; k < NB; ++k) a[k] -= a[j - 2] * prev[k].x + a[j - 1] * prev[k].y;
      }
      const double x = a[j] * r1;
      const double y = (a[j + 1] - x * l21) * i22;
      a[j] = x; a[j + 1] = y;
      if (MODE == 1 || MODE == 2 || MODE == 4) xy[s][lane] = f64x2{x, y};
      if (j + 2 < NB) {
        if (MODE == 4) {
          const f64x2 q2 = xy[s][j + 2], q3 = xy[s][j + 3];
          a[j + 2] -= x * q2.x + y * q2.y;
          a[j + 3] -= x * q3.x + y * q3.y;
        } else {
          const double x2 = lane_bcast(x, j + 2), y2 = lane_bcast(y, j + 2);
          const double x3 = lane_bcast(x, j + 3), y3 = lane_bcast(y, j + 3);
          a[j + 2] -= x * x2 + y * y2;
          a[j + 3] -= x * x3 + y * y3;
        }
      }
      if (MODE == 2) {
#pragma unroll
        for (int k = j + 4; k < NB; ++k) prev[k] = xy[s][k];
      }
    }
  }
  double sum = 0;
#pragma unroll
  for (int k = 0; k < NB; ++k) sum += a[k];
  asm volatile("" :: "v"(sum));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[lane] = sum;
  if (lane == 0) cyc[0] = t1 - t0;
}

// the product's multi-wave elimination (chol_trsm_cols): WAVES waves, wave w owns columns CW w .. CW w + CW - 1
// (CW = 32 / WAVES); stamps per wave.  FMA2: updates as two chained FMAs instead of mul + fma + sub.
template <int WAVES, int NR, int POLL_SLEEP, bool FMA2>
__global__ __launch_bounds__(64 * WAVES) void k_elim4(const double* in, double* out, unsigned long long* cyc) {
  constexpr int CW = NB / WAVES, PS = CW / 2;      // columns / pair-steps per wave
  __shared__ __attribute__((aligned(16))) f64x2 xy[16][64];
  __shared__ int flag_s;
  int* flag = &flag_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double c[CW];
#pragma unroll
  for (int u = 0; u < CW; ++u) c[u] = in[lane * NB + CW * wave + u];
  if (threadIdx.x == 0) *flag = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long tchain0 = 0, tchain1 = 0;
  auto upd = [](double cc, double x, double y, double qx, double qy) {
    if (FMA2) return __builtin_fma(-y, qy, __builtin_fma(-x, qx, cc));
    return cc - (x * qx + y * qy);
  };
#pragma unroll
  for (int seg = 0; seg < WAVES; ++seg) {
    if (wave == seg) {
      tchain0 = __builtin_amdgcn_s_memtime();
#pragma unroll
      for (int t = 0; t < PS; ++t) {
        const int j = 2 * t, s = PS * seg + t, l0 = CW * seg + j;
        const double pa = lane_bcast(c[j], l0), pb = lane_bcast(c[j], l0 + 1), pc = lane_bcast(c[j + 1], l0 + 1);
        const double det = __builtin_fma(pa, pc, -(pb * pb));
        const double r1 = rsqrt_n<NR>(pa), r2 = rsqrt_n<NR>(det);
        const double l11 = pa * r1, l21 = pb * r1, i22 = r2 * l11;
        const double x = c[j] * r1;
        const double y = (c[j + 1] - x * l21) * i22;
        c[j] = x; c[j + 1] = y;
        xy[s][lane] = f64x2{x, y};
        if (t < PS - 1) {
          const double x2 = lane_bcast(x, l0 + 2), y2 = lane_bcast(y, l0 + 2);
          const double x3 = lane_bcast(x, l0 + 3), y3 = lane_bcast(y, l0 + 3);
          c[j + 2] = upd(c[j + 2], x, y, x2, y2);
          c[j + 3] = upd(c[j + 3], x, y, x3, y3);
#pragma unroll
          for (int u = j + 4; u < CW; ++u) { const f64x2 q = xy[s][CW * seg + u]; c[u] = upd(c[u], x, y, q.x, q.y); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(flag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      tchain1 = __builtin_amdgcn_s_memtime();
    } else if (wave > seg) {
#pragma unroll
      for (int t = 0; t < PS; ++t) {
        const int s = PS * seg + t;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s) { if (POLL_SLEEP) __builtin_amdgcn_s_sleep(POLL_SLEEP); }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const f64x2 own = xy[s][lane];
#pragma unroll
        for (int u = 0; u < CW; ++u) { const f64x2 q = xy[s][CW * wave + u]; c[u] = upd(c[u], own.x, own.y, q.x, q.y); }
      }
    }
  }
  double sum = 0;
#pragma unroll
  for (int u = 0; u < CW; ++u) sum += c[u];
  asm volatile("" :: "v"(sum));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = sum;
  if (lane == 0) { cyc[3 * wave] = t1 - t0; cyc[3 * wave + 1] = tchain0 - t0; cyc[3 * wave + 2] = tchain1 - t0; }
}

// Variant 5: the consumer is software-pipelined (reads of step s issued before the FMAs of step s-1), the flag goes
// up right after (x, y) are in LDS, the chain's far columns are folded one step late inside the reciprocal-square-root
// latency, and at a hand-over the new chain wave updates only its first two columns before it starts (the other
// columns' last consumer update becomes its first deferred update).
template <int WAVES, int NR, int POLL_SLEEP, bool LATE>
__global__ __launch_bounds__(64 * WAVES) void k_elim5(const double* in, double* out, unsigned long long* cyc) {
  constexpr int CW = NB / WAVES, PS = CW / 2;
  __shared__ __attribute__((aligned(16))) f64x2 xy[16][64];
  __shared__ int flag_s;
  int* flag = &flag_s;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double c[CW];
#pragma unroll
  for (int u = 0; u < CW; ++u) c[u] = in[lane * NB + CW * wave + u];
  if (threadIdx.x == 0) *flag = 0;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long tchain0 = 0, tchain1 = 0;
  // ---- consumer of every step before my segment
  double xp = 0.0, yp = 0.0;
  f64x2 q[CW];
#pragma unroll
  for (int u = 0; u < CW; ++u) q[u] = f64x2{0.0, 0.0};
  const int nprev = PS * wave;
  for (int s = 0; s < nprev; ++s) {
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s) { if (POLL_SLEEP) __builtin_amdgcn_s_sleep(POLL_SLEEP); }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const f64x2 own = xy[s][lane];
    f64x2 qn[CW];
#pragma unroll
    for (int u = 0; u < CW; ++u) qn[u] = xy[s][CW * wave + u];
#pragma unroll
    for (int u = 0; u < CW; ++u) c[u] = __builtin_fma(-yp, q[u].y, __builtin_fma(-xp, q[u].x, c[u]));
    xp = own.x; yp = own.y;
#pragma unroll
    for (int u = 0; u < CW; ++u) q[u] = qn[u];
  }
  // ---- chain: my first two columns get the pending update now, the others inside the first step
  c[0] = __builtin_fma(-yp, q[0].y, __builtin_fma(-xp, q[0].x, c[0]));
  c[1] = __builtin_fma(-yp, q[1].y, __builtin_fma(-xp, q[1].x, c[1]));
  tchain0 = __builtin_amdgcn_s_memtime();
#pragma unroll
  for (int t = 0; t < PS; ++t) {
    const int j = 2 * t, s = PS * wave + t, l0 = CW * wave + j;
    const double pa = lane_bcast(c[j], l0), pb = lane_bcast(c[j], l0 + 1), pc = lane_bcast(c[j + 1], l0 + 1);
    const double det = __builtin_fma(pa, pc, -(pb * pb));
    const double r1 = rsqrt_n<NR>(pa), r2 = rsqrt_n<NR>(det);
    const double l11 = pa * r1, l21 = pb * r1, i22 = r2 * l11;
#pragma unroll
    for (int u = j + 2; u < CW; ++u) c[u] = __builtin_fma(-yp, q[u].y, __builtin_fma(-xp, q[u].x, c[u]));     // one step late
    const double x = c[j] * r1;
    const double y = (c[j + 1] - x * l21) * i22;
    c[j] = x; c[j + 1] = y;
    xy[s][lane] = f64x2{x, y};
    if (LATE) {
      // the LDS write drains behind the register broadcasts; the flag goes up after them
      if (t < PS - 1) {
        const double x2 = lane_bcast(x, l0 + 2), y2 = lane_bcast(y, l0 + 2);
        const double x3 = lane_bcast(x, l0 + 3), y3 = lane_bcast(y, l0 + 3);
        c[j + 2] = __builtin_fma(-y, y2, __builtin_fma(-x, x2, c[j + 2]));
        c[j + 3] = __builtin_fma(-y, y3, __builtin_fma(-x, x3, c[j + 3]));
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_store(flag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (t < PS - 1) {
#pragma unroll
        for (int u = j + 4; u < CW; ++u) q[u] = xy[s][CW * wave + u];
        xp = x; yp = y;
      }
    } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    if (lane == 0) __hip_atomic_store(flag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (t < PS - 1) {
      const double x2 = lane_bcast(x, l0 + 2), y2 = lane_bcast(y, l0 + 2);
      const double x3 = lane_bcast(x, l0 + 3), y3 = lane_bcast(y, l0 + 3);
      c[j + 2] = __builtin_fma(-y, y2, __builtin_fma(-x, x2, c[j + 2]));
      c[j + 3] = __builtin_fma(-y, y3, __builtin_fma(-x, x3, c[j + 3]));
#pragma unroll
      for (int u = j + 4; u < CW; ++u) q[u] = xy[s][CW * wave + u];
      xp = x; yp = y;
    }
    }
  }
  tchain1 = __builtin_amdgcn_s_memtime();
  double sum = 0;
#pragma unroll
  for (int u = 0; u < CW; ++u) sum += c[u];
  asm volatile("" :: "v"(sum));
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  out[threadIdx.x] = sum;
  if (lane == 0) { cyc[3 * wave] = t1 - t0; cyc[3 * wave + 1] = tchain0 - t0; cyc[3 * wave + 2] = tchain1 - t0; }
}

template <int WAVES, int NR, int POLL_SLEEP, bool LATE>
int run5(const double* din, const char* name) {
  unsigned long long* dc4; double* dout4;
  CHECK(hipMalloc(&dc4, 3 * WAVES * 8)); CHECK(hipMalloc(&dout4, 64 * WAVES * sizeof(double)));
  for (int rep = 0; rep < 3; ++rep) { k_elim5<WAVES, NR, POLL_SLEEP, LATE><<<1, 64 * WAVES>>>(din, dout4, dc4); CHECK(hipDeviceSynchronize()); }
  unsigned long long c[24];
  CHECK(hipMemcpy(c, dc4, 3 * WAVES * 8, hipMemcpyDeviceToHost));
  std::vector<double> o(64 * WAVES);
  CHECK(hipMemcpy(o.data(), dout4, o.size() * sizeof(double), hipMemcpyDeviceToHost));
  double chk = 0; for (double v : o) chk += v;
  unsigned long long end = 0;
  for (int w = 0; w < WAVES; ++w) end = c[3 * w] > end ? c[3 * w] : end;
  printf("%-44s total %5llu | chain windows:", name, end);
  for (int w = 0; w < WAVES; ++w) printf(" %llu-%llu", c[3 * w + 1], c[3 * w + 2]);
  printf("  [check %.9f]\n", chk);
  return 0;
}

template <int WAVES, int NR, int POLL_SLEEP, bool FMA2>
int run4(const double* din, const char* name) {
  unsigned long long* dc4; double* dout4;
  CHECK(hipMalloc(&dc4, 3 * WAVES * 8)); CHECK(hipMalloc(&dout4, 64 * WAVES * sizeof(double)));
  for (int rep = 0; rep < 3; ++rep) { k_elim4<WAVES, NR, POLL_SLEEP, FMA2><<<1, 64 * WAVES>>>(din, dout4, dc4); CHECK(hipDeviceSynchronize()); }
  unsigned long long c[24];
  CHECK(hipMemcpy(c, dc4, 3 * WAVES * 8, hipMemcpyDeviceToHost));
  unsigned long long end = 0;
  for (int w = 0; w < WAVES; ++w) end = c[3 * w] > end ? c[3 * w] : end;
  printf("%-44s total %5llu | chain windows:", name, end);
  for (int w = 0; w < WAVES; ++w) printf(" %llu-%llu", c[3 * w + 1], c[3 * w + 2]);
  std::vector<double> o(64 * WAVES);
  CHECK(hipMemcpy(o.data(), dout4, o.size() * sizeof(double), hipMemcpyDeviceToHost));
  double chk = 0; for (double v : o) chk += v;
  printf("  [check %.9f]\n", chk);
  return 0;
}

int main() {
  std::vector<double> h(64 * NB);
  for (int i = 0; i < 64; ++i)
    for (int k = 0; k < NB; ++k) {
      const int r = i & 31;
      double v = 0.01 * ((i * 37 + k * 11) % 17 - 8);
      if (i < 32) { const int lo = r < k ? r : k, hi = r < k ? k : r; v = 0.01 * ((hi * 5 + lo * 3) % 13 - 6); if (r == k) v = 8.0 + 0.1 * r; }
      h[i * NB + k] = v;
    }
  double *din, *dout; unsigned long long* dc;
  CHECK(hipMalloc(&din, h.size() * sizeof(double))); CHECK(hipMalloc(&dout, 64 * sizeof(double))); CHECK(hipMalloc(&dc, 8));
  CHECK(hipMemcpy(din, h.data(), h.size() * sizeof(double), hipMemcpyHostToDevice));
  const char* names[] = {"chain alone (2x2 pivots)", "+ publish to LDS", "+ deferred far columns (product)", "chain, 1 Newton step",
                         "chain, next pair via LDS", "chain, one column per step"};
  for (int m = 0; m < 6; ++m) {
    for (int rep = 0; rep < 3; ++rep) {
      switch (m) {
        case 0: k_elim<0><<<1, 64>>>(din, dout, dc); break;
        case 1: k_elim<1><<<1, 64>>>(din, dout, dc); break;
        case 2: k_elim<2><<<1, 64>>>(din, dout, dc); break;
        case 3: k_elim<3><<<1, 64>>>(din, dout, dc); break;
        case 4: k_elim<4><<<1, 64>>>(din, dout, dc); break;
        default: k_elim<5><<<1, 64>>>(din, dout, dc); break;
      }
      CHECK(hipDeviceSynchronize());
    }
    unsigned long long c; double o;
    CHECK(hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&o, dout, 8, hipMemcpyDeviceToHost));
    printf("%-36s %6llu ticks per 32 columns (%.0f per column)  [check %.6f]\n", names[m], c, c / 32.0, o);
  }
  run4<4, 2, 1, false>(din, "4 waves, 2 Newton, sleep 1, mul+fma+sub");
  run4<4, 2, 1, true>(din, "4 waves, 2 Newton, sleep 1, 2 FMAs");
  run4<4, 2, 0, true>(din, "4 waves, 2 Newton, busy poll, 2 FMAs");
  run4<4, 1, 1, true>(din, "4 waves, 1 Newton, sleep 1, 2 FMAs");
  run4<8, 2, 1, true>(din, "8 waves, 2 Newton, sleep 1, 2 FMAs");
  run4<2, 2, 1, true>(din, "2 waves, 2 Newton, sleep 1, 2 FMAs");
  run5<4, 2, 1, false>(din, "v5 pipelined: 4 waves, sleep 1");
  run5<4, 2, 0, false>(din, "v5 pipelined: 4 waves, busy poll");
  run5<4, 2, 1, true>(din, "v5 late flag: 4 waves, sleep 1");
  run5<4, 2, 0, true>(din, "v5 late flag: 4 waves, busy poll");
  run5<8, 2, 0, true>(din, "v5 late flag: 8 waves, busy poll");
  run5<2, 2, 0, true>(din, "v5 late flag: 2 waves, busy poll");
  return 0;
}
