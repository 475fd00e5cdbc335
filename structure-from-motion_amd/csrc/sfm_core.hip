// sfm_core.hip — library context + unit-level hooks + the per-point nonlinear triangulation kernel
// and the per-view nonlinear PnP kernel (gfx950).
//
//   tri_nonlinear_kernel  <-> TriangulationProcessor.nonlinear_triangulate (triangulation_processor.py:160-234)
//   pnp_nonlinear_kernel  <-> CamposeProcessor.nonlinear_estimate_cam_pose_pnp (campose_processor.py:308-459)
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "sfm_common.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

Context& ctx() {
  static Context c;
  return c;
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int hip_fail(hipError_t e, const char* what, int line) {
  set_error("HIP error '%s' in %s (line %d)", hipGetErrorString(e), what, line);
  return SFM_E_HIP;
}

PendingWork& pending_work() {
  static thread_local PendingWork w;
  return w;
}

int stream_sync(hipStream_t s) {
  SFM_HIP(hipStreamSynchronize(s));
  PendingWork& w = pending_work();
  if (w.stream == s || w.stream == nullptr) w.dirty = false;
  return SFM_OK;
}

int ensure_init() {
  if (ctx().inited) return SFM_OK;
  return sfm_init(0);
}

// ---------------------------------------------------------------------------------------------
// device-memory pool (see sfm_common.h)
//
// Three process-wide modes, chosen by the environment at first use:
//   default            blocks rounded up to a power of two, cached per size class.  NO slack behind a buffer: a read past
//                      the end is not made harmless by construction (SFM_POOL_SLACK=<bytes> adds that many bytes behind
//                      every block -- a field switch for a suspected stray read, never set by the tests).
//   SFM_POOL_REDZONE=1 every buffer between two 4 KB zones of 0xA5, checked when it returns to the pool: an out-of-bounds
//                      WRITE aborts with a message.
//   SFM_POOL_GUARD=1   every buffer is its own virtual-memory mapping (hipMemAddressReserve / hipMemCreate / hipMemMap)
//                      that ENDS (to 16 bytes) where the buffer ends, followed by a reserved, never mapped granule: an
//                      out-of-bounds READ or write past the end faults at the access instead of landing in a neighbour.
//                      No caching (every free unmaps).  For one test pass on the GPU box.
// ---------------------------------------------------------------------------------------------
namespace {
struct Block { void* base; size_t size; size_t bytes; hipMemGenericAllocationHandle_t handle; size_t mapped; };
struct Pool {
  std::mutex mu;
  std::unordered_map<void*, Block> live;                  // pointer handed out -> its block
  std::unordered_map<size_t, std::vector<void*>> free_by_size;   // rounded size -> cached block bases
  size_t cached_bytes = 0;
  long long guard_allocs = 0;
  size_t guard_reserved = 0;      // address space guard mode has reserved (never returned)
};
Pool& pool() {
  static Pool p;
  return p;
}
size_t round_up_pow2(size_t n) {
  size_t r = 256;
  while (r < n) r <<= 1;
  return r;
}
constexpr size_t kPoolCap = (size_t)2 << 30;
constexpr size_t kRedZone = 4096;
bool env_is_one(const char* name) {
  const char* e = std::getenv(name);
  return e && e[0] == '1';
}
bool redzone_on() {
  static const bool on = env_is_one("SFM_POOL_REDZONE");
  return on;
}
bool guard_on() {
  static const bool on = env_is_one("SFM_POOL_GUARD");
  return on;
}
size_t slack_bytes() {
  static const size_t n = [] { const char* e = std::getenv("SFM_POOL_SLACK"); return e ? (size_t)std::strtoull(e, nullptr, 10) : (size_t)0; }();
  return n;
}
void redzone_fill(const Block& b, void* user) {
  char* base = static_cast<char*>(b.base);
  char* end = static_cast<char*>(user) + b.bytes;
  (void)hipMemset(base, 0xA5, kRedZone);
  (void)hipMemset(end, 0xA5, (size_t)(base + b.size - end));
}
void redzone_check(const Block& b, void* user) {
  (void)hipDeviceSynchronize();
  char* base = static_cast<char*>(b.base);
  char* end = static_cast<char*>(user) + b.bytes;
  const size_t tail = (size_t)(base + b.size - end);
  std::vector<unsigned char> h(std::max(kRedZone, tail));
  auto scan = [&](const char* what, const char* dev, size_t n) {
    (void)hipMemcpy(h.data(), dev, n, hipMemcpyDeviceToHost);
    for (size_t i = 0; i < n; ++i)
      if (h[i] != 0xA5) {
        std::fprintf(stderr, "sfm pool red zone: a kernel wrote %s a %zu-byte buffer (offset %zu of the zone)\n", what, b.bytes, i);
        std::abort();
      }
  };
  scan("BEFORE", base, kRedZone);
  scan("PAST THE END OF", end, tail);
}

// ---- guard mode --------------------------------------------------------------------------------
size_t guard_granularity(int device) {
  static size_t g = 0;
  if (g) return g;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  size_t gran = 0;
  if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum) != hipSuccess || gran == 0) gran = (size_t)2 << 20;
  g = gran;
  return g;
}
hipError_t guard_alloc(void** ptr, size_t bytes, Block* out) {
  const int device = ctx().device < 0 ? 0 : ctx().device;
  const size_t gran = guard_granularity(device);
  const size_t need = (std::max<size_t>(bytes, 1) + 15) & ~(size_t)15;      // vector loads are at most 16 bytes wide
  const size_t mapped = (need + gran - 1) / gran * gran;
  void* base = nullptr;
  hipError_t e = hipMemAddressReserve(&base, mapped + gran, gran, nullptr, 0);      // + one granule that is never mapped
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = device;
  hipMemGenericAllocationHandle_t h{};
  e = hipMemCreate(&h, mapped, &prop, 0);
  if (e != hipSuccess) { (void)hipMemAddressFree(base, mapped + gran); return e; }
  e = hipMemMap(base, mapped, 0, h, 0);
  if (e != hipSuccess) { (void)hipMemRelease(h); (void)hipMemAddressFree(base, mapped + gran); return e; }
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  e = hipMemSetAccess(base, mapped, &acc, 1);
  if (e != hipSuccess) { (void)hipMemUnmap(base, mapped); (void)hipMemRelease(h); (void)hipMemAddressFree(base, mapped + gran); return e; }
  *ptr = static_cast<char*>(base) + (mapped - need);      // the buffer ends where the mapping ends
  *out = Block{base, mapped + gran, bytes, h, mapped};
  return hipSuccess;
}
void guard_free(const Block& b) {
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(b.base, b.mapped);
  (void)hipMemRelease(b.handle);
  // The address range stays RESERVED for the life of the process: hipMemAddressFree followed by a new reservation hands
  // the same addresses out again, and on this stack (ROCm 7.2, MI355X) the device then kept reading through stale
  // translations -- 85 of 200 upload / kernel / download round trips came back wrong and BA results varied from run to run
  // (profiles/r3/guard_mode_diagnostic.txt); with every range used once the same runs are exact.  A never-reused range also
  // turns a use-after-free into a fault.  The suite reserves a few hundred GB of the 47-bit address space this way.
}
}  // namespace

hipError_t pool_alloc(void** ptr, size_t bytes) {
  Pool& P = pool();
  if (bytes > ((size_t)1 << 40)) return hipErrorOutOfMemory;      // a size computed from a negative count: fail, do not loop in round_up_pow2
  if (guard_on()) {
    // guard mode never hands an address range out twice (guard_free), i.e. it leaks address space by design: it is a TEST
    // mode.  A long-lived process must not run into the end of the 47-bit space unannounced: stop at 16 TB.
    if (P.guard_reserved + bytes > ((size_t)16 << 40)) {
      std::fprintf(stderr, "sfm pool guard mode: %zu GB of address space reserved and never reused; SFM_POOL_GUARD is a test mode, not for long-lived processes\n",
                   P.guard_reserved >> 30);
      std::abort();
    }
    P.guard_reserved += bytes + (64 << 10);
    Block b{};
    const hipError_t e = guard_alloc(ptr, bytes, &b);
    if (e != hipSuccess) {
      std::fprintf(stderr, "sfm pool guard mode: the virtual-memory API failed (%s); SFM_POOL_GUARD cannot be honoured on this stack\n", hipGetErrorString(e));
      std::abort();      // a test pass that silently ran unguarded would claim what it did not check
    }
    std::lock_guard<std::mutex> g(P.mu);
    P.live[*ptr] = b;
    ++P.guard_allocs;
    return hipSuccess;
  }
  const bool rz = redzone_on();
  // size class = the request rounded up to a power of two up to 64 MB, to the next multiple of 2 MB above (VERDICT r3 item 8:
  // the 169 MB Zd of C3 took a 256 MB class, 3.7 GB of C4 took 4 GB; large buffers are few and long-lived, so a fine class
  // costs no reuse); red zones / optional slack come on top of the class, so a request that already is a class size does
  // not grow
  constexpr size_t kFineAbove = (size_t)64 << 20, kFineStep = (size_t)2 << 20;
  const size_t cls = bytes <= kFineAbove ? round_up_pow2(bytes) : (bytes + kFineStep - 1) / kFineStep * kFineStep;
  const size_t sz = cls + (rz ? 2 * kRedZone : slack_bytes());
  void* base = nullptr;
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.free_by_size.find(sz);
    if (it != P.free_by_size.end() && !it->second.empty()) {
      base = it->second.back();
      it->second.pop_back();
      P.cached_bytes -= sz;
    }
  }
  if (base == nullptr) {
    hipError_t e = hipMalloc(&base, sz);
    if (e != hipSuccess) {               // out of memory: drop the cache and retry once
      pool_release_all();
      e = hipMalloc(&base, sz);
    }
    if (e != hipSuccess) return e;
  }
  *ptr = rz ? static_cast<char*>(base) + kRedZone : base;
  const Block b{base, sz, bytes, {}, 0};
  if (rz) redzone_fill(b, *ptr);
  std::lock_guard<std::mutex> g(P.mu);
  P.live[*ptr] = b;
  return hipSuccess;
}

void pool_free(void* ptr) {
  if (ptr == nullptr) return;
  Pool& P = pool();
  Block b{};
  {
    std::lock_guard<std::mutex> g(P.mu);
    auto it = P.live.find(ptr);
    if (it == P.live.end()) { (void)hipFree(ptr); return; }     // not ours (defensive)
    b = it->second;
    P.live.erase(it);
  }
  if (b.mapped) { guard_free(b); return; }
  if (redzone_on()) redzone_check(b, ptr);
  {
    std::lock_guard<std::mutex> g(P.mu);
    if (P.cached_bytes + b.size <= kPoolCap) {
      P.free_by_size[b.size].push_back(b.base);
      P.cached_bytes += b.size;
      return;
    }
  }
  (void)hipFree(b.base);
}

void pool_release_all() {
  Pool& P = pool();
  std::lock_guard<std::mutex> g(P.mu);
  for (auto& kv : P.free_by_size)
    for (void* q : kv.second) (void)hipFree(q);
  P.free_by_size.clear();
  P.cached_bytes = 0;
}

// Diagnostics of the pool mode (sfm_pool_mode): bit 0 red zones, bit 1 guard mappings; *tail_slack = bytes between the
// end of a probe buffer of `probe_bytes` and the end of what is mapped behind it (guard mode: < 16).
int pool_mode_probe(size_t probe_bytes, long long* tail_slack, long long* guard_allocs) {
  int mode = (redzone_on() ? 1 : 0) | (guard_on() ? 2 : 0);
  if (tail_slack) {
    *tail_slack = -1;
    if (ctx().inited) {
      void* q = nullptr;
      if (pool_alloc(&q, probe_bytes) == hipSuccess) {
        Pool& P = pool();
        {
          std::lock_guard<std::mutex> g(P.mu);
          const Block& b = P.live[q];
          const char* end_mapped = static_cast<char*>(b.base) + (b.mapped ? b.mapped : b.size);
          *tail_slack = (long long)(end_mapped - (static_cast<char*>(q) + probe_bytes));
        }
        pool_free(q);
      }
    }
  }
  if (guard_allocs) *guard_allocs = pool().guard_allocs;
  return mode;
}

// ---------------------------------------------------------------------------------------------
// unit-level kernels
// ---------------------------------------------------------------------------------------------
__global__ void quat_to_rot_kernel(int n, const double* q, double* R, int* status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r[9];
  quat_to_rot(q + 4 * i, r);
  for (int k = 0; k < 9; ++k) R[9 * i + k] = r[k];
  status[i] = verify_rotation(r) ? SFM_OK : SFM_E_BAD_ROTATION;
}

__global__ void rot_to_quat_kernel(int n, const double* R, double* q, int* status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double r[9], qq[4] = {0, 0, 0, 0};
  for (int k = 0; k < 9; ++k) r[k] = R[9 * i + k];
  status[i] = rot_to_quat(r, qq);
  for (int k = 0; k < 4; ++k) q[4 * i + k] = qq[k];
}

__global__ void jac_cam_kernel(int n, const double* R, const double* C, const double* X, int quirks,
                               double* Jp, int* status) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  CamPrep c;
  double r[9], cc[3];
  for (int k = 0; k < 9; ++k) r[k] = R[9 * i + k];
  for (int k = 0; k < 3; ++k) cc[k] = C[3 * i + k];
  int st = cam_prepare_rc(r, cc, &c);
  status[i] = st;
  double jp[14];
  for (int k = 0; k < 14; ++k) jp[k] = 0;
  if (st == SFM_OK) {
    double p[3];
    const double* x = X + 4 * i;
    project_cam(c, x[0], x[1], x[2], x[3], p);
    jac_cam(c, x[0], x[1], x[2], p, quirks, jp);
  }
  for (int k = 0; k < 14; ++k) Jp[14 * i + k] = jp[k];
}

__global__ void jac_pt_kernel(int n, int n_views, const double* projs, const double* X, double* Jx) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double* x = X + 4 * i;
  for (int v = 0; v < n_views; ++v) {
    const double* P = projs + (size_t)(i * n_views + v) * 12;
    double s[3], j[6];
    for (int r = 0; r < 3; ++r) s[r] = P[4 * r] * x[0] + P[4 * r + 1] * x[1] + P[4 * r + 2] * x[2] + P[4 * r + 3] * x[3];
    jac_pt(P, s, j);
    for (int k = 0; k < 6; ++k) Jx[(size_t)(i * n_views + v) * 6 + k] = j[k];
  }
}

// ---------------------------------------------------------------------------------------------
// Nonlinear triangulation: one thread per point, every iteration in registers.
//   e = f - b, J (2V x 3) with the pixel projections, delta = inv(J^T J + lambda I3) J^T e, X -= delta
// (triangulation_processor.py:209-228).  Projections are wave-uniform -> staged in LDS; the
// uv / X arrays are SoA so consecutive lanes read consecutive doubles.
// ---------------------------------------------------------------------------------------------
constexpr int kTriLdsViews = 512;

// NV > 0: the view count is a compile-time constant (2..4, the pipeline's case is 2) and the point's keys stay
// in registers for all iterations -- re-reading them made the kernel L2/HBM-bound (4.8 GB for 10^6 points x 3
// views x 100 iterations).  NV == 0: any number of views, keys re-read per iteration.  One division per view
// and iteration (iz = 1 / s2 serves the residual and the Jacobian).
template <int NV>
__global__ __launch_bounds__(256) void tri_nonlinear_kernel(int m, int n_views, const double* __restrict__ projs,
                                                            const double* __restrict__ uv,
                                                            const double* __restrict__ Xin, double lambda, int iters,
                                                            double* __restrict__ Xout) {
  extern __shared__ double lds_proj[];
  const bool in_lds = n_views <= kTriLdsViews;
  if (in_lds) {
    for (int i = threadIdx.x; i < n_views * 12; i += blockDim.x) lds_proj[i] = projs[i];
    __syncthreads();
  }
  const double* P_all = in_lds ? lds_proj : projs;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  double x0 = Xin[p], x1 = Xin[(size_t)m + p], x2 = Xin[2 * (size_t)m + p];
  const double x3 = Xin[3 * (size_t)m + p];
  constexpr int NC = NV > 0 ? NV : 1;
  double ku[NC], kv[NC];
  if (NV > 0) {
#pragma unroll
    for (int v = 0; v < NV; ++v) { ku[v] = uv[((size_t)v * 2 + 0) * m + p]; kv[v] = uv[((size_t)v * 2 + 1) * m + p]; }
  }
  for (int it = 0; it < iters; ++it) {
    double a00 = 0, a10 = 0, a11 = 0, a20 = 0, a21 = 0, a22 = 0, b0 = 0, b1 = 0, b2 = 0;
    auto view = [&](const double* P, double ku_v, double kv_v) {
      double s[3], j[6];
#pragma unroll
      for (int r = 0; r < 3; ++r) s[r] = P[4 * r] * x0 + P[4 * r + 1] * x1 + P[4 * r + 2] * x2 + P[4 * r + 3] * x3;
      const double iz = rcp_nr(s[2]), iz2 = iz * iz;       // v_rcp_f64 + two Newton steps (a true division is ~2x the instructions)
#pragma unroll
      for (int c = 0; c < 3; ++c) {                       // tri:261-269
        j[c] = (s[2] * P[c] - s[0] * P[8 + c]) * iz2;
        j[3 + c] = (s[2] * P[4 + c] - s[1] * P[8 + c]) * iz2;
      }
      const double eu = s[0] * iz - ku_v;
      const double ev = s[1] * iz - kv_v;
      a00 += j[0] * j[0] + j[3] * j[3];
      a10 += j[1] * j[0] + j[4] * j[3];
      a11 += j[1] * j[1] + j[4] * j[4];
      a20 += j[2] * j[0] + j[5] * j[3];
      a21 += j[2] * j[1] + j[5] * j[4];
      a22 += j[2] * j[2] + j[5] * j[5];
      b0 += j[0] * eu + j[3] * ev;
      b1 += j[1] * eu + j[4] * ev;
      b2 += j[2] * eu + j[5] * ev;
    };
    if (NV > 0) {
#pragma unroll
      for (int v = 0; v < NV; ++v) view(P_all + 12 * v, ku[v], kv[v]);
    } else {
      for (int v = 0; v < n_views; ++v)
        view(P_all + 12 * v, uv[((size_t)v * 2 + 0) * m + p], uv[((size_t)v * 2 + 1) * m + p]);
    }
    a00 += lambda; a11 += lambda; a22 += lambda;
    // symmetric 3x3 inverse by adjugate (np.linalg.inv in the reference, tri:227)
    const double c00 = a11 * a22 - a21 * a21;
    const double c10 = a20 * a21 - a10 * a22;
    const double c20 = a10 * a21 - a20 * a11;
    const double det = a00 * c00 + a10 * c10 + a20 * c20;
    const double id = rcp_nr(det);
    const double c11 = a00 * a22 - a20 * a20;
    const double c21 = a10 * a20 - a00 * a21;
    const double c22 = a00 * a11 - a10 * a10;
    x0 -= (c00 * b0 + c10 * b1 + c20 * b2) * id;
    x1 -= (c10 * b0 + c11 * b1 + c21 * b2) * id;
    x2 -= (c20 * b0 + c21 * b1 + c22 * b2) * id;
  }
  Xout[p] = x0;
  Xout[(size_t)m + p] = x1;
  Xout[2 * (size_t)m + p] = x2;
  Xout[3 * (size_t)m + p] = x3;
}

static void launch_tri_nonlinear(int m, int n_views, const double* projs, const double* uv, const double* Xin, double lambda,
                                 int iters, double* Xout, size_t lds, hipStream_t s) {
  const dim3 grid((m + 255) / 256), block(256);
  switch (n_views) {
    case 2: tri_nonlinear_kernel<2><<<grid, block, lds, s>>>(m, n_views, projs, uv, Xin, lambda, iters, Xout); break;
    case 3: tri_nonlinear_kernel<3><<<grid, block, lds, s>>>(m, n_views, projs, uv, Xin, lambda, iters, Xout); break;
    case 4: tri_nonlinear_kernel<4><<<grid, block, lds, s>>>(m, n_views, projs, uv, Xin, lambda, iters, Xout); break;
    default: tri_nonlinear_kernel<0><<<grid, block, lds, s>>>(m, n_views, projs, uv, Xin, lambda, iters, Xout); break;
  }
}

// ---------------------------------------------------------------------------------------------
// Linear (DLT) triangulation, TriangulationProcessor.linear_triangulate (triangulation_processor.py:91-157):
// per point the right singular vector of the smallest singular value of A (2V x 4), rows
// u P[2,:] - P[0,:] and v P[2,:] - P[1,:], divided by its W.  One thread per point:
//   1. streaming Givens QR of the rows into a 4x4 upper-triangular R (same right singular vectors as A;
//      no Gram matrix, so the conditioning is not squared),
//   2. one-sided Jacobi (Hestenes) SVD of R in registers: columns are rotated until orthogonal, the
//      rotations accumulate in V; the column of smallest norm gives the null vector.
// Agrees with the reference's np.linalg.svd result to ~1e-13 relative on its own 1538-pair fixture.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void dlt_add_row(double (&R)[4][4], double r0, double r1, double r2, double r3) {
  double row[4] = {r0, r1, r2, r3};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const double a = R[i][i], b = row[i];
    if (b != 0.0) {
      const double h = sqrt(a * a + b * b);
      const double c = a / h, s = b / h;
#pragma unroll
      for (int k = i; k < 4; ++k) {
        const double x = R[i][k], y = row[k];
        R[i][k] = c * x + s * y;
        row[k] = -s * x + c * y;
      }
    }
  }
}

__device__ __forceinline__ void dlt_null_vector(double (&B)[4][4], double* x_out) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 16; ++sweep) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int q = p + 1; q < 4; ++q) {
        double al = 0, be = 0, ga = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { al += B[k][p] * B[k][p]; be += B[k][q] * B[k][q]; ga += B[k][p] * B[k][q]; }
        if (fabs(ga) > 1e-17 * sqrt(al * be) && ga != 0.0) {
          rotated = true;
          const double ze = (be - al) / (2.0 * ga);
          const double t = (ze == 0.0) ? 1.0 : copysign(1.0, ze) / (fabs(ze) + sqrt(1.0 + ze * ze));
          const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const double bp = B[k][p], bq = B[k][q];
            B[k][p] = c * bp - s * bq; B[k][q] = s * bp + c * bq;
            const double vp = V[k][p], vq = V[k][q];
            V[k][p] = c * vp - s * vq; V[k][q] = s * vp + c * vq;
          }
        }
      }
    }
    if (!rotated) break;
  }
  double best = 0;
  double v[4] = {0, 0, 0, 1};
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double n = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) n += B[k][c] * B[k][c];
    if (c == 0 || n < best) {
      best = n;
#pragma unroll
      for (int k = 0; k < 4; ++k) v[k] = V[k][c];
    }
  }
  x_out[0] = v[0] / v[3]; x_out[1] = v[1] / v[3]; x_out[2] = v[2] / v[3]; x_out[3] = v[3] / v[3];   // tri:152
}

__global__ __launch_bounds__(256) void tri_linear_kernel(int m, int n_views, const double* __restrict__ projs,
                                                         const double* __restrict__ uv, double* __restrict__ Xout) {
  extern __shared__ double lds_proj[];
  const bool in_lds = n_views <= kTriLdsViews;
  if (in_lds) {
    for (int i = threadIdx.x; i < n_views * 12; i += blockDim.x) lds_proj[i] = projs[i];
    __syncthreads();
  }
  const double* P_all = in_lds ? lds_proj : projs;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  double R[4][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  for (int v = 0; v < n_views; ++v) {
    const double* P = P_all + 12 * v;
    const double u = uv[((size_t)v * 2 + 0) * m + p], w = uv[((size_t)v * 2 + 1) * m + p];
    dlt_add_row(R, u * P[8] - P[0], u * P[9] - P[1], u * P[10] - P[2], u * P[11] - P[3]);    // tri:145
    dlt_add_row(R, w * P[8] - P[4], w * P[9] - P[5], w * P[10] - P[6], w * P[11] - P[7]);    // tri:146
  }
  double x[4];
  dlt_null_vector(R, x);
  Xout[p] = x[0];
  Xout[(size_t)m + p] = x[1];
  Xout[2 * (size_t)m + p] = x[2];
  Xout[3 * (size_t)m + p] = x[3];
}

// ---------------------------------------------------------------------------------------------
// Nonlinear PnP: one workgroup per view.  Per iteration every thread linearises its points and
// keeps the lower triangle of J^T J (28) and J^T e (7) in registers; a wave reduction + 4-wave LDS
// sum gives the 7x7 normal equations; one lane solves them, updates the parameter block,
// re-normalises the quaternion and rebuilds R(q) for the next iteration.
// Quirk Q1 (default): the reference stores each point's 2 rows at [pt : pt+2], so only the u-row of
// every point and the v-row of the LAST point survive (campose_processor.py:404-405).
// ---------------------------------------------------------------------------------------------
// (J^T J + lambda I) x = b for the 7x7 SPD normal equations, lower triangle packed row-wise (a[i(i+1)/2 + j]).
// Fully unrolled Cholesky in registers (no pivoting needed for an SPD matrix; the reference inverts with LU,
// campose:409 -- same solution to rounding).  a is destroyed, b becomes the solution.
__device__ __forceinline__ void solve7_spd(double (&a)[28], double (&b)[7]) {
  double inv[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    double d = a[j * (j + 1) / 2 + j];
#pragma unroll
    for (int k = 0; k < j; ++k) d -= a[j * (j + 1) / 2 + k] * a[j * (j + 1) / 2 + k];
    inv[j] = rsqrt_nr(d);
    a[j * (j + 1) / 2 + j] = d * inv[j];
#pragma unroll
    for (int i = j + 1; i < 7; ++i) {
      double v = a[i * (i + 1) / 2 + j];
#pragma unroll
      for (int k = 0; k < j; ++k) v -= a[i * (i + 1) / 2 + k] * a[j * (j + 1) / 2 + k];
      a[i * (i + 1) / 2 + j] = v * inv[j];
    }
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {          // L y = b
    double v = b[i];
#pragma unroll
    for (int k = 0; k < i; ++k) v -= a[i * (i + 1) / 2 + k] * b[k];
    b[i] = v * inv[i];
  }
#pragma unroll
  for (int i = 6; i >= 0; --i) {         // L^T x = y
    double v = b[i];
#pragma unroll
    for (int k = i + 1; k < 7; ++k) v -= a[k * (k + 1) / 2 + i] * b[k];
    b[i] = v * inv[i];
  }
}

// One workgroup per view.  The view's points (and their normalised keys, a constant of the problem) stay in
// registers across all iterations when the view has at most 256 * PNP_CACHE points; per iteration every thread
// linearises its points (35 accumulators), a DPP wave reduction + a 35-thread sum over the 4 waves gives the
// normal equations, one lane solves them and prepares the next camera.
// THREADS = 256 with four cached points per thread serves views of up to 1024 points; larger views take THREADS = 512 (two
// cached points per thread; sixteen waves would cap the kernel at 128 VGPRs and spill).  More lanes shorten only the
// linearisation: every wave pays the same 35-value reduction (~630 issue slots) and the same serial 7x7 solve, so at
// 1000 points 256 threads x 4 points take 4.6 us per iteration against 5.5 us for 512 x 2 (profiles/r3/bench_pnp_*.json).
// SPLIT (round 4, third size class): a view of more than kPnpSplitMin points is dealt to k = ceil(n / 1024) workgroups of the
// 256-thread, register-resident form (workgroup = blockIdx.x % kmax of view blockIdx.x / kmax; slices of ceil(n / k) points).
// Per iteration every workgroup stores the 35 sums of its slice, publishes its flag word (iterations published), waits until
// every sibling's flag says the same and adds the k partial vectors IN SLICE ORDER -- so every workgroup holds the same bits,
// carries out the same serial part and no camera has to be published (one hand-over per iteration, ~2 us, instead of two).
// The partial vectors are double-buffered by iteration parity: a workgroup cannot be two iterations ahead of a sibling (it
// needs the sibling's flag of the iteration in between, which the sibling sets after it has read the older vectors).
// Which kernel refines a view still depends on the view's own size only.
struct PnpSplitWs {
  double* xch;   // [n_views][2][kmax][35]
  int* ctr;      // [n_views][kmax] per-slice flag words: iterations published (monotone over a launch; cleared before it)
  int kmax;
  int slice_pts; // points per slice (<= 1024: four register-resident points per thread)
};

template <int THREADS, int PNP_CACHE, bool SPLIT = false>
__global__ __launch_bounds__(THREADS) void pnp_nonlinear_kernel(const int* __restrict__ offsets, int total,
                                                                const double* __restrict__ uv_pix,
                                                                const double* __restrict__ X,
                                                                const double* __restrict__ Kmat,
                                                                const double* __restrict__ R0,
                                                                const double* __restrict__ C0, double lambda, int iters,
                                                                int quirks, double* __restrict__ R_out,
                                                                double* __restrict__ C_out, int* __restrict__ status,
                                                                int stage_mode, int stage_cap, int n_lo, int n_hi,
                                                                PnpSplitWs ws = PnpSplitWs{nullptr, nullptr, 1, 1024}, int view0 = 0) {
  // Views too large for the register cache keep their points in LDS when they fit (stage_mode 1: X, Y, Z, W and the
  // normalised key, 48 bytes per point, SoA over stage_cap points; 2: the normalised key only, the point is re-read from
  // L2): the key normalisation -- two divisions per point -- is then done once, not in every iteration.
  extern __shared__ double pnp_stage[];
  constexpr int WAVES = THREADS / 64;
  __shared__ double kinv[9];
  __shared__ double red[4 * WAVES][35];      // one partial per 16-lane row of every wave
  __shared__ double sums[35];
  const int view = view0 + (SPLIT ? (int)blockIdx.x / ws.kmax : (int)blockIdx.x);
  const int slice = SPLIT ? (int)blockIdx.x % ws.kmax : 0;
  const int base = offsets[view];
  const int n = offsets[view + 1] - base;
  if (n < n_lo || n > n_hi) return;          // the other size class' launch refines this view (enqueue_pnp_nonlinear)
  const int nslices = SPLIT ? (n + ws.slice_pts - 1) / ws.slice_pts : 1;
  if (SPLIT && slice >= nslices) return;
  const int per = SPLIT ? (n + nslices - 1) / nslices : n;
  const int p_lo = slice * per;                          // this workgroup's points: [p_lo, p_lo + n_mine) of the view
  const int n_mine = SPLIT ? max(0, min(n, p_lo + per) - p_lo) : n;
  __shared__ int split_fail;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  if (SPLIT && tid == 0) split_fail = 0;

  if (tid == 0) {
    const double* K = Kmat + 9 * view;
    const double det = det3(K);
    const double id = 1.0 / det;
    kinv[0] = (K[4] * K[8] - K[5] * K[7]) * id; kinv[1] = (K[2] * K[7] - K[1] * K[8]) * id; kinv[2] = (K[1] * K[5] - K[2] * K[4]) * id;
    kinv[3] = (K[5] * K[6] - K[3] * K[8]) * id; kinv[4] = (K[0] * K[8] - K[2] * K[6]) * id; kinv[5] = (K[2] * K[3] - K[0] * K[5]) * id;
    kinv[6] = (K[3] * K[7] - K[4] * K[6]) * id; kinv[7] = (K[1] * K[6] - K[0] * K[7]) * id; kinv[8] = (K[0] * K[4] - K[1] * K[3]) * id;
  }
  // The serial part of an iteration -- the 7x7 solve, the parameter update, R(q) and its checks -- is carried out by EVERY
  // thread on the same reduced sums (same instruction stream, same bits): the camera stays in registers, and an iteration
  // needs two barriers instead of three plus a round trip of the expanded camera through LDS.
  // campose:361-371: q0 = q(R0) / |q(R0)|; iteration 0 linearises at (R0, C0) themselves
  CamPrep c;
  int st = cam_prepare_rc(R0 + 9 * view, C0 + 3 * view, &c);
  double params[7];
  {
    const double inq = rsqrt_nr(c.q[0] * c.q[0] + c.q[1] * c.q[1] + c.q[2] * c.q[2] + c.q[3] * c.q[3]);
#pragma unroll
    for (int k = 0; k < 3; ++k) params[k] = C0[3 * view + k];
#pragma unroll
    for (int k = 0; k < 4; ++k) params[3 + k] = c.q[k] * inq;
  }
  __syncthreads();

  // point p of this thread: homogeneous X and the normalised key (campose:390-395: inv(K) [u,v,h] / its z)
  auto load_point = [&](int p, double (&pt)[6]) {
    const size_t col = (size_t)base + p;
    pt[0] = X[col]; pt[1] = X[(size_t)total + col]; pt[2] = X[2 * (size_t)total + col]; pt[3] = X[3 * (size_t)total + col];
    const double u = uv_pix[col], v = uv_pix[(size_t)total + col], h = uv_pix[2 * (size_t)total + col];
    const double m2 = kinv[6] * u + kinv[7] * v + kinv[8] * h;
    pt[4] = (kinv[0] * u + kinv[1] * v + kinv[2] * h) / m2;
    pt[5] = (kinv[3] * u + kinv[4] * v + kinv[5] * h) / m2;
  };
  const bool cached = n_mine <= THREADS * PNP_CACHE;
  // (the 256-thread variant never stages: its branch-free loop is what views of up to 1 024 points run)
  const int staged = (THREADS > 256 && !cached && n <= stage_cap) ? stage_mode : 0;
  double pts[PNP_CACHE][6];
  if (cached) {
#pragma unroll
    for (int cc = 0; cc < PNP_CACHE; ++cc)
      if (tid + THREADS * cc < n_mine) load_point(p_lo + tid + THREADS * cc, pts[cc]);
  } else if (staged) {
    for (int p = tid; p < n; p += THREADS) {
      double pt[6];
      load_point(p, pt);
      if (staged == 1) {
#pragma unroll
        for (int k = 0; k < 6; ++k) pnp_stage[(size_t)k * stage_cap + p] = pt[k];
      } else {
        pnp_stage[p] = pt[4];
        pnp_stage[(size_t)stage_cap + p] = pt[5];
      }
    }
    __syncthreads();
  }

  // which three of the 35 sums this lane holds after the fold of an iteration
  int own[3];
  {
    int i35[35], i18[18], i9[9], i5[5];
#pragma unroll
    for (int k = 0; k < 35; ++k) i35[k] = k;
    fold_index(i35, i18, (lane & 8) != 0);
    fold_index(i18, i9, (lane & 4) != 0);
    fold_index(i9, i5, (lane & 2) != 0);
    fold_index(i5, own, (lane & 1) != 0);
  }

  for (int it = 0; it < iters && st == SFM_OK; ++it) {
    double acc[35];
#pragma unroll
    for (int k = 0; k < 35; ++k) acc[k] = 0;
    auto accumulate = [&](const double (&pt)[6], int p) {
      double pc[3], jp[14];
      project_cam(c, pt[0], pt[1], pt[2], pt[3], pc);
      const double iz = rcp_nr(pc[2]);
      const bool use_v = !(quirks & SFM_Q1_PNP_ROW_OVERLAP) || (p == n - 1);
      // quirk Q1 keeps only the u-row of every point but the last: the v-row of the Jacobian is not even formed for them
      if (use_v) jac_cam_iz(c, pt[0], pt[1], pt[2], pc, iz, quirks, jp);
      else jac_cam_iz_urow(c, pt[0], pt[1], pt[2], pc, iz, jp);
      const double eu = pt[4] - pc[0] * iz, ev = pt[5] - pc[1] * iz;
      int k = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) { acc[k] += jp[i] * jp[j]; ++k; }
      }
#pragma unroll
      for (int i = 0; i < 7; ++i) acc[28 + i] += jp[i] * eu;
      if (use_v) {
        k = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) { acc[k] += jp[7 + i] * jp[7 + j]; ++k; }
        }
#pragma unroll
        for (int i = 0; i < 7; ++i) acc[28 + i] += jp[7 + i] * ev;
      }
    };
    if (cached) {
#pragma unroll
      for (int cc = 0; cc < PNP_CACHE; ++cc)
        if (tid + THREADS * cc < n_mine) accumulate(pts[cc], p_lo + tid + THREADS * cc);
    } else {
      for (int p = tid; p < n; p += blockDim.x) {
        double pt[6];
        if (staged == 1) {
#pragma unroll
          for (int k = 0; k < 6; ++k) pt[k] = pnp_stage[(size_t)k * stage_cap + p];
        } else if (staged == 2) {
          const size_t col = (size_t)base + p;
          pt[0] = X[col]; pt[1] = X[(size_t)total + col]; pt[2] = X[2 * (size_t)total + col]; pt[3] = X[3 * (size_t)total + col];
          pt[4] = pnp_stage[p]; pt[5] = pnp_stage[(size_t)stage_cap + p];
        } else {
          load_point(p, pt);
        }
        accumulate(pt, p);
      }
    }
    // 35 sums over the workgroup: a four-step halving fold (fold_half, sfm_common.h) leaves three of the 35 totals of a
    // 16-lane row in each of its lanes; they go to LDS and 35 threads add the 4 x WAVES row partials
    {
      double f18[18], f9[9], f5[5], f3[3];
      fold_half<0x140>(acc, f18, (lane & 8) != 0);
      fold_half<0x141>(f18, f9, (lane & 4) != 0);
      fold_half<0x4E>(f9, f5, (lane & 2) != 0);
      fold_half<0xB1>(f5, f3, (lane & 1) != 0);
#pragma unroll
      for (int j = 0; j < 3; ++j) red[4 * wave + (lane >> 4)][own[j]] = f3[j];
    }
    __syncthreads();
    if (tid < 35) {
      double t = red[0][tid];
#pragma unroll
      for (int w = 1; w < 4 * WAVES; ++w) t += red[w][tid];
      if (SPLIT) ws.xch[(((size_t)view * 2 + (it & 1)) * ws.kmax + slice) * 35 + tid] = t;
      else sums[tid] = t;
    }
    if (SPLIT) {
      // hand-over (the general recipe of the CDNA guide): wave 0 stored the slice's sums -> agent-scope release -> its flag
      // word = iterations published -> lane g polls sibling g's flag (no read-modify-write on a shared counter) -> workgroup
      // barrier -> agent-scope acquire -> the partial vectors of all slices, added in slice order.
      // (Measured and not kept: the sums as sc1 stores / loads in place of the two fences, 5.44 instead of 5.66 us per iteration at
      //  5 000 points -- the guide lists that form as measured for one workgroup per CU only.)
      if (wave == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        int* flags = ws.ctr + (size_t)view * ws.kmax;
        if (tid == 0) __hip_atomic_store(flags + slice, it + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        const int want = it + 1;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
          bool mine = true;
          for (int g = lane; g < nslices; g += 64)
            mine = mine && __hip_atomic_load(flags + g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= want;
          if (__all(mine)) break;
          if (wall_clock64() - t0 > 200000000ull) { if (lane == 0) split_fail = 1; break; }      // 2 s: a sibling never ran (never seen)
          __builtin_amdgcn_s_sleep(1);
        }
      }
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      if (tid < 35 && !split_fail) {
        const double* part = ws.xch + (((size_t)view * 2 + (it & 1)) * ws.kmax) * 35 + tid;
        double t = part[0];
        for (int g = 1; g < nslices; ++g) t += part[(size_t)g * 35];
        sums[tid] = t;
      }
    }
    __syncthreads();
    if (SPLIT && split_fail) { st = SFM_E_HIP; break; }
    double a[28], b[7];
#pragma unroll
    for (int k = 0; k < 28; ++k) a[k] = sums[k];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      a[i * (i + 1) / 2 + i] += lambda;
      b[i] = sums[28 + i];
    }
    solve7_spd(a, b);
#pragma unroll
    for (int i = 0; i < 7; ++i) params[i] += b[i];
    const double inq = rsqrt_nr(params[3] * params[3] + params[4] * params[4] + params[5] * params[5] + params[6] * params[6]);
#pragma unroll
    for (int i = 3; i < 7; ++i) params[i] *= inq;
    // campose:422: R = R(q) validated -- for the quaternion normalised two lines up the determinant / inverse test cannot
    // fire (cam_prepare_dev<false>, sfm_common.h); the next Jacobian re-derives q from R (campose:464)
    st = cam_prepare_dev<false>(params, &c);
  }
  if (tid == 0 && slice == 0) {
    if (iters <= 0 && st == SFM_OK) st = cam_prepare_dev(params, &c);      // no iteration: the reference still returns R(q0) (campose:458)
    for (int k = 0; k < 9; ++k) R_out[9 * view + k] = c.R[k];
    for (int k = 0; k < 3; ++k) C_out[3 * view + k] = params[k];
    status[view] = st;
  }
}

// ---------------------------------------------------------------------------------------------
// Linear (DLT) PnP hypotheses for RANSAC, CamposeProcessor.__linear_determine_cam_pos /
// __estimate_six_pts (campose_processor.py:485-633).  The six-point samples are drawn on the host
// (the reference consumes Python's global `random` stream, campose:531); the device evaluates every
// hypothesis:
//   pnp_six_point_kernel   thread per hypothesis: 12x12 system from the 6 normalised keys and points
//                          (campose:585-613), null vector by one-sided Jacobi SVD (12x12 work arrays live
//                          in scratch: dynamic column indexing), 3x3 polar factor + largest singular value
//                          (campose:619-631), K [R^T | -R^T C] for the scoring pass
//   pnp_score_kernel       workgroup per hypothesis: pixel reprojection error of every point against the
//                          threshold (campose:544-554) -> inlier count
//   pnp_inlier_mask_kernel inlier mask of the winning hypothesis
// ---------------------------------------------------------------------------------------------
// One wave per hypothesis: lanes 0..11 build the rows of the 12x12 design matrix (campose:588-611), lanes
// 16..27 the identity; jacobi_rows_wave leaves B V in the first group and V in the second.
__global__ __launch_bounds__(64) void pnp_six_point_kernel(int n_hyp, int n, const int* __restrict__ samples /*[n_hyp][6]*/,
                                                           const double* __restrict__ uv_pix /*[3][n]*/,
                                                           const double* __restrict__ X /*[4][n]*/,
                                                           const double* __restrict__ Kmat,
                                                           double* __restrict__ R_out /*[n_hyp][9]*/,
                                                           double* __restrict__ C_out /*[n_hyp][3]*/,
                                                           double* __restrict__ proj_out /*[n_hyp][12]*/,
                                                           double* __restrict__ proj_neg_out /*[n_hyp][12] or null: K [R^T | R^T C], the pose (R, -C)*/) {
  const int h = blockIdx.x;
  const int lane = threadIdx.x;
  if (h >= n_hyp) return;
  const double* K = Kmat;
  const double idet = 1.0 / det3(K);
  double ki[9];
  ki[0] = (K[4] * K[8] - K[5] * K[7]) * idet; ki[1] = (K[2] * K[7] - K[1] * K[8]) * idet; ki[2] = (K[1] * K[5] - K[2] * K[4]) * idet;
  ki[3] = (K[5] * K[6] - K[3] * K[8]) * idet; ki[4] = (K[0] * K[8] - K[2] * K[6]) * idet; ki[5] = (K[2] * K[3] - K[0] * K[5]) * idet;
  ki[6] = (K[3] * K[7] - K[4] * K[6]) * idet; ki[7] = (K[1] * K[6] - K[0] * K[7]) * idet; ki[8] = (K[0] * K[4] - K[1] * K[3]) * idet;
  double row[12];
#pragma unroll
  for (int c = 0; c < 12; ++c) row[c] = 0.0;
  if (lane < 12) {
    const int idx = samples[6 * h + lane / 2];
    const double u = uv_pix[idx], v = uv_pix[(size_t)n + idx], w = uv_pix[2 * (size_t)n + idx];
    const double p0 = ki[0] * u + ki[1] * v + ki[2] * w;       // key in camera coordinates (campose:535, not re-normalised)
    const double p1 = ki[3] * u + ki[4] * v + ki[5] * w;
    const double p2 = ki[6] * u + ki[7] * v + ki[8] * w;
    const double x = X[idx], y = X[(size_t)n + idx], z = X[2 * (size_t)n + idx];
    const double pm = (lane & 1) ? p1 : p0;
    const int o = (lane & 1) ? 4 : 0;                          // even rows fill columns 0..3, odd rows 4..7
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const double xc = c == 0 ? x : (c == 1 ? y : (c == 2 ? z : 1.0));
      row[c] = o == 0 ? p2 * xc : 0.0;
      row[4 + c] = o == 4 ? p2 * xc : 0.0;
      row[8 + c] = -pm * xc;
    }
  } else if (lane >= 16 && lane < 28) {
#pragma unroll
    for (int c = 0; c < 12; ++c) row[c] = (c == lane - 16) ? 1.0 : 0.0;
  }
  jacobi_rows_wave<12>(row, lane, 30);
  // column of smallest norm of B V -> the null vector is that column of V (rows in lanes 16..27)
  int best = 0;
  double bn = 0;
#pragma unroll
  for (int c = 0; c < 12; ++c) {
    const double nn = wave_lane0(group_sum<16>(lane < 16 ? row[c] * row[c] : 0.0));
    if (c == 0 || nn < bn) { bn = nn; best = c; }
  }
  double mine = 0.0;
#pragma unroll
  for (int c = 0; c < 12; ++c) mine = (c == best) ? row[c] : mine;
  double cam[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) {                                // cam_mat (3x4) row-major (campose:618), on every lane
    const int lo = __builtin_amdgcn_readlane(__double2loint(mine), 16 + k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(mine), 16 + k);
    cam[k] = __hiloint2double(hi, lo);
  }
  if (lane != 0) return;
  // polar factor of the 3x3 left block and its largest singular value (campose:622-626)
  double B3[3][3], V3[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B3[i][j] = cam[4 * i + j];
  jacobi_right_vectors<3>(B3, V3, 30);
  double sig[3], smax = 0;
  for (int c = 0; c < 3; ++c) {
    sig[c] = sqrt(B3[0][c] * B3[0][c] + B3[1][c] * B3[1][c] + B3[2][c] * B3[2][c]);
    smax = fmax(smax, sig[c]);
  }
  double UV[9];   // U V^T = sum_c (B3[:,c] / sig_c) V3[:,c]^T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double a = 0;
      for (int c = 0; c < 3; ++c) a += (B3[i][c] / sig[c]) * V3[j][c];
      UV[3 * i + j] = a;
    }
  double R[9];    // rot = (uu @ vvh).T
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) R[3 * i + j] = UV[3 * j + i];
  double C[3];
  for (int i = 0; i < 3; ++i) C[i] = (R[3 * i] * -cam[3] + R[3 * i + 1] * -cam[7] + R[3 * i + 2] * -cam[11]) / smax;
  // campose:629-631 negates rot AND loc when det(rot) < 0.  C = rot (-cam[:,3]) / s is already invariant under
  // the arbitrary sign of the null vector (rot and cam flip together), so negating it makes the reference's
  // result depend on LAPACK's sign choice: about half of its hypotheses come out with -C and score ~0 inliers
  // (quirk Q13).  The device keeps the sign-invariant C; on every hypothesis the reference did not
  // ruin the two agree, and on its own PnP fixture the winning hypothesis, inlier set and pose are identical.
  if (det3(R) < 0) {
    for (int i = 0; i < 9; ++i) R[i] = -R[i];
  }
  for (int i = 0; i < 9; ++i) R_out[9 * h + i] = R[i];
  for (int i = 0; i < 3; ++i) C_out[3 * h + i] = C[i];
  // proj = K @ [R^T | R^T @ -C]  (campose:538)
  double rt[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) rt[4 * i + j] = R[3 * j + i];
    rt[4 * i + 3] = R[0 + i] * -C[0] + R[3 + i] * -C[1] + R[6 + i] * -C[2];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) proj_out[12 * h + 4 * i + j] = K[3 * i] * rt[j] + K[3 * i + 1] * rt[4 + j] + K[3 * i + 2] * rt[8 + j];
  if (proj_neg_out) {      // what the reference scores when its det(rot) < 0 branch fired (campose:629-631): rot = R, loc = -C
    for (int i = 0; i < 3; ++i) rt[4 * i + 3] = -rt[4 * i + 3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 4; ++j) proj_neg_out[12 * h + 4 * i + j] = K[3 * i] * rt[j] + K[3 * i + 1] * rt[4 + j] + K[3 * i + 2] * rt[8 + j];
  }
}

__device__ __forceinline__ bool pnp_is_inlier(const double* P, const double* uv_pix, const double* X, int n, int i,
                                              double threshold) {
  const double x = X[i], y = X[(size_t)n + i], z = X[2 * (size_t)n + i], w = X[3 * (size_t)n + i];
  const double s0 = P[0] * x + P[1] * y + P[2] * z + P[3] * w;
  const double s1 = P[4] * x + P[5] * y + P[6] * z + P[7] * w;
  const double s2 = P[8] * x + P[9] * y + P[10] * z + P[11] * w;
  const double e0 = uv_pix[i] - s0 / s2, e1 = uv_pix[(size_t)n + i] - s1 / s2, e2 = uv_pix[2 * (size_t)n + i] - s2 / s2;
  return sqrt(e0 * e0 + e1 * e1 + e2 * e2) < threshold;          // campose:550-552
}

__global__ __launch_bounds__(256) void pnp_score_kernel(int n, const double* __restrict__ proj, const double* __restrict__ uv_pix,
                                                        const double* __restrict__ X, double threshold,
                                                        int* __restrict__ counts) {
  __shared__ int wsum[4];
  const int h = blockIdx.x;
  double P[12];
  for (int k = 0; k < 12; ++k) P[k] = proj[12 * h + k];
  int cnt = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) cnt += pnp_is_inlier(P, uv_pix, X, n, i, threshold) ? 1 : 0;
  for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[h] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ void pnp_inlier_mask_kernel(int n, const double* __restrict__ proj, const double* __restrict__ uv_pix,
                                       const double* __restrict__ X, double threshold, int* __restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double P[12];
  for (int k = 0; k < 12; ++k) P[k] = proj[k];
  mask[i] = pnp_is_inlier(P, uv_pix, X, n, i, threshold) ? 1 : 0;
}

// Stable compaction of the inlier columns (campose_processor.py:236-237: key_2d_pts[:, inlier_indices], tri_3d_pts[:, inlier_indices]
// with the indices ascending): one workgroup scans the mask, then every inlier column goes to its rank.  The compacted arrays
// keep the row pitch n of the originals (only the first m columns are written); offsets = {0, m} for the nonlinear kernel.
__global__ __launch_bounds__(1024) void pnp_mask_scan_kernel(int n, const int* __restrict__ mask, int* __restrict__ pos, int* __restrict__ offsets) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const int a = i < n ? (mask[i] != 0) : 0;
    int sa = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(sa, off, 64); if (lane >= off) sa += t; }
    if (lane == 63) wsum[wave] = sa;
    __syncthreads();
    int o = carry;
    for (int w = 0; w < wave; ++w) o += wsum[w];
    if (i < n) pos[i] = o + sa - a;
    __syncthreads();
    if (tid == 1023) carry = o + sa;
    __syncthreads();
  }
  if (tid == 0) { offsets[0] = 0; offsets[1] = carry; }
}

__global__ void pnp_compact_kernel(int n, const int* __restrict__ mask, const int* __restrict__ pos, const double* __restrict__ uv_pix,
                                   const double* __restrict__ X, double* __restrict__ uv_c, double* __restrict__ X_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || mask[i] == 0) return;
  const int q = pos[i];
#pragma unroll
  for (int r = 0; r < 3; ++r) uv_c[(size_t)r * n + q] = uv_pix[(size_t)r * n + i];
#pragma unroll
  for (int r = 0; r < 4; ++r) X_c[(size_t)r * n + q] = X[(size_t)r * n + i];
}

// Columns [4][n] (X, Y, Z, 1) of the points index[0..n) of SoA point arrays (e.g. a resident BA problem's, sfm_ba_points_ptr):
// what the per-view PnP of the incremental loop consumes (ba_processor.py:184-191), built without a host round trip.
__global__ void gather_points_kernel(int n, const int* __restrict__ index, const double* __restrict__ px,
                                     const double* __restrict__ py, const double* __restrict__ pz, double* __restrict__ X) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = index[i];
  X[i] = px[p]; X[(size_t)n + i] = py[p]; X[2 * (size_t)n + i] = pz[p]; X[3 * (size_t)n + i] = 1.0;
}

// Workspace of the split PnP class, one per stream (calls on one stream are ordered, so it is reused; it only grows and
// lives until sfm_shutdown).
struct PnpSplitCache { hipStream_t stream; double* xch; int* ctr; size_t xch_doubles; int views; };
static std::vector<PnpSplitCache>& pnp_split_cache() { static std::vector<PnpSplitCache> c; return c; }
static int pnp_split_workspace(hipStream_t s, int n_views, int kmax, PnpSplitWs* out) {
  auto& cache = pnp_split_cache();
  PnpSplitCache* e = nullptr;
  for (auto& c : cache) if (c.stream == s) e = &c;
  if (!e) { cache.push_back(PnpSplitCache{s, nullptr, nullptr, 0, 0}); e = &cache.back(); }
  const size_t need = (size_t)n_views * 2 * kmax * 35;
  if (need > e->xch_doubles) {
    if (e->xch) { SFM_HIP(hipStreamSynchronize(s)); pool_free(e->xch); e->xch = nullptr; }
    SFM_HIP(pool_alloc(reinterpret_cast<void**>(&e->xch), sizeof(double) * need));
    e->xch_doubles = need;
  }
  if (n_views * kmax > e->views) {          // (views = flag words held)
    if (e->ctr) { SFM_HIP(hipStreamSynchronize(s)); pool_free(e->ctr); e->ctr = nullptr; }
    SFM_HIP(pool_alloc(reinterpret_cast<void**>(&e->ctr), sizeof(int) * (size_t)n_views * kmax));
    e->views = n_views * kmax;
  }
  out->xch = e->xch; out->ctr = e->ctr; out->kmax = kmax;
  return SFM_OK;
}
static void pnp_split_release() {
  for (auto& c : pnp_split_cache()) { if (c.xch) pool_free(c.xch); if (c.ctr) pool_free(c.ctr); }
  pnp_split_cache().clear();
}

// views of at least this many points are split over workgroups (measured: profiles/r4/time_pnp_stages.txt)
static int pnp_split_min() {
  static const int v = [] { const char* e = getenv("SFM_PNP_SPLIT_MIN"); const int x = e ? atoi(e) : 0; return x > 1024 ? x : 3000; }();
  return v;
}

// points per slice of a split view: 1 024 = four register-resident points per thread.  Smaller slices lose: every sibling adds
// ~0.25 us to the hand-over (5 000 points: 5.65 us per iteration at 1 024, 6.61 at 512, 9.60 at 256; profiles/r4/time_pnp_slices.txt)
static int pnp_split_slice() { return 1024; }

static int enqueue_pnp_nonlinear(int n_views, const int* offsets, int total, const double* uv_pix, const double* X,
                                 const double* K, const double* R0, const double* C0, double lambda, int iters, int quirks,
                                 double* R_out, double* C_out, int* status, hipStream_t s, int narrowest, int widest) {
  // Three size classes, each with its own launch over all views (a workgroup whose view belongs to another class returns at
  // once), so that the kernel a view runs on -- and with it every bit of its result -- depends on the view's own size and
  // not on what else is in the batch (a shard of a batch returns what the whole batch returns):
  //   up to 1024 points        256 threads keep four points each in registers;
  //   up to pnp_split_min()-1  512 threads work out of LDS: all six values of a point up to 3 200 points (150 KB);
  //   above                    the view is split over ceil(n / 1024) workgroups of the first form that exchange their 35
  //                            sums once per iteration (pnp_nonlinear_kernel<256, 4, true>).
  // `narrowest` / `widest`: the smallest and the largest view (the host entry point has the offsets; the device-pointer
  // entry is told the largest, or reads the offsets back when it is not).  A class no view can be in is not launched.
  constexpr int kSmall = 1024, kAll = 0x7fffffff;
  const int split_min = pnp_split_min();
  const bool small_class = narrowest <= kSmall;
  const bool mid_class = widest > kSmall && narrowest < split_min;
  const bool split_class = widest >= split_min;
  if (small_class)
    pnp_nonlinear_kernel<256, 4><<<n_views, 256, 0, s>>>(offsets, total, uv_pix, X, K, R0, C0, lambda, iters, quirks, R_out, C_out, status, 0, 0,
                                                         0, kSmall);
  if (mid_class) {
    // dynamic LDS the kernel may ask for = what the device offers a workgroup minus the kernel's static arrays; the staging
    // capacities follow from it (the attribute call's result was dropped and the capacities were constants), and
    // a kernel that cannot have it re-reads its points (mode 0: the staging changes where values come from, not the values)
    static const int dyn_lds = [] {
      hipFuncAttributes fa{};
      hipDeviceProp_t prop{};
      const void* fn = reinterpret_cast<const void*>(pnp_nonlinear_kernel<512, 2>);
      if (hipFuncGetAttributes(&fa, fn) != hipSuccess || hipGetDeviceProperties(&prop, ctx().device) != hipSuccess) return 0;
      const long long room = (long long)prop.sharedMemPerBlock - (long long)fa.sharedSizeBytes - 256;
      const int want = (int)std::max(0LL, std::min(room, 150LL * 1024));
      if (want <= 0 || hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, want) != hipSuccess) return 0;
      return want;
    }();
    const int cap_all = dyn_lds / (6 * (int)sizeof(double)), cap_key = dyn_lds / (2 * (int)sizeof(double));      // 3 200 / 9 600 at 150 KB
    const int widest_mid = std::min(widest, split_min - 1);
    const int mode = widest_mid <= cap_all ? 1 : (widest_mid <= cap_key ? 2 : 0);
    const int cap = mode == 1 ? cap_all : (mode == 2 ? cap_key : 0);
    const size_t lds = sizeof(double) * (size_t)cap * (mode == 1 ? 6 : 2);
    pnp_nonlinear_kernel<512, 2><<<n_views, 512, lds, s>>>(offsets, total, uv_pix, X, K, R0, C0, lambda, iters, quirks, R_out, C_out, status, mode, cap,
                                                          kSmall + 1, split_min - 1);
  }
  if (split_class) {
    const int slice_pts = pnp_split_slice();
    const int kmax = (widest + slice_pts - 1) / slice_pts;
    PnpSplitWs ws;
    SFM_TRY(pnp_split_workspace(s, n_views, kmax, &ws));
    ws.slice_pts = slice_pts;
    SFM_HIP(hipMemsetAsync(ws.ctr, 0, sizeof(int) * (size_t)n_views * kmax, s));
    // the slices of a view wait for each other inside the launch: keep a launch's workgroups within what is resident at
    // once (sibling workgroups are neighbours in the grid; workgroups of views of other classes leave at once)
    const int views_per_launch = std::max(1, 2 * ctx().num_cus / kmax);
    for (int v0 = 0; v0 < n_views; v0 += views_per_launch) {
      const int nv = std::min(views_per_launch, n_views - v0);
      pnp_nonlinear_kernel<256, 4, true><<<nv * kmax, 256, 0, s>>>(offsets, total, uv_pix, X, K, R0, C0, lambda, iters, quirks, R_out, C_out, status,
                                                                  0, 0, split_min, kAll, ws, v0);
    }
  }
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

static int first_bad(const std::vector<int>& st) {
  for (size_t i = 0; i < st.size(); ++i)
    if (st[i] != SFM_OK) return st[i];
  return SFM_OK;
}

}  // namespace sfm

using namespace sfm;

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

int sfm_version(void) { return 100; }

const char* sfm_last_error(void) { return g_err; }

int sfm_init(int device) {
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0) {
    set_error("no HIP device visible (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
    return SFM_E_NO_DEVICE;
  }
  if (device < 0 || device >= count) {
    set_error("device %d out of range (0..%d)", device, count - 1);
    return SFM_E_NO_DEVICE;
  }
  Context& c = ctx();
  if (c.inited && c.device == device) return SFM_OK;
  if (c.inited) sfm_shutdown();
  SFM_HIP(hipSetDevice(device));
  hipDeviceProp_t prop;
  SFM_HIP(hipGetDeviceProperties(&prop, device));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
    set_error("device %d is %s; this library ships gfx950 (MI355X) code objects only", device, prop.gcnArchName);
    return SFM_E_NO_DEVICE;
  }
  c.num_cus = prop.multiProcessorCount;
  SFM_HIP(hipStreamCreateWithFlags(&c.own, hipStreamNonBlocking));
  c.stream = c.own;
  c.device = device;
  c.inited = true;
  return SFM_OK;
}

int sfm_shutdown(void) {
  Context& c = ctx();
  if (!c.inited) return SFM_OK;
  (void)hipStreamSynchronize(c.stream);
  if (c.stream != c.own) (void)hipStreamSynchronize(c.own);
  (void)hipDeviceSynchronize();      // the split-PnP workspaces belong to callers' streams
  pnp_split_release();
  pool_release_all();
  if (c.own) (void)hipStreamDestroy(c.own);
  c = Context();
  return SFM_OK;
}

int sfm_set_stream(void* hip_stream) {
  SFM_TRY(ensure_init());
  Context& c = ctx();
  // the library's own stream lives from sfm_init to sfm_shutdown, so a problem that captured it as its default
  // (sfm_ba_create) never holds a dangling handle; an installed stream stays the caller's
  SFM_HIP(hipStreamSynchronize(c.stream));
  c.stream = hip_stream == nullptr ? c.own : reinterpret_cast<hipStream_t>(hip_stream);
  return SFM_OK;
}

int sfm_synchronize(void) {
  SFM_TRY(ensure_init());
  SFM_TRY(stream_sync(ctx().stream));
  return SFM_OK;
}

int sfm_pool_redzone_active(void) { return redzone_on() ? 1 : 0; }

int sfm_pool_mode(int64_t probe_bytes, int64_t* tail_slack, int64_t* guard_allocs) {
  long long slack = -1, allocs = 0;
  const int mode = pool_mode_probe(probe_bytes > 0 ? (size_t)probe_bytes : 1, tail_slack ? &slack : nullptr, &allocs);
  if (tail_slack) *tail_slack = slack;
  if (guard_allocs) *guard_allocs = allocs;
  return mode;
}

int sfm_quat_to_rot(int n, const double* q, double* R, int* status) {
  SFM_TRY(ensure_init());
  if (n < 0) { set_error("sfm_quat_to_rot: n < 0"); return SFM_E_SHAPE; }
  if (n == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dq, dR; DevBuf<int> dst;
  SFM_TRY(dq.upload(q, 4 * (size_t)n, s)); SFM_TRY(dR.alloc(9 * (size_t)n)); SFM_TRY(dst.alloc(n));
  quat_to_rot_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dq.p, dR.p, dst.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dR.download(R, 9 * (size_t)n, s)); SFM_TRY(dst.download(status, n, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_rot_to_quat(int n, const double* R, double* q, int* status) {
  SFM_TRY(ensure_init());
  if (n < 0) { set_error("sfm_rot_to_quat: n < 0"); return SFM_E_SHAPE; }
  if (n == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dq, dR; DevBuf<int> dst;
  SFM_TRY(dR.upload(R, 9 * (size_t)n, s)); SFM_TRY(dq.alloc(4 * (size_t)n)); SFM_TRY(dst.alloc(n));
  rot_to_quat_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dR.p, dq.p, dst.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dq.download(q, 4 * (size_t)n, s)); SFM_TRY(dst.download(status, n, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_jac_cam(int n, const double* R, const double* C, const double* X, int quirks, double* Jp, int* status) {
  SFM_TRY(ensure_init());
  if (n < 0) { set_error("sfm_jac_cam: n < 0"); return SFM_E_SHAPE; }
  if (n == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dR, dC, dX, dJ; DevBuf<int> dst;
  SFM_TRY(dR.upload(R, 9 * (size_t)n, s)); SFM_TRY(dC.upload(C, 3 * (size_t)n, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s));
  SFM_TRY(dJ.alloc(14 * (size_t)n)); SFM_TRY(dst.alloc(n));
  jac_cam_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dR.p, dC.p, dX.p, quirks, dJ.p, dst.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dJ.download(Jp, 14 * (size_t)n, s)); SFM_TRY(dst.download(status, n, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_jac_pt(int n, int n_views, const double* projs, const double* X, double* Jx) {
  SFM_TRY(ensure_init());
  if (n < 0 || n_views < 1) { set_error("sfm_jac_pt: bad sizes n=%d n_views=%d", n, n_views); return SFM_E_SHAPE; }
  if (n == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dP, dX, dJ;
  SFM_TRY(dP.upload(projs, 12 * (size_t)n * n_views, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s));
  SFM_TRY(dJ.alloc(6 * (size_t)n * n_views));
  jac_pt_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, n_views, dP.p, dX.p, dJ.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dJ.download(Jx, 6 * (size_t)n * n_views, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

// ---- DEVICE-pointer, stream-ordered forms: enqueue on the caller's stream and return ---------------------------
static hipStream_t pick_stream(void* hip_stream) { return hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx().stream; }

int sfm_tri_nonlinear_dev(int m, int n_views, const double* d_projs, const double* d_uv, const double* d_X_in, double lambda,
                          int iters, double* d_X_out, void* hip_stream) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1 || iters < 0) {
    set_error("sfm_tri_nonlinear_dev: bad sizes m=%d n_views=%d iters=%d", m, n_views, iters);
    return SFM_E_SHAPE;
  }
  if (m == 0) return SFM_OK;
  if (!d_projs || !d_uv || !d_X_in || !d_X_out) { set_error("sfm_tri_nonlinear_dev: null device pointer"); return SFM_E_SHAPE; }
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  launch_tri_nonlinear(m, n_views, d_projs, d_uv, d_X_in, lambda, iters, d_X_out, lds, pick_stream(hip_stream));
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

int sfm_tri_linear_dev(int m, int n_views, const double* d_projs, const double* d_uv, double* d_X_out, void* hip_stream) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1) { set_error("sfm_tri_linear_dev: bad sizes m=%d n_views=%d", m, n_views); return SFM_E_SHAPE; }
  if (m == 0) return SFM_OK;
  if (!d_projs || !d_uv || !d_X_out) { set_error("sfm_tri_linear_dev: null device pointer"); return SFM_E_SHAPE; }
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  tri_linear_kernel<<<(m + 255) / 256, 256, lds, pick_stream(hip_stream)>>>(m, n_views, d_projs, d_uv, d_X_out);
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

int sfm_triangulate_dev(int m, int n_views, const double* d_projs, const double* d_uv, double lambda, int iters,
                        double* d_X_out, void* hip_stream) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1 || iters < 0) {
    set_error("sfm_triangulate_dev: bad sizes m=%d n_views=%d iters=%d", m, n_views, iters);
    return SFM_E_SHAPE;
  }
  if (m == 0) return SFM_OK;
  if (!d_projs || !d_uv || !d_X_out) { set_error("sfm_triangulate_dev: null device pointer"); return SFM_E_SHAPE; }
  hipStream_t s = pick_stream(hip_stream);
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  // the DLT result lands in X_out and is refined in place: a thread reads and writes only its own point
  tri_linear_kernel<<<(m + 255) / 256, 256, lds, s>>>(m, n_views, d_projs, d_uv, d_X_out);            // tri:85
  launch_tri_nonlinear(m, n_views, d_projs, d_uv, d_X_out, lambda, iters, d_X_out, lds, s);           // tri:86
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

int sfm_pnp_nonlinear_batch_dev(int n_views, const int* d_offsets, int total, const double* d_uv_pix, const double* d_X,
                                const double* d_K, const double* d_R0, const double* d_C0, double lambda, int iters,
                                int quirks, double* d_R_out, double* d_C_out, int* d_status, int max_view_points,
                                void* hip_stream) {
  SFM_TRY(ensure_init());
  if (n_views < 0 || total < 0 || iters < 0) {
    set_error("sfm_pnp_nonlinear_batch_dev: bad sizes n_views=%d total=%d iters=%d", n_views, total, iters);
    return SFM_E_SHAPE;
  }
  if (n_views == 0) return SFM_OK;
  if (!d_offsets || !d_K || !d_R0 || !d_C0 || !d_R_out || !d_C_out || !d_status || (total > 0 && (!d_uv_pix || !d_X))) {
    set_error("sfm_pnp_nonlinear_batch_dev: null device pointer");
    return SFM_E_SHAPE;
  }
  // the offsets live on the device: the caller says how large its largest view is (0: unknown -- both size classes are
  // launched; a view larger than the caller said is still refined, by the small-view kernel re-reading its points)
  hipStream_t s = pick_stream(hip_stream);
  int widest = max_view_points, narrowest = 0;
  if (widest <= 0) {      // not told: the size classes are a function of the views' sizes, so read them (blocking once)
    std::vector<int> off((size_t)n_views + 1);
    SFM_HIP(hipMemcpyAsync(off.data(), d_offsets, sizeof(int) * ((size_t)n_views + 1), hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    widest = 1; narrowest = 0x7fffffff;
    for (int v = 0; v < n_views; ++v) { widest = std::max(widest, off[v + 1] - off[v]); narrowest = std::min(narrowest, off[v + 1] - off[v]); }
  }
  SFM_TRY(enqueue_pnp_nonlinear(n_views, d_offsets, total, d_uv_pix, d_X, d_K, d_R0, d_C0, lambda, iters, quirks, d_R_out,
                                d_C_out, d_status, s, narrowest, widest));
  return SFM_OK;
}

int sfm_gather_points_dev(int n, const int* d_index, const double* d_px, const double* d_py, const double* d_pz,
                          double* d_X_out, void* hip_stream) {
  SFM_TRY(ensure_init());
  if (n < 0) { set_error("sfm_gather_points_dev: n < 0"); return SFM_E_SHAPE; }
  if (n == 0) return SFM_OK;
  if (!d_index || !d_px || !d_py || !d_pz || !d_X_out) { set_error("sfm_gather_points_dev: null device pointer"); return SFM_E_SHAPE; }
  gather_points_kernel<<<(n + 255) / 256, 256, 0, pick_stream(hip_stream)>>>(n, d_index, d_px, d_py, d_pz, d_X_out);
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

int sfm_tri_nonlinear(int m, int n_views, const double* projs, const double* uv, const double* X_in, double lambda,
                      int iters, double* X_out) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1 || iters < 0) {
    set_error("sfm_tri_nonlinear: bad sizes m=%d n_views=%d iters=%d", m, n_views, iters);
    return SFM_E_SHAPE;
  }
  if (m == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dP, dUV, dX, dO;
  SFM_TRY(dP.upload(projs, 12 * (size_t)n_views, s));
  SFM_TRY(dUV.upload(uv, 2 * (size_t)n_views * m, s));
  SFM_TRY(dX.upload(X_in, 4 * (size_t)m, s));
  SFM_TRY(dO.alloc(4 * (size_t)m));
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  launch_tri_nonlinear(m, n_views, dP.p, dUV.p, dX.p, lambda, iters, dO.p, lds, s);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dO.download(X_out, 4 * (size_t)m, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_tri_linear(int m, int n_views, const double* projs, const double* uv, double* X_out) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1) { set_error("sfm_tri_linear: bad sizes m=%d n_views=%d", m, n_views); return SFM_E_SHAPE; }
  if (m == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dP, dUV, dO;
  SFM_TRY(dP.upload(projs, 12 * (size_t)n_views, s));
  SFM_TRY(dUV.upload(uv, 2 * (size_t)n_views * m, s));
  SFM_TRY(dO.alloc(4 * (size_t)m));
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  tri_linear_kernel<<<(m + 255) / 256, 256, lds, s>>>(m, n_views, dP.p, dUV.p, dO.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dO.download(X_out, 4 * (size_t)m, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_triangulate(int m, int n_views, const double* projs, const double* uv, double lambda, int iters, double* X_out) {
  SFM_TRY(ensure_init());
  if (m < 0 || n_views < 1 || iters < 0) {
    set_error("sfm_triangulate: bad sizes m=%d n_views=%d iters=%d", m, n_views, iters);
    return SFM_E_SHAPE;
  }
  if (m == 0) return SFM_OK;
  hipStream_t s = ctx().stream;
  DevBuf<double> dP, dUV, dL, dO;
  SFM_TRY(dP.upload(projs, 12 * (size_t)n_views, s));
  SFM_TRY(dUV.upload(uv, 2 * (size_t)n_views * m, s));
  SFM_TRY(dL.alloc(4 * (size_t)m)); SFM_TRY(dO.alloc(4 * (size_t)m));
  const size_t lds = n_views <= kTriLdsViews ? sizeof(double) * 12 * n_views : 0;
  tri_linear_kernel<<<(m + 255) / 256, 256, lds, s>>>(m, n_views, dP.p, dUV.p, dL.p);          // tri:85
  launch_tri_nonlinear(m, n_views, dP.p, dUV.p, dL.p, lambda, iters, dO.p, lds, s);   // tri:86
  SFM_HIP(hipGetLastError());
  SFM_TRY(dO.download(X_out, 4 * (size_t)m, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_pnp_six_point_hypotheses(int n, const double* uv_pix, const double* X, const double K[9], int n_hyp,
                                 const int* samples, double threshold, double* R_out, double* C_out, int* counts) {
  SFM_TRY(ensure_init());
  if (n < 6 || n_hyp < 1) { set_error("sfm_pnp_six_point_hypotheses: need n >= 6 points and n_hyp >= 1 (n=%d n_hyp=%d)", n, n_hyp); return SFM_E_SHAPE; }
  for (int i = 0; i < 6 * n_hyp; ++i)
    if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_pnp_six_point_hypotheses: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<double> dUV, dX, dK, dR, dC, dP;
  DevBuf<int> dS, dCnt;
  SFM_TRY(dUV.upload(uv_pix, 3 * (size_t)n, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s)); SFM_TRY(dK.upload(K, 9, s));
  SFM_TRY(dS.upload(samples, 6 * (size_t)n_hyp, s));
  SFM_TRY(dR.alloc(9 * (size_t)n_hyp)); SFM_TRY(dC.alloc(3 * (size_t)n_hyp)); SFM_TRY(dP.alloc(12 * (size_t)n_hyp));
  SFM_TRY(dCnt.alloc(n_hyp));
  pnp_six_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, n, dS.p, dUV.p, dX.p, dK.p, dR.p, dC.p, dP.p, nullptr);
  pnp_score_kernel<<<n_hyp, 256, 0, s>>>(n, dP.p, dUV.p, dX.p, threshold, dCnt.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dR.download(R_out, 9 * (size_t)n_hyp, s)); SFM_TRY(dC.download(C_out, 3 * (size_t)n_hyp, s));
  SFM_TRY(dCnt.download(counts, n_hyp, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_pnp_linear_ransac(int n, const double* uv_pix, const double* X, const double K[9], int n_hyp, const int* samples,
                          double threshold, double R_out[9], double C_out[3], int* inlier_mask, int* n_inliers,
                          int* best_hypothesis) {
  SFM_TRY(ensure_init());
  if (n < 6 || n_hyp < 1) { set_error("sfm_pnp_linear_ransac: need n >= 6 points and n_hyp >= 1 (n=%d n_hyp=%d)", n, n_hyp); return SFM_E_SHAPE; }
  for (int i = 0; i < 6 * n_hyp; ++i)
    if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_pnp_linear_ransac: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<double> dUV, dX, dK, dR, dC, dP;
  DevBuf<int> dS, dCnt, dMask;
  SFM_TRY(dUV.upload(uv_pix, 3 * (size_t)n, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s)); SFM_TRY(dK.upload(K, 9, s));
  SFM_TRY(dS.upload(samples, 6 * (size_t)n_hyp, s));
  SFM_TRY(dR.alloc(9 * (size_t)n_hyp)); SFM_TRY(dC.alloc(3 * (size_t)n_hyp)); SFM_TRY(dP.alloc(12 * (size_t)n_hyp));
  SFM_TRY(dCnt.alloc(n_hyp)); SFM_TRY(dMask.alloc(n));
  pnp_six_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, n, dS.p, dUV.p, dX.p, dK.p, dR.p, dC.p, dP.p, nullptr);
  pnp_score_kernel<<<n_hyp, 256, 0, s>>>(n, dP.p, dUV.p, dX.p, threshold, dCnt.p);
  SFM_HIP(hipGetLastError());
  std::vector<int> counts(n_hyp);
  SFM_TRY(dCnt.download(counts.data(), n_hyp, s));
  SFM_TRY(stream_sync(s));
  // the reference keeps the FIRST hypothesis with a strictly larger count, starting from 0 inliers / identity
  // pose (campose:524-560)
  int best = -1, best_cnt = 0;
  for (int h = 0; h < n_hyp; ++h)
    if (counts[h] > best_cnt) { best_cnt = counts[h]; best = h; }
  if (best_hypothesis) *best_hypothesis = best;
  if (n_inliers) *n_inliers = best_cnt;
  if (best < 0) {
    const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) R_out[i] = I[i];
    for (int i = 0; i < 3; ++i) C_out[i] = 0.0;
    for (int i = 0; i < n; ++i) inlier_mask[i] = 0;
    return SFM_OK;
  }
  pnp_inlier_mask_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dP.p + 12 * (size_t)best, dUV.p, dX.p, threshold, dMask.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dMask.download(inlier_mask, n, s));
  SFM_HIP(hipMemcpyAsync(R_out, dR.p + 9 * (size_t)best, 9 * sizeof(double), hipMemcpyDeviceToHost, s));
  SFM_HIP(hipMemcpyAsync(C_out, dC.p + 3 * (size_t)best, 3 * sizeof(double), hipMemcpyDeviceToHost, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_pnp_ransac_evaluate(int n, const double* uv_pix, const double* X, const double K[9], int n_hyp, const int* samples,
                            double threshold, double* R_out, double* C_out, int* counts, int* counts_neg) {
  SFM_TRY(ensure_init());
  if (n < 6 || n_hyp < 1) { set_error("sfm_pnp_ransac_evaluate: need n >= 6 points and n_hyp >= 1 (n=%d n_hyp=%d)", n, n_hyp); return SFM_E_SHAPE; }
  for (int i = 0; i < 6 * n_hyp; ++i)
    if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_pnp_ransac_evaluate: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<double> dUV, dX, dK, dR, dC, dP;
  DevBuf<int> dS, dCnt;
  SFM_TRY(dUV.upload(uv_pix, 3 * (size_t)n, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s)); SFM_TRY(dK.upload(K, 9, s));
  SFM_TRY(dS.upload(samples, 6 * (size_t)n_hyp, s));
  SFM_TRY(dR.alloc(9 * (size_t)n_hyp)); SFM_TRY(dC.alloc(3 * (size_t)n_hyp)); SFM_TRY(dP.alloc(24 * (size_t)n_hyp));
  SFM_TRY(dCnt.alloc(2 * (size_t)n_hyp));
  // projections of the pose (R, C) in the first n_hyp blocks, of (R, -C) behind them; one scoring launch over both
  pnp_six_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, n, dS.p, dUV.p, dX.p, dK.p, dR.p, dC.p, dP.p, dP.p + 12 * (size_t)n_hyp);
  pnp_score_kernel<<<2 * n_hyp, 256, 0, s>>>(n, dP.p, dUV.p, dX.p, threshold, dCnt.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dR.download(R_out, 9 * (size_t)n_hyp, s)); SFM_TRY(dC.download(C_out, 3 * (size_t)n_hyp, s));
  SFM_HIP(hipMemcpyAsync(counts, dCnt.p, sizeof(int) * n_hyp, hipMemcpyDeviceToHost, s));
  SFM_HIP(hipMemcpyAsync(counts_neg, dCnt.p + n_hyp, sizeof(int) * n_hyp, hipMemcpyDeviceToHost, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_pnp_inlier_mask(int n, const double* uv_pix, const double* X, const double K[9], const double R[9], const double C[3],
                        double threshold, int* inlier_mask, int* n_inliers) {
  SFM_TRY(ensure_init());
  if (n < 1) { set_error("sfm_pnp_inlier_mask: n < 1"); return SFM_E_SHAPE; }
  // proj = K @ [R^T | R^T @ -C]  (campose:538), 12 doubles: formed on the host side of the library, scored on the device
  double rt[12], P[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) rt[4 * i + j] = R[3 * j + i];
    rt[4 * i + 3] = R[0 + i] * -C[0] + R[3 + i] * -C[1] + R[6 + i] * -C[2];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) P[4 * i + j] = K[3 * i] * rt[j] + K[3 * i + 1] * rt[4 + j] + K[3 * i + 2] * rt[8 + j];
  hipStream_t s = ctx().stream;
  DevBuf<double> dUV, dX, dP;
  DevBuf<int> dMask;
  SFM_TRY(dUV.upload(uv_pix, 3 * (size_t)n, s)); SFM_TRY(dX.upload(X, 4 * (size_t)n, s)); SFM_TRY(dP.upload(P, 12, s));
  SFM_TRY(dMask.alloc(n));
  pnp_inlier_mask_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dP.p, dUV.p, dX.p, threshold, dMask.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dMask.download(inlier_mask, n, s));
  SFM_TRY(stream_sync(s));
  int cnt = 0;
  for (int i = 0; i < n; ++i) cnt += inlier_mask[i] != 0;
  if (n_inliers) *n_inliers = cnt;
  return SFM_OK;
}

// ---- estimate_cam_pose_pnp as two calls around the host's decision (quirk Q13) with the view RESIDENT in between ----------------
// campose_processor.py:192-246 = RANSAC (which hypothesis wins is the host's call: q13.py) -> the winner's inliers -> 300
// nonlinear iterations on them.  sfm_pnp_ransac_begin = sfm_pnp_ransac_evaluate that keeps the keys and points on the
// device; sfm_pnp_ransac_finish scores the chosen pose, compacts the inlier columns on the device (ascending, as the
// reference's list indexes them) and refines on them: the keys and points cross PCIe once instead of three times and the
// host never gathers columns.
struct sfm_pnp_session {
  unsigned magic;
  int n;
  double *dUV, *dX, *dK;
  double K[9];       // host copy (the projection of the chosen pose is formed on the host side of the library)
};
namespace sfm { constexpr unsigned kPnpSessionMagic = 0x5F3B5E55u; }

int sfm_pnp_session_destroy(sfm_pnp_session* ses) {
  if (ses == nullptr) return SFM_OK;
  if (ses->magic != kPnpSessionMagic) { set_error("sfm_pnp_session_destroy: invalid handle"); return SFM_E_HANDLE; }
  if (ctx().inited) (void)hipStreamSynchronize(ctx().stream);
  pool_free(ses->dUV); pool_free(ses->dX); pool_free(ses->dK);
  ses->magic = 0;
  delete ses;
  return SFM_OK;
}

int sfm_pnp_ransac_begin(int n, const double* uv_pix, const double* X, const double K[9], int n_hyp, const int* samples,
                         double threshold, double* R_out, double* C_out, int* counts, int* counts_neg, sfm_pnp_session** out) {
  SFM_TRY(ensure_init());
  if (out == nullptr) { set_error("sfm_pnp_ransac_begin: out is null"); return SFM_E_SHAPE; }
  *out = nullptr;
  if (n < 6 || n_hyp < 1) { set_error("sfm_pnp_ransac_begin: need n >= 6 points and n_hyp >= 1 (n=%d n_hyp=%d)", n, n_hyp); return SFM_E_SHAPE; }
  for (int i = 0; i < 6 * n_hyp; ++i)
    if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_pnp_ransac_begin: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  sfm_pnp_session* ses = new sfm_pnp_session{kPnpSessionMagic, n, nullptr, nullptr, nullptr, {0, 0, 0, 0, 0, 0, 0, 0, 0}};
  for (int i = 0; i < 9; ++i) ses->K[i] = K[i];
  auto fail = [&](int st) { (void)sfm_pnp_session_destroy(ses); return st; };
  if (pool_alloc(reinterpret_cast<void**>(&ses->dUV), sizeof(double) * 3 * (size_t)n) != hipSuccess ||
      pool_alloc(reinterpret_cast<void**>(&ses->dX), sizeof(double) * 4 * (size_t)n) != hipSuccess ||
      pool_alloc(reinterpret_cast<void**>(&ses->dK), sizeof(double) * 9) != hipSuccess) {
    set_error("sfm_pnp_ransac_begin: out of device memory");
    return fail(SFM_E_HIP);
  }
  auto run = [&]() -> int {
    SFM_HIP(hipMemcpyAsync(ses->dUV, uv_pix, sizeof(double) * 3 * (size_t)n, hipMemcpyHostToDevice, s));
    SFM_HIP(hipMemcpyAsync(ses->dX, X, sizeof(double) * 4 * (size_t)n, hipMemcpyHostToDevice, s));
    SFM_HIP(hipMemcpyAsync(ses->dK, K, sizeof(double) * 9, hipMemcpyHostToDevice, s));
    DevBuf<double> dR, dC, dP;
    DevBuf<int> dS, dCnt;
    SFM_TRY(dS.upload(samples, 6 * (size_t)n_hyp, s));
    SFM_TRY(dR.alloc(9 * (size_t)n_hyp)); SFM_TRY(dC.alloc(3 * (size_t)n_hyp)); SFM_TRY(dP.alloc(24 * (size_t)n_hyp));
    SFM_TRY(dCnt.alloc(2 * (size_t)n_hyp));
    pnp_six_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, n, dS.p, ses->dUV, ses->dX, ses->dK, dR.p, dC.p, dP.p, dP.p + 12 * (size_t)n_hyp);
    pnp_score_kernel<<<2 * n_hyp, 256, 0, s>>>(n, dP.p, ses->dUV, ses->dX, threshold, dCnt.p);
    SFM_HIP(hipGetLastError());
    SFM_TRY(dR.download(R_out, 9 * (size_t)n_hyp, s)); SFM_TRY(dC.download(C_out, 3 * (size_t)n_hyp, s));
    SFM_HIP(hipMemcpyAsync(counts, dCnt.p, sizeof(int) * n_hyp, hipMemcpyDeviceToHost, s));
    SFM_HIP(hipMemcpyAsync(counts_neg, dCnt.p + n_hyp, sizeof(int) * n_hyp, hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    return SFM_OK;
  };
  const int st = run();
  if (st != SFM_OK) return fail(st);
  *out = ses;
  return SFM_OK;
}

int sfm_pnp_ransac_finish(sfm_pnp_session* ses, const double R[9], const double C[3], double threshold, double lambda, int iters,
                          int quirks, int* inlier_mask, int* n_inliers, double R_out[9], double C_out[3]) {
  SFM_TRY(ensure_init());
  if (ses == nullptr || ses->magic != kPnpSessionMagic) { set_error("sfm_pnp_ransac_finish: invalid session handle"); return SFM_E_HANDLE; }
  if (iters < 0) { set_error("sfm_pnp_ransac_finish: iters < 0"); return SFM_E_SHAPE; }
  const int n = ses->n;
  // proj = K @ [R^T | R^T @ -C]  (campose:538) of the chosen pose
  const double* Kh = ses->K;
  hipStream_t s = ctx().stream;
  double rt[12], P[12];
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) rt[4 * i + j] = R[3 * j + i];
    rt[4 * i + 3] = R[0 + i] * -C[0] + R[3 + i] * -C[1] + R[6 + i] * -C[2];
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 4; ++j) P[4 * i + j] = Kh[3 * i] * rt[j] + Kh[3 * i + 1] * rt[4 + j] + Kh[3 * i + 2] * rt[8 + j];
  auto run = [&]() -> int {
    DevBuf<double> dP, dUVc, dXc, dR0, dC0, dR, dC;
    DevBuf<int> dMask, dPos, dOff, dSt;
    SFM_TRY(dP.upload(P, 12, s)); SFM_TRY(dR0.upload(R, 9, s)); SFM_TRY(dC0.upload(C, 3, s));
    SFM_TRY(dMask.alloc(n)); SFM_TRY(dPos.alloc(n)); SFM_TRY(dOff.alloc(2)); SFM_TRY(dSt.alloc(1));
    SFM_TRY(dUVc.alloc(3 * (size_t)n)); SFM_TRY(dXc.alloc(4 * (size_t)n)); SFM_TRY(dR.alloc(9)); SFM_TRY(dC.alloc(3));
    pnp_inlier_mask_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dP.p, ses->dUV, ses->dX, threshold, dMask.p);
    pnp_mask_scan_kernel<<<1, 1024, 0, s>>>(n, dMask.p, dPos.p, dOff.p);
    pnp_compact_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dMask.p, dPos.p, ses->dUV, ses->dX, dUVc.p, dXc.p);
    SFM_HIP(hipGetLastError());
    int off[2] = {0, 0};
    SFM_TRY(dMask.download(inlier_mask, n, s));
    SFM_HIP(hipMemcpyAsync(off, dOff.p, sizeof(off), hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    const int m = off[1];
    if (n_inliers) *n_inliers = m;
    // the refinement of the reference's call chain on exactly those columns (the size class follows m, as in sfm_pnp_nonlinear)
    SFM_TRY(enqueue_pnp_nonlinear(1, dOff.p, n, dUVc.p, dXc.p, ses->dK, dR0.p, dC0.p, lambda, iters, quirks, dR.p, dC.p, dSt.p, s, m, std::max(m, 1)));
    int st = SFM_OK;
    SFM_TRY(dR.download(R_out, 9, s)); SFM_TRY(dC.download(C_out, 3, s));
    SFM_HIP(hipMemcpyAsync(&st, dSt.p, sizeof(int), hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    if (st != SFM_OK) set_error("sfm_pnp_ransac_finish: rotation check failed on device (status %d)", st);
    return st;
  };
  const int st = run();
  const int sd = sfm_pnp_session_destroy(ses);
  return st != SFM_OK ? st : sd;
}

int sfm_pnp_nonlinear_batch(int n_views, const int* offsets, int total, const double* uv_pix, const double* X,
                            const double* K, const double* R0, const double* C0, double lambda, int iters, int quirks,
                            double* R_out, double* C_out, int* status) {
  SFM_TRY(ensure_init());
  if (n_views < 0 || total < 0 || iters < 0) {
    set_error("sfm_pnp_nonlinear_batch: bad sizes n_views=%d total=%d iters=%d", n_views, total, iters);
    return SFM_E_SHAPE;
  }
  if (n_views == 0) return SFM_OK;
  if (offsets[0] != 0 || offsets[n_views] != total) { set_error("sfm_pnp_nonlinear_batch: offsets do not span [0,total]"); return SFM_E_SHAPE; }
  for (int v = 0; v < n_views; ++v)
    if (offsets[v + 1] < offsets[v]) { set_error("sfm_pnp_nonlinear_batch: offsets not monotone"); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<int> dOff, dSt;
  DevBuf<double> dUV, dX, dK, dR0, dC0, dR, dC;
  SFM_TRY(dOff.upload(offsets, (size_t)n_views + 1, s));
  SFM_TRY(dUV.upload(uv_pix, 3 * (size_t)total, s)); SFM_TRY(dX.upload(X, 4 * (size_t)total, s));
  SFM_TRY(dK.upload(K, 9 * (size_t)n_views, s)); SFM_TRY(dR0.upload(R0, 9 * (size_t)n_views, s));
  SFM_TRY(dC0.upload(C0, 3 * (size_t)n_views, s));
  SFM_TRY(dR.alloc(9 * (size_t)n_views)); SFM_TRY(dC.alloc(3 * (size_t)n_views)); SFM_TRY(dSt.alloc(n_views));
  int widest = 0, narrowest = 0x7fffffff;
  for (int v = 0; v < n_views; ++v) {
    widest = std::max(widest, offsets[v + 1] - offsets[v]);
    narrowest = std::min(narrowest, offsets[v + 1] - offsets[v]);
  }
  SFM_TRY(enqueue_pnp_nonlinear(n_views, dOff.p, total, dUV.p, dX.p, dK.p, dR0.p, dC0.p, lambda, iters, quirks, dR.p, dC.p,
                                dSt.p, s, narrowest, std::max(widest, 1)));
  SFM_TRY(dR.download(R_out, 9 * (size_t)n_views, s)); SFM_TRY(dC.download(C_out, 3 * (size_t)n_views, s));
  SFM_TRY(dSt.download(status, n_views, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_pnp_nonlinear(int n, const double* uv_pix, const double* X, const double K[9], const double R0[9],
                      const double C0[3], double lambda, int iters, int quirks, double R_out[9], double C_out[3]) {
  if (n < 0) { set_error("sfm_pnp_nonlinear: n < 0"); return SFM_E_SHAPE; }
  int offsets[2] = {0, n};
  int st = SFM_OK;
  SFM_TRY(sfm_pnp_nonlinear_batch(1, offsets, n, uv_pix, X, K, R0, C0, lambda, iters, quirks, R_out, C_out, &st));
  if (st != SFM_OK) set_error("sfm_pnp_nonlinear: rotation check failed on device (status %d)", st);
  return st;
}

}  // extern "C"
