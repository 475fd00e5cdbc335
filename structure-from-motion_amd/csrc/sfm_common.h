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
  hipStream_t stream = nullptr;       // default stream of the library: `own` or the one installed by sfm_set_stream
  hipStream_t own = nullptr;          // created at sfm_init, lives until sfm_shutdown
  int num_cus = 256;
};

Context& ctx();
int ensure_init();

// Device-memory pool: hipMalloc/hipFree cost tens of microseconds each and a host-buffer call makes ~25 of
// them, which dominates small problems (incremental SfM sizes).  Blocks are rounded up to a power of two
// (>= 256 B) and kept on per-size free lists (at most 2 GiB cached); sfm_shutdown releases them.
hipError_t pool_alloc(void** ptr, size_t bytes);
void pool_free(void* ptr);
void pool_release_all();
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

// Temporary device buffers go back to the caching pool when their DevBuf dies.  An entry point that returns
// early with an error may still have copies or kernels in flight on its stream that use them, and the next
// pool_alloc of that size class would hand the block out again: every DevBuf allocation therefore marks the
// calling thread "unsynchronised" on its stream, stream_sync() clears the mark (the normal end of an entry point),
// and the first DevBuf that dies while the mark is set synchronises that stream before releasing its block.
struct PendingWork { bool dirty = false; hipStream_t stream = nullptr; };
PendingWork& pending_work();
int stream_sync(hipStream_t s);

// RAII device buffer for the host-pointer convenience entry points.
template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  ~DevBuf() {
    if (!p) return;
    PendingWork& w = pending_work();
    if (w.dirty) { (void)hipStreamSynchronize(w.stream); w.dirty = false; }
    pool_free(p);
  }
  int alloc(size_t count, hipStream_t s = nullptr) {
    n = count;
    if (count == 0) count = 1;
    hipError_t e = pool_alloc(reinterpret_cast<void**>(&p), count * sizeof(T));
    if (e != hipSuccess) return hip_fail(e, "hipMalloc", __LINE__);
    PendingWork& w = pending_work();
    w.dirty = true;
    w.stream = s ? s : ctx().stream;
    return SFM_OK;
  }
  int upload(const T* host, size_t count, hipStream_t s) {
    SFM_TRY(alloc(count, s));
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

// Cross-lane moves on the DPP path (no LDS crossbar): quad_perm xor-1 / xor-2, row_half_mirror, row_mirror.
// bound_ctrl = 1: every source lane of these patterns is valid, so the flag changes no value -- but it tells the
// compiler that the destination's old contents are dead, which saves a v_mov_b32 0 per half (8 of the 20
// instructions of a 16-lane sum).
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}

// Sum over aligned groups of G lanes (G = 4, 8, 16, 32, 64); every lane of the group gets the sum.
// Up to 16 lanes stay on DPP: xor-1, xor-2 inside quads, then i <-> 7-i inside half rows, then
// i <-> 15-i inside rows (mirrors pair each partial sum with a disjoint one, so four steps cover a row);
// only the 32- and 64-lane steps go through ds_bpermute.
template <int G>
__device__ __forceinline__ double group_sum(double v) {
  v += dpp_f64<0xB1>(v);                       // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);                       // quad_perm [2,3,0,1]
  if (G >= 8) v += dpp_f64<0x141>(v);          // row_half_mirror
  if (G >= 16) v += dpp_f64<0x140>(v);         // row_mirror
  if (G >= 32) v += __shfl_xor(v, 16, 64);
  if (G >= 64) v += __shfl_xor(v, 32, 64);
  return v;
}

// Row totals of MANY values at once (the 35 normal-equation sums of the PnP kernel): a halving fold instead of N
// independent group sums.  One step pairs every lane with a partner in the other half of its group; the lane keeps the
// lower half of the value list if its bit `hi` is clear, the upper half otherwise, and receives the partner's partial sums
// of exactly those values -- so the list halves while the number of lanes summed doubles.  Four steps (row_mirror split by
// lane bit 3, row_half_mirror by bit 2, quad xor-2 by bit 1, quad xor-1 by bit 0: each partner agrees with the lane in
// the bits already used) leave ceil(N / 16) row totals per lane: 35 -> 18 -> 9 -> 5 -> 3 values, 233 instructions where 35
// four-step sums take 420 (and the compiler keeps each of those a dependent chain).  `fold_index` applies the same
// selection to the value numbers, once per kernel, so that a lane knows which totals it ends up holding (an odd list
// leaves its middle value in both lanes of a pair: both then hold, and may store, the same total).
template <int CTRL, int N>
__device__ __forceinline__ void fold_half(const double (&v)[N], double (&w)[(N + 1) / 2], bool hi) {
  constexpr int H = (N + 1) / 2;
#pragma unroll
  for (int k = 0; k < N - H; ++k) {
    const double keep = hi ? v[k + H] : v[k], send = hi ? v[k] : v[k + H];
    w[k] = keep + dpp_f64<CTRL>(send);
  }
  if (N & 1) w[H - 1] = v[H - 1] + dpp_f64<CTRL>(v[H - 1]);
}
template <int N>
__device__ __forceinline__ void fold_index(const int (&v)[N], int (&w)[(N + 1) / 2], bool hi) {
  constexpr int H = (N + 1) / 2;
#pragma unroll
  for (int k = 0; k < N - H; ++k) w[k] = hi ? v[k + H] : v[k];
  if (N & 1) w[H - 1] = v[H - 1];
}

// 1/sqrt(d) to full double precision: v_rsq_f64 seed (5e-8) + ONE third-order step r (1 + e/2 + 3 e^2 / 8),
// e = 1 - d r^2: five operations where two Newton steps take seven, and closer -- 1.4e-16 against 2.4e-16 maximal
// relative error (tools/microbench_solve.hip).  (sqrt / division expand to ~50 dependent instructions each.)
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double r = __builtin_amdgcn_rsq(d);
  const double e = __builtin_fma(-(d * r), r, 1.0);
  return __builtin_fma(r, e * __builtin_fma(e, 0.375, 0.5), r);
}

// chol3_inv without sqrt or division: a = (a00,a10,a11,a20,a21,a22) SPD -> Li = L^-1 (lower, packed).
__device__ __forceinline__ void chol3_inv_fast(const double* a, double* li) {
  const double i00 = rsqrt_nr(a[0]);
  const double l10 = a[1] * i00;
  const double i11 = rsqrt_nr(a[2] - l10 * l10);
  const double l20 = a[3] * i00;
  const double l21 = (a[4] - l20 * l10) * i11;
  const double i22 = rsqrt_nr(a[5] - l20 * l20 - l21 * l21);
  li[0] = i00;
  li[1] = -l10 * i00 * i11;
  li[2] = i11;
  li[3] = (-l20 * i00 - l21 * li[1]) * i22;
  li[4] = -l21 * i11 * i22;
  li[5] = i22;
}

__device__ __forceinline__ double wave_sum(double v) { return group_sum<64>(v); }

// One-sided Jacobi on the columns of B (N x N, row-major B[row][col]): on return the columns of B are
// mutually orthogonal (B_out = B_in V, column c = sigma_c u_c) and V holds the right singular vectors as
// columns.  Column order is whatever the sweeps leave; callers pick columns by norm.
template <int N>
__device__ void jacobi_right_vectors(double (&B)[N][N], double (&V)[N][N], int max_sweeps) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    bool rotated = false;
    for (int p = 0; p < N - 1; ++p) {
      for (int q = p + 1; q < N; ++q) {
        double al = 0, be = 0, ga = 0;
        for (int k = 0; k < N; ++k) { al += B[k][p] * B[k][p]; be += B[k][q] * B[k][q]; ga += B[k][p] * B[k][q]; }
        if (ga != 0.0 && fabs(ga) > 1e-17 * sqrt(al * be)) {
          rotated = true;
          const double ze = (be - al) / (2.0 * ga);
          const double t = (ze == 0.0) ? 1.0 : copysign(1.0, ze) / (fabs(ze) + sqrt(1.0 + ze * ze));
          const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
          for (int k = 0; k < N; ++k) {
            const double bp = B[k][p], bq = B[k][q];
            B[k][p] = c * bp - sn * bq; B[k][q] = sn * bp + c * bq;
            const double vp = V[k][p], vq = V[k][q];
            V[k][p] = c * vp - sn * vq; V[k][q] = sn * vp + c * vq;
          }
        }
      }
    }
    if (!rotated) break;
  }
}

// Wave-cooperative version for the RANSAC solvers (one wave per hypothesis): lanes 0..N-1 hold the rows of B
// (lanes N..15 must hold zero rows), lanes 16..16+N-1 the rows of V (the identity on entry), N <= 16; other
// lanes carry zero rows along.  Same cyclic sweep as jacobi_right_vectors; the three column inner products
// of a pair are one DPP row reduction each, the rotation is computed redundantly on every lane from the
// broadcast sums (rsqrt + Newton instead of sqrt / division: Jacobi rotations only have to be orthogonal,
// which c = rsqrt(1 + t^2), s = c t guarantee to rounding) and applied to the lane's own row.  All column
// indices are compile-time constants, so the rows stay in registers (the thread-per-hypothesis version keeps
// its N x N arrays in scratch memory and took 5.7 ms for 300 12x12 problems; this one ~0.2 ms).
__device__ __forceinline__ double wave_lane0(double v) {
  const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
  const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
  return __hiloint2double(hi, lo);
}

// 1/d: v_rcp_f64 seed (5e-8) + one third-order step r (1 + e + e^2), e = 1 - d r (1.1e-16; two Newton steps: 1.9e-16)
__device__ __forceinline__ double rcp_nr(double d) {
  const double r = __builtin_amdgcn_rcp(d);
  const double e = __builtin_fma(-d, r, 1.0);
  return __builtin_fma(r, __builtin_fma(e, e, e), r);
}

// cam_prepare (sfm_math.h) for serial sections of a kernel: the same predicate and the same formulas with the divisions and
// the square root replaced by v_rcp_f64 / v_rsq_f64 + one third-order step (1e-16 relative; ~10 instructions each instead
// of ~40).  verify_rotation's one-sided 1e-8 thresholds are ten orders of magnitude above that difference.
//
// VERIFY = false leaves out verify_rotation's determinant and inverse (≈70 of the ≈130 instructions) and is only for a
// quaternion that the caller has JUST normalised with rsqrt_nr: |q|^2 = 1 + e with |e| of a few 1e-16, and R(q) = Q + e (Q - I)
// for an orthonormal Q, so det R - 1 and inv(R) - R^T are of the order of e, eight orders of magnitude below the 1e-8
// the predicate tests -- it cannot fire.  A NaN / Inf quaternion does not reach it either way: every comparison of the
// predicate is written so that NaN passes (as the reference's `>` comparisons do), and such a camera ends in
// SFM_E_QW_ZERO below in both forms.
template <bool VERIFY = true>
__device__ __forceinline__ int cam_prepare_dev(const double* cam7, CamPrep* out) {
  out->C[0] = cam7[0]; out->C[1] = cam7[1]; out->C[2] = cam7[2];
  quat_to_rot(cam7 + 3, out->R);
  const double* R = out->R;
  for (int j = 0; j < 3; ++j) out->t[j] = R[0 + j] * -cam7[0] + R[3 + j] * -cam7[1] + R[6 + j] * -cam7[2];
  if (VERIFY) {
    const double d = det3(R);
    bool ok = !(d - 1 >= kRotTol);
    const double id = rcp_nr(d);
    const double inv[9] = {(R[4] * R[8] - R[5] * R[7]) * id, (R[2] * R[7] - R[1] * R[8]) * id, (R[1] * R[5] - R[2] * R[4]) * id,
                           (R[5] * R[6] - R[3] * R[8]) * id, (R[0] * R[8] - R[2] * R[6]) * id, (R[2] * R[3] - R[0] * R[5]) * id,
                           (R[3] * R[7] - R[4] * R[6]) * id, (R[1] * R[6] - R[0] * R[7]) * id, (R[0] * R[4] - R[1] * R[3]) * id};
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) ok = ok && !(inv[3 * i + j] - R[3 * j + i] > kRotTol);
    if (!ok) return SFM_E_BAD_ROTATION;
  }
  const double tr1 = 1 + R[0] + R[4] + R[8];
  if (tr1 < 0) return SFM_E_SQRT_DOMAIN;
  const double qw = tr1 > 0 ? 0.5 * tr1 * rsqrt_nr(tr1) : 0.0;
  if (fabs(qw) < kQwMin) return SFM_E_QW_ZERO;
  const double i4 = rcp_nr(4 * qw);
  out->q[0] = qw;
  out->q[1] = (R[7] - R[5]) * i4;
  out->q[2] = (R[2] - R[6]) * i4;
  out->q[3] = (R[3] - R[1]) * i4;
  return SFM_OK;
}

// Round r of the round-robin ("circle") ordering of M players (M even): pair i of the round, as (lower, higher) index.
// Player M-1 stays, the others rotate; over rounds 0..M-2 every pair meets exactly once and the M/2 pairs of a round are
// disjoint.  For odd N the extra player M-1 = N is a bye.
constexpr int rr_first(int M, int r, int i) {
  const int a = i == 0 ? M - 1 : (r + i) % (M - 1);
  const int b = i == 0 ? r : (r - i + (M - 1)) % (M - 1);
  return a < b ? a : b;
}
constexpr int rr_second(int M, int r, int i) {
  const int a = i == 0 ? M - 1 : (r + i) % (M - 1);
  const int b = i == 0 ? r : (r - i + (M - 1)) % (M - 1);
  return a < b ? b : a;
}

// One-sided Jacobi on the columns of [B; V] held as rows-in-lanes (lanes 0..15: rows of B, lanes 16..31: rows of V), register
// c of a lane = column c.  The rotations of a sweep are taken round by round in the round-robin order: the N/2 pairs of a
// round touch disjoint columns, so their dot products (a 16-lane DPP sum each), their angles (three reciprocal / square-root
// refinements each) and their updates are independent instruction streams that the scheduler interleaves -- the cyclic
// (p, q) order made every rotation wait for the previous one (~900 cycles each: 200 us for the 300 12 x 12 problems of a
// PnP RANSAC, 66 rotations x ~7 sweeps).  Column norms are formed once per sweep and carried through the rotations
// (|b_p|^2 -= t g, |b_q|^2 += t g), which leaves one reduction per rotation instead of three.  A pair below the
// threshold gets the identity rotation (c = 1, s = 0: exact), so a round has no branches.
template <int N>
__device__ __forceinline__ void jacobi_rows_wave(double (&row)[N], int lane, int max_sweeps) {
  constexpr int M = (N + 1) & ~1, H = M / 2;
  const bool is_b = lane < 16;
#define SFM_EACH(i) _Pragma("unroll") for (int i = 0; i < H; ++i)
  for (int sweep = 0; sweep < max_sweeps; ++sweep) {
    double nrm[M];
    {
      double x[M];
#pragma unroll
      for (int c = 0; c < M; ++c) x[c] = (is_b && c < N) ? row[c < N ? c : 0] * row[c < N ? c : 0] : 0.0;
#pragma unroll
      for (int c = 0; c < M; ++c) x[c] += dpp_f64<0xB1>(x[c]);
#pragma unroll
      for (int c = 0; c < M; ++c) x[c] += dpp_f64<0x4E>(x[c]);
#pragma unroll
      for (int c = 0; c < M; ++c) x[c] += dpp_f64<0x141>(x[c]);
#pragma unroll
      for (int c = 0; c < M; ++c) x[c] += dpp_f64<0x140>(x[c]);
#pragma unroll
      for (int c = 0; c < M; ++c) nrm[c] = wave_lane0(x[c]);
    }
    bool rotated = false;
#pragma unroll
    for (int r = 0; r < M - 1; ++r) {
      // the H pairs of the round in lockstep, stage by stage (the source order is what the scheduler keeps: written pair
      // after pair, each pair's chain stayed a dependent sequence).  A bye (odd N: the pair with player N) works on a
      // zero column and leaves the identity rotation.
      constexpr int kM = M;
      double bp[H], bq[H], ga[H], t[H], cs[H], sn[H];
      SFM_EACH(i) {
        const int p = rr_first(kM, r, i), q = rr_second(kM, r, i);
        bp[i] = row[p];
        bq[i] = q < N ? row[q < N ? q : 0] : 0.0;
        ga[i] = is_b ? bp[i] * bq[i] : 0.0;
      }
      SFM_EACH(i) ga[i] += dpp_f64<0xB1>(ga[i]);
      SFM_EACH(i) ga[i] += dpp_f64<0x4E>(ga[i]);
      SFM_EACH(i) ga[i] += dpp_f64<0x141>(ga[i]);
      SFM_EACH(i) ga[i] += dpp_f64<0x140>(ga[i]);
      SFM_EACH(i) ga[i] = wave_lane0(ga[i]);
      double d[H], rc[H], e[H], ze[H], az[H], w[H];
      bool turn[H];
      SFM_EACH(i) {
        const int p = rr_first(kM, r, i), q = rr_second(kM, r, i);
        // |cos(angle)| > 3e-16: a few ulp above what the Newton-refined rotations can reach
        turn[i] = ga[i] != 0.0 && ga[i] * ga[i] > 1e-31 * (nrm[p] * nrm[q]);       // wave-uniform
        rotated = rotated || turn[i];
        d[i] = 2.0 * (turn[i] ? ga[i] : 1.0);
      }
      // ze = (|b_q|^2 - |b_p|^2) / (2 g): rcp_nr, stage by stage
      SFM_EACH(i) rc[i] = __builtin_amdgcn_rcp(d[i]);
      SFM_EACH(i) e[i] = __builtin_fma(-d[i], rc[i], 1.0);
      SFM_EACH(i) rc[i] = __builtin_fma(rc[i], __builtin_fma(e[i], e[i], e[i]), rc[i]);
      SFM_EACH(i) {
        const int p = rr_first(kM, r, i), q = rr_second(kM, r, i);
        ze[i] = (nrm[q] - nrm[p]) * rc[i];
        // t = sign(ze) / (|ze| + sqrt(1 + ze^2)) without a conditional: every quantity here is wave-uniform and the
        // compiler turns a ?: with an expensive arm into a scalar branch, which would fence the pairs of a round off
        // from each other.  |ze| is clamped (beyond 1e100 the rotation is the identity to 200 digits either way);
        // ze = +-0 (equal norms) gives t = +-1, a 45-degree rotation in either direction.
        az[i] = fmin(fabs(ze[i]), 1e100);
        w[i] = __builtin_fma(az[i], az[i], 1.0);
      }
      // hyp = w rsqrt(w)
      SFM_EACH(i) rc[i] = __builtin_amdgcn_rsq(w[i]);
      SFM_EACH(i) e[i] = __builtin_fma(-(w[i] * rc[i]), rc[i], 1.0);
      SFM_EACH(i) rc[i] = __builtin_fma(rc[i], e[i] * __builtin_fma(e[i], 0.375, 0.5), rc[i]);
      SFM_EACH(i) d[i] = __builtin_fma(w[i], rc[i], az[i]);                   // |ze| + sqrt(1 + ze^2)
      SFM_EACH(i) rc[i] = __builtin_amdgcn_rcp(d[i]);
      SFM_EACH(i) e[i] = __builtin_fma(-d[i], rc[i], 1.0);
      SFM_EACH(i) rc[i] = __builtin_fma(rc[i], __builtin_fma(e[i], e[i], e[i]), rc[i]);
      SFM_EACH(i) t[i] = copysign(rc[i], ze[i]) * (turn[i] ? 1.0 : 0.0);      // (a product: a ?: here lets the compiler branch around the pair)
      // c = 1 / sqrt(1 + t^2), s = c t  (t = 0: c = 1, s = 0 exactly)
      SFM_EACH(i) w[i] = __builtin_fma(t[i], t[i], 1.0);
      SFM_EACH(i) rc[i] = __builtin_amdgcn_rsq(w[i]);
      SFM_EACH(i) e[i] = __builtin_fma(-(w[i] * rc[i]), rc[i], 1.0);
      SFM_EACH(i) cs[i] = __builtin_fma(rc[i], e[i] * __builtin_fma(e[i], 0.375, 0.5), rc[i]);
      SFM_EACH(i) sn[i] = cs[i] * t[i];
      SFM_EACH(i) {
        const int p = rr_first(kM, r, i), q = rr_second(kM, r, i);
        row[p] = cs[i] * bp[i] - sn[i] * bq[i];
        if (q < N) row[q < N ? q : 0] = sn[i] * bp[i] + cs[i] * bq[i];
        nrm[p] -= t[i] * ga[i];
        nrm[q] += t[i] * ga[i];
      }
    }
    if (!rotated) break;
  }
#undef SFM_EACH
}

}  // namespace sfm
