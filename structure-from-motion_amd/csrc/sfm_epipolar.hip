// sfm_epipolar.hip — two-view initialisation: eight-point fundamental-matrix
// RANSAC, essential matrix, the four pose candidates and the cheirality vote.
//
//   fund_normalize_kernel    one workgroup: centroid + mean-distance normalisation of both point sets
//                            (epipolar_processor.py:97-137) -> pairs [n][4] and the two 3x3 transforms
//   fund_eight_point_kernel  wave per hypothesis: 8x9 design matrix -> null vector by a wave-cooperative one-sided
//                            Jacobi (9x9 with a zero row), rank-2 projection by a 3x3 Jacobi SVD, / f[2][2]
//                            (epipolar:140-193)
//   fund_score_kernel        workgroup per hypothesis: |x_r^T F x_l| < threshold over all pairs (epipolar:231-239)
//   fund_finish_kernel       inlier mask of the winner + de-normalisation (epipolar:251-267)
//   essential_kernel / pose_candidates_kernel   single-thread 3x3 algebra (epipolar:60-95, campose:29-100)
//   cheirality_kernel        both depths positive, per candidate (campose:133-189)
// Everything is once-per-sequence work; the kernels are written for clarity, not throughput.
#include <cmath>
#include <vector>

#include "sfm_common.h"

namespace sfm {

// 3x3 SVD pieces: columns of B become sigma_c u_c, V the right singular vectors; ord[] sorts sigma descending.
__device__ void svd3(double (&B)[3][3], double (&V)[3][3], double (&sig)[3], int (&ord)[3]) {
  jacobi_right_vectors<3>(B, V, 40);
  for (int c = 0; c < 3; ++c) sig[c] = sqrt(B[0][c] * B[0][c] + B[1][c] * B[1][c] + B[2][c] * B[2][c]);
  ord[0] = 0; ord[1] = 1; ord[2] = 2;
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (sig[ord[j]] > sig[ord[i]]) { const int t = ord[i]; ord[i] = ord[j]; ord[j] = t; }
}

__global__ __launch_bounds__(256) void fund_normalize_kernel(int n, const double* __restrict__ left, const double* __restrict__ right,
                                                             double* __restrict__ pairs /*[n][4]*/, double* __restrict__ T /*[2][9]*/) {
  __shared__ double red[4][4];
  __shared__ double stat[6];     // mean_lx, mean_ly, mean_rx, mean_ry, scale_l, scale_r
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  auto block_sum4 = [&](double (&v)[4]) {
    for (int k = 0; k < 4; ++k) v[k] = wave_sum(v[k]);
    if (lane == 0) for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    for (int k = 0; k < 4; ++k) v[k] = red[0][k] + red[1][k] + red[2][k] + red[3][k];
    __syncthreads();
  };
  double s[4] = {0, 0, 0, 0};
  for (int i = tid; i < n; i += blockDim.x) { s[0] += left[i]; s[1] += left[n + i]; s[2] += right[i]; s[3] += right[n + i]; }
  block_sum4(s);
  const double ml[2] = {s[0] / n, s[1] / n}, mr[2] = {s[2] / n, s[3] / n};
  double d[4] = {0, 0, 0, 0};
  for (int i = tid; i < n; i += blockDim.x) {
    const double ax = left[i] - ml[0], ay = left[n + i] - ml[1], bx = right[i] - mr[0], by = right[n + i] - mr[1];
    d[0] += sqrt(ax * ax + ay * ay);
    d[1] += sqrt(bx * bx + by * by);
  }
  block_sum4(d);
  const double sl = sqrt(2.0 * n) / d[0], sr = sqrt(2.0 * n) / d[1];      // epipolar:123-124
  if (tid == 0) {
    stat[0] = ml[0]; stat[1] = ml[1]; stat[2] = mr[0]; stat[3] = mr[1]; stat[4] = sl; stat[5] = sr;
    const double tl[9] = {sl, 0, -ml[0] * sl, 0, sl, -ml[1] * sl, 0, 0, 1};
    const double tr[9] = {sr, 0, -mr[0] * sr, 0, sr, -mr[1] * sr, 0, 0, 1};
    for (int k = 0; k < 9; ++k) { T[k] = tl[k]; T[9 + k] = tr[k]; }
  }
  for (int i = tid; i < n; i += blockDim.x) {
    pairs[4 * i + 0] = sl * left[i] + -ml[0] * sl;
    pairs[4 * i + 1] = sl * left[n + i] + -ml[1] * sl;
    pairs[4 * i + 2] = sr * right[i] + -mr[0] * sr;
    pairs[4 * i + 3] = sr * right[n + i] + -mr[1] * sr;
  }
}

// epipolar_processor.py:140-193, one wave per hypothesis: lanes 0..7 hold the rows of the 8x9 design matrix
// (lane 8 a zero row), lanes 16..24 the identity; after jacobi_rows_wave the null vector is the column of V
// whose B V column has the smallest norm.  The 3x3 rank-2 projection runs on lane 0.
__global__ __launch_bounds__(64) void fund_eight_point_kernel(int n_hyp, const int* __restrict__ samples, const double* __restrict__ pairs,
                                                              double* __restrict__ F_out, int* __restrict__ status) {
  const int h = blockIdx.x;
  const int lane = threadIdx.x;
  if (h >= n_hyp) return;
  double row[9];
#pragma unroll
  for (int c = 0; c < 9; ++c) row[c] = 0.0;
  if (lane < 8) {
    const int idx = samples ? samples[8 * h + lane] : lane;
    const double* p = pairs + 4 * (size_t)idx;
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    row[0] = x1 * x2; row[1] = y1 * x2; row[2] = x2;
    row[3] = x1 * y2; row[4] = y1 * y2; row[5] = y2;
    row[6] = x1;      row[7] = y1;      row[8] = 1.0;
  } else if (lane >= 16 && lane < 25) {
#pragma unroll
    for (int c = 0; c < 9; ++c) row[c] = (c == lane - 16) ? 1.0 : 0.0;
  }
  jacobi_rows_wave<9>(row, lane, 40);
  int best = 0;
  double bn = 0;
#pragma unroll
  for (int c = 0; c < 9; ++c) {
    const double nn = wave_lane0(group_sum<16>(lane < 16 ? row[c] * row[c] : 0.0));
    if (c == 0 || nn < bn) { bn = nn; best = c; }
  }
  double mine = 0.0;
#pragma unroll
  for (int c = 0; c < 9; ++c) mine = (c == best) ? row[c] : mine;
  double f[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) {                                  // f_ = reshape(null vector, (3, 3))
    const int lo = __builtin_amdgcn_readlane(__double2loint(mine), 16 + k);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(mine), 16 + k);
    f[k] = __hiloint2double(hi, lo);
  }
  if (lane != 0) return;
  double B[3][3], V3[3][3], sig[3];
  int ord[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[i][j] = f[3 * i + j];
  svd3(B, V3, sig, ord);
  // matrix_rank tolerance: sigma_max * max(M, N) * eps (numpy.linalg.matrix_rank)
  const double tol = sig[ord[0]] * 3.0 * 2.220446049250313e-16;
  int st = (sig[ord[1]] > tol) ? SFM_OK : SFM_E_RANK;
  if (!(sig[ord[0]] == sig[ord[0]])) st = SFM_E_RANK;
  double f2[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)       // u diag(s0, s1, 0) v^T = sum over the two largest of (sigma u)_i v_j
      f2[3 * i + j] = B[i][ord[0]] * V3[j][ord[0]] + B[i][ord[1]] * V3[j][ord[1]];
  for (int k = 0; k < 9; ++k) F_out[9 * (size_t)h + k] = f2[k] / f2[8];
  status[h] = st;
}

__device__ __forceinline__ bool fund_is_inlier(const double* F, const double* p, double threshold) {
  // |[x_r y_r 1] F [x_l y_l 1]^T| < threshold  (epipolar:233-236)
  const double a0 = F[0] * p[0] + F[1] * p[1] + F[2];
  const double a1 = F[3] * p[0] + F[4] * p[1] + F[5];
  const double a2 = F[6] * p[0] + F[7] * p[1] + F[8];
  return fabs(p[2] * a0 + p[3] * a1 + a2) < threshold;
}

__global__ __launch_bounds__(256) void fund_score_kernel(int n, const double* __restrict__ F_all, const double* __restrict__ pairs,
                                                         double threshold, int* __restrict__ counts) {
  __shared__ int wsum[4];
  const int h = blockIdx.x;
  double F[9];
  for (int k = 0; k < 9; ++k) F[k] = F_all[9 * (size_t)h + k];
  int cnt = 0;
  for (int i = threadIdx.x; i < n; i += blockDim.x) cnt += fund_is_inlier(F, pairs + 4 * (size_t)i, threshold) ? 1 : 0;
  for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
  if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) counts[h] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// mask of the winning hypothesis (mode 0) or all ones (mode 1: the n == 8 case); thread 0 also de-normalises
// F = T_r^T F T_l / [2][2]  (epipolar:265-266)
__global__ void fund_finish_kernel(int n, const double* __restrict__ F, const double* __restrict__ pairs, const double* __restrict__ T,
                                   double threshold, int all_inliers, int* __restrict__ mask, double* __restrict__ F_pix) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double f[9];
  for (int k = 0; k < 9; ++k) f[k] = F[k];
  if (i < n) mask[i] = all_inliers ? 1 : (fund_is_inlier(f, pairs + 4 * (size_t)i, threshold) ? 1 : 0);
  if (i == 0) {
    const double* tl = T;
    const double* tr = T + 9;
    double m[9], g[9];
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) m[3 * a + b] = tr[0 + a] * f[b] + tr[3 + a] * f[3 + b] + tr[6 + a] * f[6 + b];   // T_r^T F
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) g[3 * a + b] = m[3 * a] * tl[b] + m[3 * a + 1] * tl[3 + b] + m[3 * a + 2] * tl[6 + b];
    for (int k = 0; k < 9; ++k) F_pix[k] = g[k] / g[8];
  }
}

// epipolar_processor.py:60-95
__global__ void essential_kernel(const double* __restrict__ F, const double* __restrict__ Kl, const double* __restrict__ Kr,
                                 double* __restrict__ E_out, int* __restrict__ status) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double m[9], B[3][3], V[3][3], sig[3];
  int ord[3];
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) m[3 * a + b] = Kr[0 + a] * F[b] + Kr[3 + a] * F[3 + b] + Kr[6 + a] * F[6 + b];      // K_r^T F
  for (int a = 0; a < 3; ++a)
    for (int b = 0; b < 3; ++b) B[a][b] = m[3 * a] * Kl[b] + m[3 * a + 1] * Kl[3 + b] + m[3 * a + 2] * Kl[6 + b];
  svd3(B, V, sig, ord);
  // u diag(1, 1, 0) v^T: unit left vectors of the two largest singular values
  double e[9];
  bool ok = sig[ord[1]] > 0.0 && sig[ord[0]] == sig[ord[0]];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      e[3 * i + j] = B[i][ord[0]] / sig[ord[0]] * V[j][ord[0]] + B[i][ord[1]] / sig[ord[1]] * V[j][ord[1]];
  for (int k = 0; k < 9; ++k) E_out[k] = e[k] / e[8];
  *status = ok ? SFM_OK : SFM_E_RANK;
}

// campose_processor.py:29-100.  E = s (u1 v1^T + u2 v2^T); with right-handed completions u3 = u1 x u2, v3 = v1 x v2:
// U W V^T = u2 v1^T - u1 v2^T + u3 v3^T =: A + T (det +1) and U W^T V^T = -A + T; the returned matrices are their
// transposes and c1 = u3 (left null vector).
__global__ void pose_candidates_kernel(const double* __restrict__ E, double* __restrict__ R_out /*[2][9]*/, double* __restrict__ C1) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double B[3][3], V[3][3], sig[3];
  int ord[3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) B[i][j] = E[3 * i + j];
  svd3(B, V, sig, ord);
  double u[2][3], v[2][3], u3[3], v3[3];
  for (int k = 0; k < 2; ++k)
    for (int i = 0; i < 3; ++i) { u[k][i] = B[i][ord[k]] / sig[ord[k]]; v[k][i] = V[i][ord[k]]; }
  u3[0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u3[1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u3[2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
  v3[0] = v[0][1] * v[1][2] - v[0][2] * v[1][1]; v3[1] = v[0][2] * v[1][0] - v[0][0] * v[1][2]; v3[2] = v[0][0] * v[1][1] - v[0][1] * v[1][0];
  double ra[9], rb[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      const double a = u[1][i] * v[0][j] - u[0][i] * v[1][j], t = u3[i] * v3[j];
      ra[3 * i + j] = a + t;
      rb[3 * i + j] = -a + t;
    }
  if (det3(ra) < 0) for (int k = 0; k < 9; ++k) ra[k] = -ra[k];      // campose:74-77 (never taken with the completions above)
  if (det3(rb) < 0) for (int k = 0; k < 9; ++k) rb[k] = -rb[k];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { R_out[3 * i + j] = ra[3 * j + i]; R_out[9 + 3 * i + j] = rb[3 * j + i]; }   // campose:95-96
  for (int i = 0; i < 3; ++i) C1[i] = u3[i];
}

// campose_processor.py:133-189; blockIdx.y = candidate
__global__ __launch_bounds__(256) void cheirality_kernel(int n, const double* __restrict__ P1, const double* __restrict__ P2 /*[k][12]*/,
                                                         const double* __restrict__ X /*[k][4][n]*/, int* __restrict__ mask,
                                                         int* __restrict__ counts) {
  const int c = blockIdx.y;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const double* x = X + (size_t)c * 4 * n;
  const double* p2 = P2 + 12 * c;
  int ok = 0;
  if (i < n) {
    const double x0 = x[i], x1 = x[(size_t)n + i], x2 = x[2 * (size_t)n + i], x3 = x[3 * (size_t)n + i];
    const double z1 = P1[8] * x0 + P1[9] * x1 + P1[10] * x2 + P1[11] * x3;
    const double z2 = p2[8] * x0 + p2[9] * x1 + p2[10] * x2 + p2[11] * x3;
    ok = (z1 > 0 && z2 > 0) ? 1 : 0;
    mask[(size_t)c * n + i] = ok;
  }
  int cnt = ok;
  for (int s = 32; s > 0; s >>= 1) cnt += __shfl_xor(cnt, s, 64);
  if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counts[c], cnt);
}

}  // namespace sfm

using namespace sfm;

extern "C" {

int sfm_fundamental_eight_point(int n, const double* pairs, int n_hyp, const int* samples, double* F_out, int* status) {
  SFM_TRY(ensure_init());
  if (n < 8 || n_hyp < 1) { set_error("sfm_fundamental_eight_point: need n >= 8 pairs and n_hyp >= 1 (n=%d n_hyp=%d)", n, n_hyp); return SFM_E_SHAPE; }
  for (int i = 0; i < 8 * n_hyp; ++i)
    if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_fundamental_eight_point: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<double> dP, dF;
  DevBuf<int> dS, dSt;
  SFM_TRY(dP.upload(pairs, 4 * (size_t)n, s)); SFM_TRY(dS.upload(samples, 8 * (size_t)n_hyp, s));
  SFM_TRY(dF.alloc(9 * (size_t)n_hyp)); SFM_TRY(dSt.alloc(n_hyp));
  fund_eight_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, dS.p, dP.p, dF.p, dSt.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dF.download(F_out, 9 * (size_t)n_hyp, s)); SFM_TRY(dSt.download(status, n_hyp, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_fundamental_ransac(int n, const double* left, const double* right, int n_hyp, const int* samples, double threshold,
                           double F_out[9], int* inlier_mask, int* n_inliers, int* best_hypothesis) {
  SFM_TRY(ensure_init());
  if (n < 8) { set_error("Insufficient matched pairs : %d", n); return SFM_E_SHAPE; }      // epipolar:213-215
  const bool exact = n == 8;
  if (exact) n_hyp = 1;
  if (n_hyp < 1) { set_error("sfm_fundamental_ransac: n_hyp < 1"); return SFM_E_SHAPE; }
  if (!exact)
    for (int i = 0; i < 8 * n_hyp; ++i)
      if (samples[i] < 0 || samples[i] >= n) { set_error("sfm_fundamental_ransac: sample index %d out of range", samples[i]); return SFM_E_SHAPE; }
  hipStream_t s = ctx().stream;
  DevBuf<double> dL, dR, dP, dT, dF, dFp;
  DevBuf<int> dS, dSt, dCnt, dMask;
  SFM_TRY(dL.upload(left, 2 * (size_t)n, s)); SFM_TRY(dR.upload(right, 2 * (size_t)n, s));
  if (!exact) SFM_TRY(dS.upload(samples, 8 * (size_t)n_hyp, s));
  SFM_TRY(dP.alloc(4 * (size_t)n)); SFM_TRY(dT.alloc(18)); SFM_TRY(dF.alloc(9 * (size_t)n_hyp)); SFM_TRY(dFp.alloc(9));
  SFM_TRY(dSt.alloc(n_hyp)); SFM_TRY(dCnt.alloc(n_hyp)); SFM_TRY(dMask.alloc(n));
  fund_normalize_kernel<<<1, 256, 0, s>>>(n, dL.p, dR.p, dP.p, dT.p);
  fund_eight_point_kernel<<<n_hyp, 64, 0, s>>>(n_hyp, exact ? nullptr : dS.p, dP.p, dF.p, dSt.p);
  if (!exact) fund_score_kernel<<<n_hyp, 256, 0, s>>>(n, dF.p, dP.p, threshold, dCnt.p);
  SFM_HIP(hipGetLastError());
  std::vector<int> st(n_hyp), counts(n_hyp, 8);
  SFM_TRY(dSt.download(st.data(), n_hyp, s));
  if (!exact) SFM_TRY(dCnt.download(counts.data(), n_hyp, s));
  SFM_TRY(stream_sync(s));
  for (int h = 0; h < n_hyp; ++h)
    if (st[h] != SFM_OK) { set_error("f__ rank is not equal to 2 (hypothesis %d)", h); return SFM_E_RANK; }
  // the reference keeps the FIRST hypothesis with a strictly larger count, starting from 0 inliers (epipolar:222-245)
  int best = exact ? 0 : -1, best_cnt = exact ? 8 : 0;
  if (!exact)
    for (int h = 0; h < n_hyp; ++h)
      if (counts[h] > best_cnt) { best_cnt = counts[h]; best = h; }
  if (best_hypothesis) *best_hypothesis = best;
  if (n_inliers) *n_inliers = best_cnt;
  if (best < 0) {
    for (int i = 0; i < n; ++i) inlier_mask[i] = 0;
    for (int k = 0; k < 9; ++k) F_out[k] = NAN;
    return SFM_OK;
  }
  fund_finish_kernel<<<(n + 255) / 256, 256, 0, s>>>(n, dF.p + 9 * (size_t)best, dP.p, dT.p, threshold, exact ? 1 : 0, dMask.p, dFp.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dMask.download(inlier_mask, n, s));
  SFM_TRY(dFp.download(F_out, 9, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_essential_from_fundamental(const double F[9], const double K_left[9], const double K_right[9], double E_out[9]) {
  SFM_TRY(ensure_init());
  hipStream_t s = ctx().stream;
  DevBuf<double> dF, dKl, dKr, dE;
  DevBuf<int> dSt;
  SFM_TRY(dF.upload(F, 9, s)); SFM_TRY(dKl.upload(K_left, 9, s)); SFM_TRY(dKr.upload(K_right, 9, s));
  SFM_TRY(dE.alloc(9)); SFM_TRY(dSt.alloc(1));
  essential_kernel<<<1, 64, 0, s>>>(dF.p, dKl.p, dKr.p, dE.p, dSt.p);
  SFM_HIP(hipGetLastError());
  int st = SFM_OK;
  SFM_TRY(dE.download(E_out, 9, s)); SFM_TRY(dSt.download(&st, 1, s));
  SFM_TRY(stream_sync(s));
  if (st != SFM_OK) set_error("esse_mat rank is not equal to 2");
  return st;
}

int sfm_pose_candidates(const double E[9], double R_out[18], double C1_out[3]) {
  SFM_TRY(ensure_init());
  hipStream_t s = ctx().stream;
  DevBuf<double> dE, dR, dC;
  SFM_TRY(dE.upload(E, 9, s)); SFM_TRY(dR.alloc(18)); SFM_TRY(dC.alloc(3));
  pose_candidates_kernel<<<1, 64, 0, s>>>(dE.p, dR.p, dC.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dR.download(R_out, 18, s)); SFM_TRY(dC.download(C1_out, 3, s));
  SFM_TRY(stream_sync(s));
  return SFM_OK;
}

int sfm_cheirality(int k, int n, const double P1[12], const double* P2, const double* X, int* mask, int* counts, int* best) {
  SFM_TRY(ensure_init());
  if (k < 1 || n < 0) { set_error("sfm_cheirality: bad sizes k=%d n=%d", k, n); return SFM_E_SHAPE; }
  int b = 0, bc = 0;
  if (n > 0) {
    hipStream_t s = ctx().stream;
    DevBuf<double> dP1, dP2, dX;
    DevBuf<int> dM, dC;
    SFM_TRY(dP1.upload(P1, 12, s)); SFM_TRY(dP2.upload(P2, 12 * (size_t)k, s)); SFM_TRY(dX.upload(X, 4 * (size_t)k * n, s));
    SFM_TRY(dM.alloc((size_t)k * n)); SFM_TRY(dC.alloc(k));
    SFM_HIP(hipMemsetAsync(dC.p, 0, sizeof(int) * k, s));
    cheirality_kernel<<<dim3((n + 255) / 256, k), 256, 0, s>>>(n, dP1.p, dP2.p, dX.p, dM.p, dC.p);
    SFM_HIP(hipGetLastError());
    SFM_TRY(dM.download(mask, (size_t)k * n, s)); SFM_TRY(dC.download(counts, k, s));
    SFM_TRY(stream_sync(s));
  } else {
    for (int c = 0; c < k; ++c) counts[c] = 0;
  }
  for (int c = 0; c < k; ++c)        // campose:121-129: first strictly larger count, starting from (0, 0)
    if (counts[c] > bc) { bc = counts[c]; b = c; }
  if (best) *best = b;
  return SFM_OK;
}

}  // extern "C"
