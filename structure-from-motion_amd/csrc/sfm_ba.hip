// sfm_ba.hip — bundle-adjustment kernels and the device-resident problem object (gfx950).
//
// One damped Gauss-Newton iteration of BaProcessor.__execute_bundle_adjustment
// (ba_processor.py:297-406), restructured as a block-sparse Schur solve that never builds the
// reference's dense J:
//
//   ba_cam_prep        per camera: R(q) (+ validity), canonical q^ from R, t = R^T(-C)     (ba:321-328, campose:464)
//   ba_linearize<G>    G lanes per point, one observation per lane: r, Jp, Jx; wave-segment reduction of
//                      V_p = sum Jx^T Jx + lambda I and g_p = sum Jx^T r; L_p = chol(V_p);
//                      Z_o = (Jp^T Jx) L_p^-T -> HBM (21 doubles / obs);
//                      U_c = sum Jp^T Jp and rhs_c = sum (Jp^T r - Z_o L_p^-1 g_p) accumulated in LDS
//                      (ds_add_f64) per workgroup and flushed once with global f64 atomics            (ba:355-379)
//   ba_schur_*         S -= sum_p Z_p Z_p^T  (= B D^-1 B^T, ba:382)                         [sfm_ba_schur.hip]
//   ba_chol_panel      blocked left-looking Cholesky of S + lambda I with the rhs carried as an extra row
//   ba_back_solve      L^T dp = y; cams += dp; q /= |q| (ba:383-392); cam_prep for the next iteration
//   ba_backsub<G>      per point: recompute the linearisation from the 20 B/observation inputs and
//                      dX = V_p^-1 (g_p - sum_o W_o^T dp_c); X += dX                          (ba:405-406)
//
// HBM layout: observations sorted by (point, camera) as a CSR over points; u[], v[] (normalised
// keys), cam_idx[] SoA; points SoA X[], Y[], Z[]; cameras AoS [V][7]; reduced system
// [S (ld x ld, lower triangle valid) | rhs (ld)] contiguous so one all-reduce covers both.
#include <algorithm>
#include <vector>

#include "sfm_ba.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
__global__ void ba_cam_prep_kernel(int V, const double* __restrict__ cams, CamPrep* __restrict__ prep,
                                   int* __restrict__ status) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= V) return;
  CamPrep out;
  const int st = cam_prepare(cams + 7 * c, &out);
  prep[c] = out;
  report_status(status, st, c);
}

// Residual + Jacobians of one observation at the prepared camera (ba_processor.py:317-349).
__device__ __forceinline__ void obs_terms(const CamPrep& c, double X, double Y, double Z, double u, double v,
                                          int quirks, double* r, double* Jp, double* Jx) {
  double p[3];
  project_cam(c, X, Y, Z, 1.0, p);
  const double iz = rcp_nr(p[2]);        // one reciprocal (v_rcp_f64 + two Newton steps) per observation; f = p * iz (ba:339-342)
  jac_cam_iz(c, X, Y, Z, p, iz, quirks, Jp);
  jac_pt_cam_iz(c, p, iz, Jx);
  r[0] = u - p[0] * iz;        // b - f (ba:376)
  r[1] = v - p[1] * iz;
}

__device__ __forceinline__ void load_cam(CamPrep& dst, const CamPrep* src) {
  const double* s = reinterpret_cast<const double*>(src);
  double* d = reinterpret_cast<double*>(&dst);
#pragma unroll
  for (int k = 0; k < 19; ++k) d[k] = s[k];
}

// ---------------------------------------------------------------------------------------------
// ba_linearize: G lanes per point (G = power of two <= 64 chosen from the mean track length).
// LDS_MODE 2: [V][19] prepared cameras + [V][35] camera-side accumulators in LDS (V <= 151);
// LDS_MODE 1: accumulators only, cameras read from global/L2 (V <= 234); LDS_MODE 0: global atomics.
// ---------------------------------------------------------------------------------------------
// DENSE_Z: Z_o goes to its 7x3 slot of the dense Zd (MFMA product); otherwise to the AoS Z of the sparse product.
template <int G, int LDS_MODE, bool DENSE_Z>
__global__ __launch_bounds__(256) void ba_linearize_kernel(BaDev d, int cur, double lambda, int quirks) {
  extern __shared__ double lds[];
  unsigned long long* stamp = (d.stamps && blockIdx.x == 0 && threadIdx.x == 0) ? d.stamps + 192 : nullptr;
  int sidx = 0;
  if (stamp) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  constexpr bool PREP_LDS = LDS_MODE == 2, ACC_LDS = LDS_MODE >= 1;
  double* lds_prep = lds;                                        // V * 19 (mode 2)
  double* lds_acc = lds + (PREP_LDS ? (size_t)d.V * 19 : 0);     // V * 35
  const CamPrep* gprep = d.prep[cur];
  if (ACC_LDS) {
    if (PREP_LDS) {
      const double* src = reinterpret_cast<const double*>(gprep);
      for (int i = threadIdx.x; i < d.V * 19; i += blockDim.x) lds_prep[i] = src[i];
    }
    for (int i = threadIdx.x; i < d.V * 35; i += blockDim.x) lds_acc[i] = 0.0;
    __syncthreads();
  }
  if (stamp) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  constexpr int GPB = 256 / G;                 // point groups per block
  const int lane_g = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  double* S = d.red;
  double* rhs = d.red + (size_t)d.ld * d.ld;

  for (int p0 = blockIdx.x * GPB; p0 < d.N; p0 += gridDim.x * GPB) {
    const int p = p0 + grp;
    int beg = 0, end = 0;
    double X = 0, Y = 0, Z = 0;
    if (p < d.N) {
      beg = d.pt_ptr[p]; end = d.pt_ptr[p + 1];
      X = d.px[p]; Y = d.py[p]; Z = d.pz[p];
    }
    double v6[6] = {0, 0, 0, 0, 0, 0}, g3[3] = {0, 0, 0};
    double r[2], Jp[14], Jx[6];
    int cam = 0;
    for (int o = beg + lane_g; o < end; o += G) {
      cam = d.cam_idx[o];
      CamPrep c;
      load_cam(c, PREP_LDS ? reinterpret_cast<const CamPrep*>(lds_prep) + cam : gprep + cam);
      obs_terms(c, X, Y, Z, d.u[o], d.v[o], quirks, r, Jp, Jx);
      v6[0] += Jx[0] * Jx[0] + Jx[3] * Jx[3];
      v6[1] += Jx[1] * Jx[0] + Jx[4] * Jx[3];
      v6[2] += Jx[1] * Jx[1] + Jx[4] * Jx[4];
      v6[3] += Jx[2] * Jx[0] + Jx[5] * Jx[3];
      v6[4] += Jx[2] * Jx[1] + Jx[5] * Jx[4];
      v6[5] += Jx[2] * Jx[2] + Jx[5] * Jx[5];
      g3[0] += Jx[0] * r[0] + Jx[3] * r[1];
      g3[1] += Jx[1] * r[0] + Jx[4] * r[1];
      g3[2] += Jx[2] * r[0] + Jx[5] * r[1];
    }
    if (stamp && sidx < 60) { asm volatile("" :: "v"(v6[0])); stamp[sidx++] = __builtin_amdgcn_s_memtime(); }
#pragma unroll
    for (int k = 0; k < 6; ++k) v6[k] = group_sum<G>(v6[k]);
#pragma unroll
    for (int k = 0; k < 3; ++k) g3[k] = group_sum<G>(g3[k]);
    v6[0] += lambda; v6[2] += lambda; v6[5] += lambda;     // ba:359
    double li[6];
    chol3_inv_fast(v6, li);
    const double y0 = li[0] * g3[0];
    const double y1 = li[1] * g3[0] + li[2] * g3[1];
    const double y2 = li[3] * g3[0] + li[4] * g3[1] + li[5] * g3[2];
    // h = L^-T y = V^-1 g
    const double h2 = li[5] * y2;
    const double h1 = li[2] * y1 + li[4] * y2;
    const double h0 = li[0] * y0 + li[1] * y1 + li[3] * y2;
    if (stamp && sidx < 60) { asm volatile("" :: "v"(h0)); stamp[sidx++] = __builtin_amdgcn_s_memtime(); }
    const bool single = (end - beg) <= G;
    for (int o = beg + lane_g; o < end; o += G) {
      if (!single) {
        cam = d.cam_idx[o];
        CamPrep c;
        load_cam(c, PREP_LDS ? reinterpret_cast<const CamPrep*>(lds_prep) + cam : gprep + cam);
        obs_terms(c, X, Y, Z, d.u[o], d.v[o], quirks, r, Jp, Jx);
      }
      double acc[35];
      {
        // Z_o = (Jp^T Jx) L^-T = Jp^T (Jx L^-T): the 2x3 product first (ba:379 block B_{c,p})
        double m0[3], m1[3];
        m0[0] = Jx[0] * li[0];
        m0[1] = Jx[0] * li[1] + Jx[1] * li[2];
        m0[2] = Jx[0] * li[3] + Jx[1] * li[4] + Jx[2] * li[5];
        m1[0] = Jx[3] * li[0];
        m1[1] = Jx[3] * li[1] + Jx[4] * li[2];
        m1[2] = Jx[3] * li[3] + Jx[4] * li[4] + Jx[5] * li[5];
        if (DENSE_Z) {
          const int blk = cam / kSchurCB;
          double* zr = d.Zd + (size_t)(3 * p) * d.zp + blk * kSchurRB + 7 * (cam - blk * kSchurCB);
          // 7 consecutive doubles per row, 8-byte aligned: three 16-byte stores + one 8-byte store per row
          // (12 write requests per observation instead of 21; the request rate bounds this kernel's tail)
          typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
#pragma unroll
          for (int j = 0; j < 3; ++j) {
            double zz[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) zz[i] = Jp[i] * m0[j] + Jp[7 + i] * m1[j];
            double* row = zr + (size_t)j * d.zp;
#pragma unroll
            for (int i = 0; i < 6; i += 2) *reinterpret_cast<d2u*>(row + i) = d2u{zz[i], zz[i + 1]};
            row[6] = zz[6];
          }
        } else {
          double* zo = d.Z + (size_t)o * 21;          // AoS: the 21 elements of observation o are contiguous
          typedef double d2u __attribute__((ext_vector_type(2), aligned(8)));
          double zz[22];
#pragma unroll
          for (int i = 0; i < 7; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) zz[3 * i + j] = Jp[i] * m0[j] + Jp[7 + i] * m1[j];
          }
#pragma unroll
          for (int e = 0; e < 20; e += 2) *reinterpret_cast<d2u*>(zo + e) = d2u{zz[e], zz[e + 1]};
          zo[20] = zz[20];
        }
      }
      // rhs_c -= W V^-1 g = Jp^T (Jx h) with h = V^-1 g: folded into the residual, e = r - Jx h
      const double e0 = r[0] - (Jx[0] * h0 + Jx[1] * h1 + Jx[2] * h2);
      const double e1 = r[1] - (Jx[3] * h0 + Jx[4] * h1 + Jx[5] * h2);
      int k = 0;
#pragma unroll
      for (int i = 0; i < 7; ++i) {
#pragma unroll
        for (int j = 0; j <= i; ++j) { acc[k] = Jp[i] * Jp[j] + Jp[7 + i] * Jp[7 + j]; ++k; }
        acc[28 + i] = Jp[i] * e0 + Jp[7 + i] * e1;
      }
      if (ACC_LDS) {
        double* a = lds_acc + (size_t)cam * 35;
#pragma unroll
        for (int q = 0; q < 35; ++q) atomicAdd(a + q, acc[q]);
      } else {
        k = 0;
        for (int i = 0; i < 7; ++i)
          for (int j = 0; j <= i; ++j) { atomicAdd(&S[(size_t)(7 * cam + i) * d.ld + 7 * cam + j], acc[k]); ++k; }
        for (int i = 0; i < 7; ++i) atomicAdd(&rhs[7 * cam + i], acc[28 + i]);
      }
    }
  }
  if (stamp && sidx < 62) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  if (ACC_LDS) {
    // per-workgroup partial sums -> workspace row (plain coalesced stores); ba_schur_reduce_kernel adds
    // them into S / rhs.  (Flushing with global atomics made 512 workgroups collide on the same 1750
    // addresses: ~18 G atomics/s on MI355X, 50 us at C3.)
    __syncthreads();
    double* row = d.lin_ws + (size_t)blockIdx.x * d.V * 35;
    for (int t = threadIdx.x; t < d.V * 35; t += blockDim.x) row[t] = lds_acc[t];
  }
  if (stamp && sidx < 63) { stamp[sidx++] = __builtin_amdgcn_s_memtime(); stamp[63] = sidx; }
}

// ---------------------------------------------------------------------------------------------
// ba_backsub: dX_p = V_p^-1 (g_p - sum_o W_o^T dp_c), recomputing the linearisation at the
// iteration's starting state (prep[cur] and the not-yet-updated points).
// ---------------------------------------------------------------------------------------------
template <int G, bool CAMS_IN_LDS>
__global__ __launch_bounds__(256) void ba_backsub_kernel(BaDev d, int cur, double lambda, int quirks) {
  extern __shared__ double lds[];
  double* lds_prep = lds;                      // V * 19
  double* lds_delta = lds + (size_t)d.V * 19;  // V * 7
  const CamPrep* gprep = d.prep[cur];
  if (CAMS_IN_LDS) {
    const double* src = reinterpret_cast<const double*>(gprep);
    for (int i = threadIdx.x; i < d.V * 19; i += blockDim.x) lds_prep[i] = src[i];
    for (int i = threadIdx.x; i < d.V * 7; i += blockDim.x) lds_delta[i] = d.delta[i];
    __syncthreads();
  }
  constexpr int GPB = 256 / G;
  const int lane_g = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  // [S | rhs] is dead once ba_back_solve has run: clear it here (a few hundred bytes per workgroup) so the
  // next iteration's linearisation needs no separate memset
  {
    const size_t n_red = (size_t)d.ld * d.ld + d.ld;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_red; i += (size_t)gridDim.x * blockDim.x) d.red[i] = 0.0;
  }
  for (int p0 = blockIdx.x * GPB; p0 < d.N; p0 += gridDim.x * GPB) {
    const int p = p0 + grp;
    int beg = 0, end = 0;
    double X = 0, Y = 0, Z = 0;
    if (p < d.N) {
      beg = d.pt_ptr[p]; end = d.pt_ptr[p + 1];
      X = d.px[p]; Y = d.py[p]; Z = d.pz[p];
    }
    double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // v6 | (g - W^T dp)
    for (int o = beg + lane_g; o < end; o += G) {
      const int cam = d.cam_idx[o];
      CamPrep c;
      load_cam(c, CAMS_IN_LDS ? reinterpret_cast<const CamPrep*>(lds_prep) + cam : gprep + cam);
      double r[2], Jp[14], Jx[6];
      obs_terms(c, X, Y, Z, d.u[o], d.v[o], quirks, r, Jp, Jx);
      const double* dp = CAMS_IN_LDS ? lds_delta + 7 * cam : d.delta + 7 * cam;
      double e0 = r[0], e1 = r[1];               // r - Jp dp
#pragma unroll
      for (int i = 0; i < 7; ++i) { e0 -= Jp[i] * dp[i]; e1 -= Jp[7 + i] * dp[i]; }
      a[0] += Jx[0] * Jx[0] + Jx[3] * Jx[3];
      a[1] += Jx[1] * Jx[0] + Jx[4] * Jx[3];
      a[2] += Jx[1] * Jx[1] + Jx[4] * Jx[4];
      a[3] += Jx[2] * Jx[0] + Jx[5] * Jx[3];
      a[4] += Jx[2] * Jx[1] + Jx[5] * Jx[4];
      a[5] += Jx[2] * Jx[2] + Jx[5] * Jx[5];
      a[6] += Jx[0] * e0 + Jx[3] * e1;
      a[7] += Jx[1] * e0 + Jx[4] * e1;
      a[8] += Jx[2] * e0 + Jx[5] * e1;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = group_sum<G>(a[k]);
    if (lane_g == 0 && p < d.N) {
      a[0] += lambda; a[2] += lambda; a[5] += lambda;
      double li[6];
      chol3_inv_fast(a, li);
      const double y0 = li[0] * a[6];
      const double y1 = li[1] * a[6] + li[2] * a[7];
      const double y2 = li[3] * a[6] + li[4] * a[7] + li[5] * a[8];
      d.px[p] = X + (li[0] * y0 + li[1] * y1 + li[3] * y2);     // L^-T y
      d.py[p] = Y + (li[2] * y1 + li[4] * y2);
      d.pz[p] = Z + (li[5] * y2);
    }
  }
}

// Parity hook: per-observation r / Jp / Jx (thread per observation).
__global__ void ba_residual_jacobian_kernel(BaDev d, int cur, int quirks, const int* __restrict__ obs_pt,
                                            double* __restrict__ r_out, double* __restrict__ Jp_out,
                                            double* __restrict__ Jx_out) {
  const long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (o >= d.M) return;
  const int p = obs_pt[o];
  CamPrep c;
  load_cam(c, d.prep[cur] + d.cam_idx[o]);
  double r[2], Jp[14], Jx[6];
  obs_terms(c, d.px[p], d.py[p], d.pz[p], d.u[o], d.v[o], quirks, r, Jp, Jx);
  r_out[2 * o] = r[0]; r_out[2 * o + 1] = r[1];
  for (int k = 0; k < 14; ++k) Jp_out[14 * o + k] = Jp[k];
  for (int k = 0; k < 6; ++k) Jx_out[6 * o + k] = Jx[k];
}

// ---------------------------------------------------------------------------------------------
// Reduced solve.  S + lambda I = L L^T by a blocked right-looking Cholesky on 32x32 blocks, one launch
// per block column j.  Workgroup (r, c), j <= c <= r, first applies the previous panel's update
// A[r][c] -= L[r][j-1] L[c][j-1]^T (K = 32, all 256 threads).  Blocks right of column j store the
// result and leave.  Blocks in column j then need the factor of the diagonal block: every one of them
// recomputes D = A[j][j] - L[j][j-1] L[j][j-1]^T + lambda I locally (nobody writes S(j,j) in this
// launch, so there is no race) and hands [D; T] to ONE wave that runs a register-resident elimination
// with lanes 0..31 = rows of D and lanes 32..63 = rows of T: the column steps that factor D
// (l_jj = sqrt(d_jj), rank-1 trailing update, operands broadcast with v_readlane) perform the
// triangular solve X L_d^T = T on the other 32 lanes in the same instruction stream.  The rhs vector
// rides along as block row `nbk` (one valid row), so L y = rhs needs no pass of its own.
// L overwrites the strictly-lower blocks of S in place; diagonal factors go to d.ldiag.
// ---------------------------------------------------------------------------------------------
constexpr int NB = 32;

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// One wave: a[] = this lane's row (lanes 0..31: rows of the SPD block D, lower part valid; lanes
// 32..63: rows of T).  On return lanes 0..31 hold the rows of L_d (lower) and lanes 32..63 the rows of
// X = T L_d^-T.  Column j of L_d is published through a 32-double LDS buffer and read back with
// wave-uniform addresses (LDS broadcast): one ds_read + one FMA per trailing entry, no SGPR hazards.
// Single wave => its LDS operations execute in order; no barrier is needed.
__device__ __forceinline__ void chol_trsm_rows(double (&a)[NB], double (*colbuf)[64], int lane) {
  // Software-pipelined over columns.  Per column j:
  //   1. pivot chain: inv = rsqrt(d_jj), scale column j                       (critical path)
  //   2. deferred bulk update with column j-1, whose entries were read back from LDS one step ago
  //   3. fast path: the FAST trailing columns the next pivots depend on, through register broadcasts
  //   4. publish column j in its own LDS row (all 64 lanes store: no exec masking, one basic block) and
  //      issue the LDS reads whose values step j+1 consumes in (2)
  // Every LDS row is written once and read once, so there are no WAR hazards for the scheduler to respect.
  constexpr int FAST = 2;
  double lk[NB];
#pragma unroll
  for (int k = 0; k < NB; ++k) lk[k] = 0.0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    const double inv = rsqrt_nr(lane_bcast(a[j], j));
    if (j > 0) {
#pragma unroll
      for (int k = j + FAST; k < NB; ++k) a[k] -= a[j - 1] * lk[k];      // column j-1, entries k >= (j-1)+1+FAST
    }
    a[j] *= inv;
#pragma unroll
    for (int k = j + 1; k < NB && k <= j + FAST; ++k) a[k] -= a[j] * lane_bcast(a[j], k);
    colbuf[j][lane] = a[j];
#pragma unroll
    for (int k = j + 1 + FAST; k < NB; ++k) lk[k] = colbuf[j][k];
  }
}

typedef double chol_f64x4 __attribute__((ext_vector_type(4)));
constexpr int LP = NB + 2;   // LDS pitch 34 doubles: the (row, k) operand reads of v_mfma_f64_16x16x4 hit 32 distinct bank pairs

// Trailing role of a column step: a 64x64 super-tile (2x2 blocks, one block per wave) of the blocks right of
// column j gets the previous panel's update A[r][c] -= L[r][j-1] L[c][j-1]^T on the matrix pipe.  Four times
// fewer workgroups than one per block and each L tile is loaded once for two blocks, which is what matters
// when 7V is in the thousands (V = 200: ~1000 blocks per step).
__device__ __forceinline__ void chol_trailing_supertile(const BaDev& d, int j, int sr, int sc, double (*La2)[LP],
                                                        double (*Lb2)[LP]) {
  const int P = d.P, ld = d.ld;
  const int nbk = (P + NB - 1) / NB;
  double* S = d.red;
  double* rhs = d.red + (size_t)ld * ld;
  const int k0 = (j - 1) * NB;
  const int rbase = j + 1 + 2 * sr, cbase = j + 1 + 2 * sc;       // block indices of the super-tile's corner
  const int tid = threadIdx.x, ti = tid / NB, tj = tid % NB;
  const int lane = tid & 63, wave = tid >> 6;
  const int br = wave >> 1, bc = wave & 1;
  const int r = rbase + br, c = cbase + bc;
  const bool valid = c <= nbk - 1 && r >= c && r <= nbk;
  const bool is_rhs = r == nbk;
  const int lr = lane & 15, lk = lane >> 4;
  // old block values in the MFMA C/D layout (issued before the LDS hand-over so they overlap it)
  double old[2][2][4];
  if (valid) {
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
      for (int sy = 0; sy < 2; ++sy)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int i = 16 * sx + lk + 4 * g;
          const double* pr = is_rhs ? rhs : S + (size_t)(r * NB + i) * ld;
          old[sx][sy][g] = pr[c * NB + 16 * sy + lr];
        }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = ti + 8 * e;                 // 0..63: two stacked blocks
    const int rb = rbase + (i >> 5), cb = cbase + (i >> 5), ii = i & 31;
    const double* pa = rb == nbk ? rhs : S + (size_t)(min(rb, nbk - 1) * NB + ii) * ld;
    const double va = pa[k0 + tj];
    La2[i][tj] = (rb < nbk || (rb == nbk && ii == 0)) ? va : 0.0;
    const double vb = S[(size_t)(min(cb, nbk - 1) * NB + ii) * ld + k0 + tj];
    Lb2[i][tj] = cb <= nbk - 1 ? vb : 0.0;
  }
  __syncthreads();
  if (!valid) return;
  chol_f64x4 acc[2][2];
#pragma unroll
  for (int sx = 0; sx < 2; ++sx)
#pragma unroll
    for (int sy = 0; sy < 2; ++sy) acc[sx][sy] = chol_f64x4{0, 0, 0, 0};
#pragma unroll
  for (int kk = 0; kk < NB; kk += 4) {
    const double a0 = La2[32 * br + lr][kk + lk], a1 = La2[32 * br + 16 + lr][kk + lk];
    const double b0 = Lb2[32 * bc + lr][kk + lk], b1 = Lb2[32 * bc + 16 + lr][kk + lk];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int sx = 0; sx < 2; ++sx)
#pragma unroll
    for (int sy = 0; sy < 2; ++sy)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = 16 * sx + lk + 4 * g, col = c * NB + 16 * sy + lr;
        const bool row_ok = is_rhs ? (i == 0) : (r * NB + i < P);
        if (row_ok && col < P) (is_rhs ? rhs : S + (size_t)(r * NB + i) * ld)[col] = old[sx][sy][g] - acc[sx][sy][g];
      }
}

__global__ __launch_bounds__(256) void ba_chol_step_kernel(BaDev d, int j, double lambda) {
  // one LDS arena, carved per role
  constexpr int kTile = NB * LP, kSq = NB * (NB + 1);
  __shared__ double arena[3 * kTile + 2 * kSq + NB * 64];
  const int P = d.P, ld = d.ld;
  const int nbk = (P + NB - 1) / NB;
  // workgroups 0 .. nbk-j are the column role (block rows j .. nbk, the last one being the rhs row); the rest
  // are trailing super-tiles (sr >= sc) with two 64 x 34 operand tiles
  const int ncol = nbk - j + 1;
  if ((int)blockIdx.x >= ncol) {
    int t = blockIdx.x - ncol, sr = 0;
    while (t > sr) { t -= sr + 1; ++sr; }
    static_assert(4 * kTile <= 3 * kTile + 2 * kSq + NB * 64, "trailing tiles must fit the arena");
    chol_trailing_supertile(d, j, sr, t, reinterpret_cast<double(*)[LP]>(arena),
                            reinterpret_cast<double(*)[LP]>(arena + 2 * kTile));
    return;
  }
  double(*La)[LP] = reinterpret_cast<double(*)[LP]>(arena);                  // L[r][j-1]
  double(*Lb)[LP] = reinterpret_cast<double(*)[LP]>(arena + kTile);          // L[c][j-1]
  double(*Lj)[LP] = reinterpret_cast<double(*)[LP]>(arena + 2 * kTile);      // L[j][j-1] (blocks with r != j)
  double(*Tm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(arena + 3 * kTile);
  double(*Dm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(arena + 3 * kTile + kSq);
  double(*colbuf)[64] = reinterpret_cast<double(*)[64]>(arena + 3 * kTile + 2 * kSq);   // one row per elimination column
  const int c = j, r = j + blockIdx.x;
  const bool is_rhs = r == nbk;
  // diagnostic stamps (SFM_OPT_DEBUG bit 8): shader-clock reads of one column workgroup's phases
  unsigned long long* stamp = (d.stamps && blockIdx.x == 1 && threadIdx.x == 0) ? d.stamps + 8 * j : nullptr;
  if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();
  double* S = d.red;
  double* rhs = d.red + (size_t)ld * ld;
  const int r0 = r * NB, c0 = c * NB, j0 = j * NB, k0 = (j - 1) * NB;
  const int tid = threadIdx.x, ti = tid / NB, tj = tid % NB;
  const int lane = tid & 63, wave = tid >> 6;
  const bool need_d = r != j;

  // This wave's 16x16 part of the 32x32 block, in the C/D layout of v_mfma_f64_16x16x4_f64:
  // element reg of lane l is (row = 16 sx + (l >> 4) + 4 reg, col = 16 sy + (l & 15)).
  const int sx = wave >> 1, sy = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int ocol = 16 * sy + lr;
  const bool col_ok = c0 + ocol < P;

  // Every global load of the step is issued unconditionally and before the first wait: S has ld >= 32 nbk
  // rows and columns and is zero outside P x P (memset per iteration, never written there), so padded
  // rows / columns simply read as zero; the 31 non-existent rows of the rhs block are masked by a select.
  double aT[4], aD[4] = {0, 0, 0, 0};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = 16 * sx + lk + 4 * g;
    const double* pr = is_rhs ? rhs : S + (size_t)(r0 + i) * ld;
    const double t = pr[c0 + ocol];
    aT[g] = (!is_rhs || i == 0) ? t : 0.0;
    if (need_d) aD[g] = S[(size_t)(j0 + i) * ld + j0 + ocol];
  }
  if (j > 0) {
    double la[4], lb[4], lj[4] = {0, 0, 0, 0};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = ti + 8 * e;
      const double* pr = is_rhs ? rhs : S + (size_t)(r0 + i) * ld;
      const double a = pr[k0 + tj];
      la[e] = (!is_rhs || i == 0) ? a : 0.0;
      lb[e] = S[(size_t)(c0 + i) * ld + k0 + tj];
      if (need_d) lj[e] = S[(size_t)(j0 + i) * ld + k0 + tj];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int i = ti + 8 * e;
      La[i][tj] = la[e];
      Lb[i][tj] = lb[e];
      if (need_d) Lj[i][tj] = lj[e];
    }
    __syncthreads();
    if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();
    // previous-panel update on the matrix pipe: T -= L[r][j-1] L[c][j-1]^T, D -= L[j][j-1] L[j][j-1]^T.
    // A operand: lane l holds A[row = l&15][k = l>>4]; B operand: B[k = l>>4][col = l&15] = Lb[col][k].
    chol_f64x4 pT = {0, 0, 0, 0}, pD = {0, 0, 0, 0};
#pragma unroll
    for (int kk = 0; kk < NB; kk += 4)
      pT = __builtin_amdgcn_mfma_f64_16x16x4f64(La[16 * sx + lr][kk + lk], Lb[16 * sy + lr][kk + lk], pT, 0, 0, 0);
    if (need_d) {
#pragma unroll
      for (int kk = 0; kk < NB; kk += 4)
        pD = __builtin_amdgcn_mfma_f64_16x16x4f64(Lj[16 * sx + lr][kk + lk], Lj[16 * sy + lr][kk + lk], pD, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) { aT[g] -= pT[g]; aD[g] -= pD[g]; }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = 16 * sx + lk + 4 * g;
    const bool row_ok = is_rhs ? (i == 0) : (r0 + i < P);
    double t = (row_ok && col_ok) ? aT[g] : 0.0;
    if (r == j) {                      // this block IS the diagonal block: D = T + lambda I (identity on padding)
      if (i == ocol) t = col_ok ? t + lambda : 1.0;
      Dm[i][ocol] = t;
      Tm[i][ocol] = (i == ocol) ? 1.0 : 0.0;     // the T half of the diagonal workgroup carries I: X = L_d^-T for free
    } else {
      Tm[i][ocol] = t;
      double dv = (j0 + i < P && col_ok) ? aD[g] : 0.0;
      if (i == ocol) dv = col_ok ? dv + lambda : 1.0;
      Dm[i][ocol] = dv;
    }
  }
  if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  if (tid >= 64) return;
  double a[NB];
  const double(*src)[NB + 1] = lane < NB ? Dm : Tm;
  const int row = lane & (NB - 1);
#pragma unroll
  for (int k = 0; k < NB; ++k) a[k] = src[row][k];
  if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
  chol_trsm_rows(a, colbuf, lane);
  if (stamp) { asm volatile("" :: "v"(a[NB - 1])); stamp[4] = __builtin_amdgcn_s_memtime(); }
  if (r == j) {
    if (lane >= NB) {
      // row `row` of X = L_d^-T (upper triangular), stored k-major so that the back substitution's lane i reads
      // its row with coalesced loads: ldiag[j][k][i] = X[i][k]
      double* out = d.ldiag + (size_t)j * NB * NB + row;
#pragma unroll
      for (int k = 0; k < NB; ++k) out[k * NB] = (k >= row) ? a[k] : 0.0;
    }
  } else if (lane >= NB) {
    if (is_rhs) {
      if (row == 0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) if (c0 + k < P) rhs[c0 + k] = a[k];
      }
    } else if (r0 + row < P) {
      double* out = S + (size_t)(r0 + row) * ld + c0;
#pragma unroll
      for (int k = 0; k < NB; ++k) out[k] = a[k];
    }
  }
  if (stamp) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[5] = __builtin_amdgcn_s_memtime(); }
}

// L^T dp = y (y sits in rhs after the column steps), blocked back substitution in one workgroup:
// per block, wave 0 computes x_b = L_d^-T y_b as a 32x32 mat-vec with the inverse factor the factorisation left
// in ldiag (lane i holds row i; the next block's factor is prefetched) while the other waves hold the L rows of
// the blocks above in registers (two adjacent rows per thread, 32 independent 16-byte loads issued before x_b
// exists) and fold x_b into their y.  Then the camera update of ba:383-392 and the preparation of the next
// iteration.
// Rows beyond the 384 that ba_back_solve's update waves hold in registers; only the BIG instantiation
// (7V > 416) contains it, so the small-system kernel keeps its register allocation.
template <bool Y_LDS>
__device__ __forceinline__ void back_solve_far_rows(const double* __restrict__ S, int ld, int c0, int utid,
                                                              const double* xb, double* y) {
  for (int i = utid + 384; i < c0; i += 192) {
    double w[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) w[k] = S[(size_t)(c0 + k) * ld + i];
    double s = 0;
#pragma unroll
    for (int k = 0; k < NB; ++k) s += w[k] * xb[k];
    y[i] -= s;
  }
}

template <bool Y_LDS, bool BIG>
__global__ __launch_bounds__(256) void ba_back_solve_kernel(BaDev d, int cur) {
  extern __shared__ double ylds[];     // [ld] working copy of y when it fits (Y_LDS)
  __shared__ double xb[NB];
  const int P = d.P, ld = d.ld;
  const double* S = d.red;
  double* yg = d.red + (size_t)ld * ld;
  const int tid = threadIdx.x;
  const int lane = tid & (NB - 1);
  const int nbk = (P + NB - 1) / NB;
  if (Y_LDS) {
    for (int i = tid; i < ld; i += blockDim.x) ylds[i] = yg[i];
  }
  auto yref = [&](int i) -> double& { return Y_LDS ? ylds[i] : yg[i]; };
  // wave 0 owns the diagonal blocks: x_b = L_d^-T y_b as a 32x32 mat-vec with the inverse factor the
  // factorisation left in ldiag (lane i holds row i, static register indexing; no sequential column chain);
  // waves 1-3 fold x_b into the y of the blocks above, their 32 independent coalesced row loads per thread
  // issued before x_b exists
  __shared__ double yb[NB];
  double col[NB];
  if (tid < 64) {
    const double* Xd = d.ldiag + (size_t)(nbk - 1) * NB * NB;
#pragma unroll
    for (int k = 0; k < NB; ++k) col[k] = Xd[k * NB + lane];      // row `lane` of L_d^-T
  }
  __syncthreads();
  unsigned long long* stamp = (d.stamps && tid == 0) ? d.stamps + 128 : nullptr;
  const int utid = tid - 64;            // 0..191 for the update waves
  for (int b = nbk - 1; b >= 0; --b) {
    const int c0 = b * NB;
    if (stamp) stamp[4 * b + 0] = __builtin_amdgcn_s_memtime();
    double v[2][NB];
    if (tid >= 64) {
      // this thread's two ADJACENT rows 2 utid, 2 utid + 1 as one 16-byte load per k (half the requests of
      // two 8-byte loads: the hand-over to wave 0 waits for exactly these loads)
      if (2 * utid < c0) {
#pragma unroll
        for (int k = 0; k < NB; ++k) {
          const double2 t = *reinterpret_cast<const double2*>(&S[(size_t)(c0 + k) * ld + 2 * utid]);   // rows >= P are zero
          v[0][k] = t.x; v[1][k] = t.y;
        }
      }
    } else {
      if (tid < NB) yb[lane] = (c0 + lane < P) ? yref(c0 + lane) : 0.0;      // single wave: LDS in order
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
#pragma unroll
      for (int k = 0; k < NB; k += 4) {
        x0 += col[k] * yb[k]; x1 += col[k + 1] * yb[k + 1]; x2 += col[k + 2] * yb[k + 2]; x3 += col[k + 3] * yb[k + 3];
      }
      const double xi = (x0 + x1) + (x2 + x3);
      if (tid < NB) {
        xb[lane] = (c0 + lane < P) ? xi : 0.0;
        if (c0 + lane < P) d.delta[c0 + lane] = xi;
      }
      if (stamp) stamp[4 * b + 1] = __builtin_amdgcn_s_memtime();
      if (b > 0) {                       // next diagonal factor; lands during the update below
        const double* Xd = d.ldiag + (size_t)(b - 1) * NB * NB;
#pragma unroll
        for (int k = 0; k < NB; ++k) col[k] = Xd[k * NB + lane];
      }
    }
    __syncthreads();
    if (stamp) stamp[4 * b + 2] = __builtin_amdgcn_s_memtime();
    if (tid >= 64) {
      if (2 * utid < c0) {
        double s0 = 0, s1 = 0;
#pragma unroll
        for (int k = 0; k < NB; ++k) { s0 += v[0][k] * xb[k]; s1 += v[1][k] * xb[k]; }
        yref(2 * utid) -= s0;
        yref(2 * utid + 1) -= s1;
      }
      if (BIG) back_solve_far_rows<Y_LDS>(S, ld, c0, utid, xb, Y_LDS ? ylds : yg);
    }
    __syncthreads();
    if (stamp) stamp[4 * b + 3] = __builtin_amdgcn_s_memtime();
  }
  for (int c = tid; c < d.V; c += blockDim.x) {
    double cam[7];
    for (int k = 0; k < 7; ++k) cam[k] = d.cams[7 * c + k] + d.delta[7 * c + k];          // ba:383
    const double nq = sqrt(cam[3] * cam[3] + cam[4] * cam[4] + cam[5] * cam[5] + cam[6] * cam[6]);   // ba:388-392
    for (int k = 3; k < 7; ++k) cam[k] /= nq;
    for (int k = 0; k < 7; ++k) d.cams[7 * c + k] = cam[k];
    CamPrep out;
    const int st = cam_prepare(cam, &out);      // ba:323 of the next iteration / ba:412 after the last one
    d.prep[cur ^ 1][c] = out;
    report_status(d.status, st, c);
  }
}

// S (lower) -> dense symmetric host-visible copy for the parity hook.
__global__ void ba_symmetrize_kernel(const double* __restrict__ S, int ld, int P, double lambda, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * P) return;
  const int i = idx / P, j = idx % P;
  const int a = max(i, j), b = min(i, j);
  out[idx] = S[(size_t)a * ld + b] + (i == j ? lambda : 0.0);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int pick_group(const sfm_ba_problem* p) {
  // lanes per point: smallest power of two >= mean track length (clamped to [4, 64]); longer tracks loop
  double mean = p->dev.N > 0 ? (double)p->dev.M / p->dev.N : 1.0;
  int g = 4;
  while (g < 64 && g < mean) g <<= 1;
  return g;
}

template <int LDS, bool WZ>
static void launch_linearize(const sfm_ba_problem* p, int g, int grid, size_t lds, hipStream_t s, double lambda, int quirks) {
  const BaDev& d = p->dev;
  switch (g) {
    case 4: ba_linearize_kernel<4, LDS, WZ><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 8: ba_linearize_kernel<8, LDS, WZ><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 16: ba_linearize_kernel<16, LDS, WZ><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 32: ba_linearize_kernel<32, LDS, WZ><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    default: ba_linearize_kernel<64, LDS, WZ><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
  }
}

template <bool LDS>
static void launch_backsub(const sfm_ba_problem* p, int g, int grid, size_t lds, hipStream_t s, double lambda, int quirks) {
  const BaDev& d = p->dev;
  switch (g) {
    case 4: ba_backsub_kernel<4, LDS><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 8: ba_backsub_kernel<8, LDS><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 16: ba_backsub_kernel<16, LDS><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    case 32: ba_backsub_kernel<32, LDS><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
    default: ba_backsub_kernel<64, LDS><<<grid, 256, lds, s>>>(d, p->cur, lambda, quirks); break;
  }
}

void ba_tick(sfm_ba_problem* p, int kid, bool begin, hipStream_t s) {
  if (!(p->timing & (1 << kid))) return;
  KernelTimer& t = p->timers[kid];
  if (begin) {
    if (t.used == (int)t.ev.size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      t.ev.push_back({a, b});
    }
    (void)hipEventRecord(t.ev[t.used].first, s);
  } else {
    (void)hipEventRecord(t.ev[t.used].second, s);
    t.used++;
  }
}

int ba_enqueue_prep(sfm_ba_problem* p) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
  ba_tick(p, SFM_K_PREP, true, s);
  ba_cam_prep_kernel<<<(d.V + 63) / 64, 64, 0, s>>>(d.V, d.cams, d.prep[p->cur], d.status);
  ba_tick(p, SFM_K_PREP, false, s);
  SFM_HIP(hipGetLastError());
  p->prep_valid = true;
  return SFM_OK;
}

int ba_enqueue_linearize_reduce(sfm_ba_problem* p, double lambda, int quirks) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
  if (!p->prep_valid) SFM_TRY(ba_enqueue_prep(p));
  if (!p->red_clean) SFM_HIP(hipMemsetAsync(d.red, 0, sizeof(double) * ((size_t)d.ld * d.ld + d.ld), s));
  p->red_clean = false;
  const int g = pick_group(p);
  const int gpb = 256 / g;
  int grid = std::min((d.N + gpb - 1) / gpb, kLinGridPerCu * ctx().num_cus);
  if (grid < 1) grid = 1;
  const size_t lds = sizeof(double) * (size_t)d.V * (19 + 35);
  ba_tick(p, SFM_K_LINEARIZE, true, s);
  p->quirks = quirks;
  const bool dense_z = ba_schur_uses_mfma(p);
  if (dense_z) SFM_TRY(ba_schur_prepare_dense(p, s));
  if (!dense_z && d.Z == nullptr && d.M > 0) {     // sparse-product path: Z as AoS [M][21] (168 B/obs)
    SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->dev.Z), sizeof(double) * 21 * (size_t)d.M));
  }
  const size_t lds_acc = sizeof(double) * (size_t)d.V * 35;
  const int mode = lds <= 64 * 1024 ? 2 : (lds_acc <= 64 * 1024 ? 1 : 0);
  if (mode == 2) {
    if (dense_z) launch_linearize<2, true>(p, g, grid, lds, s, lambda, quirks);
    else launch_linearize<2, false>(p, g, grid, lds, s, lambda, quirks);
  } else if (mode == 1) {
    if (dense_z) launch_linearize<1, true>(p, g, grid, lds_acc, s, lambda, quirks);
    else launch_linearize<1, false>(p, g, grid, lds_acc, s, lambda, quirks);
  } else {
    if (dense_z) launch_linearize<0, true>(p, g, grid, 0, s, lambda, quirks);
    else launch_linearize<0, false>(p, g, grid, 0, s, lambda, quirks);
  }
  p->lin_rows = mode >= 1 ? grid : 0;
  ba_tick(p, SFM_K_LINEARIZE, false, s);
  SFM_HIP(hipGetLastError());
  SFM_TRY(ba_enqueue_schur(p, s));
  return SFM_OK;
}

int ba_enqueue_solve_update(sfm_ba_problem* p, double lambda, int quirks) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
  ba_tick(p, SFM_K_SOLVE, true, s);
  const int nbk = (d.P + NB - 1) / NB;
  for (int j = 0; j < nbk; ++j) {
    const int ncol = nbk - j + 1;                        // column role: block rows j .. nbk (nbk = the rhs row)
    // trailing role (from the second step on): 64x64 super-tiles over block rows j+1 .. nbk x block columns
    // j+1 .. nbk-1, lower part only
    const int srn = j > 0 ? (nbk - j + 1) / 2 : 0;
    ba_chol_step_kernel<<<ncol + srn * (srn + 1) / 2, 256, 0, s>>>(d, j, lambda);
  }
  {
    const size_t ybytes = sizeof(double) * (size_t)d.ld;
    const bool big = d.P > 416;      // more rows above a block than the update waves hold in registers
    if (ybytes <= 48 * 1024) {
      if (big) ba_back_solve_kernel<true, true><<<1, 256, ybytes, s>>>(d, p->cur);
      else ba_back_solve_kernel<true, false><<<1, 256, ybytes, s>>>(d, p->cur);
    } else {
      ba_back_solve_kernel<false, true><<<1, 256, 0, s>>>(d, p->cur);
    }
  }
  ba_tick(p, SFM_K_SOLVE, false, s);
  SFM_HIP(hipGetLastError());
  const int g = pick_group(p);
  const int gpb = 256 / g;
  int grid = std::min((d.N + gpb - 1) / gpb, 4 * ctx().num_cus);
  if (grid < 1) grid = 1;
  const size_t lds = sizeof(double) * (size_t)d.V * (19 + 7);
  ba_tick(p, SFM_K_BACKSUB, true, s);
  if (lds <= 64 * 1024) launch_backsub<true>(p, g, grid, lds, s, lambda, quirks);
  else launch_backsub<false>(p, g, grid, 0, s, lambda, quirks);
  ba_tick(p, SFM_K_BACKSUB, false, s);
  SFM_HIP(hipGetLastError());
  p->red_clean = true;      // ba_backsub_kernel cleared [S | rhs]
  p->cur ^= 1;      // ba_back_solve_kernel prepared the updated cameras into the other slot
  return SFM_OK;
}

// parity hooks (sfm_ba_residual_jacobian / sfm_ba_reduced_system)
void ba_enqueue_residual_jacobian(sfm_ba_problem* p, int quirks, double* r, double* Jp, double* Jx) {
  const BaDev& d = p->dev;
  ba_residual_jacobian_kernel<<<(unsigned)((d.M + 255) / 256), 256, 0, p->stream>>>(d, p->cur, quirks, d.obs_pt, r, Jp, Jx);
}

void ba_enqueue_symmetrize(sfm_ba_problem* p, double lambda, double* S_out, double* rhs_out) {
  const BaDev& d = p->dev;
  ba_symmetrize_kernel<<<(d.P * d.P + 255) / 256, 256, 0, p->stream>>>(d.red, d.ld, d.P, lambda, S_out);
  (void)hipMemcpyAsync(rhs_out, d.red + (size_t)d.ld * d.ld, sizeof(double) * d.P, hipMemcpyDeviceToDevice, p->stream);
}

}  // namespace sfm
