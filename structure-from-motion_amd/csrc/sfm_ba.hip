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
// DENSE_Z: Z_o goes to its 7x3 slot of the dense Zd (MFMA product); otherwise to the AoS Z of the sparse product.
// FUSED (LDS_MODE 2 only): the back substitution of the PREVIOUS iteration runs first, on the same lanes and the
// same 20 bytes per observation: dX_p = V_p^-1 (g_p - sum_o W_o^T dp_c) at the old cameras prep[cur ^ 1] and the old
// point, X += dX (ba:405-406), then the linearisation at the new cameras prep[cur] and the new point -- one pass
// over the observation list per iteration instead of two (ba_backsub + ba_linearize).  The kernel also clears
// [S | rhs], which is dead once the reduced solve has run.
// ---------------------------------------------------------------------------------------------
// THREADS = 64 (one wave per workgroup) is the deterministic variant: the camera accumulators in LDS then receive
// their ds_add_f64 from a single instruction stream, in program order.
template <int G, int LDS_MODE, bool DENSE_Z, bool FUSED, int THREADS = 256>
__global__ __launch_bounds__(THREADS) void ba_linearize_kernel(BaDev d, int cur, double lambda, int quirks) {
  static_assert(!FUSED || LDS_MODE == 2, "the fused kernel keeps both camera sets in LDS");
  extern __shared__ double lds[];
  unsigned long long* stamp = (d.stamps && blockIdx.x == 0 && threadIdx.x == 0) ? d.stamps + 192 : nullptr;
  int sidx = 0;
  if (stamp) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  constexpr bool PREP_LDS = LDS_MODE == 2, ACC_LDS = LDS_MODE >= 1;
  double* lds_prep = lds;                                        // V * 19 (mode 2)
  double* lds_acc = lds + (PREP_LDS ? (size_t)d.V * 19 : 0);     // V * 35
  double* lds_old = lds + (size_t)d.V * (19 + 35);               // V * 19 (FUSED): cameras the last step linearised at
  double* lds_delta = lds_old + (size_t)d.V * 19;                // V * 7  (FUSED): their update
  const CamPrep* gprep = d.prep[cur];
  if (ACC_LDS) {
    if (PREP_LDS) {
      const double* src = reinterpret_cast<const double*>(gprep);
      for (int i = threadIdx.x; i < d.V * 19; i += blockDim.x) lds_prep[i] = src[i];
    }
    if (FUSED) {
      const double* src = reinterpret_cast<const double*>(d.prep[cur ^ 1]);
      for (int i = threadIdx.x; i < d.V * 19; i += blockDim.x) lds_old[i] = src[i];
      for (int i = threadIdx.x; i < d.V * 7; i += blockDim.x) lds_delta[i] = d.delta[i];
    }
    for (int i = threadIdx.x; i < d.V * 35; i += blockDim.x) lds_acc[i] = 0.0;
    __syncthreads();
  }
  if (FUSED) {
    const size_t n_red = red_size(d.nbk);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_red; i += (size_t)gridDim.x * blockDim.x) d.red[i] = 0.0;
  }
  if (stamp) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  constexpr int GPB = THREADS / G;             // point groups per block
  const int lane_g = threadIdx.x % G;
  const int grp = threadIdx.x / G;
  double* S = d.red;
  double* rhs = d.red + red_rhs_off(d.nbk);
  double cost = 0.0;                           // sum |b - f|^2 of this lane's observations (sfm_ba_get_stats)

  for (int p0 = blockIdx.x * GPB; p0 < d.N; p0 += gridDim.x * GPB) {
    const int p = p0 + grp;
    int beg = 0, end = 0;
    double X = 0, Y = 0, Z = 0;
    if (p < d.N) {
      beg = d.pt_ptr[p]; end = d.pt_ptr[p + 1];
      X = d.px[p]; Y = d.py[p]; Z = d.pz[p];
    }
    const bool single = (end - beg) <= G;       // the whole track fits the lane group: every lane keeps its observation
    double r[2], Jp[14], Jx[6];
    int cam = 0;
    double uo = 0, vo = 0;
    const int o1 = beg + lane_g;
    if (single && o1 < end) { cam = d.cam_idx[o1]; uo = d.u[o1]; vo = d.v[o1]; }
    if (FUSED) {
      double a[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // v6 | (g - W^T dp)
      for (int o = o1; o < end; o += G) {
        if (!single) { cam = d.cam_idx[o]; uo = d.u[o]; vo = d.v[o]; }
        CamPrep c;
        load_cam(c, reinterpret_cast<const CamPrep*>(lds_old) + cam);
        obs_terms(c, X, Y, Z, uo, vo, quirks, r, Jp, Jx);
        const double* dp = lds_delta + 7 * cam;
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
      a[0] += lambda; a[2] += lambda; a[5] += lambda;
      double li[6];
      chol3_inv_fast(a, li);
      const double y0 = li[0] * a[6];
      const double y1 = li[1] * a[6] + li[2] * a[7];
      const double y2 = li[3] * a[6] + li[4] * a[7] + li[5] * a[8];
      X += li[0] * y0 + li[1] * y1 + li[3] * y2;     // L^-T y, the same expression on every lane of the group
      Y += li[2] * y1 + li[4] * y2;
      Z += li[5] * y2;
      if (lane_g == 0 && p < d.N) { d.px[p] = X; d.py[p] = Y; d.pz[p] = Z; }
    }
    double v6[6] = {0, 0, 0, 0, 0, 0}, g3[3] = {0, 0, 0};
    for (int o = o1; o < end; o += G) {
      if (!single) { cam = d.cam_idx[o]; uo = d.u[o]; vo = d.v[o]; }
      CamPrep c;
      load_cam(c, PREP_LDS ? reinterpret_cast<const CamPrep*>(lds_prep) + cam : gprep + cam);
      obs_terms(c, X, Y, Z, uo, vo, quirks, r, Jp, Jx);
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
    for (int o = o1; o < end; o += G) {
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
          double* zr = d.Zd + (size_t)(3 * p) * d.zp + 7 * cam;
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
      cost += r[0] * r[0] + r[1] * r[1];
      // rhs_c -= W V^-1 g = Jp^T (Jx h) with h = V^-1 g: folded into the residual, e = r - Jx h
      const double e0 = r[0] - (Jx[0] * h0 + Jx[1] * h1 + Jx[2] * h2);
      const double e1 = r[1] - (Jx[3] * h0 + Jx[4] * h1 + Jx[5] * h2);
      int k = 0;
      if (ACC_LDS) {
        // row by row, every row's products handed to the LDS atomics before the next row is formed: all 35 values at once
        // kept 70 VGPRs live across the 35 ds_add_f64 (168 VGPRs, three waves per SIMD)
        double* a = lds_acc + (size_t)cam * 35;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) { atomicAdd(a + k, Jp[i] * Jp[j] + Jp[7 + i] * Jp[7 + j]); ++k; }
          atomicAdd(a + 28 + i, Jp[i] * e0 + Jp[7 + i] * e1);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
#pragma unroll
          for (int j = 0; j <= i; ++j) { acc[k] = Jp[i] * Jp[j] + Jp[7 + i] * Jp[7 + j]; ++k; }
          acc[28 + i] = Jp[i] * e0 + Jp[7 + i] * e1;
        }
        k = 0;
        for (int i = 0; i < 7; ++i)
          for (int j = 0; j <= i; ++j) { atomicAdd(&S[red_index(7 * cam + i, 7 * cam + j)], acc[k]); ++k; }
        for (int i = 0; i < 7; ++i) atomicAdd(&rhs[7 * cam + i], acc[28 + i]);
      }
    }
  }
  if (stamp && sidx < 62) stamp[sidx++] = __builtin_amdgcn_s_memtime();
  {
    // one partial cost per workgroup, plain store; ba_schur_reduce adds them up (3000 waves adding to ONE address
    // with global atomics cost 18 us at C3: same-address atomics serialise at the memory side)
    __shared__ double wcost[THREADS / 64];
    cost = wave_sum(cost);
    if ((threadIdx.x & 63) == 0) wcost[threadIdx.x >> 6] = cost;
    __syncthreads();
    if (threadIdx.x == 0) {
      double t = 0;
#pragma unroll
      for (int w = 0; w < THREADS / 64; ++w) t += wcost[w];
      d.cost_ws[blockIdx.x] = t;
    }
  }
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
    const size_t n_red = red_size(d.nbk);
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
// host side
// ---------------------------------------------------------------------------------------------
static int pick_group(const sfm_ba_problem* p) {
  // lanes per point: smallest power of two >= mean track length (clamped to [4, 64]); longer tracks loop
  double mean = p->dev.N > 0 ? (double)p->dev.M / p->dev.N : 1.0;
  int g = 4;
  while (g < 64 && g < mean) g <<= 1;
  return g;
}

// `cur` = prep slot of the cameras to linearise at (FUSED: the back substitution uses the other slot)
template <int LDS, bool WZ, bool FUSED = false>
static void launch_linearize(const sfm_ba_problem* p, int cur, int g, int grid, size_t lds, hipStream_t s, double lambda, int quirks) {
  const BaDev& d = p->dev;
  if (p->deterministic) {      // one wave per workgroup (ordered LDS accumulation)
    switch (g) {
      case 4: ba_linearize_kernel<4, LDS, WZ, FUSED, 64><<<grid, 64, lds, s>>>(d, cur, lambda, quirks); break;
      case 8: ba_linearize_kernel<8, LDS, WZ, FUSED, 64><<<grid, 64, lds, s>>>(d, cur, lambda, quirks); break;
      case 16: ba_linearize_kernel<16, LDS, WZ, FUSED, 64><<<grid, 64, lds, s>>>(d, cur, lambda, quirks); break;
      case 32: ba_linearize_kernel<32, LDS, WZ, FUSED, 64><<<grid, 64, lds, s>>>(d, cur, lambda, quirks); break;
      default: ba_linearize_kernel<64, LDS, WZ, FUSED, 64><<<grid, 64, lds, s>>>(d, cur, lambda, quirks); break;
    }
    return;
  }
  switch (g) {
    case 4: ba_linearize_kernel<4, LDS, WZ, FUSED><<<grid, 256, lds, s>>>(d, cur, lambda, quirks); break;
    case 8: ba_linearize_kernel<8, LDS, WZ, FUSED><<<grid, 256, lds, s>>>(d, cur, lambda, quirks); break;
    case 16: ba_linearize_kernel<16, LDS, WZ, FUSED><<<grid, 256, lds, s>>>(d, cur, lambda, quirks); break;
    case 32: ba_linearize_kernel<32, LDS, WZ, FUSED><<<grid, 256, lds, s>>>(d, cur, lambda, quirks); break;
    default: ba_linearize_kernel<64, LDS, WZ, FUSED><<<grid, 256, lds, s>>>(d, cur, lambda, quirks); break;
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
    t.open = (t.calls++ % p->timing_stride) == 0;
    if (!t.open) return;
    if (t.used == (int)t.ev.size()) {
      hipEvent_t a, b;
      (void)hipEventCreate(&a); (void)hipEventCreate(&b);
      t.ev.push_back({a, b});
    }
    (void)hipEventRecord(t.ev[t.used].first, s);
  } else if (t.open) {
    (void)hipEventRecord(t.ev[t.used].second, s);
    t.used++;
    t.open = false;
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

// The fused kernel keeps both camera sets, the camera update and the camera accumulators in LDS: 80 doubles per
// camera (V <= 102 within the 64 KB the other kernels of the iteration leave room for).
bool ba_can_fuse(const sfm_ba_problem* p) {
  return !(p->debug & 16) && sizeof(double) * (size_t)p->dev.V * (19 + 35 + 19 + 7) <= 64 * 1024;
}

static int enqueue_backsub(sfm_ba_problem* p, double lambda, int quirks) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
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
  p->cur ^= 1;              // ba_back_solve_kernel prepared the updated cameras into the other slot
  return SFM_OK;
}

// The back substitution of an iteration is DEFERRED when the fused kernel can take it (ba_can_fuse): if the next
// thing the caller does is linearise again with the same lambda / quirks -- sfm_ba_iterate, or the multi-GPU loop
// linearize_reduce / all-reduce / solve_update -- it rides in that launch (one pass over the observations per
// iteration instead of two).  Anything else that looks at or replaces the state first completes it with the
// stand-alone kernel (ba_flush; also sfm_ba_flush, which sfm_ba_iterate calls before it returns so that every
// call enqueues complete iterations).
int ba_flush(sfm_ba_problem* p) {
  if (!p->backsub_pending) return SFM_OK;
  p->backsub_pending = false;
  return enqueue_backsub(p, p->pending_lambda, p->pending_quirks);
}

int ba_enqueue_linearize_reduce(sfm_ba_problem* p, double lambda, int quirks, bool allow_defer) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
  bool fused = p->backsub_pending && ba_can_fuse(p) && lambda == p->pending_lambda && quirks == p->pending_quirks;
  if (!fused) {
    SFM_TRY(ba_flush(p));
    if (!p->prep_valid) SFM_TRY(ba_enqueue_prep(p));
    if (!p->red_clean) SFM_HIP(hipMemsetAsync(d.red, 0, sizeof(double) * red_size(d.nbk), s));
  }
  p->red_clean = false;
  const int g = pick_group(p);
  const int gpb = (p->deterministic ? 64 : 256) / g;
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
  if (fused) {
    const size_t lds_f = sizeof(double) * (size_t)d.V * (19 + 35 + 19 + 7);
    if (dense_z) launch_linearize<2, true, true>(p, p->cur ^ 1, g, grid, lds_f, s, lambda, quirks);
    else launch_linearize<2, false, true>(p, p->cur ^ 1, g, grid, lds_f, s, lambda, quirks);
    p->backsub_pending = false;
    p->cur ^= 1;            // ba_back_solve_kernel prepared the updated cameras into the other slot
  } else if (mode == 2) {
    if (dense_z) launch_linearize<2, true>(p, p->cur, g, grid, lds, s, lambda, quirks);
    else launch_linearize<2, false>(p, p->cur, g, grid, lds, s, lambda, quirks);
  } else if (mode == 1) {
    if (dense_z) launch_linearize<1, true>(p, p->cur, g, grid, lds_acc, s, lambda, quirks);
    else launch_linearize<1, false>(p, p->cur, g, grid, lds_acc, s, lambda, quirks);
  } else {
    if (dense_z) launch_linearize<0, true>(p, p->cur, g, grid, 0, s, lambda, quirks);
    else launch_linearize<0, false>(p, p->cur, g, grid, 0, s, lambda, quirks);
  }
  p->lin_rows = mode >= 1 ? grid : 0;
  p->lin_grid = grid;
  ba_tick(p, SFM_K_LINEARIZE, false, s);
  SFM_HIP(hipGetLastError());
  SFM_TRY(ba_enqueue_schur(p, s, allow_defer));
  return SFM_OK;
}

int ba_enqueue_solve_update(sfm_ba_problem* p, double lambda, int quirks) {
  hipStream_t s = p->stream;
  SFM_TRY(ba_flush(p));           // a caller that solves twice without linearising in between
  ba_tick(p, SFM_K_SOLVE, true, s);
  SFM_TRY(ba_enqueue_reduced_solve(p, lambda));
  ba_tick(p, SFM_K_SOLVE, false, s);
  SFM_HIP(hipGetLastError());
  if (ba_can_fuse(p)) {
    p->backsub_pending = true;
    p->pending_lambda = lambda;
    p->pending_quirks = quirks;
    return SFM_OK;
  }
  return enqueue_backsub(p, lambda, quirks);
}

void ba_graph_drop(sfm_ba_problem* p) {
  for (auto& g : p->body_graph) {
    if (g) (void)hipGraphExecDestroy(g);
    g = nullptr;
  }
}

// One steady-state iteration (the previous back substitution pending, so the linearisation is the fused kernel) as a
// hipGraph: captured from the very enqueue functions the eager path uses -- which also advance the host-side state
// exactly as an eager iteration does -- then launched once to carry that iteration out.
static int capture_body(sfm_ba_problem* p, double lambda, int quirks) {
  hipStream_t s = p->stream;
  const int slot = p->cur;
  SFM_HIP(hipStreamBeginCapture(s, hipStreamCaptureModeRelaxed));
  int st = ba_enqueue_linearize_reduce(p, lambda, quirks, true);      // graphs are used without a communicator only
  if (st == SFM_OK) st = ba_enqueue_solve_update(p, lambda, quirks);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (st != SFM_OK) { if (graph) (void)hipGraphDestroy(graph); return st; }
  SFM_HIP(e);
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  SFM_HIP(ei);
  p->body_graph[slot] = exec;
  SFM_HIP(hipGraphLaunch(exec, s));
  return SFM_OK;
}

// `iters` complete iterations: between two of them the back substitution rides in the next linearisation's launch,
// the last one is flushed before returning.  With SFM_OPT_GRAPH the iterations after the first are graph replays
// (two graphs: the camera slots alternate).
int ba_enqueue_iterations(sfm_ba_problem* p, double lambda, int iters, int quirks) {
  const bool graphs = p->use_graph && p->timing == 0 && p->stream != nullptr && ba_can_fuse(p) && p->comm == nullptr;
  if (graphs && (lambda != p->graph_lambda || quirks != p->graph_quirks)) {
    ba_graph_drop(p);
    p->graph_lambda = lambda;
    p->graph_quirks = quirks;
  }
  for (int it = 0; it < iters; ++it) {
    const bool steady = p->backsub_pending && lambda == p->pending_lambda && quirks == p->pending_quirks;
    if (graphs && steady) {
      if (p->body_graph[p->cur] == nullptr) {
        SFM_TRY(capture_body(p, lambda, quirks));
      } else {
        SFM_HIP(hipGraphLaunch(p->body_graph[p->cur], p->stream));
        p->cur ^= 1;              // what the captured enqueue calls did to the host-side state: the fused launch
        p->red_clean = false;     // switched the camera slot; the back substitution is pending again
        ++p->graph_replays;
      }
      continue;
    }
    SFM_TRY(ba_enqueue_linearize_reduce(p, lambda, quirks, p->comm == nullptr));      // nothing but the solve reads S: its reduce may ride in the solve's launch
    // sharded loop inside the library: this rank's partial [S | rhs] -> the sum over all ranks, on the problem's stream
    if (p->comm) SFM_TRY(comm_all_reduce_f64(p->comm, p->dev.red, red_size(p->dev.nbk), p->stream));
    SFM_TRY(ba_enqueue_solve_update(p, lambda, quirks));
  }
  return ba_flush(p);
}

// parity hooks (sfm_ba_residual_jacobian / sfm_ba_reduced_system)
void ba_enqueue_residual_jacobian(sfm_ba_problem* p, int quirks, double* r, double* Jp, double* Jx) {
  const BaDev& d = p->dev;
  ba_residual_jacobian_kernel<<<(unsigned)((d.M + 255) / 256), 256, 0, p->stream>>>(d, p->cur, quirks, d.obs_pt, r, Jp, Jx);
}

}  // namespace sfm
