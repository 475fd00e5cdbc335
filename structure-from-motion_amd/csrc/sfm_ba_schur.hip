// sfm_ba_schur.hip — Schur-complement product  S -= sum_p Z_p Z_p^T  (= B D^-1 B^T of
// ba_processor.py:382, in the symmetric form Z_o = W_o L_p^-T).  Only the lower triangle of S
// (row >= column) is produced; block (c_a, c_b) with c_a > c_b of point p is Z_a Z_b^T.
//
//   ba_schur_pairs_kernel   one wave per point: the point's Z rows staged in LDS, every camera pair
//                           (a >= b) x 49 entries as lane tasks, f64 atomics into S.  Work is
//                           proportional to sum_p k_p (k_p+1)/2 (sparse-optimal); bound by the f64
//                           atomic rate.  Used for small or sparse scenes.
//   ba_schur_mfma_kernel    dense  v_mfma_f64_16x16x4_f64  SYRK over zero-filled LDS tiles of Z^T,
//                           output-stationary 64x64 tiles x split-K chunks of points; no atomics on
//                           the inner loop.  Used when visibility is high enough that dense wins.
#include <algorithm>

#include "sfm_ba.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ba_schur_pairs_kernel(BaDev d) {
  extern __shared__ double zl[];            // [max_track][21]
  double* S = d.red;
  const int lane = threadIdx.x;
  for (int p = blockIdx.x; p < d.N; p += gridDim.x) {
    const int beg = d.pt_ptr[p], k = d.pt_ptr[p + 1] - beg;
    const double* zg = d.Z + (size_t)beg * 21;
    for (int t = lane; t < k * 21; t += 64) zl[t] = zg[t];
    __syncthreads();
    for (int a = 0; a < k; ++a) {
      const int ca = d.cam_idx[beg + a];
      const double* za = zl + a * 21;
      const int ntask = (a + 1) * 49;
      for (int t = lane; t < ntask; t += 64) {
        const int b = t / 49, e = t - b * 49;
        const int i = e / 7, j = e - i * 7;
        if (b == a && j > i) continue;                   // diagonal block: lower part only
        const double* zb = zl + b * 21;
        const double val = za[3 * i] * zb[3 * j] + za[3 * i + 1] * zb[3 * j + 1] + za[3 * i + 2] * zb[3 * j + 2];
        const int cb = d.cam_idx[beg + b];
        atomicAdd(&S[(size_t)(7 * ca + i) * d.ld + 7 * cb + j], -val);
      }
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------
// Dense MFMA product.  Grid = (lower-triangular 64x64 output tiles) x (point chunks).  A workgroup
// (4 waves) walks its chunk of points in slabs of KS point-columns: the slab of Z^T restricted to the
// tile's 64 row-range and 64 column-range is built zero-filled in LDS ([k][row], k = 3*point + j)
// by scattering the compact per-observation Z rows, then each wave issues
// v_mfma_f64_16x16x4_f64 on its 32x32 quarter (2x2 MFMA tiles, 16 accumulator doubles per lane).
// A operand: lane l holds A[row = l&15][k = l>>4]; B operand: B[k = l>>4][col = l&15]; both are the
// same [k][row] LDS image, so one ds_read_b64 per operand per k-step.  C/D: col = l&15,
// row = (l>>4) + 4*reg.  Partial tiles go to per-(chunk, tile) slabs, summed by ba_schur_reduce_kernel.
// ---------------------------------------------------------------------------------------------
typedef double double4_ __attribute__((ext_vector_type(4)));

constexpr int TS = 64;       // output tile edge
constexpr int KP = 16;       // points per LDS slab  -> 48 k-columns
constexpr int ZLD = TS + 16;  // row pitch = 16 (mod 32) doubles: the four k-rows of one ds_read_b64 hit disjoint banks
constexpr int KS = 3 * KP;

__global__ __launch_bounds__(256) void ba_schur_mfma_kernel(BaDev d, double* __restrict__ ws, int pts_per_chunk) {
  __shared__ double za[KS][ZLD];     // Z^T slab restricted to the tile's row range
  __shared__ double zb[KS][ZLD];     // ... and to its column range (unused on diagonal tiles)
  // decode the lower-triangular tile index
  int tile = blockIdx.x, ti = 0;
  while (tile >= ti + 1) { tile -= ti + 1; ++ti; }
  const int tj = tile;                   // ti >= tj
  const int row0 = ti * TS, col0 = tj * TS;
  const bool diag = ti == tj;
  const int chunk = blockIdx.y;
  const int p_beg = chunk * pts_per_chunk;
  const int p_end = min(d.N, p_beg + pts_per_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;     // this wave's 32x32 quarter
  const int lr = lane & 15, lk = lane >> 4;

  double4_ acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_{0, 0, 0, 0};

  for (int ps = p_beg; ps < p_end; ps += KP) {
    const int pe = min(p_end, ps + KP);
    for (int t = tid; t < KS * ZLD; t += 256) {
      (&za[0][0])[t] = 0.0;
      if (!diag) (&zb[0][0])[t] = 0.0;
    }
    __syncthreads();
    const int o_beg = d.pt_ptr[ps], o_end = d.pt_ptr[pe];
    // scatter: one (observation, element) per thread-iteration
    for (int t = o_beg * 21 + tid; t < o_end * 21; t += 256) {
      const int o = t / 21, e = t - o * 21;
      const int r = 7 * d.cam_idx[o] + e / 3;
      const int kk = 3 * (d.obs_pt[o] - ps) + e % 3;
      const double val = d.Z[t];
      if (r >= row0 && r < row0 + TS) za[kk][r - row0] = val;
      if (!diag && r >= col0 && r < col0 + TS) zb[kk][r - col0] = val;
    }
    __syncthreads();
    const int nk = 3 * (pe - ps);
    const double(*zcol)[ZLD] = diag ? za : zb;
    for (int k0 = 0; k0 < nk; k0 += 4) {
      const int kk = k0 + lk;            // k0 + lk < KS always (KS multiple of 4); rows >= nk are zero-filled
      const double a0 = za[kk][wr + lr], a1 = za[kk][wr + 16 + lr];
      const double b0 = zcol[kk][wc + lr], b1 = zcol[kk][wc + 16 + lr];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
  // partial tile -> this (chunk, tile)'s slab; ba_schur_reduce_kernel sums the chunks (no atomics:
  // contended f64 atomics on a 352x352 target run at ~18 G/s on MI355X, plain stores at HBM rate)
  double* slab = ws + ((size_t)chunk * gridDim.x + blockIdx.x) * (TS * TS);
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * a + lk + 4 * r;
        const int col = wc + 16 * b + lr;
        slab[row * TS + col] = acc[a][b][r];
      }
}

// S(lower) -= sum over chunks of the slabs.  One thread per tile element.
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BaDev d, const double* __restrict__ ws, int ntiles,
                                                              int chunks) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntiles * TS * TS) return;
  int tile = idx / (TS * TS);
  const int e = idx - tile * (TS * TS);
  const int tile_id = tile;
  int ti = 0;
  while (tile >= ti + 1) { tile -= ti + 1; ++ti; }
  const int row = ti * TS + e / TS, col = tile * TS + e % TS;
  if (row >= d.P || col > row) return;
  double s = 0;
  for (int c = 0; c < chunks; ++c) s += ws[((size_t)c * ntiles + tile_id) * (TS * TS) + e];
  d.red[(size_t)row * d.ld + col] -= s;
}

int ba_schur_plan(sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  const int ntr = (d.P + TS - 1) / TS;
  const int ntiles = ntr * (ntr + 1) / 2;
  // enough (tile, chunk) workgroups for ~4 per CU, chunks a multiple of the LDS slab
  int chunks = std::max(1, (4 * ctx().num_cus + ntiles - 1) / ntiles);
  int ppc = (d.N + chunks - 1) / std::max(1, chunks);
  ppc = std::max(KP, ((ppc + KP - 1) / KP) * KP);
  chunks = std::max(1, (d.N + ppc - 1) / ppc);
  p->schur_chunks = chunks;
  p->schur_pts_per_chunk = ppc;
  if (p->schur_ws) { (void)hipFree(p->schur_ws); p->schur_ws = nullptr; }
  SFM_HIP(hipMalloc(&p->schur_ws, sizeof(double) * (size_t)chunks * ntiles * TS * TS));
  return SFM_OK;
}

static bool use_mfma(const sfm_ba_problem* p) {
  if (p->schur_mode == SFM_SCHUR_MFMA) return true;
  if (p->schur_mode == SFM_SCHUR_PAIRS) return false;
  const BaDev& d = p->dev;
  if (d.N == 0 || d.M == 0) return false;
  // dense cost ~ (7V)^2/2 * 3N MACs; pair cost ~ 49 * sum k(k+1)/2 atomics (~50x dearer each)
  const double dense = 0.5 * (double)d.P * d.P * 3.0 * d.N;
  const double kbar = (double)d.M / d.N;
  const double pairs = 49.0 * 0.5 * kbar * (kbar + 1) * d.N;
  return dense < 40.0 * pairs;
}

int ba_enqueue_schur(sfm_ba_problem* p, hipStream_t s) {
  const BaDev& d = p->dev;
  if (d.N == 0 || d.M == 0) return SFM_OK;
  if (use_mfma(p)) {
    const int ntr = (d.P + TS - 1) / TS;
    const int ntiles = ntr * (ntr + 1) / 2;
    dim3 grid(ntiles, p->schur_chunks);
    double* ws = static_cast<double*>(p->schur_ws);
    ba_schur_mfma_kernel<<<grid, 256, 0, s>>>(d, ws, p->schur_pts_per_chunk);
    ba_schur_reduce_kernel<<<(ntiles * TS * TS + 255) / 256, 256, 0, s>>>(d, ws, ntiles, p->schur_chunks);
  } else {
    const size_t lds = sizeof(double) * 21 * (size_t)std::max(1, p->max_track);
    if (lds > 64 * 1024) {
      set_error("track of %d observations exceeds the pair kernel's LDS staging", p->max_track);
      return SFM_E_SHAPE;
    }
    const int grid = std::min(d.N, 16 * ctx().num_cus);
    ba_schur_pairs_kernel<<<grid, 64, lds, s>>>(d);
  }
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

}  // namespace sfm
