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
#include <vector>

#include "sfm_ba.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void ba_schur_pairs_kernel(BaDev d) {
  extern __shared__ double zl[];            // [max_track][21]
  double* S = d.red;
  const int lane = threadIdx.x;
  for (int p = blockIdx.x; p < d.N; p += gridDim.x) {
    const int beg = d.pt_ptr[p], k = d.pt_ptr[p + 1] - beg;
    for (int t = lane; t < k * 21; t += 64) {      // Z is SoA [21][M]; LDS image is [track slot][21]
      const int e = t / k, a = t - e * k;
      zl[a * 21 + e] = d.Z[(size_t)e * d.M + beg + a];
    }
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
// Dense MFMA product.  Cameras are grouped into row blocks of CB = 18 (126 rows, padded to RB = 128
// = 8 MFMA row strips), so block boundaries never cut a camera.  Grid = (lower-triangular 128x128
// output tiles) x (point chunks).  A workgroup (8 waves) walks its chunk in slabs of SP = 8 points
// (24 k-columns): thread (which, point, camera slot) looks the observation up in the slot table
// (slot_obs[p][cam] = observation index or -1, built at create time) and writes that camera's
// 7x3 block of Z -- or zeros -- straight to its fixed place in the [k][row] LDS image: no zero-fill
// pass, no filtering, coalesced SoA reads of Z.  Each wave owns a 64x32 part (4x2 MFMA tiles):
// 6 ds_read_b64 + 8 v_mfma_f64_16x16x4_f64 per k-step.  A operand: lane l holds A[row = l&15][k = l>>4];
// B operand: B[k = l>>4][col = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.  Waves of a diagonal
// tile that lie entirely above the diagonal skip their MFMAs.  Partial tiles go to per-(chunk, tile)
// slabs (plain stores), summed into S by ba_schur_reduce_kernel.
// ---------------------------------------------------------------------------------------------
typedef double double4_ __attribute__((ext_vector_type(4)));

constexpr int CB = 18;          // cameras per row block
constexpr int RB = 128;         // padded rows per block
constexpr int SP = 8;           // points per LDS slab
constexpr int KSL = 3 * SP;     // k-columns per slab
constexpr int ZLD = RB + 16;    // row pitch = 16 (mod 32) doubles: the k-rows of one ds_read_b64 hit disjoint banks

__global__ __launch_bounds__(512, 4) void ba_schur_mfma_kernel(BaDev d, const int* __restrict__ slot_obs, int vpad,
                                                            double* __restrict__ ws, int pts_per_chunk) {
  __shared__ double za[KSL][ZLD];     // Z^T slab restricted to the tile's row block
  __shared__ double zb[KSL][ZLD];     // ... and to its column block (unused on diagonal tiles)
  int tile = blockIdx.x, ti = 0;
  while (tile >= ti + 1) { tile -= ti + 1; ++ti; }
  const int tj = tile;                   // ti >= tj
  const bool diag = ti == tj;
  const int chunk = blockIdx.y;
  const int p_beg = chunk * pts_per_chunk;
  const int p_end = min(d.N, p_beg + pts_per_chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = (wave >> 2) * 64, wc = (wave & 3) * 32;     // this wave's 64x32 part
  const int lr = lane & 15, lk = lane >> 4;
  const bool active = !(diag && wr == 0 && wc >= 64);
  const size_t M = (size_t)d.M;

  double4_ acc[4][2];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) acc[a][b] = double4_{0, 0, 0, 0};

  for (int t = tid; t < KSL * ZLD; t += 512) {     // padding rows 126,127 (+pitch pad) stay zero for good
    (&za[0][0])[t] = 0.0;
    (&zb[0][0])[t] = 0.0;
  }
  __syncthreads();

  const int ntask = (diag ? 1 : 2) * SP * CB;
  for (int ps = p_beg; ps < p_end; ps += SP) {
    for (int t = tid; t < ntask; t += 512) {
      const int which = t / (SP * CB);
      const int tt = t - which * (SP * CB);
      const int pl = tt / CB, cs = tt - pl * CB;
      const int p = ps + pl;
      const int cam = (which ? tj : ti) * CB + cs;
      const int o = (p < p_end) ? slot_obs[(size_t)p * vpad + cam] : -1;
      double(*dst)[ZLD] = which ? zb : za;
      double z[21];
#pragma unroll
      for (int e = 0; e < 21; ++e) z[e] = (o >= 0) ? d.Z[e * M + o] : 0.0;
#pragma unroll
      for (int e = 0; e < 21; ++e) dst[3 * pl + e % 3][7 * cs + e / 3] = z[e];
    }
    __syncthreads();
    if (active) {
      const double(*zcol)[ZLD] = diag ? za : zb;
#pragma unroll
      for (int k0 = 0; k0 < KSL; k0 += 4) {
        const int kk = k0 + lk;
        double a[4], b[2];
#pragma unroll
        for (int x = 0; x < 4; ++x) a[x] = za[kk][wr + 16 * x + lr];
#pragma unroll
        for (int y = 0; y < 2; ++y) b[y] = zcol[kk][wc + 16 * y + lr];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  double* slab = ws + ((size_t)chunk * gridDim.x + blockIdx.x) * (RB * RB);
#pragma unroll
  for (int x = 0; x < 4; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int row = wr + 16 * x + lk + 4 * r;
        const int col = wc + 16 * y + lr;
        slab[row * RB + col] = acc[x][y][r];
      }
}

// S(lower) -= sum over chunks of the slabs, un-padding block coordinates (block b, row r) -> camera
// b*CB + r/7, parameter r%7.  One thread per padded tile element.
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BaDev d, const double* __restrict__ ws, int ntiles,
                                                              int chunks) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntiles * RB * RB) return;
  int tile = idx / (RB * RB);
  const int e = idx - tile * (RB * RB);
  const int tile_id = tile;
  int ti = 0;
  while (tile >= ti + 1) { tile -= ti + 1; ++ti; }
  const int r = e / RB, c = e - r * RB;
  if (r >= 7 * CB || c >= 7 * CB) return;
  const int cam_r = ti * CB + r / 7, cam_c = tile * CB + c / 7;
  if (cam_r >= d.V || cam_c >= d.V) return;
  const int row = 7 * cam_r + r % 7, col = 7 * cam_c + c % 7;
  if (col > row) return;
  double s = 0;
  for (int k = 0; k < chunks; ++k) s += ws[((size_t)k * ntiles + tile_id) * (RB * RB) + e];
  d.red[(size_t)row * d.ld + col] -= s;
}

static int schur_nblk(const BaDev& d) { return (d.V + CB - 1) / CB; }

// Plan of the MFMA product: chunking of the points, slab workspace and the slot table.
int ba_schur_plan(sfm_ba_problem* p, const int* pt_ptr, const int* cam_idx) {
  const BaDev& d = p->dev;
  const int nblk = schur_nblk(d);
  const int ntiles = nblk * (nblk + 1) / 2;
  // ~2 workgroups of 8 waves per CU; chunks are whole slabs
  int chunks = std::max(1, (2 * ctx().num_cus + ntiles - 1) / ntiles);
  int ppc = (d.N + chunks - 1) / std::max(1, chunks);
  ppc = std::max(SP, ((ppc + SP - 1) / SP) * SP);
  chunks = std::max(1, (d.N + ppc - 1) / ppc);
  p->schur_chunks = chunks;
  p->schur_pts_per_chunk = ppc;
  p->schur_vpad = nblk * CB;
  const size_t ws_bytes = sizeof(double) * (size_t)chunks * ntiles * RB * RB;
  const size_t slot_bytes = sizeof(int) * (size_t)std::max(1, d.N) * p->schur_vpad;
  // the dense path is only ever chosen when it is cheaper than the pair path; do not reserve
  // gigabytes for scenes that will never take it
  p->schur_mfma_ok = ws_bytes + slot_bytes <= ((size_t)4 << 30);
  if (!p->schur_mfma_ok) return SFM_OK;
  SFM_HIP(hipMalloc(&p->schur_ws, ws_bytes));
  SFM_HIP(hipMalloc(reinterpret_cast<void**>(&p->schur_slot), slot_bytes));
  std::vector<int> slot((size_t)std::max(1, d.N) * p->schur_vpad, -1);
  for (int pt = 0; pt < d.N; ++pt)
    for (int o = pt_ptr[pt]; o < pt_ptr[pt + 1]; ++o) slot[(size_t)pt * p->schur_vpad + cam_idx[o]] = o;
  SFM_HIP(hipMemcpy(p->schur_slot, slot.data(), slot_bytes, hipMemcpyHostToDevice));
  return SFM_OK;
}

static bool use_mfma(const sfm_ba_problem* p) {
  if (!p->schur_mfma_ok) return false;
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
    const int nblk = schur_nblk(d);
    const int ntiles = nblk * (nblk + 1) / 2;
    dim3 grid(ntiles, p->schur_chunks);
    double* ws = static_cast<double*>(p->schur_ws);
    ba_schur_mfma_kernel<<<grid, 512, 0, s>>>(d, p->schur_slot, p->schur_vpad, ws, p->schur_pts_per_chunk);
    ba_schur_reduce_kernel<<<(ntiles * RB * RB + 255) / 256, 256, 0, s>>>(d, ws, ntiles, p->schur_chunks);
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
