// sfm_ba_schur.hip — Schur-complement product  S -= sum_p Z_p Z_p^T  (= B D^-1 B^T of
// ba_processor.py:382, in the symmetric form Z_o = W_o L_p^-T).  Only the lower triangle of S
// (row >= column) is produced; block (c_a, c_b) with c_a > c_b of point p is Z_a Z_b^T.
//
//   ba_schur_pairs_kernel   one wave per point: the point's Z rows staged in LDS, every camera pair
//                           (a >= b) x 49 entries as lane tasks, f64 atomics into S.  Work is
//                           proportional to sum_p k_p (k_p+1)/2 (sparse-optimal); bound by the f64
//                           atomic rate.  Used for small or sparse scenes.
//   ba_schur_mfma_kernel    dense  v_mfma_f64_16x16x4_f64  SYRK over LDS images of Z^T that producer waves
//                           re-derive from the 20 B/observation inputs (Z never touches HBM);
//                           output-stationary 128x128 tiles x split-K chunks of points; no atomics.
//                           Used when visibility is high enough that dense wins.
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
// Dense MFMA product, fused with the re-linearisation (no Z in HBM).
//
// Cameras are grouped into row blocks of CB = 18 (126 rows, padded to RB = 128 = 8 MFMA row strips),
// so block boundaries never cut a camera.  Grid = (lower-triangular 128x128 output tiles) x (point
// chunks).  One workgroup = 13 waves with fixed roles:
//   * 5 producer waves: thread (which, point, camera slot) looks its observation up in the slot table
//     (slot_obs[p][cam] = observation index or -1), reads the 16 B key and the point, re-derives
//     Jp, Jx, W = Jp^T Jx and Z = W L_p^-T (L_p^-1 comes from ba_linearize, 48 B/point) and writes the
//     camera's 7x3 block of Z -- or zeros -- to its fixed place in the [k][row] LDS image of the NEXT
//     slab (8 points = 24 k-columns); the loads of the slab after that are already in flight.
//   * 8 consumer waves: v_mfma_f64_16x16x4_f64 on the CURRENT slab.  Off-diagonal tiles: each wave
//     owns 64x32 (4x2 MFMA tiles, 6 ds_read_b64 per 8 MFMAs).  Diagonal tiles: the 36 lower MFMA tiles
//     are dealt round-robin (5,5,5,5,4,4,4,4: every SIMD gets 9), the 28 upper ones are never computed.
// One __syncthreads per slab hands the double-buffered LDS image over.  A operand: lane l holds
// A[row = l&15][k = l>>4]; B operand: B[k = l>>4][col = l&15]; C/D: col = l&15, row = (l>>4) + 4*reg.
// Partial tiles go to per-(chunk, tile) slabs (plain stores), summed into S by ba_schur_reduce_kernel.
// ---------------------------------------------------------------------------------------------
typedef double double4_ __attribute__((ext_vector_type(4)));

constexpr int CB = 18;          // cameras per row block
constexpr int RB = 128;         // padded rows per block
constexpr int SP = 8;           // points per LDS slab
constexpr int KSL = 3 * SP;     // k-columns per slab
constexpr int ZLD = RB + 16;    // row pitch = 16 (mod 32) doubles: the k-rows of one ds_read_b64 hit disjoint banks
constexpr int N_CONS = 8;       // consumer (MFMA) waves
constexpr int N_PROD = 5;       // producer waves: 320 threads >= 2 * SP * CB = 288 staging tasks
constexpr int SCHUR_THREADS = 64 * (N_CONS + N_PROD);
constexpr int STAGE = KSL * ZLD;   // doubles per LDS image

struct SlabIn {     // what one producer task needs for one slab
  int o;            // observation index or -1
  double u, v, X, Y, Z, li[6];
};

// The slot lookup and the loads that depend on it are issued in DIFFERENT slabs (slot two slabs ahead,
// data one slab ahead), so no load latency is ever exposed on the producers' per-slab path.
__device__ __forceinline__ int producer_slot(const BaDev& d, const int* __restrict__ slot_obs, int vpad, int p, int p_end,
                                             int cam) {
  return (p < p_end && cam < d.V) ? slot_obs[(size_t)p * vpad + cam] : -1;
}

__device__ __forceinline__ void producer_fetch(const BaDev& d, int o, int p, SlabIn& in) {
  in.o = o;
  if (o >= 0) {
    in.u = d.u[o]; in.v = d.v[o];
    in.X = d.px[p]; in.Y = d.py[p]; in.Z = d.pz[p];
#pragma unroll
    for (int k = 0; k < 6; ++k) in.li[k] = d.lip[(size_t)p * 6 + k];
  }
}

__device__ __forceinline__ void producer_emit(const SlabIn& in, const double* __restrict__ cam_lds, int quirks,
                                              double* __restrict__ dst /* &image[3*pl][7*cs] */) {
  double z[21];
  if (in.o >= 0) {
    CamPrep c;
    double* cd = reinterpret_cast<double*>(&c);
#pragma unroll
    for (int k = 0; k < 19; ++k) cd[k] = cam_lds[k];
    double pc[3], Jp[14], Jx[6];
    project_cam(c, in.X, in.Y, in.Z, 1.0, pc);
    jac_cam(c, in.X, in.Y, in.Z, pc, quirks, Jp);
    jac_pt_cam(c, pc, Jx);
    // Z = (Jp^T Jx) Li^T = Jp^T (Jx Li^T): 2x3 product first, then 7 rows of 2 FMAs x 3
    double m0[3], m1[3];
    m0[0] = Jx[0] * in.li[0];
    m0[1] = Jx[0] * in.li[1] + Jx[1] * in.li[2];
    m0[2] = Jx[0] * in.li[3] + Jx[1] * in.li[4] + Jx[2] * in.li[5];
    m1[0] = Jx[3] * in.li[0];
    m1[1] = Jx[3] * in.li[1] + Jx[4] * in.li[2];
    m1[2] = Jx[3] * in.li[3] + Jx[4] * in.li[4] + Jx[5] * in.li[5];
#pragma unroll
    for (int i = 0; i < 7; ++i) {
#pragma unroll
      for (int j = 0; j < 3; ++j) z[3 * i + j] = Jp[i] * m0[j] + Jp[7 + i] * m1[j];
    }
  } else {
#pragma unroll
    for (int e = 0; e < 21; ++e) z[e] = 0.0;
  }
#pragma unroll
  for (int e = 0; e < 21; ++e) dst[(e % 3) * ZLD + e / 3] = z[e];
}

// Slab hand-over barrier.  __syncthreads() would also wait for vmcnt(0) and so drain the producers'
// prefetch of the slab after next; here only the LDS traffic has to be complete.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool DIAG>
__device__ __forceinline__ void schur_tile_body(const BaDev& d, int cur, int quirks, const int* __restrict__ slot_obs,
                                                int vpad, double* __restrict__ slab, int ti, int tj, int p_beg,
                                                int p_end, double* __restrict__ img /*[2 stages][2][STAGE]*/,
                                                double* __restrict__ cam_lds /*[2][CB][19]*/, int dbg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool consumer = wave < N_CONS;
  const int lr = lane & 15, lk = lane >> 4;

  // ---- one-time LDS setup: zero both stage images (padding rows stay zero), stage the cameras of both blocks
  for (int t = tid; t < 4 * STAGE; t += SCHUR_THREADS) img[t] = 0.0;
  {
    const double* gprep = reinterpret_cast<const double*>(d.prep[cur]);
    for (int t = tid; t < 2 * CB * 19; t += SCHUR_THREADS) {
      const int which = t / (CB * 19), rem = t - which * (CB * 19);
      const int cam = (which ? tj : ti) * CB + rem / 19;
      cam_lds[t] = cam < d.V ? gprep[(size_t)cam * 19 + rem % 19] : 0.0;
    }
  }
  __syncthreads();

  // Producer and consumer waves run DIFFERENT loops with the same number of barriers, so the register
  // allocator sees max(producer, consumer) pressure instead of their sum.
  if (!consumer) {
    // MFMA-f64 and VALU-f64 share one FP64 pipe per SIMD and the (older) consumer waves always have an
    // MFMA ready: at equal priority the producers only get the pipe once the consumers sit at the slab
    // barrier, which serialises the two phases.  With raised priority the producers' short dependent
    // chains slot in between MFMAs and the slab is ready before the consumers need it.
    __builtin_amdgcn_s_setprio(3);
    constexpr int NTASK = (DIAG ? 1 : 2) * SP * CB;
    const int ptid = tid - 64 * N_CONS;
    const bool has_task = ptid < NTASK;
    const int which = has_task ? ptid / (SP * CB) : 0;
    const int tt = ptid - which * (SP * CB);
    const int pl = has_task ? tt / CB : 0, cs = has_task ? tt - pl * CB : 0;
    const int cam = (which ? tj : ti) * CB + cs;
    const double* my_cam = cam_lds + (which * CB + cs) * 19;
    const int dst_off = which * STAGE + 3 * pl * ZLD + 7 * cs;
    SlabIn in;
    in.o = -1;
    int o_next = -1;
    // prologue: image 0 <- slab 0; data of slab 1 and slot of slab 2 go in flight
    if (has_task) {
      producer_fetch(d, producer_slot(d, slot_obs, vpad, p_beg + pl, p_end, cam), p_beg + pl, in);
      o_next = producer_slot(d, slot_obs, vpad, p_beg + SP + pl, p_end, cam);
      producer_emit(in, my_cam, quirks, img + dst_off);
      producer_fetch(d, o_next, p_beg + SP + pl, in);
      o_next = producer_slot(d, slot_obs, vpad, p_beg + 2 * SP + pl, p_end, cam);
    }
    lds_barrier();
    int stage = 0;
    for (int ps = p_beg; ps < p_end; ps += SP, stage ^= 1) {
      if (has_task && ps + SP < p_end) {
        // image stage^1 <- slab ps+SP (data loaded one slab ago); data of slab ps+2SP (its slot was
        // loaded one slab ago) and slot of slab ps+3SP go in flight
        if (!(dbg & 2)) producer_emit(in, my_cam, quirks, img + (stage ^ 1) * 2 * STAGE + dst_off);
        if (!(dbg & 4)) {
          producer_fetch(d, o_next, ps + 2 * SP + pl, in);
          o_next = producer_slot(d, slot_obs, vpad, ps + 3 * SP + pl, p_end, cam);
        }
      }
      lds_barrier();
    }
    return;
  }

  // ---- consumer waves
  double4_ acc[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) acc[s] = double4_{0, 0, 0, 0};
  int sx[5] = {0, 0, 0, 0, 0}, sy[5] = {0, 0, 0, 0, 0};
  int nsub = 0;
  if (DIAG) {
    for (int s = 0; s < 5; ++s) {
      const int idx = wave + N_CONS * s;
      if (idx < 36) {
        int x = 0;
        while ((x + 1) * (x + 2) / 2 <= idx) ++x;
        sx[s] = __builtin_amdgcn_readfirstlane(x);
        sy[s] = __builtin_amdgcn_readfirstlane(idx - x * (x + 1) / 2);
        nsub = s + 1;
      }
    }
    nsub = __builtin_amdgcn_readfirstlane(nsub);
  }
  const int wr = (wave >> 2) * 64, wc = (wave & 3) * 32;     // off-diagonal tiles: this wave's 64x32 part
  lds_barrier();                                           // prologue barrier (image 0 ready)
  int stage = 0;
  for (int ps = p_beg; ps < p_end; ps += SP, stage ^= 1) {
    const double* za = img + stage * 2 * STAGE;
    const double* zc = DIAG ? za : za + STAGE;
    if (!(dbg & 1))
#pragma unroll
    for (int k0 = 0; k0 < KSL; k0 += 4) {
      const double* ra = za + (k0 + lk) * ZLD + lr;
      const double* rc = zc + (k0 + lk) * ZLD + lr;
      if (DIAG) {
#pragma unroll
        for (int s = 0; s < 5; ++s)
          if (s < nsub) acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(ra[16 * sx[s]], rc[16 * sy[s]], acc[s], 0, 0, 0);
      } else {
        double a[4], b[2];
#pragma unroll
        for (int x = 0; x < 4; ++x) a[x] = ra[wr + 16 * x];
#pragma unroll
        for (int y = 0; y < 2; ++y) b[y] = rc[wc + 16 * y];
#pragma unroll
        for (int x = 0; x < 4; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[2 * x + y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[2 * x + y], 0, 0, 0);
      }
    }
    lds_barrier();
  }
  if (DIAG) {
#pragma unroll
    for (int s = 0; s < 5; ++s)
      if (s < nsub) {
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(16 * sx[s] + lk + 4 * r) * RB + 16 * sy[s] + lr] = acc[s][r];
      }
  } else {
#pragma unroll
    for (int x = 0; x < 4; ++x)
#pragma unroll
      for (int y = 0; y < 2; ++y)
#pragma unroll
        for (int r = 0; r < 4; ++r) slab[(wr + 16 * x + lk + 4 * r) * RB + wc + 16 * y + lr] = acc[2 * x + y][r];
  }
}

// Work split of the MFMA product.  A diagonal tile issues 9 MFMAs per SIMD and k-step, an
// off-diagonal one 16, so diagonal tiles get proportionally longer point chunks: every workgroup then
// carries the same MFMA load and the ~2 x CUs workgroups finish in two even rounds.  Workgroup w:
//   w <  n_off * chunks_off : off-diagonal tile w / chunks_off (pairs ti > tj in row-major order)
//   else                    : diagonal tile (w - n_off * chunks_off) / chunks_diag
// and its partial tile goes to slab w.
struct SchurPlan {
  int nblk, n_off;
  int chunks_off, ppc_off;      // chunks per off-diagonal tile, points per chunk
  int chunks_diag, ppc_diag;
  int dbg;                      // profiling ablations (SFM_OPT_DEBUG): 1 = no MFMA, 2 = no producer math, 4 = no producer loads
};

__global__ __launch_bounds__(SCHUR_THREADS) void ba_schur_mfma_kernel(BaDev d, int cur, int quirks,
                                                                    const int* __restrict__ slot_obs, int vpad,
                                                                    double* __restrict__ ws, SchurPlan plan) {
  extern __shared__ double lds_dyn[];
  double* img = lds_dyn;                        // [2 stages][za, zb][KSL][ZLD]
  double* cam_lds = lds_dyn + 4 * STAGE;        // [2 blocks][CB][19]
  const int w = blockIdx.x;
  const int off_wgs = plan.n_off * plan.chunks_off;
  double* slab = ws + (size_t)w * (RB * RB);
  if (w < off_wgs) {
    int t = w / plan.chunks_off, ti = 1;
    const int chunk = w - t * plan.chunks_off;
    while (t >= ti) { t -= ti; ++ti; }           // t-th pair (ti, tj) with ti > tj
    const int p_beg = chunk * plan.ppc_off;
    const int p_end = min(d.N, p_beg + plan.ppc_off);
    schur_tile_body<false>(d, cur, quirks, slot_obs, vpad, slab, ti, t, p_beg, p_end, img, cam_lds, plan.dbg);
  } else {
    const int w2 = w - off_wgs;
    const int ti = w2 / plan.chunks_diag;
    const int chunk = w2 - ti * plan.chunks_diag;
    const int p_beg = chunk * plan.ppc_diag;
    const int p_end = min(d.N, p_beg + plan.ppc_diag);
    schur_tile_body<true>(d, cur, quirks, slot_obs, vpad, slab, ti, ti, p_beg, p_end, img, cam_lds, plan.dbg);
  }
}

constexpr size_t kSchurLdsBytes = sizeof(double) * (4 * STAGE + 2 * CB * 19);

// S(lower) -= sum over the tile's chunk slabs, un-padding block coordinates (block b, row r) -> camera
// b*CB + r/7, parameter r%7.  One thread per padded tile element; blockIdx.y slices the chunk range
// (more loads in flight), one f64 atomic per slice and element.
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BaDev d, const double* __restrict__ ws, SchurPlan plan, int tile_blocks,
                                                              int lin_rows) {
  if ((int)blockIdx.x >= tile_blocks) {
    // extra blocks: the camera-side sums of ba_linearize (U_c, rhs_c), 12 x gridDim.y slices
    const int cam_blocks = (d.V * 35 + 255) / 256;
    const int e = blockIdx.x - tile_blocks;
    cam_reduce_slice(d, lin_rows, (e % cam_blocks) * 256 + threadIdx.x, (e / cam_blocks) * gridDim.y + blockIdx.y, 12 * gridDim.y);
    return;
  }
  const int ntiles = plan.n_off + plan.nblk;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntiles * RB * RB) return;
  const int tile = idx / (RB * RB);
  const int e = idx - tile * (RB * RB);
  int ti, tj, first, chunks;
  if (tile < plan.n_off) {
    int t = tile;
    ti = 1;
    while (t >= ti) { t -= ti; ++ti; }
    tj = t;
    first = tile * plan.chunks_off;
    chunks = plan.chunks_off;
  } else {
    ti = tj = tile - plan.n_off;
    first = plan.n_off * plan.chunks_off + ti * plan.chunks_diag;
    chunks = plan.chunks_diag;
  }
  const int r = e / RB, c = e - r * RB;
  if (r >= 7 * CB || c >= 7 * CB) return;
  const int cam_r = ti * CB + r / 7, cam_c = tj * CB + c / 7;
  if (cam_r >= d.V || cam_c >= d.V) return;
  const int row = 7 * cam_r + r % 7, col = 7 * cam_c + c % 7;
  if (col > row) return;
  const int per = (chunks + gridDim.y - 1) / gridDim.y;
  const int k0 = blockIdx.y * per, k1 = min(chunks, k0 + per);
  double s = 0;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) s += ws[(size_t)(first + k) * (RB * RB) + e];
  if (s != 0.0) atomicAdd(&d.red[(size_t)row * d.ld + col], -s);
}

static SchurPlan make_plan(const BaDev& d) {
  SchurPlan pl;
  pl.dbg = 0;
  pl.nblk = (d.V + CB - 1) / CB;
  pl.n_off = pl.nblk * (pl.nblk - 1) / 2;
  const int slabs = std::max(1, (d.N + SP - 1) / SP);
  // ONE workgroup per CU in total (each needs 116 KB of LDS, so a CU hosts one at a time): a single even
  // round pays the per-workgroup setup / prologue / slab write once and halves the split-K slab traffic
  // compared with two rounds.  chunks_diag : chunks_off = 9 : 16; shrink until everything fits one round.
  auto fit = [&](double want, int& chunks, int& ppc) {
    int c = std::max(1, std::min(slabs, (int)(want + 0.5)));
    const int slabs_per = (slabs + c - 1) / c;
    ppc = slabs_per * SP;
    chunks = (slabs + slabs_per - 1) / slabs_per;
  };
  double target = (double)ctx().num_cus;
  for (int attempt = 0; attempt < 64; ++attempt) {
    const double a = target / (pl.n_off + (9.0 / 16.0) * pl.nblk);
    fit(a, pl.chunks_off, pl.ppc_off);
    fit(a * 9.0 / 16.0, pl.chunks_diag, pl.ppc_diag);
    if (pl.n_off * pl.chunks_off + pl.nblk * pl.chunks_diag <= ctx().num_cus || target < 8) break;
    target -= 1.0;
  }
  if (pl.n_off == 0) { pl.chunks_off = 0; pl.ppc_off = SP; }
  return pl;
}

// Plan of the MFMA product: chunking of the points, slab workspace and the slot table.
int ba_schur_plan(sfm_ba_problem* p, const int* pt_ptr, const int* cam_idx) {
  const BaDev& d = p->dev;
  const SchurPlan pl = make_plan(d);
  const int wgs = pl.n_off * pl.chunks_off + pl.nblk * pl.chunks_diag;
  p->schur_vpad = pl.nblk * CB;
  const size_t ws_bytes = sizeof(double) * (size_t)wgs * RB * RB;
  const size_t slot_bytes = sizeof(int) * (size_t)std::max(1, d.N) * p->schur_vpad;
  // the dense path is only ever chosen when it is cheaper than the pair path; do not reserve
  // gigabytes for scenes that will never take it
  p->schur_mfma_ok = ws_bytes + slot_bytes <= ((size_t)4 << 30);
  if (!p->schur_mfma_ok) return SFM_OK;
  SFM_HIP(pool_alloc(&p->schur_ws, ws_bytes));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->schur_slot), slot_bytes));
  std::vector<int> slot((size_t)std::max(1, d.N) * p->schur_vpad, -1);
  for (int pt = 0; pt < d.N; ++pt)
    for (int o = pt_ptr[pt]; o < pt_ptr[pt + 1]; ++o) slot[(size_t)pt * p->schur_vpad + cam_idx[o]] = o;
  SFM_HIP(hipMemcpy(p->schur_slot, slot.data(), slot_bytes, hipMemcpyHostToDevice));
  SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_mfma_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchurLdsBytes));
  return SFM_OK;
}

bool ba_schur_uses_mfma(const sfm_ba_problem* p) {
  if (!p->schur_mfma_ok) return false;
  if (p->schur_mode == SFM_SCHUR_MFMA) return true;
  if (p->schur_mode == SFM_SCHUR_PAIRS) return false;
  const BaDev& d = p->dev;
  if (d.N == 0 || d.M == 0) return false;
  // Measured on MI355X at C3 (profiles/r01p): the dense product retires ~22 T MACs/s of its (7V)^2/2 * 3N
  // MACs, the pair kernel ~0.08 T f64 atomic adds/s of its 49 * sum k(k+1)/2 adds: one add costs ~280 MACs.
  const double dense = 0.5 * (double)d.P * d.P * 3.0 * d.N;
  const double kbar = (double)d.M / d.N;
  const double pairs = 49.0 * 0.5 * kbar * (kbar + 1) * d.N;
  return dense < 250.0 * pairs;
}

int ba_enqueue_schur(sfm_ba_problem* p, hipStream_t s) {
  const BaDev& d = p->dev;
  if (d.N == 0 || d.M == 0) return SFM_OK;
  if (ba_schur_uses_mfma(p)) {
    SchurPlan pl = make_plan(d);
    pl.dbg = p->debug;
    const int wgs = pl.n_off * pl.chunks_off + pl.nblk * pl.chunks_diag;
    const int ntiles = pl.n_off + pl.nblk;
    double* ws = static_cast<double*>(p->schur_ws);
    ba_schur_mfma_kernel<<<wgs, SCHUR_THREADS, kSchurLdsBytes, s>>>(d, p->cur, p->quirks, p->schur_slot, p->schur_vpad, ws, pl);
    const int tile_blocks = (ntiles * RB * RB + 255) / 256;
    const int cam_blocks = p->lin_rows > 0 ? 12 * ((d.V * 35 + 255) / 256) : 0;
    ba_schur_reduce_kernel<<<dim3(tile_blocks + cam_blocks, 4), 256, 0, s>>>(d, ws, pl, tile_blocks, p->lin_rows);
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
