// sfm_ba_schur_rows.hip — sparse Schur-complement product, row-panel form (gfx950):
//   S(lower) -= sum_p Z_p Z_p^T     (= B D^-1 B^T of ba_processor.py:382, Z_o = W_o L_p^-T)
// for low visibility and small scenes, where the dense MFMA product (sfm_ba_schur.hip) multiplies mostly zeros.
//
// The unit of work is one OBSERVATION a = (camera c, point p): it owns block row c of the contribution of p,
//   S[c][c'] -= Z_a Z_b^T   for every observation b of p with camera c' <= c,
// and the observations of a point are sorted by camera, so those b are the contiguous range [pt_ptr[p], a] -- no
// search, no empty visits.  A workgroup owns a group of R cameras (R = as many block rows of S as fit in LDS: 7R rows
// x 7V columns of doubles, R = 2 at V = 200, 7 at V = 50) and a chunk of that group's observations (from the
// camera-major observation list built once on the device, ba_cam_major_*); its 16 waves take one observation each:
// the 21 values of Z_a arrive through the scalar data cache as SGPR operands, lanes = (b, column j) hold Z_b -- nine
// B-observations per lane round, and a point seen by 30 cameras fills most of the 64 lanes where the 18x18-camera
// tiles of the first sparse kernel kept 19 busy -- and every product is one ds_add_f64 into the panel.  Panels go to
// split-K slabs (plain stores) and ba_schur_rows_reduce adds them into the packed S.
#include <algorithm>
#include <vector>

#include "sfm_ba.h"

namespace sfm {

constexpr int ROWS_WAVES = 16;
constexpr int ROWS_THREADS = 64 * ROWS_WAVES;
constexpr size_t kRowsLdsBudget = 156 * 1024;

// ---- camera-major observation list (static structure, built at create / append) ------------------------------
__global__ void ba_cam_major_count_kernel(long long M, const int* __restrict__ cam_idx, const int* __restrict__ obs_pt,
                                          const int* __restrict__ pt_ptr, int* __restrict__ cnt,
                                          unsigned long long* __restrict__ pairs) {
  const long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (o >= M) return;
  const int c = cam_idx[o];
  atomicAdd(&cnt[c], 1);
  atomicAdd(&pairs[c], (unsigned long long)(o - pt_ptr[obs_pt[o]] + 1));      // camera pairs this observation owns
}

__global__ __launch_bounds__(1024) void ba_cam_major_scan_kernel(int V, const int* __restrict__ cnt, int* __restrict__ cam_ptr) {
  __shared__ int wsum[16];
  __shared__ int carry;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < V; base += 1024) {
    const int q = base + tid;
    const int a = q < V ? cnt[q] : 0;
    int sa = a;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(sa, off, 64); if (lane >= off) sa += t; }
    if (lane == 63) wsum[wave] = sa;
    __syncthreads();
    int o = carry;
    for (int w = 0; w < wave; ++w) o += wsum[w];
    if (q < V) cam_ptr[q] = o + sa - a;
    __syncthreads();
    if (tid == 1023) carry = o + sa;
    __syncthreads();
  }
  if (tid == 0) cam_ptr[V] = carry;
}

// entry of the camera-major list: everything a visit needs about its A-observation in ONE 16-byte load (the chain
// cam_obs -> obs_pt -> pt_ptr of dependent loads cost three L2 round trips per visit: 410 us at the C4 share)
__global__ void ba_cam_major_fill_kernel(long long M, const int* __restrict__ cam_idx, const int* __restrict__ obs_pt,
                                         const int* __restrict__ pt_ptr, const int* __restrict__ cam_ptr,
                                         int* __restrict__ fill, int4* __restrict__ cam_ent) {
  const long long o = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (o >= M) return;
  const int c = cam_idx[o];
  const int b0 = pt_ptr[obs_pt[o]];
  cam_ent[cam_ptr[c] + atomicAdd(&fill[c], 1)] = int4{(int)o, b0, c, (int)o - b0 + 1};      // observation, first of its track, camera, k_B
}

// ---- the product ---------------------------------------------------------------------------------------------
struct RowsWg { int group, e_beg, e_end, pad; };

__global__ __launch_bounds__(ROWS_THREADS) void ba_schur_rows_kernel(BaDev d, const int4* __restrict__ cam_ent,
                                                                    const RowsWg* __restrict__ table, double* __restrict__ ws,
                                                                    int R, int tpr, int tpl, int cp) {
  extern __shared__ double panel[];          // [7 R][tpl]: camera c' at column cp * c' (cp = 7, or 8: fewer LDS bank conflicts)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const RowsWg wg = table[blockIdx.x];
  const int gbase = wg.group * R;
  const int nrows = 7 * min(R, d.V - gbase);
  const int ncols = 7 * min(d.V, gbase + R);           // columns beyond the group's last camera are never touched
  for (int t = tid; t < nrows * tpl; t += ROWS_THREADS) panel[t] = 0.0;
  __syncthreads();
  const double* __restrict__ Z = d.Z;
  typedef const __attribute__((address_space(4))) double ConstF64;
  ConstF64* Zc = (ConstF64*)d.Z;
  const int lb = lane / 7, lj = lane - 7 * lb;          // lanes 0..62: B-observation slot, column; lane 63 idles
  const bool slot_ok = lane < 63;
  // one visit = one A-observation; its entry is fetched two visits ahead, the first lane round of the B side one
  struct Visit { int oa, b0, kB, rowoff; double zb0, zb1, zb2; int col; };
  auto fetch_entry = [&](int e) -> int4 { return e < wg.e_end ? cam_ent[e] : int4{0, 0, 0, 0}; };
  auto load_meta = [&](const int4& en, Visit& v) {
    v.oa = __builtin_amdgcn_readfirstlane(en.x);
    v.b0 = __builtin_amdgcn_readfirstlane(en.y);
    v.kB = __builtin_amdgcn_readfirstlane(en.w);
    v.rowoff = 7 * (__builtin_amdgcn_readfirstlane(en.z) - gbase) * tpl;
  };
  auto load_round = [&](const Visit& v, int bb, double& z0, double& z1, double& z2, int& col) {
    const bool on = slot_ok && bb + lb < v.kB;
    const int ob = v.b0 + bb + (on ? lb : 0);
    const double* q = Z + (size_t)ob * 21 + 3 * lj;
    z0 = q[0]; z1 = q[1]; z2 = q[2];
    col = on ? cp * d.cam_idx[ob] + lj : -1;
  };
  Visit cur, nxt;
  int e = wg.e_beg + wave;
  int4 en1 = fetch_entry(e + ROWS_WAVES);
  if (e < wg.e_end) { load_meta(fetch_entry(e), cur); load_round(cur, 0, cur.zb0, cur.zb1, cur.zb2, cur.col); }
  for (; e < wg.e_end; e += ROWS_WAVES) {
    const bool more = e + ROWS_WAVES < wg.e_end;
    const int4 en2 = fetch_entry(e + 2 * ROWS_WAVES);
    if (more) { load_meta(en1, nxt); load_round(nxt, 0, nxt.zb0, nxt.zb1, nxt.zb2, nxt.col); }
    en1 = en2;
    ConstF64* za = Zc + (size_t)cur.oa * 21;             // wave-uniform: SGPR operands of the FMAs below
    double* prow = panel + cur.rowoff;
    double z0 = cur.zb0, z1 = cur.zb1, z2 = cur.zb2;
    int col = cur.col;
    for (int bb = 0; bb < cur.kB; bb += 9) {
      double n0 = 0, n1 = 0, n2 = 0;
      int ncol = -1;
      if (bb + 9 < cur.kB) load_round(cur, bb + 9, n0, n1, n2, ncol);     // the next lane round's loads go first
      if (col >= 0) {
        double* pc = prow + col;
#pragma unroll
        for (int i = 0; i < 7; ++i) atomicAdd(pc + i * tpl, za[3 * i] * z0 + za[3 * i + 1] * z1 + za[3 * i + 2] * z2);
      }
      z0 = n0; z1 = n1; z2 = n2; col = ncol;
    }
    if (more) cur = nxt;
  }
  __syncthreads();
  double* slab = ws + (size_t)blockIdx.x * ((size_t)7 * R * tpr);
  for (int t = tid; t < nrows * ncols; t += ROWS_THREADS) {
    const int r = t / ncols, c = t - r * ncols;
    const int cam = c / 7;
    slab[(size_t)r * tpr + c] = panel[r * tpl + cp * cam + (c - 7 * cam)];
  }
}

// S(lower) -= sum over a group's chunk slabs.  Thread per (row of S, column): coalesced along the column.
__global__ __launch_bounds__(256) void ba_schur_rows_reduce_kernel(BaDev d, const double* __restrict__ ws, const int* __restrict__ group_first,
                                                                   int R, int tpr, int lin_rows, int lin_grid) {
  const int row = blockIdx.y;                            // 0 .. 7V-1; beyond: the camera-side sums of ba_linearize and its cost
  if (row >= d.P) {
    // what ba_schur_reduce's extra blocks do, in this launch (the two reduces overlap instead of following each other): block e of
    // 12 x cam_blocks sums slice (e / cam_blocks, blockIdx.z) of 12 x gridDim.z of the accumulator rows for 256 accumulators
    const int cam_blocks = (d.V * 35 + 255) / 256;
    const int e = (row - d.P) * gridDim.x + blockIdx.x;
    if (lin_rows > 0 && e < 12 * cam_blocks) cam_reduce_slice(d, lin_rows, (e % cam_blocks) * 256 + threadIdx.x, (e / cam_blocks) * gridDim.z + blockIdx.z, 12 * gridDim.z);
    else if (e == (lin_rows > 0 ? 12 * cam_blocks : 0) && blockIdx.z == 0 && threadIdx.x < 64) cost_reduce(d, lin_grid);
    return;
  }
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col > row) return;
  const int g = (row / 7) / R;
  const int r = row - 7 * g * R;
  const int first = group_first[g], last = group_first[g + 1];
  const int per = (last - first + gridDim.z - 1) / gridDim.z;      // blockIdx.z slices the chunk range (more loads in flight)
  const int w0 = first + blockIdx.z * per, w1 = min(last, w0 + per);
  const size_t slab = (size_t)7 * R * tpr;
  double s = 0;
  for (int w = w0; w < w1; ++w) s += ws[(size_t)w * slab + (size_t)r * tpr + col];
  if (s != 0.0) atomicAdd(&d.red[red_index(row, col)], -s);
}

// ---- host side -----------------------------------------------------------------------------------------------
// Camera-major list of the problem's observations (cam_ptr / cam_obs on the device, per-camera counts and pair
// counts on the host for the work split).  Enqueued behind the structure kernel; ba_rows_finish reads it back.
int ba_rows_enqueue_build(sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  hipStream_t s = p->stream;
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->cam_ptr), sizeof(int) * ((size_t)d.V + 1)));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->cam_ent), sizeof(int4) * std::max<size_t>(1, (size_t)d.M)));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->cam_pairs), sizeof(unsigned long long) * (size_t)d.V));
  DevBuf<int> cnt, fill;
  SFM_TRY(cnt.alloc((size_t)d.V, s)); SFM_TRY(fill.alloc((size_t)d.V, s));
  SFM_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int) * d.V, s));
  SFM_HIP(hipMemsetAsync(fill.p, 0, sizeof(int) * d.V, s));
  SFM_HIP(hipMemsetAsync(p->cam_pairs, 0, sizeof(unsigned long long) * d.V, s));
  if (d.M > 0) ba_cam_major_count_kernel<<<(unsigned)((d.M + 255) / 256), 256, 0, s>>>(d.M, d.cam_idx, d.obs_pt, d.pt_ptr, cnt.p, p->cam_pairs);
  ba_cam_major_scan_kernel<<<1, 1024, 0, s>>>(d.V, cnt.p, p->cam_ptr);
  if (d.M > 0) ba_cam_major_fill_kernel<<<(unsigned)((d.M + 255) / 256), 256, 0, s>>>(d.M, d.cam_idx, d.obs_pt, d.pt_ptr, p->cam_ptr, fill.p,
                                                                                       static_cast<int4*>(p->cam_ent));
  SFM_HIP(hipGetLastError());
  p->h_cam_ptr.assign((size_t)d.V + 1, 0);
  p->h_cam_pairs.assign((size_t)d.V, 0);
  SFM_HIP(hipMemcpyAsync(p->h_cam_ptr.data(), p->cam_ptr, sizeof(int) * ((size_t)d.V + 1), hipMemcpyDeviceToHost, s));
  SFM_HIP(hipMemcpyAsync(p->h_cam_pairs.data(), p->cam_pairs, sizeof(unsigned long long) * (size_t)d.V, hipMemcpyDeviceToHost, s));
  SFM_TRY(stream_sync(s));            // cnt / fill go back to the pool; the host copies are complete
  return SFM_OK;
}

// Work split: R cameras per group (LDS), chunks per group in proportion to the camera pairs it owns, about two
// workgroups per CU in total (one is resident per CU).  Uploads the workgroup table.
int ba_rows_plan(sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  p->rows_ok = false;
  if (d.M == 0 || d.N == 0) return SFM_OK;
  const int tpr = ((7 * d.V + 1) / 2) * 2;
  // Camera pitch of the LDS panel.  A lane round adds 9 x 7 columns of one panel row; with the natural pitch of 7 the nine
  // 7-wide groups of a sparse track sit at random offsets of the 32 bank pairs (worst bank pair 4.1 lanes on average at 15 %
  // visibility, ideal 2), with a pitch of 8 they fall into four classes (3.7).  Measured (profiles/r4/ab_rows_pitch.txt,
  // profiles/r4/time_schur_rows_pitch.txt; us per launch, pitch 7 / 8): 200 cameras @ 0.15 (the C4 share) 325 / 285;
  // 100 @ 0.25: 202 / 191; 120 @ 0.1: 62 / 62; 60 @ 0.3: 97 / 99; 80 @ 0.4: 281 / 290; 50 @ 0.6 (C3 through this kernel)
  // 428 / 452 -- dense tracks are the other way round (consecutive cameras tile the banks exactly with pitch 7) -- so the
  // pitch follows the scene's visibility.  SFM_ROWS_PITCH8 = 0 / 1 forces it (A/B runs).
  static const int forced = [] { const char* e = getenv("SFM_ROWS_PITCH8"); return e ? (atoi(e) == 1 ? 8 : 7) : 0; }();
  const double visibility = (double)d.M / ((double)d.N * (double)d.V);
  int cp = forced ? forced : (visibility <= 0.25 ? 8 : 7);
  if (cp == 8 && kRowsLdsBudget / ((size_t)7 * 8 * d.V * sizeof(double)) < 1) cp = 7;      // 349 ... 398 cameras: one camera still fits with pitch 7
  const int tpl = cp == 7 ? tpr : 8 * d.V;
  const int R = (int)std::min<size_t>((size_t)d.V, kRowsLdsBudget / ((size_t)7 * tpl * sizeof(double)));
  if (R < 1) return SFM_OK;                 // more than ~2800 cameras: the 18-camera tile kernel takes over
  const int G = (d.V + R - 1) / R;
  double total = 0;
  for (int c = 0; c < d.V; ++c) total += (double)p->h_cam_pairs[c] + 4.0 * (p->h_cam_ptr[c + 1] - p->h_cam_ptr[c]);
  const int target = std::max(G, 2 * ctx().num_cus);
  std::vector<RowsWg> table;
  std::vector<int> first((size_t)G + 1, 0);
  for (int g = 0; g < G; ++g) {
    const int c0 = g * R, c1 = std::min(d.V, c0 + R);
    const int e0 = p->h_cam_ptr[c0], e1 = p->h_cam_ptr[c1];
    double w = 0;
    for (int c = c0; c < c1; ++c) w += (double)p->h_cam_pairs[c] + 4.0 * (p->h_cam_ptr[c + 1] - p->h_cam_ptr[c]);
    int chunks = total > 0 ? (int)(w / total * target + 0.5) : 1;
    // at least four visits per wave and at most 128 chunks per group (each is one more slab the reduce reads)
    chunks = std::max(1, std::min(std::min(chunks, 128), (e1 - e0 + 4 * ROWS_WAVES - 1) / (4 * ROWS_WAVES)));
    first[g] = (int)table.size();
    const int per = (e1 - e0 + chunks - 1) / std::max(1, chunks);
    for (int k = 0; k < chunks; ++k) {
      const int b = e0 + k * per, e = std::min(e1, b + per);
      table.push_back(RowsWg{g, b, std::max(b, e), 0});
    }
  }
  first[G] = (int)table.size();
  const size_t ws_bytes = sizeof(double) * table.size() * (size_t)7 * R * tpr;
  if (ws_bytes > ((size_t)8 << 30)) return SFM_OK;
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->rows_table), sizeof(RowsWg) * table.size()));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->rows_first), sizeof(int) * first.size()));
  SFM_HIP(pool_alloc(&p->rows_ws, ws_bytes));
  SFM_HIP(hipMemcpyAsync(p->rows_table, table.data(), sizeof(RowsWg) * table.size(), hipMemcpyHostToDevice, p->stream));
  SFM_HIP(hipMemcpyAsync(p->rows_first, first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice, p->stream));
  SFM_TRY(stream_sync(p->stream));
  SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kRowsLdsBudget));
  p->rows_R = R; p->rows_tpr = tpr; p->rows_wgs = (int)table.size(); p->rows_groups = G;
  p->rows_tpl = tpl; p->rows_cp = cp;
  p->rows_ok = true;
  return SFM_OK;
}

// The product + its reduce; the same launch adds the camera accumulators of ba_linearize and sums its cost (extra block rows).
int ba_rows_enqueue(sfm_ba_problem* p, hipStream_t s) {
  const BaDev& d = p->dev;
  const size_t lds = sizeof(double) * (size_t)7 * p->rows_R * p->rows_tpl;
  ba_tick(p, SFM_K_SCHUR, true, s);
  ba_schur_rows_kernel<<<p->rows_wgs, ROWS_THREADS, lds, s>>>(d, static_cast<const int4*>(p->cam_ent), static_cast<const RowsWg*>(p->rows_table),
                                                             static_cast<double*>(p->rows_ws), p->rows_R, p->rows_tpr, p->rows_tpl, p->rows_cp);
  ba_tick(p, SFM_K_SCHUR, false, s);
  ba_tick(p, SFM_K_REDUCE, true, s);       // closed by the caller
  const int gx = (d.P + 255) / 256;
  const int extra = (p->lin_rows > 0 ? 12 * ((d.V * 35 + 255) / 256) : 0) + 1;      // blocks of the camera-side sums + the cost
  ba_schur_rows_reduce_kernel<<<dim3(gx, d.P + (extra + gx - 1) / gx, 4), 256, 0, s>>>(d, static_cast<const double*>(p->rows_ws), p->rows_first,
                                                                                       p->rows_R, p->rows_tpr, p->lin_rows, p->lin_grid);
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

}  // namespace sfm
