// sfm_ba.h — device-side view and host-side object of a resident bundle-adjustment problem.
#pragma once

#include <utility>
#include <vector>

#include "sfm_common.h"

namespace sfm {

// Blocks of the Schur products (sfm_ba_schur.hip): the sparse tiles hold CB whole cameras = 126 rows in an RB = 128 pitch;
// the dense product cuts the rows of S into blocks of RB regardless of cameras.
constexpr int kSchurCB = 18;
constexpr int kSchurRB = 128;
constexpr int kSchurMaxChunks = 128;  // split-K slabs of the dense product's tile when it is the only one (up to 18 cameras)
constexpr int kSchurKSL = 16;    // Z rows per LDS slab; the row count of Zd is padded to a multiple of it
constexpr int kInvRowsMaxNbk = 52;  // block columns up to which the column steps carry the identity rows (X = L^-T; sfm_ba_solve.hip)
constexpr int kLinGridPerCu = 3; // ba_linearize workgroups per CU (132 VGPRs -> 3 waves/SIMD; forcing 4 measured 10 % slower)

// ---------------------------------------------------------------------------------------------
// Layout of the reduced camera system [S | rhs] (BaDev::red), the buffer one all-reduce covers in the
// multi-GPU split.  S is symmetric and only its lower triangle is ever formed, so only the lower-triangular
// 32x32 BLOCKS are stored, block (br, bc), bc <= br, at ((br (br + 1)) / 2 + bc) * 1024 doubles, row-major inside
// the block (element (i, k) at 32 i + k): 8.1 MB at V = 200 where the full square was 15.9 MB (0.54 MB vs 0.99 MB
// at V = 50).  rhs (padded to 32 nbk) follows the blocks.  Entries above the diagonal inside diagonal blocks and
// everything beyond P = 7V stay zero (the buffer is cleared once per iteration).
//
// The factorisation overwrites block (r, j) with L[r][j] in a DIFFERENT, k-interleaved order (red_lblk_off): lane
// (row i, k-phase lk = k & 3) of a v_mfma_f64_16x16x4 A / B operand consumes k = lk, lk + 4, ..., lk + 28 over the
// eight k-steps of a 32-deep product; that is four 16-byte pairs (k, k + 4), and pair m of the sixteen rows of a
// tile is 256 contiguous bytes, so the block products of the later steps load their operands straight from global
// memory with four fully coalesced 16-byte loads per tile and never stage them in LDS.  A row of L is written
// as sixteen such pairs, coalesced across the 32 rows; columns c and c + 4 of one block row are neighbours as
// well (what the back substitution reads).  Nothing outside sfm_ba_solve.hip sees that order.
// ---------------------------------------------------------------------------------------------
constexpr int kStatSlots = 256;   // iterations sfm_ba_get_stats can report between two state uploads
constexpr int kNB = 32;
constexpr int kBlk = kNB * kNB;
__host__ __device__ inline size_t red_blk_base(int br, int bc) { return ((size_t)br * (br + 1) / 2 + bc) * kBlk; }
__host__ __device__ inline int red_blk_off(int i, int k) { return i * kNB + k; }
__host__ __device__ inline int red_lblk_off(int i, int k) { return (k >> 3) * 256 + (k & 3) * 64 + i * 2 + ((k >> 2) & 1); }
__host__ __device__ inline size_t red_index(int row, int col) {      // row >= col
  return red_blk_base(row >> 5, col >> 5) + red_blk_off(row & 31, col & 31);
}
__host__ __device__ inline size_t red_rhs_off(int nbk) { return (size_t)nbk * (nbk + 1) / 2 * kBlk; }
__host__ __device__ inline size_t red_size(int nbk) { return red_rhs_off(nbk) + (size_t)nbk * kNB; }

// Plain-data view passed by value to kernels (all pointers are device memory).
struct BaDev {
  int V = 0, N = 0;
  long long M = 0;
  int P = 0;    // 7 V
  int nbk = 0;  // 32-wide block rows / columns of S: ceil(P / 32)
  // static structure: observations sorted by (point, camera)
  int* pt_ptr = nullptr;    // [N+1]
  int* cam_idx = nullptr;   // [M]
  int* obs_pt = nullptr;    // [M] point of each observation (thread-per-observation kernels)
  double* u = nullptr;      // [M] normalised keys
  double* v = nullptr;      // [M]
  // state
  double* cams = nullptr;   // [V][7]
  double* px = nullptr;     // [N] SoA points
  double* py = nullptr;
  double* pz = nullptr;
  // per-iteration scratch
  CamPrep* prep[2] = {nullptr, nullptr};   // double-buffered: back-substitution still needs the old one
  double* Z = nullptr;      // [M][21] (AoS)  Z_o = (Jp^T Jx) L_p^-T, element e = 3*i + j; sparse-product path only (lazy)
  double* Zd = nullptr;     // [zrows][zp] dense Z^T for the MFMA product: row 3p + j, column 7 cam + i (= the row of S);
                            // entries of invisible (point, camera) pairs and all padding stay zero for the problem's lifetime
  int zp = 0;               // row pitch of Zd = 7 V rounded up to a multiple of 16 (the MFMA strip)
  int zrows = 0;            // 3N rounded up to a multiple of kSchurKSL
  double* lin_ws = nullptr; // [linearize workgroups][V][35] per-workgroup camera accumulators (U lower 28 | rhs 7)
  double* red = nullptr;    // [red_size(nbk)] reduced system S (lower-triangular 32x32 blocks) | rhs
  double* delta = nullptr;  // [32 nbk] camera update
  double* ldiag = nullptr;  // [ceil(P/32)][32][32] INVERSE transposed Cholesky factors L_d^-T of the diagonal blocks, k-major
  double* xinv = nullptr;   // [nbk (nbk + 1) / 2 blocks] X = L^-T, block (e, c >= e) at red_blk_base(c, e): the identity carried
                            // through the column steps as extra block rows (sfm_ba_solve.hip); dp = X y is one launch
  int* sync_ctr = nullptr;  // [1] workgroups of ba_inv_apply that have stored their part of dp (self-resetting)
  unsigned* flow = nullptr; // [flow_words(nbk)] epoch / arrival / abort words and one flag per block of the data-flow solve (sfm_ba_flow.h); null beyond kFlowMaxNbk
  const void* flow_tasks = nullptr;  // [flow_ntasks] FlowTask table of the data-flow solve, sorted by column (one per device and nbk, owned by sfm_ba_solve.hip)
  int flow_ntasks = 0;
  int debug = 0;            // copy of sfm_ba_problem::debug for kernels that switch on it (diagnostic stamps, code-path switches)
  int* status = nullptr;    // [2] first failure code, camera index
  int* sinfo = nullptr;     // [4] structure check: first failure code, its index, longest track, unused
  double* cost = nullptr;   // [kStatSlots] sum |b - f|^2 over this problem's observations at the start of iteration i
  double* cost_ws = nullptr;  // [linearize workgroups] per-workgroup partial cost of the last linearisation
  int* iter_count = nullptr;  // [1] iterations completed since the state was last set (advanced by ba_back_solve)
  unsigned long long* stamps = nullptr;   // diagnostic shader-clock stamps (SFM_OPT_DEBUG bit 8), else null
};

// Sum of ba_linearize's per-workgroup partial costs -> cost[iteration] (one wave, fixed order).
__device__ __forceinline__ void cost_reduce(const BaDev& d, int nrows) {
  double s = 0;
  for (int r = threadIdx.x; r < nrows; r += 64) s += d.cost_ws[r];
  s = wave_sum(s);
  if (threadIdx.x == 0) d.cost[min(*d.iter_count, kStatSlots - 1)] = s;
}

// Slice `slice` of `nslices` of the sum over ba_linearize's per-workgroup camera accumulators (lin_ws rows)
// for accumulator element t, added into the diagonal blocks of S (lower part) / rhs with one f64 atomic.
__device__ __forceinline__ void cam_reduce_slice(const BaDev& d, int nrows, int t, int slice, int nslices) {
  if (t >= d.V * 35) return;
  const int per = (nrows + nslices - 1) / nslices;
  const int r0 = slice * per, r1 = min(nrows, r0 + per);
  double s = 0;
  for (int r = r0; r < r1; ++r) s += d.lin_ws[(size_t)r * d.V * 35 + t];
  if (s == 0.0) return;
  const int c = t / 35, e = t % 35;
  double* S = d.red;
  double* rhs = d.red + red_rhs_off(d.nbk);
  if (e >= 28) {
    atomicAdd(&rhs[7 * c + (e - 28)], s);
  } else {
    int i = 0, base = 0;                   // e = i(i+1)/2 + j
    while (base + i + 1 <= e) { base += i + 1; ++i; }
    atomicAdd(&S[red_index(7 * c + i, 7 * c + (e - base))], s);
  }
}

// Work split of both products.  Tiles are the lower-triangular pairs of blocks (128 rows of S in the dense product, 18
// whole cameras in the sparse one), in four classes with
// their own chunking: off-diagonal tiles (ti > tj, row-major order) whose row block is full, off-diagonal tiles
// of the LAST block (which may be partly empty: its empty 16-row MFMA strips are skipped), full
// diagonal tiles and the last diagonal tile.  A dense off-diagonal tile issues 2 x (strips of its row block)
// MFMAs per SIMD and k-step (16 when full), a diagonal one the larger half of its lower sub-tiles (9 when full);
// rows per chunk are inversely proportional, so every workgroup carries the same MFMA load.  Workgroup w's
// partial tile goes to slab w; workgroups are numbered class by class, tile by tile, chunk by chunk.
struct SchurPlan {
  int nblk, n_off;
  int chunks[4], rpc[4];        // per class (0 off full, 1 off last row, 2 diag full, 3 diag last): chunks per tile,
                                // rows of Zd (dense) or points (sparse) per chunk
  int ra_last;                  // 16-row strips of the last block that hold cameras (1..8)
  int cam_blocks;               // 1: blocks of 18 whole cameras, block row r = camera 18 b + r / 7 (sparse tiles);
                                // 0: blocks of 128 consecutive rows of S, block row r = row 128 b + r (dense product)
  int dbg;                      // profiling ablations (SFM_OPT_DEBUG): 1 = no MFMA, 4 = no staging loads
};

struct SchurTileRef { int ti, tj, cls, first, chunk; };

__host__ __device__ inline int plan_tiles_in_class(const SchurPlan& pl, int cls) {
  const int last_row = pl.nblk - 1;                       // off-diagonal tiles with ti == nblk - 1
  return cls == 0 ? pl.n_off - last_row : (cls == 1 ? last_row : (cls == 2 ? pl.nblk - 1 : 1));
}
__host__ __device__ inline int plan_wgs(const SchurPlan& pl) {
  int w = 0;
  for (int c = 0; c < 4; ++c) w += plan_tiles_in_class(pl, c) * pl.chunks[c];
  return w;
}
// tile index (off-diagonal tiles first in row-major (ti, tj) order, then the diagonal ones) -> blocks, class,
// first workgroup
__host__ __device__ inline SchurTileRef plan_tile(const SchurPlan& pl, int tile) {
  SchurTileRef r;
  r.chunk = 0;
  const int n0 = plan_tiles_in_class(pl, 0), n1 = plan_tiles_in_class(pl, 1), n2 = plan_tiles_in_class(pl, 2);
  if (tile < pl.n_off) {
    int t = tile;
    r.ti = 1;
    while (t >= r.ti) { t -= r.ti; ++r.ti; }
    r.tj = t;
    if (tile < n0) { r.cls = 0; r.first = tile * pl.chunks[0]; }
    else { r.cls = 1; r.first = n0 * pl.chunks[0] + (tile - n0) * pl.chunks[1]; }
  } else {
    r.ti = r.tj = tile - pl.n_off;
    const int base = n0 * pl.chunks[0] + n1 * pl.chunks[1];
    if (r.ti < n2) { r.cls = 2; r.first = base + r.ti * pl.chunks[2]; }
    else { r.cls = 3; r.first = base + n2 * pl.chunks[2]; }
  }
  return r;
}
// workgroup -> tile and chunk
__host__ __device__ inline SchurTileRef plan_locate(const SchurPlan& pl, int w) {
  int tile0 = 0, w0 = 0;
  for (int c = 0; c < 4; ++c) {
    const int nt = plan_tiles_in_class(pl, c), span = nt * pl.chunks[c];
    if (w < w0 + span || c == 3) {
      const int t = (w - w0) / pl.chunks[c];
      // classes are stored off-full, off-last, diag-full, diag-last = exactly the tile index order
      SchurTileRef r = plan_tile(pl, tile0 + t);
      r.chunk = (w - w0) - t * pl.chunks[c];
      return r;
    }
    w0 += span;
    tile0 += nt;
  }
  return SchurTileRef{0, 0, 0, 0, 0};
}

struct KernelTimer {
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
  int used = 0;
  int calls = 0;         // brackets seen since the last reset (SFM_OPT_TIMING_STRIDE samples every stride-th)
  bool open = false;     // the current bracket is a sampled one
};

constexpr unsigned kBaMagic = 0x5F3BA001u;

}  // namespace sfm

struct sfm_ba_problem {
  unsigned magic = sfm::kBaMagic;
  sfm::BaDev dev;
  hipStream_t stream = nullptr;   // every copy and kernel of this problem goes here (sfm_ba_set_stream)
  long long upload_bytes = 0;     // host -> device bytes moved on behalf of this handle (SFM_INFO_UPLOAD_BYTES)
  int cur = 0;               // which prep slot holds the cameras of the current state
  bool prep_valid = false;
  int lin_rows = 0;          // rows of lin_ws the last ba_linearize wrote (0: it used global atomics)
  int lin_grid = 0;          // workgroups of the last ba_linearize (rows of cost_ws)
  bool red_clean = false;    // [S | rhs] is known to be all zero (cleared by the last ba_backsub)
  bool backsub_pending = false;   // the reduced solve ran, its back substitution waits for the next launch (ba_flush)
  double pending_lambda = 0;      // ... with these parameters
  int pending_quirks = 0;
  int max_track = 0;         // longest track (observations of one point)
  int schur_mode = SFM_SCHUR_AUTO;
  int quirks = SFM_QUIRKS_REFERENCE;   // of the linearisation in flight
  int debug = 0;             // SFM_OPT_DEBUG: profiling ablations (results are wrong when set)
  int deterministic = 0;     // SFM_OPT_DETERMINISTIC: fixed summation order everywhere (bitwise repeatable results)
  int timing = 0;            // bitmask over SFM_K_* of the kernel classes bracketed by hipEvents
  int timing_stride = 1;     // ... every stride-th time they run (an event pair costs ~11 us of stream bubbles on this stack)
  double* own_red = nullptr; // library-owned reduced buffer (dev.red may point to a caller's tensor)
  sfm_comm* comm = nullptr;  // library-owned RCCL communicator (sfm_ba_set_comm): the iterations all-reduce [S | rhs] themselves
  // Schur-product plan (sfm_ba_schur.hip)
  void* schur_ws = nullptr;      // [chunks][tiles][128][128] split-K partial tiles
  const void* flow_tasks_red = nullptr;  // task table of the data-flow solve with the reduce deferred into it (sfm_ba_solve.hip)
  int flow_ntasks_red = 0;
  double* flow_camsum = nullptr; // [4][35 V] partial camera sums of the deferred reduce
  bool last_reduce_deferred = false;   // what the last ba_enqueue_schur decided (SFM_INFO_REDUCE_IN_SOLVE; survives the solve and graph replays)
  bool reduce_deferred = false;  // the last ba_enqueue_schur left its slabs unsummed: the next solve is the data-flow launch with FlowRed
  int* schur_blk_ptr = nullptr;  // [N][nblk + 1] first observation of a point in each 18-camera block (sparse path)
  bool schur_mfma_ok = false;
  // row-panel sparse product (sfm_ba_schur_rows.hip): camera-major observation list and work split, built on first use
  bool rows_built = false, rows_ok = false;
  int* cam_ptr = nullptr;                  // [V+1] device
  void* cam_ent = nullptr;                 // [M] device int4: the observations grouped by camera (observation, first of its track, camera, k_B)
  unsigned long long* cam_pairs = nullptr; // [V] device: camera pairs the observations of a camera own
  std::vector<int> h_cam_ptr;
  std::vector<unsigned long long> h_cam_pairs;
  void* rows_table = nullptr;              // [rows_wgs] workgroup -> (camera group, observation range)
  int* rows_first = nullptr;               // [groups+1] first workgroup of every camera group
  void* rows_ws = nullptr;                 // [rows_wgs][7 R][tpr] split-K panels
  int rows_R = 0, rows_tpr = 0, rows_wgs = 0, rows_groups = 0;
  int rows_tpl = 0, rows_cp = 7;          // LDS row pitch and camera pitch of the panel (experiment: 8)
  // SFM_OPT_GRAPH: the steady-state iteration body (fused linearise + Schur + reduce + solve) captured once per
  // camera-slot parity and replayed by sfm_ba_iterate; dropped whenever an option, the stream, the reduced buffer or
  // the structure changes
  int use_graph = 0;
  hipGraphExec_t body_graph[2] = {nullptr, nullptr};
  double graph_lambda = 0;
  int graph_quirks = 0;
  long long graph_replays = 0;    // SFM_INFO_GRAPH_REPLAYS
  sfm::KernelTimer timers[SFM_K_COUNT];
};

namespace sfm {
int ba_schur_plan(sfm_ba_problem* p);
int ba_rows_enqueue_build(sfm_ba_problem* p);
int ba_rows_plan(sfm_ba_problem* p);
int ba_rows_enqueue(sfm_ba_problem* p, hipStream_t s);
int ba_enqueue_structure(sfm_ba_problem* p);      // validate the CSR, fill obs_pt / per-point block offsets / longest track (device)
int ba_schur_prepare_dense(sfm_ba_problem* p, hipStream_t s);
int ba_enqueue_schur(sfm_ba_problem* p, hipStream_t s, bool allow_defer);
SchurPlan ba_schur_dense_plan(const sfm_ba_problem* p);
bool ba_solve_can_defer_reduce(const sfm_ba_problem* p);      // sfm_ba_solve.hip
void ba_tick(sfm_ba_problem* p, int kernel_class, bool begin, hipStream_t s);   // hipEvent bracket of a kernel class (SFM_OPT_TIMING)
bool ba_schur_uses_mfma(const sfm_ba_problem* p);
int ba_schur_choice(const sfm_ba_problem* p);
int ba_enqueue_prep(sfm_ba_problem* p);
int ba_enqueue_linearize_reduce(sfm_ba_problem* p, double lambda, int quirks, bool allow_defer = false);
int ba_enqueue_solve_update(sfm_ba_problem* p, double lambda, int quirks);
int ba_flush(sfm_ba_problem* p);      // complete a deferred back substitution
void ba_graph_drop(sfm_ba_problem* p); // forget the captured iteration bodies
int ba_enqueue_iterations(sfm_ba_problem* p, double lambda, int iters, int quirks);
bool ba_can_fuse(const sfm_ba_problem* p);
int ba_enqueue_reduced_solve(sfm_ba_problem* p, double lambda);      // sfm_ba_solve.hip: factor, solve, update cameras
int ba_flow_setup(sfm_ba_problem* p);      // sfm_ba_solve.hip: flag words and task table of the data-flow solve (2 <= nbk <= kFlowMaxNbk)
int comm_all_reduce_f64(sfm_comm* comm, double* buf, size_t count, hipStream_t s);      // sfm_comm.hip
int comm_attach(sfm_comm* comm, int delta);      // a problem takes (+1) / gives back (-1) its hold on a communicator
void ba_enqueue_residual_jacobian(sfm_ba_problem* p, int quirks, double* r, double* Jp, double* Jx);
void ba_enqueue_symmetrize(sfm_ba_problem* p, double lambda, double* S_out, double* rhs_out);
}  // namespace sfm
