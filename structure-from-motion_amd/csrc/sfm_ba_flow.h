// sfm_ba_flow.h — the reduced camera solve as ONE persistent data-flow launch (included by sfm_ba_solve.hip, which
// defines chol_trsm_cols and the operand layouts this file builds on).
//
// Why: the column steps of ba_chol_step are a chain of nbk dependent launches; a step is 0.7 us of launch, 1.5 us of
// operand wait + previous-panel update, 3.35 us of elimination and 0.3 us of stores (profiles/r4/probe_solve.json) — half of it
// is the round trip of the diagonal block through L2 and the launch boundary.  Here ONE workgroup (the chain) keeps
// everything the pivot chain depends on in its own LDS and never leaves it: per block column j it eliminates [D_j; I]
// (chol_trsm_cols: L_jj and X_j = L_jj^-T in one instruction stream), forms L[j+1][j] = T W_j^T and D_j+1 -= L L^T on the
// matrix pipe out of LDS, and goes on.  Everything else — the blocks of L more than two block rows below the diagonal, the
// right-hand side, the identity rows that leave X = L^-T behind for ba_inv_apply — is plain matrix-pipe work: one TASK per
// block, LEFT-looking (S_ik - sum_m L[i][m] L[k][m]^T, then the product with W_k = L_kk^-1), taken by the other workgroups
// of the launch one after the other in column order (a ticket per task).  Blocks travel through global memory: write-through (sc1) stores, one flag word per block,
// sc1 loads on the consuming side (MI355X_MICROARCH.md, inter-workgroup visibility; every workgroup of this launch owns its
// CU: 121 KB of LDS).
//
// Who computes what (the chain keeps the two sub-diagonals to itself):
//   chain, step j:   elimination waves (0-3, LDS only): [D_j; I] -> X_j, W_j = X_j^T; then L[j+1][j] = T W_j^T (TRSM as a product),
//                    D_j+1 -= L[j+1][j] L[j+1][j]^T.
//                    preparation waves (4-7), concurrently with the elimination, row r = j+1: take over the blocks
//                    (r, r-2), (r, r-1), (r, r) (fetched a step ahead), L[r][r-2] = T W_j-1^T, and apply columns r-3 / r-2 to
//                    (r, r-1) and column r-2 to (r, r); behind the elimination they publish X_j -> xinv(j, j), W_j -> ldiag[j], L[j+1][j].
//   tasks of block row i >= 3:  L[i][k] for k <= i-4; the "closer" L[i][i-3] together with the hand-over blocks (i, i-2) and
//                    (i, i) through column i-3 (sums through column i-4 formed before W_i-3 exists -- but for the one term that needs
//                    the chain's L[i-2][i-4], announced with W_i-3 --, the last terms out of LDS);
//                    the hand-over block (i, i-1) through column i-4 — levels chosen so that no hand-over waits for a block
//                    the chain publishes later than W_i-3.
//   tasks of the rhs row and of the identity rows e = 0 .. nbk-2: the same recurrence with their own blocks; the task that finishes
//                    an identity row forms its part of dp = X y, the last one to arrive updates the cameras.
//   with the split-K reduce riding in the launch (FlowRed, below): tasks ahead of all these sum ba_linearize's camera accumulators
//                    (one camera, a quarter of the rows) and the product's slabs (8 rows of a block of S); the chain sums D_0 itself.
//   Which workgroup is the chain is decided by arrival: the first to take a ticket; the others take tasks in table order.
// Every block is produced by exactly one task with a fixed summation order: the result does not depend on timing or placement.
// All products are formed TRANSPOSED (the 16x16x4 accumulator layout of M^T is the operand layout of M: rows lk + 4g), so a
// block goes accumulator -> k-interleaved block (red_lblk_off) with two 16-byte stores per tile and comes back as an A or
// B operand with four 16-byte loads, in LDS and in global memory alike.
#pragma once

namespace sfm {

constexpr int kFlowMaxNbk = kInvRowsMaxNbk;      // wherever xinv is allocated (X = L^-T for dp = X y): 52 block columns, 237 cameras
constexpr int kFlowHdr = 128;        // header words of BaDev::flow, the hot ones on 128-byte lines of their own: [0] epoch of the last finished solve,
                                     // [2] abort (read-mostly, polled rarely), [32] workgroups done, [64] tasks taken (one atomic per task); flags from [128]
constexpr int kFlowDone = 32, kFlowTicket = 64;
constexpr unsigned kFlowSpinLimit = 4000000u;   // polls (~0.5-1 us each) before a wait gives up and the solve reports SFM_E_HIP
__host__ __device__ inline int flow_fl(int nbk, int i, int k) { return kFlowHdr + i * nbk + k; }                      // L[i][k] published
__host__ __device__ inline int flow_fx(int nbk, int e, int m) { return kFlowHdr + nbk * nbk + e * nbk + m; }          // X[e][m] published
__host__ __device__ inline int flow_fw(int nbk, int k) { return kFlowHdr + 2 * nbk * nbk + k; }                       // W_k published
__host__ __device__ inline int flow_fy(int nbk, int k) { return kFlowHdr + 2 * nbk * nbk + nbk + k; }                 // y_k published
__host__ __device__ inline int flow_fh(int nbk, int i, int t) { return kFlowHdr + 2 * nbk * nbk + 2 * nbk + 3 * i + t; }   // hand-over block (i, i-2+t)
__host__ __device__ inline int flow_words_solve(int nbk) { return kFlowHdr + 2 * nbk * nbk + 5 * nbk; }
// deferred reduce (S summed from the product's split-K slabs inside this launch): camera c's accumulators, part 0..3; quarter q of block (i, k) of S
__host__ __device__ inline int flow_fc(int nbk, int c, int part) { return flow_words_solve(nbk) + 4 * c + part; }
__host__ __device__ inline int flow_fs(int nbk, int i, int k, int q) { return flow_words_solve(nbk) + 4 * (5 * nbk + 1) + 4 * (i * (i + 1) / 2 + k) + q; }
__host__ __device__ inline int flow_fcost(int nbk) { return flow_fs(nbk, nbk, 0, 0); }      // this linearisation's cost is stored
__host__ __device__ inline int flow_words(int nbk) { return flow_fcost(nbk) + 1; }

// task table (host-built once per problem, sorted by the column a task waits for last)
enum { FLOW_T1 = 0, FLOW_CLOSER = 1, FLOW_H1 = 2, FLOW_RHS = 3, FLOW_IDENT = 4, FLOW_CAMSUM = 5, FLOW_SRED = 6, FLOW_COST = 7 };
struct FlowTask { int type, i, k, key; };      // CAMSUM: i = camera, k = part;  SRED: block (i, k), quarter key & 3 (after sorting)

// Deferred reduce (sfm_ba_iterate on one GPU, dense product): ba_schur_reduce is not launched; the first tasks of this launch sum
// ba_linearize's per-workgroup camera accumulators (CAMSUM: one camera, a quarter of the rows) and the product's split-K slabs
// (SRED: 8 rows of a 32x32 block of S, + the camera sums on the camera-diagonal, fixed order: deterministic), block column by
// block column, and the chain starts as soon as D_0 is there instead of behind the whole reduce and a launch boundary.
struct FlowRed {
  const double* ws;      // the product's slabs (null: S and rhs are in `red` already)
  double* camsum;        // [4][35 V] partial sums of lin_ws
  SchurPlan plan;
  int lin_rows, lin_grid;
};

// LDS carve (doubles)
constexpr int FS_DM = 0;                  // [32][33] D_j, row-major (elimination: lane = row)
constexpr int FS_XM = 1056;               // [32][33] X_j rows as the elimination leaves them
constexpr int FS_XY = 2112;               // f64x2[16][64] published pivots of chol_trsm_cols
constexpr int FS_W = 4160;                // [2][1024] W_j, W_j-1 (k-interleaved)
constexpr int FS_LA = 6208;               // [2][1024] L[j+1][j] of this and the previous step
constexpr int FS_LB = 8256;               // [2][1024] L[r][r-2] of this and the previous step
constexpr int FS_B = 10304;               // [3][1024] the row being taken over: (r, r-2), (r, r-1), (r, r); tasks: T and L
constexpr int FS_XL = 13376;              // [1024] X_j in the operand layout, for the preparation waves to publish
constexpr int FS_XM2 = 14400;             // [32][33] the second X_j row buffer
constexpr int FS_INT = 15456;             // ints: [0..4] the chain's elimination flag, group-barrier counters and step counters; task workgroups: [15] last-arriver flag of dp, [20] ticket
constexpr int FS_TOTAL = 15472;
constexpr size_t kFlowLdsBytes = FS_TOTAL * sizeof(double);

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct FlowBuf { __amdgpu_buffer_rsrc_t r; };
__device__ __forceinline__ FlowBuf flow_buf(const void* base) {
  return FlowBuf{__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000)};
}
// write-through / L1-bypassing 16-byte accesses (buffer_load / store_dwordx4 ... sc1); offsets in doubles
__device__ __forceinline__ f64x2 ld2_sc1(FlowBuf b, size_t off) {
  return __builtin_bit_cast(f64x2, __builtin_amdgcn_raw_buffer_load_b128(b.r, (int)(off * 8), 0, 16));
}
__device__ __forceinline__ void st2_sc1(FlowBuf b, size_t off, f64x2 v) {
  __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), b.r, (int)(off * 8), 0, 16);
}
__device__ __forceinline__ void flow_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// operand tile `tile` (rows 16 tile .. +15) of a k-interleaved block: from global memory (sc1) / from LDS
__device__ __forceinline__ void op_sc1(FlowBuf b, size_t blk, int tile, int lr, int lk, double (&o)[8]) {
  const size_t p = blk + lk * 64 + (16 * tile + lr) * 2;
#pragma unroll
  for (int m = 0; m < 4; ++m) { const f64x2 v = ld2_sc1(b, p + 256 * m); o[2 * m] = v.x; o[2 * m + 1] = v.y; }
}
__device__ __forceinline__ void op_lds(const double* blk, int tile, int lr, int lk, double (&o)[8]) {
  const double* p = blk + lk * 64 + (16 * tile + lr) * 2;
#pragma unroll
  for (int m = 0; m < 4; ++m) { const f64x2 v = *reinterpret_cast<const f64x2*>(p + 256 * m); o[2 * m] = v.x; o[2 * m + 1] = v.y; }
}
// the rhs row as a block with one valid row: y_m sits at rhs[32 m ..]
__device__ __forceinline__ void op_rhs_sc1(const double* seg, bool row0, int lk, double (&o)[8]) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) o[ks] = row0 ? __hip_atomic_load(seg + 4 * ks + lk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
}
// acc(kappa, a) += sum_k A(kappa, k) B(a, k): A, B operand tiles of two k-interleaved blocks
__device__ __forceinline__ void mfma8(f64x4& acc, const double (&a)[8], const double (&b)[8]) {
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks], b[ks], acc, 0, 0, 0);
}
// Accumulator tile (sx, sy) of M^T -- register g of lane (lr, lk) is M(a = 16 sy + lr, kappa = 16 sx + lk + 4 g) -- inside the
// k-interleaved block of M: g = 0, 1 at ctile_off, g = 2, 3 256 doubles further.
__device__ __forceinline__ int ctile_off(int sx, int sy, int lr, int lk) { return 2 * sx * 256 + lk * 64 + 2 * (16 * sy + lr); }
__device__ __forceinline__ void ctile_st_lds(double* blk, int off, const f64x4& v) {
  *reinterpret_cast<f64x2*>(blk + off) = f64x2{v[0], v[1]};
  *reinterpret_cast<f64x2*>(blk + off + 256) = f64x2{v[2], v[3]};
}
__device__ __forceinline__ f64x4 ctile_ld_lds(const double* blk, int off) {
  const f64x2 p = *reinterpret_cast<const f64x2*>(blk + off), q = *reinterpret_cast<const f64x2*>(blk + off + 256);
  return f64x4{p.x, p.y, q.x, q.y};
}
__device__ __forceinline__ void ctile_st_sc1(FlowBuf b, size_t blk, int off, const f64x4& v) {
  st2_sc1(b, blk + off, f64x2{v[0], v[1]});
  st2_sc1(b, blk + off + 256, f64x2{v[2], v[3]});
}

// Waiting.  Lane 0.. poll flags, lane 63 the abort word, relaxed agent-scope loads (sc1).  After a give-up (a sibling never
// scheduled, or the spin limit reached) every further wait returns at once: the control flow -- and with it the number of barriers
// every wave passes -- stays what it is, the results are garbage and d.status says so.
struct FlowWait {
  unsigned* flow; unsigned epoch; int* status; bool dead; unsigned limit;
  // the abort word sits on a line every waiting wave of the launch would hammer: it is looked at every 64th poll only
  __device__ __forceinline__ bool aborted() {
    unsigned v = 0;
    if ((threadIdx.x & 63) == 0) v = __hip_atomic_load(flow + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(v) != 0;
  }
  __device__ __forceinline__ void give_up() {
    if ((threadIdx.x & 63) == 0) { __hip_atomic_store(flow + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); report_status(status, SFM_E_HIP, -2); }
    dead = true;
  }
  // up to four single flags at once
  __device__ __forceinline__ void wait(int i0, int i1 = -1, int i2 = -1, int i3 = -1) {
    if (dead) return;
    const int lane = threadIdx.x & 63;
    const int mine = lane == 0 ? i0 : lane == 1 ? i1 : lane == 2 ? i2 : lane == 3 ? i3 : -1;
    for (unsigned spins = 0;; ++spins) {
      unsigned v = epoch;
      if (mine >= 0) v = __hip_atomic_load(flow + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long miss = __ballot(mine >= 0 && v != epoch);
      if (!miss) break;
      if ((spins & 63) == 63 && (aborted() || spins > limit)) { give_up(); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");      // no instruction: keeps the compiler from hoisting loads above the poll
  }
  // one poll of up to three flags: all there?
  __device__ __forceinline__ bool poll(int i0, int i1, int i2) {
    if (dead) return true;
    const int lane = threadIdx.x & 63;
    const int mine = lane == 0 ? i0 : lane == 1 ? i1 : lane == 2 ? i2 : -1;
    unsigned v = epoch;
    if (mine >= 0) v = __hip_atomic_load(flow + mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool all = __ballot(v != epoch) == 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return all;
  }
  // Terms m0 .. m0+n-1 of a row sum; term t is ready when flag idx[q] + t carries the epoch for every one of the nrows <= 3 flag
  // rows (row q polled by lanes 16 q .. 16 q + 15).  Waits until `need` leading terms are ready, returns how many are.
  __device__ __forceinline__ int wait_terms(int idxa, int idxb, int idxc, int nrows, int n, int need) {
    if (dead || n <= 0) return n;
    const int lane = threadIdx.x & 63, q = lane >> 4, t = lane & 15;
    const bool mine = q < nrows && t < n;
    const int idx = (q == 0 ? idxa : q == 1 ? idxb : idxc) + t;
    int have = 0;
    for (unsigned spins = 0;; ++spins) {
      unsigned v = epoch;
      if (mine) v = __hip_atomic_load(flow + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned long long miss = __ballot(mine && v != epoch);
      const unsigned m16 = (unsigned)((miss | (miss >> 16) | (miss >> 32)) & 0xffffu);
      have = m16 ? __builtin_ctz(m16) : n;
      if (have >= need) break;
      if ((spins & 63) == 63 && (aborted() || spins > limit)) { give_up(); have = n; break; }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    return have;
  }
  // n <= 64 consecutive flags, one per lane: all there
  __device__ __forceinline__ void wait_run(int first, int n) {
    if (dead) return;
    const int lane = threadIdx.x & 63;
    for (unsigned spins = 0;; ++spins) {
      unsigned v = epoch;
      if (lane < n) v = __hip_atomic_load(flow + first + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (!__ballot(v != epoch)) break;
      if ((spins & 63) == 63 && (aborted() || spins > limit)) { give_up(); break; }
      __builtin_amdgcn_s_sleep(2);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
};
__device__ __forceinline__ void flow_publish(unsigned* flow, int idx, unsigned epoch) {
  __hip_atomic_store(flow + idx, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// barrier of the four preparation waves alone (the elimination waves run beside them and meet no barrier): every wave
// finishes its LDS writes, one lane arrives on an LDS counter, all spin until the round's arrivals are in
__device__ __forceinline__ void flow_group_sync(int* ctr, int& target) {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  target += 4;
  if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

struct FlowCtx {
  FlowBuf red, xinv, ldiag;
  double* redp; double* rhs;      // plain views (S blocks, the rhs segment)
  unsigned* flow; unsigned epoch; int nbk;
  double* sm; int* smi;
  int lane, lr, lk, gw, sx, sy;
  FlowWait w;
  unsigned long long* stamps;      // SFM_OPT_DEBUG bit 8 (thread 0 of a task workgroup), else null
  bool sred;                       // S comes from SRED tasks of this launch (wait for them, sc1 loads), rhs from the camera sums
  const double* camsum; int ncam35;
};

// Blocks (i, k0 .. k1) of S are there (deferred reduce: the 4 (k1 - k0 + 1) quarter flags are neighbours -- one poll loop, so that
// the loads of all of them travel together afterwards)
__device__ __forceinline__ void flow_wait_S(FlowCtx& c, int i, int k0, int k1) {
  if (!c.sred) return;
  for (int n = 4 * (k1 - k0 + 1), f = flow_fs(c.nbk, i, k0, 0); n > 0; n -= 64, f += 64) c.w.wait_run(f, min(64, n));
}
// tile (sx, sy) of S_blk^T of block (i, k): as the reduce kernel left it (plain loads), or as this launch's SRED tasks published it
__device__ __forceinline__ f64x4 flow_ld_S(const FlowCtx& c, int i, int k, int sx, int sy) {
  const double* p = c.redp + red_blk_base(i, k) + (16 * sy + c.lr) * kNB + 16 * sx + c.lk;
  if (!c.sred) return f64x4{p[0], p[4], p[8], p[12]};
  f64x4 v;
#pragma unroll
  for (int g = 0; g < 4; ++g) v[g] = __hip_atomic_load(p + 4 * g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}
// every part of the camera sums of cameras c0 .. c1 is there
__device__ __forceinline__ void flow_wait_cams(FlowCtx& c, int c0, int c1) {
  for (int n = 4 * (c1 - c0 + 1), f = flow_fc(c.nbk, c0, 0); n > 0; n -= 64, f += 64) c.w.wait_run(f, min(64, n));
}
__device__ __forceinline__ double flow_camsum(const FlowCtx& c, int t) {      // element t of [V][35], parts added in order
  double u = 0;
#pragma unroll
  for (int part = 0; part < 4; ++part) u += __hip_atomic_load(c.camsum + (size_t)part * c.ncam35 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return u;
}

// CAMSUM: rows [part/4, (part+1)/4) of lin_ws for the 35 accumulators of one camera (thread = accumulator e, row residue g mod 7)
__device__ __forceinline__ void flow_task_camsum(FlowCtx& c, const BaDev& d, const FlowRed& fr, int cam, int part) {
  const int tid = threadIdx.x, T = 35 * d.V, per = (fr.lin_rows + 3) / 4, r0 = part * per, r1 = min(fr.lin_rows, r0 + per);
  double* sums = c.sm + FS_B;      // [7][35]
  unsigned long long* stamp = (c.stamps && cam == 0 && part == 0) ? c.stamps + 128 : nullptr;      // 100 MHz clock: [0] start, [1] rows summed, [2] published
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
  if (tid < 245) {
    const int e = tid % 35, g = tid / 35;
    const double* src = d.lin_ws + (size_t)cam * 35 + e;
    double s = 0;
    for (int rb = r0 + g; rb < r1; rb += 7 * 28) {      // 28 loads in flight (all of a part at 768 rows), added in row order
      double v[28];
#pragma unroll
      for (int u = 0; u < 28; ++u) v[u] = rb + 7 * u < r1 ? src[(size_t)(rb + 7 * u) * T] : 0.0;
#pragma unroll
      for (int u = 0; u < 28; ++u) s += v[u];
    }
    sums[tid] = s;
  }
  __syncthreads();
  if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  if (tid < 35) {
    double t = 0;
#pragma unroll
    for (int g = 0; g < 7; ++g) t += sums[35 * g + tid];
    __hip_atomic_store(fr.camsum + (size_t)part * T + cam * 35 + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  flow_drain();
  __syncthreads();
  if (tid == 0) flow_publish(c.flow, flow_fc(c.nbk, cam, part), c.epoch);
  if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
}

// COST: the cost of this linearisation (what ba_schur_reduce's last block does); the workgroup that advances iter_count at the end of
// the solve waits for it
__device__ __forceinline__ void flow_task_cost(FlowCtx& c, const BaDev& d, const FlowRed& fr) {
  if (threadIdx.x < 64) {
    double s = 0;
    for (int r = c.lane; r < fr.lin_grid; r += 64) s += d.cost_ws[r];
    s = wave_sum(s);
    if (c.lane == 0) d.cost[min(*d.iter_count, kStatSlots - 1)] = s;
  }
  flow_drain();
  __syncthreads();
  if (threadIdx.x == 0) flow_publish(c.flow, flow_fcost(c.nbk), c.epoch);
}

// SRED: rows 8q .. 8q+7 of block (i, k) of S = camera sums (on a camera's own 7x7 block) - the tile's slabs in chunk order;
// zero above the diagonal and on the padding (what the reduce kernel leaves of a cleared `red`)
__device__ __forceinline__ void flow_task_sred(FlowCtx& c, const BaDev& d, const FlowRed& fr, int i, int k, int q) {
  const int tid = threadIdx.x, r = tid >> 5, cc = tid & 31, row = kNB * i + 8 * q + r, col = kNB * k + cc;
  const int ti = row / kSchurRB, tj = col / kSchurRB;
  const SchurTileRef tr = plan_tile(fr.plan, ti > tj ? ti * (ti - 1) / 2 + tj : fr.plan.n_off + ti);
  const int chunks = fr.plan.chunks[tr.cls];
  unsigned long long* stamp = (c.stamps && i == 1 && k == 0 && q == 0) ? c.stamps + 136 : nullptr;      // [0] start, [1] slabs summed, [2] camera sums seen, [3] published
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const double* src = fr.ws + (size_t)tr.first * (kSchurRB * kSchurRB) + (row % kSchurRB) * kSchurRB + (col % kSchurRB);
  double s = 0;
  for (int cb = 0; cb < chunks; cb += 40) {      // 40 loads in flight (a diagonal tile's 37 slabs at C3 in one trip), added in chunk order
    double v[40];
#pragma unroll
    for (int u = 0; u < 40; ++u) v[u] = cb + u < chunks ? src[(size_t)(cb + u) * (kSchurRB * kSchurRB)] : 0.0;
#pragma unroll
    for (int u = 0; u < 40; ++u) s += v[u];
  }
  double v = -s;
  if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  if (k >= i - 1) {
    const int c0 = (kNB * i + 8 * q) / 7, c1 = min(d.V - 1, (kNB * i + 8 * q + 7) / 7);
    if (c0 <= c1) flow_wait_cams(c, c0, c1);
    if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
    if (row < d.P && col <= row && row / 7 == col / 7) {
      const int ii = row % 7, jj = col % 7;
      v += flow_camsum(c, (row / 7) * 35 + ii * (ii + 1) / 2 + jj);
    }
  }
  if (row >= d.P || col > row) v = 0;
  __hip_atomic_store(c.redp + red_blk_base(i, k) + (8 * q + r) * kNB + cc, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  flow_drain();
  __syncthreads();
  if (tid == 0) flow_publish(c.flow, flow_fs(c.nbk, i, k, q), c.epoch);
  if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
}

// ---------------------------------------------------------------------------------------------
// Tasks (four waves, wave (sx, sy) owns accumulator tile (sx, sy)).
// A row sum: acc(kappa, a) += sum_{m = m0}^{m1 - 1} L[arow][m](kappa, .) . own[m](a, .), terms taken in order, up to four of
// them loaded at a time as soon as their flags are up.
// OWN 0: a block row of L (red);  1: the rhs row (y_m);  2: identity row `own` (xinv).
// ---------------------------------------------------------------------------------------------
template <int OWN>
__device__ __forceinline__ void flow_own_operand(FlowCtx& c, int own, int m, double (&b)[8]) {
  if (OWN == 0) op_sc1(c.red, red_blk_base(own, m), c.sy, c.lr, c.lk, b);
  else if (OWN == 1) op_rhs_sc1(c.rhs + m * kNB, c.sy == 0 && c.lr == 0, c.lk, b);
  else op_sc1(c.xinv, red_blk_base(m, own), c.sy, c.lr, c.lk, b);
}
template <int OWN>
__device__ __forceinline__ int flow_own_flag(const FlowCtx& c, int own, int m) {
  return OWN == 0 ? flow_fl(c.nbk, own, m) : OWN == 1 ? flow_fy(c.nbk, m) : flow_fx(c.nbk, own, m);
}
template <int OWN>
__device__ __forceinline__ void flow_row_sum(FlowCtx& c, f64x4& acc, int arow, int own, int m0, int m1) {
  int ready = m0;
  for (int m = m0; m < m1;) {
    if (m >= ready) {
      const int ws = m0 + ((m - m0) & ~15);      // one poll covers a window of 16 terms
      ready = ws + c.w.wait_terms(flow_fl(c.nbk, arow, ws), flow_own_flag<OWN>(c, own, ws), 0, (OWN == 0 && arow == own) ? 1 : 2, min(16, m1 - ws), m - ws + 1);
    }
    const int n = min(ready - m, 4);
    double a[4][8], b[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < n) { op_sc1(c.red, red_blk_base(arow, m + q), c.sx, c.lr, c.lk, a[q]); flow_own_operand<OWN>(c, own, m + q, b[q]); }
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < n) mfma8(acc, a[q], b[q]);
    m += n;
  }
}

// second half of a plain task: T^T (accumulator layout) -> LDS -> L^T = W_k T^T -> its k-interleaved block in global memory
template <int OWN>
__device__ __forceinline__ void flow_finish(FlowCtx& c, const f64x4& t, int own, int k, unsigned long long* stamp = nullptr) {
  double* Tx = c.sm + FS_B;
  const int coff = ctile_off(c.sx, c.sy, c.lr, c.lk);
  ctile_st_lds(Tx, coff, t);
  __syncthreads();
  c.w.wait(flow_fw(c.nbk, k));
  if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
  double a[8], b[8];
  op_sc1(c.ldiag, (size_t)k * kBlk, c.sx, c.lr, c.lk, a);
  op_lds(Tx, c.sy, c.lr, c.lk, b);
  f64x4 o = {0, 0, 0, 0};
  mfma8(o, a, b);
  if (OWN == 0) ctile_st_sc1(c.red, red_blk_base(own, k), coff, o);
  else if (OWN == 2) ctile_st_sc1(c.xinv, red_blk_base(k, own), coff, o);
  else if (c.sy == 0 && c.lr == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) __hip_atomic_store(c.rhs + k * kNB + 16 * c.sx + c.lk + 4 * g, o[g], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  flow_drain();
  __syncthreads();      // every wave's stores are in memory (and Tx is free again)
  if (threadIdx.x == 0) flow_publish(c.flow, flow_own_flag<OWN>(c, own, k), c.epoch);
}

__device__ __forceinline__ void flow_task_t1(FlowCtx& c, int i, int k) {
  unsigned long long* stamp = (c.stamps && k == i - 4) ? c.stamps + 128 + 8 * i : nullptr;      // the block the closer of row i waits for last
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
  flow_wait_S(c, i, k, k);
  const f64x4 s = flow_ld_S(c, i, k, c.sx, c.sy);
  f64x4 acc = {0, 0, 0, 0};
  flow_row_sum<0>(c, acc, k, i, 0, k);
  if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  flow_finish<0>(c, s - acc, i, k, stamp);
  if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
}

// hand-over block (i, i-1) through column i-4
__device__ __forceinline__ void flow_task_h1(FlowCtx& c, int i) {
  const int coff = ctile_off(c.sx, c.sy, c.lr, c.lk);
  flow_wait_S(c, i, i - 1, i - 1);
  const f64x4 s = flow_ld_S(c, i, i - 1, c.sx, c.sy);
  f64x4 acc = {0, 0, 0, 0};
  flow_row_sum<0>(c, acc, i - 1, i, 0, i - 3);
  __syncthreads();      // every wave has its part of S before any wave overwrites the block
  ctile_st_sc1(c.red, red_blk_base(i, i - 1), coff, s - acc);
  flow_drain();
  __syncthreads();
  if (threadIdx.x == 0) flow_publish(c.flow, flow_fh(c.nbk, i, 1), c.epoch);
}

// L[i][i-3] and the hand-over blocks (i, i-2), (i, i) through column i-3: the three sums through column i-4 share the operand
// tiles of row i and are complete before W_i-3 exists; what W_i-3 releases is one product, an LDS round trip and two terms.
__device__ __forceinline__ void flow_task_closer(FlowCtx& c, int i) {
  const int coff = ctile_off(c.sx, c.sy, c.lr, c.lk), k = i - 3, nbk = c.nbk;
  flow_wait_S(c, i, k, i);
  const f64x4 sT = flow_ld_S(c, i, k, c.sx, c.sy);
  const f64x4 s0 = flow_ld_S(c, i, i - 2, c.sx, c.sy);
  const f64x4 s2 = flow_ld_S(c, i, i, c.sx, c.sy);
  f64x4 accT = {0, 0, 0, 0}, acc0 = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
  unsigned long long* stamp = c.stamps ? c.stamps + 640 + 8 * i : nullptr;      // 100 MHz clock (s_memrealtime): comparable across workgroups
  if (stamp) stamp[0] = __builtin_amdgcn_s_memrealtime();
  // The sums through column k-1 -- except ONE term: L[i-2][k-1] is the block (r, r-2) of the chain's row r = i-2, which the chain
  // forms during step k and announces together with W_k (the cross-workgroup timeline showed the closer still summing 1.9 us
  // after W_k was out because it waited for that flag with all three sums).  That term joins the two last ones behind W_k.
  int ready = 0;
  for (int m = 0; m < k - 1;) {            // columns 0 .. k-2: all three sums
    if (m >= ready) {
      const int ws = m & ~15;
      ready = ws + c.w.wait_terms(flow_fl(nbk, k, ws), flow_fl(nbk, i - 2, ws), flow_fl(nbk, i, ws), 3, min(16, k - 1 - ws), m - ws + 1);
    }
    const int n = min(ready - m, 2);
    double b[2][8], at[2][8], a0[2][8], a2[2][8];
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < n) {
        op_sc1(c.red, red_blk_base(i, m + q), c.sy, c.lr, c.lk, b[q]);
        op_sc1(c.red, red_blk_base(k, m + q), c.sx, c.lr, c.lk, at[q]);
        op_sc1(c.red, red_blk_base(i - 2, m + q), c.sx, c.lr, c.lk, a0[q]);
        op_sc1(c.red, red_blk_base(i, m + q), c.sx, c.lr, c.lk, a2[q]);
      }
#pragma unroll
    for (int q = 0; q < 2; ++q)
      if (q < n) { mfma8(accT, at[q], b[q]); mfma8(acc0, a0[q], b[q]); mfma8(acc2, a2[q], b[q]); }
    m += n;
  }
  double bl[8];      // operand tile of L[i][k-1], kept for the deferred term
#pragma unroll
  for (int q = 0; q < 8; ++q) bl[q] = 0;
  if (k >= 1) {                            // column k-1: the sums of L[i][k] and of (i, i)
    c.w.wait(flow_fl(nbk, k, k - 1), flow_fl(nbk, i, k - 1));
    if (stamp) stamp[6] = __builtin_amdgcn_s_memrealtime();
    double at[8], a2[8];
    op_sc1(c.red, red_blk_base(i, k - 1), c.sy, c.lr, c.lk, bl);
    op_sc1(c.red, red_blk_base(k, k - 1), c.sx, c.lr, c.lk, at);
    op_sc1(c.red, red_blk_base(i, k - 1), c.sx, c.lr, c.lk, a2);
    mfma8(accT, at, bl);
    mfma8(acc2, a2, bl);
  }
  double* Tx = c.sm + FS_B;
  double* Lx = c.sm + FS_B + kBlk;
  ctile_st_lds(Tx, coff, sT - accT);
  __syncthreads();
  if (stamp) stamp[1] = __builtin_amdgcn_s_memrealtime();
  double a[8], b[8], a0[8], a0p[8];
  // W_k and L[i-2][k-1] are announced together; L[i-2][k] ~0.6 us later: the loads of the first two travel while it is polled
  if (k >= 1) c.w.wait(flow_fw(nbk, k), flow_fl(nbk, i - 2, k - 1)); else c.w.wait(flow_fw(nbk, k));
  if (stamp) stamp[5] = __builtin_amdgcn_s_memrealtime();
  op_sc1(c.ldiag, (size_t)k * kBlk, c.sx, c.lr, c.lk, a);
  if (k >= 1) op_sc1(c.red, red_blk_base(i - 2, k - 1), c.sx, c.lr, c.lk, a0p);
  c.w.wait(flow_fl(nbk, i - 2, k));
  if (stamp) stamp[2] = __builtin_amdgcn_s_memrealtime();
  op_sc1(c.red, red_blk_base(i - 2, k), c.sx, c.lr, c.lk, a0);
  op_lds(Tx, c.sy, c.lr, c.lk, b);
  f64x4 o = {0, 0, 0, 0};
  mfma8(o, a, b);
  ctile_st_sc1(c.red, red_blk_base(i, k), coff, o);
  ctile_st_lds(Lx, coff, o);
  if (k >= 1) mfma8(acc0, a0p, bl);      // the deferred term
  __syncthreads();
  if (stamp) stamp[3] = __builtin_amdgcn_s_memrealtime();
  op_lds(Lx, c.sy, c.lr, c.lk, b);
  op_lds(Lx, c.sx, c.lr, c.lk, a);
  mfma8(acc0, a0, b);
  mfma8(acc2, a, b);
  ctile_st_sc1(c.red, red_blk_base(i, i - 2), coff, s0 - acc0);
  ctile_st_sc1(c.red, red_blk_base(i, i), coff, s2 - acc2);
  flow_drain();
  __syncthreads();
  if (threadIdx.x == 0) {
    flow_publish(c.flow, flow_fl(nbk, i, k), c.epoch);
    flow_publish(c.flow, flow_fh(nbk, i, 0), c.epoch);
    flow_publish(c.flow, flow_fh(nbk, i, 2), c.epoch);
  }
  if (stamp) stamp[4] = __builtin_amdgcn_s_memrealtime();
}

__device__ __forceinline__ void flow_task_rhs(FlowCtx& c, int k, int P) {
  f64x4 s = {0, 0, 0, 0};
  if (c.sred) flow_wait_cams(c, (kNB * k) / 7, min(P - 1, kNB * k + kNB - 1) / 7);
  if (c.sy == 0 && c.lr == 0) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = k * kNB + 16 * c.sx + c.lk + 4 * g;
      if (!c.sred) s[g] = c.rhs[row];
      else if (row < P) s[g] = flow_camsum(c, (row / 7) * 35 + 28 + row % 7);
    }
  }
  f64x4 acc = {0, 0, 0, 0};
  flow_row_sum<1>(c, acc, k, 0, 0, k);
  flow_finish<1>(c, s - acc, 0, k);
}

__device__ __forceinline__ void flow_task_ident(FlowCtx& c, int e, int k) {
  f64x4 acc = {0, 0, 0, 0};
  flow_row_sum<2>(c, acc, k, e, e, k);
  flow_finish<2>(c, -acc, e, k);
}

// dp_e = sum_{c >= e} X[e][c] y_c for identity row e (what ba_inv_apply does in its own launch), run by the workgroup that
// has just published the last block of the row (e < nbk-1) or the last block of y (e = nbk-1); the workgroup whose part
// arrives last updates the cameras (ba:383-392) and prepares the next iteration's.  Thread (row i, slice): every 8th block.
__device__ __forceinline__ void flow_task_dp(FlowCtx& c, const BaDev& d, int e, int cur) {
  const int nbk = c.nbk, tid = threadIdx.x, i = tid & 31, slice = tid >> 5, n = nbk - e;
  double* ys = c.sm + FS_B;                   // y_e .. y_nbk-1
  double* part = c.sm + FS_B + 2 * kBlk;      // [8][33]
  for (int w0 = 0; w0 < n; w0 += 16)          // every block of the row and of y is there
    (void)c.w.wait_terms(flow_fx(nbk, e, e + w0), flow_fy(nbk, e + w0), 0, 2, min(16, n - w0), min(16, n - w0));
  for (int t = tid; t < n * kNB; t += 256) ys[t] = __hip_atomic_load(c.rhs + e * kNB + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  f64x2 w[2][16];
  double s = 0;
  __syncthreads();
  for (int cc = e + slice; cc < nbk; cc += 16) {
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (cc + 8 * h < nbk) {
        const size_t blk = red_blk_base(cc + 8 * h, e) + 2 * i;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) w[h][4 * m + q] = ld2_sc1(c.xinv, blk + m * 256 + q * 64);      // columns 8m+q, 8m+q+4 of row i
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (cc + 8 * h < nbk) {
        const double* yc = ys + (cc + 8 * h - e) * kNB;
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) s = __builtin_fma(w[h][4 * m + q].y, yc[8 * m + q + 4], __builtin_fma(w[h][4 * m + q].x, yc[8 * m + q], s));
      }
  }
  part[slice * 33 + i] = s;
  __syncthreads();
  if (tid < kNB) {
    double t = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) t += part[q * 33 + tid];
    __hip_atomic_store(d.delta + e * kNB + tid, t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  flow_drain();
  __syncthreads();
  int* is_last = c.smi + 15;
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(d.sync_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *is_last = done == nbk - 1;
    if (done == nbk - 1) __hip_atomic_store(d.sync_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next solve
  }
  __syncthreads();
  if (!*is_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (c.sred) c.w.wait(flow_fcost(nbk));      // (this linearisation's has been stored)
  if (tid == 0) *d.iter_count += 1;      // the next linearisation's cost goes to the next slot (sfm_ba_get_stats)
  for (int v = tid; v < d.V; v += 256) {
    double cam[7];
    for (int k = 0; k < 7; ++k) cam[k] = d.cams[7 * v + k] + __hip_atomic_load(d.delta + 7 * v + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ba:383
    const double nq = sqrt(cam[3] * cam[3] + cam[4] * cam[4] + cam[5] * cam[5] + cam[6] * cam[6]);   // ba:388-392
    for (int k = 3; k < 7; ++k) cam[k] /= nq;
    for (int k = 0; k < 7; ++k) d.cams[7 * v + k] = cam[k];
    CamPrep out;
    const int st = cam_prepare(cam, &out);      // ba:323 of the next iteration / ba:412 after the last one
    d.prep[cur ^ 1][v] = out;
    report_status(d.status, st, v);
  }
}

// ---------------------------------------------------------------------------------------------
// The chain.  512 threads.  Waves 0-3 (raised priority) eliminate and never touch global memory: elimination -> X_j rows in LDS ->
// L[j+1][j] (operand tile of W_j read straight from those rows) -> D_j+1 -> next elimination, with barriers of their own group in
// between.  Waves 4-7 do everything that waits for memory: they publish what the elimination waves leave in LDS (X_j,
// W_j in the operand layout, L[j+1][j]: copy, drain, flag) and prepare the next row.  The two groups meet at ONE workgroup barrier per
// step (X_j rows complete, row r prepared) and hand each other the LDS buffers through two step counters.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void flow_lds_wait(const int* word, int value) {
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < value) __builtin_amdgcn_s_sleep(1);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
__device__ __forceinline__ void flow_lds_post(int* word, int value) {      // after a group barrier: the group's LDS writes are complete
  if ((threadIdx.x & 255) == 0) __hip_atomic_store(word, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

__device__ __forceinline__ void flow_chain(FlowCtx& c, const BaDev& d, double lambda, const FlowRed& fr) {
  const int nbk = c.nbk, P = d.P;
  const int tid = threadIdx.x, lane = c.lane, grp = tid >> 8;
  const int sx = c.sx, sy = c.sy, lr = c.lr, lk = c.lk, gw = c.gw;
  double* sm = c.sm;
  double(*Dm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(sm + FS_DM);
  f64x2(*xy)[64] = reinterpret_cast<f64x2(*)[64]>(sm + FS_XY);
  double* Xl = sm + FS_XL;
  int* eflag = c.smi;            // chol_trsm_cols: pair-steps published
  int* pctr = c.smi + 1;         // barrier counter of the preparation waves
  int* ectr = c.smi + 2;         // barrier counter of the elimination waves
  int* la_ready = c.smi + 3;     // steps whose L[j+1][j] sits in LDS
  int* c2_done = c.smi + 4;      // steps whose D_j+1 is formed (the row buffers B0..B2 are free again)
  int ptarget = 0, etarget = 0;
  const int coff = ctile_off(sx, sy, lr, lk);
  unsigned long long* stamp = (d.stamps && nbk <= 16 && (tid == 0 || tid == 256)) ? d.stamps + (tid == 0 ? 0 : 512) : nullptr;

  // D(a, kappa) of diagonal block `blk` from accumulator tile (sx, sy) of D^T: + lambda, identity on the padding
  auto put_d = [&](int blk, const f64x4& v) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int a = 16 * sy + lr, kap = 16 * sx + lk + 4 * g;
      const bool ok_a = blk * NB + a < P, ok_k = blk * NB + kap < P;
      double t = (ok_a && ok_k) ? v[g] : 0.0;
      if (a == kap) t = ok_a ? t + lambda : 1.0;
      Dm[a][kap] = t;
    }
  };
  // a k-interleaved block from LDS to global memory by the 256 preparation threads
  auto push_block = [&](const double* src, FlowBuf dst, size_t base) {
    const int u0 = tid - 256;
#pragma unroll
    for (int h = 0; h < 2; ++h) { const int u = 2 * (u0 + 256 * h); st2_sc1(dst, base + u, *reinterpret_cast<const f64x2*>(src + u)); }
  };
  // preparation waves: the next row's hand-over blocks (and the operand tiles of L[r][r-3]) fetched one step ahead
  f64x2 pf[3][2];
  double pfb[8];
  bool pf_ok = false;
#pragma unroll
  for (int t = 0; t < 3; ++t) { pf[t][0] = f64x2{0, 0}; pf[t][1] = f64x2{0, 0}; }
#pragma unroll
  for (int t = 0; t < 8; ++t) pfb[t] = 0;
  if (c.sred) {
    // Deferred reduce: D_0 is summed here, by all eight waves, straight into LDS -- no task, no flag, no second trip through memory
    // between the slabs and the first elimination.  Thread = (row, column pair): one 16-byte load per slab, all of them in flight.
    const int row = tid >> 4, cp = 2 * (tid & 15);
    const SchurTileRef tr = plan_tile(fr.plan, fr.plan.n_off);
    const int chunks = fr.plan.chunks[tr.cls];
    const double* src = fr.ws + (size_t)tr.first * (kSchurRB * kSchurRB) + row * kSchurRB + cp;
    f64x2 s = {0, 0};
    for (int cb = 0; cb < chunks; cb += 40) {
      f64x2 v[40];
#pragma unroll
      for (int u = 0; u < 40; ++u) v[u] = (cb + u < chunks && cp <= row) ? *reinterpret_cast<const f64x2*>(src + (size_t)(cb + u) * (kSchurRB * kSchurRB)) : f64x2{0, 0};
#pragma unroll
      for (int u = 0; u < 40; ++u) s += v[u];
    }
    if (stamp && tid == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
    flow_wait_cams(c, 0, min(d.V - 1, (kNB - 1) / 7));
    if (stamp && tid == 0) stamp[13] = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int col = cp + h;
      double t = 0;
      if (row < P && col <= row) {
        t = -s[h];
        if (row / 7 == col / 7) { const int ii = row % 7, jj = col % 7; t += flow_camsum(c, (row / 7) * 35 + ii * (ii + 1) / 2 + jj); }
      }
      if (row == col) t = row < P ? t + lambda : 1.0;
      Dm[row][col] = t;
    }
    if (stamp && tid == 0) stamp[14] = __builtin_amdgcn_s_memrealtime();
    if (grp == 0) __builtin_amdgcn_s_setprio(3);
  } else if (grp == 0) { __builtin_amdgcn_s_setprio(3); put_d(0, flow_ld_S(c, 0, 0, sx, sy)); }
  if (tid == 0) { eflag[0] = 0; pctr[0] = 0; ectr[0] = 0; la_ready[0] = 0; c2_done[0] = 0; }
  __syncthreads();

  for (int j = 0; j < nbk; ++j) {
    const int r = j + 1;
    const bool has_r = r < nbk;
    double* Wcur = sm + FS_W + (j & 1) * kBlk;
    const double* Wprev = sm + FS_W + ((j & 1) ^ 1) * kBlk;
    double* LAcur = sm + FS_LA + (j & 1) * kBlk;
    const double* LAprev = sm + FS_LA + ((j & 1) ^ 1) * kBlk;
    double* LBcur = sm + FS_LB + (j & 1) * kBlk;
    const double* LBprev = sm + FS_LB + ((j & 1) ^ 1) * kBlk;
    double* B0 = sm + FS_B, *B1 = sm + FS_B + kBlk, *B2 = sm + FS_B + 2 * kBlk;
    // X_j rows of the elimination: two buffers in turn (the preparation waves read step j's while the elimination waves are in step j+1)
    double(*Xm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(sm + ((j & 1) ? FS_XM2 : FS_XM));
    if (stamp) stamp[8 * j + 0] = __builtin_amdgcn_s_memtime();
    if (stamp && tid == 256) stamp[384 + 8 * j + 0] = __builtin_amdgcn_s_memrealtime();
    if (grp == 0) {
      // ---- elimination of [D_j; I]
      const int row = lane & (NB - 1);
      double cc[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) cc[u] = lane < NB ? Dm[row][8 * gw + u] : ((row == 8 * gw + u) ? 1.0 : 0.0);
      chol_trsm_cols(cc, xy, eflag, lane, gw);
      if (lane >= NB) {
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = 8 * gw + u; Xm[row][k] = (k >= row) ? cc[u] : 0.0; }
      }
    } else if (has_r) {
      // ---- preparation of row r = j + 1 beside the elimination: wave 4 + s owns tile s of every product.  (Its 64-cycle matrix
      // instructions share the SIMDs with the elimination waves and cost the elimination ~2 000 cycles per step; keeping a wave
      // from issuing them while the elimination wave of its SIMD owns the pivot chain, or giving the work to the two waves whose
      // SIMDs are idle after the first chain segments, only made the preparation the longer path: EXPERIMENTS.md.)
      const int u0 = tid - 256;
      if (r >= 3 && !pf_ok) {
        c.w.wait(flow_fh(nbk, r, 0), flow_fh(nbk, r, 1), flow_fh(nbk, r, 2));
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) pf[t][h] = ld2_sc1(c.red, red_blk_base(r, r - 2 + t) + 2 * (u0 + 256 * h));
        op_sc1(c.red, red_blk_base(r, r - 3), sy, lr, lk, pfb);
      }
      if (stamp) { stamp[8 * j + 5] = __builtin_amdgcn_s_memtime(); stamp[384 + 8 * j + 1] = __builtin_amdgcn_s_memrealtime(); }
      flow_lds_wait(c2_done, j);            // step j-1 has read the last of B0..B2
      if (r >= 3) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) *reinterpret_cast<f64x2*>(B0 + t * kBlk + 2 * (u0 + 256 * h)) = pf[t][h];
      } else {
        flow_wait_S(c, r, 0, r);
        if (r == 2) ctile_st_lds(B0, coff, flow_ld_S(c, 2, 0, sx, sy));
        ctile_st_lds(B1, coff, flow_ld_S(c, r, r - 1, sx, sy));
        ctile_st_lds(B2, coff, flow_ld_S(c, r, r, sx, sy));
      }
      pf_ok = false;
      if (stamp) { stamp[256 + 8 * j + 0] = __builtin_amdgcn_s_memtime(); stamp[384 + 8 * j + 2] = __builtin_amdgcn_s_memrealtime(); }
      flow_group_sync(pctr, ptarget);
      if (stamp) stamp[256 + 8 * j + 1] = __builtin_amdgcn_s_memtime();
      f64x4 acc1 = {0, 0, 0, 0};
      double a[8], b[8];
      if (r >= 3) {
        // (r, r-1) -= L[r][r-3] L[r-1][r-3]^T (kept in the accumulator until the second term is there)
        op_lds(LBprev, sx, lr, lk, a);
        mfma8(acc1, a, pfb);
      }
      if (r >= 2) {
        // L[r][r-2]^T = W_j-1 T^T
        op_lds(Wprev, sx, lr, lk, a);
        op_lds(B0, sy, lr, lk, b);
        f64x4 o = {0, 0, 0, 0};
        mfma8(o, a, b);
        ctile_st_lds(LBcur, coff, o);
        ctile_st_sc1(c.red, red_blk_base(r, r - 2), coff, o);
        if (stamp) stamp[256 + 8 * j + 3] = __builtin_amdgcn_s_memtime();
        flow_group_sync(pctr, ptarget);
        if (stamp) stamp[256 + 8 * j + 4] = __builtin_amdgcn_s_memtime();
        // (r, r-1) -= L[r][r-2] L[r-1][r-2]^T ; (r, r) -= L[r][r-2] L[r][r-2]^T (lower tiles)
        op_lds(LBcur, sy, lr, lk, b);
        op_lds(LAprev, sx, lr, lk, a);
        mfma8(acc1, a, b);
        ctile_st_lds(B1, coff, ctile_ld_lds(B1, coff) - acc1);
        if (gw != 2) {
          op_lds(LBcur, sx, lr, lk, a);
          f64x4 acc2 = {0, 0, 0, 0};
            mfma8(acc2, a, b);
          ctile_st_lds(B2, coff, ctile_ld_lds(B2, coff) - acc2);
        }
      }
      if (stamp) stamp[8 * j + 6] = __builtin_amdgcn_s_memtime();
      // the next row's hand-over, if it is there already: its loads travel while the elimination finishes
      if (r + 1 < nbk && r + 1 >= 3 && c.w.poll(flow_fh(nbk, r + 1, 0), flow_fh(nbk, r + 1, 1), flow_fh(nbk, r + 1, 2))) {
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int h = 0; h < 2; ++h) pf[t][h] = ld2_sc1(c.red, red_blk_base(r + 1, r - 1 + t) + 2 * (u0 + 256 * h));
        op_sc1(c.red, red_blk_base(r + 1, r - 2), sy, lr, lk, pfb);
        pf_ok = true;
      }
    }
    __syncthreads();                                                  // B: X_j rows in Xm, row r prepared
    if (stamp) stamp[8 * j + 1] = __builtin_amdgcn_s_memtime();
    if (stamp) stamp[8 * j + 2] = __builtin_amdgcn_s_memtime();
    if (grp == 0) {
      if (!has_r) break;
      // L[r][j]^T = W_j T^T with W_j(c, kappa) = X_j(kappa, c) taken straight from the rows the elimination left (8-byte reads, two
      // lanes per bank pair): the elimination waves do not wait for the operand layout of X_j / W_j, which only the
      // preparation waves need (publishing, the next step's product)
      double a[8], b[8];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) a[ks] = Xm[lk + 4 * ks][16 * sx + lr];
      op_lds(B1, sy, lr, lk, b);
      f64x4 o = {0, 0, 0, 0};
      mfma8(o, a, b);
      ctile_st_lds(LAcur, coff, o);
      if (tid == 0) eflag[0] = 0;
      flow_group_sync(ectr, etarget);
      flow_lds_post(la_ready, j + 1);
      if (stamp) stamp[8 * j + 3] = __builtin_amdgcn_s_memtime();
      if (gw != 2) {
        // D_r -= L[r][j] L[r][j]^T, lower tiles only
        op_lds(LAcur, sx, lr, lk, a);
        op_lds(LAcur, sy, lr, lk, b);
        f64x4 acc = {0, 0, 0, 0};
        mfma8(acc, a, b);
        put_d(r, ctile_ld_lds(B2, coff) - acc);
      }
      flow_group_sync(ectr, etarget);
      flow_lds_post(c2_done, j + 1);
      if (stamp) stamp[8 * j + 4] = __builtin_amdgcn_s_memtime();
    } else {
      // X_j and W_j = X_j^T into the operand layout (LDS): W_j for the next step's preparation, both to be published (ldiag[j]: the
      // tasks; xinv(j, j): ba_inv_apply and the tasks of identity row j), then L[r][j] as soon as the elimination waves have it
      {
        const int u0 = tid - 256;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int idx = u0 + 256 * h, m = idx >> 7, q = (idx >> 5) & 3, col = idx & 31, kap = 8 * m + q;
          const int off = m * 256 + q * 64 + 2 * col;
          *reinterpret_cast<f64x2*>(Wcur + off) = f64x2{Xm[kap][col], Xm[kap + 4][col]};
          *reinterpret_cast<f64x2*>(Xl + off) = f64x2{Xm[col][kap], Xm[col][kap + 4]};
        }
      }
      flow_group_sync(pctr, ptarget);
      push_block(Xl, c.xinv, red_blk_base(j, j));
      push_block(Wcur, c.ldiag, (size_t)j * kBlk);
      flow_drain();
      flow_group_sync(pctr, ptarget);
      // SFM_OPT_DEBUG bit 8192 (test of the bounded waits): W_1 is never announced, every wait gives up after 20 000 polls
      if (tid == 256 && !((d.debug & 8192) && j == 1)) {
        flow_publish(c.flow, flow_fw(nbk, j), c.epoch);
        flow_publish(c.flow, flow_fx(nbk, j, j), c.epoch);
        if (has_r && r >= 2) flow_publish(c.flow, flow_fl(nbk, r, r - 2), c.epoch);      // stored by the preparation above, drained here
        if (stamp) stamp[384 + 8 * j + 3] = __builtin_amdgcn_s_memrealtime();
      }
      if (!has_r) break;
      flow_lds_wait(la_ready, j + 1);
      push_block(LAcur, c.red, red_blk_base(r, j));
      flow_drain();
      flow_group_sync(pctr, ptarget);
      if (tid == 256) { flow_publish(c.flow, flow_fl(nbk, r, j), c.epoch); if (stamp) stamp[384 + 8 * j + 4] = __builtin_amdgcn_s_memrealtime(); }
      if (stamp) stamp[8 * j + 7] = __builtin_amdgcn_s_memtime();
    }
  }
}

// Workgroup 0 is the chain; the others take the tasks of the table one after the other (sorted by column: a task waits only for
// tasks before it).
__global__ __launch_bounds__(512) void ba_chol_flow_kernel(BaDev d, unsigned* flow, const FlowTask* tasks, int ntasks, double lambda, int cur, FlowRed fr) {
  extern __shared__ __attribute__((aligned(16))) double flow_sm[];
  const int tid = threadIdx.x;
  FlowCtx c;
  c.red = flow_buf(d.red); c.xinv = flow_buf(d.xinv); c.ldiag = flow_buf(d.ldiag);
  c.redp = d.red; c.rhs = d.red + red_rhs_off(d.nbk);
  c.flow = flow; c.nbk = d.nbk;
  c.epoch = __hip_atomic_load(flow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
  c.sm = flow_sm; c.smi = reinterpret_cast<int*>(flow_sm + FS_INT);
  c.lane = tid & 63; c.lr = c.lane & 15; c.lk = c.lane >> 4;
  c.gw = (tid >> 6) & 3; c.sx = c.gw >> 1; c.sy = c.gw & 1;
  c.w = FlowWait{flow, c.epoch, d.status, false, (d.debug & 8192) ? 20000u : kFlowSpinLimit};
  c.stamps = (d.stamps && tid == 0 && d.nbk <= 16) ? d.stamps : nullptr;      // the 1024 stamp slots are laid out for up to 16 block columns
  c.sred = fr.ws != nullptr; c.camsum = fr.camsum; c.ncam35 = 35 * d.V;
  // Roles and tasks are TAKEN (one agent-scope ticket each), not dealt by workgroup index: the first workgroup to arrive is the chain
  // (workgroups of a launch start over several microseconds, and not in index order), the others take the tasks in table order.  A
  // task waits only for the chain and for tasks earlier in the table, and those have been taken by workgroups that are running -- so
  // the launch makes progress with any number of resident workgroups (two problems on two streams, several ranks rehearsed on one
  // GPU), not only when all of its workgroups hold a CU at the same time.
  int* slot = c.smi + 20;
  const bool dealt = (d.debug & 4096) != 0;      // SFM_OPT_DEBUG bit 4096 (A/B only): roles and tasks by workgroup index; needs every workgroup resident
  if (tid == 0) *slot = dealt ? (int)blockIdx.x : (int)__hip_atomic_fetch_add(flow + kFlowTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  const int first = *slot;
  if (first == 0) {
    if (c.stamps) { c.stamps[5] = __builtin_amdgcn_s_memtime(); c.stamps[6] = __builtin_amdgcn_s_memrealtime(); }      // kernel entry: step 0 starts when D_0 is there
    flow_chain(c, d, lambda, fr);
  } else {
    if (tid >= 256) return;      // tasks are run by four waves
    for (int round = 0;; ++round) {
      if (round > 0) {
        if (tid == 0) *slot = dealt ? first + round * ((int)gridDim.x - 1) : (int)__hip_atomic_fetch_add(flow + kFlowTicket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
      }
      const int t = *slot - 1;      // every task below passes a barrier before this word is written again
      if (t >= ntasks) break;
      const FlowTask tk = tasks[t];
      switch (tk.type) {
        case FLOW_T1: flow_task_t1(c, tk.i, tk.k); break;
        case FLOW_CLOSER: flow_task_closer(c, tk.i); break;
        case FLOW_H1: flow_task_h1(c, tk.i); break;
        case FLOW_CAMSUM: flow_task_camsum(c, d, fr, tk.i, tk.k); break;
        case FLOW_SRED: flow_task_sred(c, d, fr, tk.i, tk.k, tk.key & 3); break;
        case FLOW_COST: flow_task_cost(c, d, fr); break;
        case FLOW_RHS:
          flow_task_rhs(c, tk.k, d.P);
          if (cur >= 0 && tk.k == d.nbk - 1) flow_task_dp(c, d, d.nbk - 1, cur);
          break;
        default:
          flow_task_ident(c, tk.i, tk.k);
          if (cur >= 0 && tk.k == d.nbk - 1) flow_task_dp(c, d, tk.i, cur);
          break;
      }
    }
  }
  // the last workgroup to leave closes the epoch
  __syncthreads();
  if (tid == 0) {
    const unsigned done = __hip_atomic_fetch_add(flow + kFlowDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (done == gridDim.x - 1) {
      __hip_atomic_store(flow + kFlowDone, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flow + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flow + kFlowTicket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(flow, c.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// The task table of a problem with nbk block columns, in the order the workgroups take it.  Keys: 64 x (the last column a task waits
// for) + its place among that column's tasks; a task waits only for the chain and for tasks before it in the table.
// V > 0: the table of the deferred reduce (V cameras): CAMSUM and SRED tasks ahead of their readers -- block (i, k) of S is read by
// the chain (i < 3), by the task of L[i][k] (k <= i-4), by the closer of row i (k = i-3, i-2, i) or by its hand-over task (k = i-1).
inline std::vector<FlowTask> flow_build_tasks(int nbk, int V = 0) {
  std::vector<FlowTask> t;
  if (V > 0) {
    t.push_back(FlowTask{FLOW_COST, 0, 0, 9});      // behind the first block rows' sums
    // ahead of everything: what the chain's first steps read, block row by block row (the camera sums of a row, then its blocks of S)
    for (int c = 0; c < V; ++c) {
      const int ic = (7 * c) / kNB;
      for (int part = 0; part < 4; ++part) t.push_back(FlowTask{FLOW_CAMSUM, c, part, 64 * std::max(0, ic - 4) + (ic <= 4 ? 2 * ic : 0)});
    }
    for (int k = 0; k < nbk; ++k)
      for (int i = std::max(k, 1); i < nbk; ++i)      // (block (0, 0) is summed by the chain itself)
        for (int q = 0; q < 4; ++q) {
          const int g = k <= i - 4 ? k : std::max(0, i - 4);
          t.push_back(FlowTask{FLOW_SRED, i, k, 64 * g + (g == 0 ? 2 * std::min(i, 15) + 1 : 1)});
        }
  }
  for (int i = 3; i < nbk; ++i) {
    t.push_back(FlowTask{FLOW_CLOSER, i, i - 3, 64 * (i - 3) + 40});
    t.push_back(FlowTask{FLOW_H1, i, i - 1, 64 * std::max(0, i - 4) + 48});      // after the block (i, i-4) of the same column
    for (int k = 0; k <= i - 4; ++k) t.push_back(FlowTask{FLOW_T1, i, k, 64 * k + 44});
  }
  for (int k = 0; k < nbk; ++k) t.push_back(FlowTask{FLOW_RHS, 0, k, 64 * k + 52});
  for (int e = 0; e + 1 < nbk; ++e)
    for (int k = e + 1; k < nbk; ++k) t.push_back(FlowTask{FLOW_IDENT, e, k, 64 * k + 56});
  std::stable_sort(t.begin(), t.end(), [](const FlowTask& x, const FlowTask& y) { return x.key < y.key; });
  int q = 0;
  for (FlowTask& x : t)
    if (x.type == FLOW_SRED) x.key = (x.key & ~3) | (q++ & 3);      // the four quarters of a block stay adjacent (stable sort): 0, 1, 2, 3
  return t;
}

}  // namespace sfm
