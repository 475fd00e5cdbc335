// sfm_ba_solve.hip — the reduced camera solve of one bundle-adjustment iteration (gfx950):
//   dp = inv(A - B D^-1 B^T) (ep - B D^-1 ex)        (ba_processor.py:382)
//   cams += dp ; q <- q / |q|                        (ba_processor.py:383-392)
// as a blocked right-looking Cholesky of S + lambda I on the packed 32x32 blocks of BaDev::red (layout:
// sfm_ba.h), the right-hand side carried as an extra block row, and dp = L^-T y either as one product with the
// explicitly carried L^-T or as a blocked back substitution.
//
//   ba_chol_step   one launch per block column j.  Column-role workgroup (r, j), r = j .. nbk (nbk = the rhs
//                  row): T = A[r][j] - L[r][j-1] L[j][j-1]^T and D = A[j][j] - L[j][j-1] L[j][j-1]^T on the matrix
//                  pipe, the operand tiles loaded straight from the k-interleaved blocks into registers (four
//                  16-byte loads per 16x32 tile, no LDS staging, no barrier before the MFMAs); [D; T] then goes
//                  through LDS to a four-wave, column-split elimination (chol_trsm_cols) that factors D and solves X L_d^T = T in the same instruction stream (lanes 0-31 = rows
//                  of D, lanes 32-63 = rows of T), two columns per step (2x2 pivots: the two reciprocal square
//                  roots of a step are independent).  Trailing-role workgroups give the 64x64
//                  super-tiles right of column j the update of panel j-1, again register-to-register.
//                  Identity rows (nbk <= 52): the P x P identity rides through the same column steps as extra block
//                  rows, so the launches that factor S also leave X = L^-T behind.
//   ba_inv_apply   dp = X y (y = L^-1 rhs sits in the rhs row): one launch of independent block rows; the last workgroup
//                  updates the cameras and prepares the next iteration's.
//   ba_back_solve  (nbk > 52, or SFM_OPT_DEBUG bit 512) L^T dp = y block row by block row with the inverse diagonal
//                  factors the factorisation left behind, ba_back_update between groups of block rows.
//   ba_small_solve P <= 56: the whole solve in one single-workgroup launch (whole-matrix multi-wave elimination).
#include <algorithm>
#include <map>
#include <mutex>

#include "sfm_ba.h"

namespace sfm {

constexpr int NB = kNB;
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef double f64x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}

// The eight k-step values of one lane of a v_mfma_f64_16x16x4 A (or B) operand tile taken from a 32x32 block:
// lane (lr = lane & 15, lk = lane >> 4) of 16-row tile `tile` holds element [16 tile + lr][4 ks + lk] for ks = 0..7:
// four 16-byte pairs (ks = 2m, 2m + 1), each load instruction reading 4 x 256 contiguous bytes across the wave.
__device__ __forceinline__ void load_operand(const double* __restrict__ blk, int tile, int lr, int lk, double (&o)[8]) {
  const double* p = blk + lk * 64 + (16 * tile + lr) * 2;
#pragma unroll
  for (int m = 0; m < 4; ++m) { const f64x2 v = *reinterpret_cast<const f64x2*>(p + 256 * m); o[2 * m] = v.x; o[2 * m + 1] = v.y; }
}
// the rhs "block row" has one valid row: the 32 values of block column c sit at rhs[32 c ..]
__device__ __forceinline__ void load_operand_rhs(const double* __restrict__ seg, int tile, int lr, int lk, double (&o)[8]) {
  const bool row0 = (16 * tile + lr) == 0;
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) { const double v = seg[4 * ks + lk]; o[ks] = row0 ? v : 0.0; }
}

// ---------------------------------------------------------------------------------------------
// The elimination of one column step, spread over the FOUR waves of the column workgroup (one per SIMD): lanes 0..31 = rows of
// the SPD block D (lower part valid), lanes 32..63 = rows of T; on return lanes 0..31 hold the rows of L_d and lanes 32..63
// the rows of X = T L_d^-T.  Two columns per step: with the 2x2 pivot [[a, b], [b, c]] of rows / columns j, j+1,
//   r1 = 1/sqrt(a), r2 = 1/sqrt(a c - b^2)    -- independent of each other: ONE reciprocal-square-root latency
//   l11 = a r1, l21 = b r1, 1/l22 = r2 l11    (l22 = sqrt(c - l21^2) = sqrt(det / a))
//   every row:  x = a_j r1,  y = (a_j+1 - x l21) / l22,   a_k -= x X_k + y Y_k  for k > j+1  (X_k, Y_k = x, y of row k)
// A single wave issues one FP64 instruction every ~7 cycles whatever the dependencies (tools/microbench_solve.hip), so
// a one-wave elimination is bound by its ~85 instructions per pair-step, half of them the update of the far columns
// (rounds 1-2 carried that variant behind a debug switch; its 192 extra VGPRs capped the whole kernel at two waves per
// SIMD and it is gone).  Here wave w owns columns 8w .. 8w+7 of all 64 rows (lane = row, 8 registers).  The pivot chain
// walks through the waves: wave `seg` runs the four pair-steps of its columns (pivot broadcast, the two reciprocal
// square roots, x / y of every row, its own remaining columns) and publishes (x, y) of every row in LDS followed
// by a step counter; the waves to its right apply each published step to their eight columns as it appears
// (one 16-byte broadcast read + two FMAs per column) and take the chain over once it reaches them.  The flag and the
// data are LDS writes of one wave, which the LDS executes in order; readers poll the counter, then read.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void chol_trsm_cols(double (&c)[8], f64x2 (*xy)[64], int* flag, int lane, int wave) {
#pragma unroll
  for (int seg = 0; seg < 4; ++seg) {
    if (wave == seg) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int j = 2 * t, s = 4 * seg + t, l0 = 8 * seg + j;      // l0: the lane that holds row (= column) 8 seg + j of D
        const double pa = lane_bcast(c[j], l0), pb = lane_bcast(c[j], l0 + 1), pc = lane_bcast(c[j + 1], l0 + 1);
        const double det = __builtin_fma(pa, pc, -(pb * pb));
        const double r1 = rsqrt_nr(pa), r2 = rsqrt_nr(det);
        const double l11 = pa * r1, l21 = pb * r1, i22 = r2 * l11;
        const double x = c[j] * r1;
        const double y = (c[j + 1] - x * l21) * i22;
        c[j] = x; c[j + 1] = y;
        xy[s][lane] = f64x2{x, y};
        if (t < 3) {
          // the next pair's two columns at once through register broadcasts, my other columns through LDS
          const double x2 = lane_bcast(x, l0 + 2), y2 = lane_bcast(y, l0 + 2);
          const double x3 = lane_bcast(x, l0 + 3), y3 = lane_bcast(y, l0 + 3);
          c[j + 2] = __builtin_fma(-y, y2, __builtin_fma(-x, x2, c[j + 2]));
          c[j + 3] = __builtin_fma(-y, y3, __builtin_fma(-x, x3, c[j + 3]));
#pragma unroll
          for (int u = j + 4; u < 8; ++u) { const f64x2 q = xy[s][8 * seg + u]; c[u] = __builtin_fma(-y, q.y, __builtin_fma(-x, q.x, c[u])); }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (lane == 0) __hip_atomic_store(flag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    } else if (wave > seg) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int s = 4 * seg + t;
        while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const f64x2 own = xy[s][lane];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const f64x2 q = xy[s][8 * wave + u]; c[u] = __builtin_fma(-own.y, q.y, __builtin_fma(-own.x, q.x, c[u])); }
      }
    }
  }
}

// Trailing role of a column step: a 64x64 super-tile (2x2 blocks, one block per wave) of the blocks right of
// column j gets the previous panel's update A[r][c] -= L[r][j-1] L[c][j-1]^T.  No LDS: every wave loads its four
// operand tiles and the old block in the MFMA layouts directly.
// INV = true: the same update for the identity rows, E[e][c] -= X[e][j-1] L[c][j-1]^T (e <= j-1 < c; rectangle, not
// triangle).  E starts as the identity, so the block (e, c), c > e, is all zero until panel e contributes: the first
// update (j - 1 == e) writes instead of accumulating and the buffer never needs clearing.
// The super-tile is named by ABSOLUTE pair coordinates (block rows 2 sr, 2 sr + 1; block columns 2 sc, 2 sc + 1), so that the
// same workgroup slot -- and with it the same XCD, see trail_pick -- updates a block in every column step.
template <bool INV>
__device__ __forceinline__ void chol_trailing_supertile(const BaDev& d, int j, int sr, int sc) {
  const int nbk = d.nbk;
  double* red = d.red;
  double* rhs = d.red + red_rhs_off(nbk);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int r = 2 * sr + (wave >> 1), c = 2 * sc + (wave & 1);
  if (c < j + 1) return;
  if (INV) { if (!(c <= nbk - 1 && r <= j - 1)) return; }
  else if (!(c <= nbk - 1 && r >= c && r <= nbk)) return;
  const bool is_rhs = !INV && r == nbk;
  const bool fresh = INV && r == j - 1;
  const int lr = lane & 15, lk = lane >> 4;
  double* blk = is_rhs ? nullptr : (INV ? d.xinv + red_blk_base(c, r) : red + red_blk_base(r, c));
  const double* Lc = red + red_blk_base(c, j - 1);
  // every global load of the block update is issued before the first MFMA; the two 16-column strips of the block are
  // finished one after the other so that only half of the accumulators and old values are live at a time (the kernel's
  // register budget decides how many trailing workgroups a CU holds, and at 200 cameras the trailing traffic, not the
  // pivot chain, bounds the middle steps)
  double la[2][8];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (is_rhs) load_operand_rhs(rhs + (j - 1) * NB, t, lr, lk, la[t]);
    else load_operand(INV ? d.xinv + red_blk_base(j - 1, r) : red + red_blk_base(r, j - 1), t, lr, lk, la[t]);
  }
#pragma unroll 1
  for (int sy = 0; sy < 2; ++sy) {
    double lb[8], old[2][4];
    load_operand(Lc, sy, lr, lk, lb);
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = 16 * sx + lk + 4 * g, col = 16 * sy + lr;
        old[sx][g] = is_rhs ? (i == 0 ? rhs[c * NB + col] : 0.0) : (fresh ? 0.0 : blk[red_blk_off(i, col)]);
      }
    f64x4 acc[2] = {f64x4{0, 0, 0, 0}, f64x4{0, 0, 0, 0}};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[0][ks], lb[ks], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(la[1][ks], lb[ks], acc[1], 0, 0, 0);
    }
#pragma unroll
    for (int sx = 0; sx < 2; ++sx)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int i = 16 * sx + lk + 4 * g, col = 16 * sy + lr;
        const double v = old[sx][g] - acc[sx][g];
        if (is_rhs) { if (i == 0) rhs[c * NB + col] = v; }
        else blk[red_blk_off(i, col)] = v;
      }
  }
}

// XCD-affine placement of the trailing work (round 4).  Workgroups are dealt round-robin to the 8 XCDs and an XCD's L2 keeps
// the lines a launch touched for the next launch on the stream (tools/microbench_l2_across_launches.hip: a 16 MB
// read-modify-write pass takes 1.4 us when the same XCD made the previous pass, 9 us when another did).  A trailing block is
// read-modified-written in every column step until its own column comes up, so its super-tile (absolute pair coordinates
// (R, C)) always goes to XCD (R + C) % 8: the trailing region of a step's grid is 8 x slots workgroups, workgroup
// 8 slot + x takes the slot-th super-tile of XCD x in the enumeration below (or leaves at once).  At 200 cameras the
// read-modify-write traffic of the trailing blocks was 90 us of the 411 us solve (profiles/r4/ablate_trailing_c4share.txt).
// trail_pick: S-part, super-tiles (R >= C) with block columns >= j + 1 (C >= (j+1)/2), rows up to the rhs row nbk.
__host__ __device__ inline bool trail_pick(int nbk, int j, int x, int slot, int* R_out, int* C_out, int* count_out) {
  const int cmin = (j + 1) / 2, cmax = (nbk - 1) / 2, rmax = nbk / 2;
  int count = 0;
  for (int C = cmin; C <= cmax; ++C) {
    const int first = C + ((x - 2 * C) % 8 + 8) % 8;          // smallest R >= C with (R + C) % 8 == x
    if (first > rmax) continue;
    const int n = (rmax - first) / 8 + 1;
    if (slot >= 0 && slot < count + n) { *R_out = first + 8 * (slot - count); *C_out = C; return true; }
    count += n;
  }
  if (count_out) *count_out = count;
  return false;
}
// inv_pick: identity rows, pairs E (block rows 2E, 2E + 1 <= j - 1) x pairs C (block columns j + 1 .. nbk - 1).
__host__ __device__ inline bool inv_pick(int nbk, int j, int x, int slot, int* E_out, int* C_out, int* count_out) {
  const int cmin = (j + 1) / 2, cmax = (nbk - 1) / 2, emax = (j - 1) / 2;
  int count = 0;
  if (j >= 1 && cmin <= cmax) {
    for (int E = 0; E <= emax; ++E) {
      const int first = cmin + ((x - E - cmin) % 8 + 8) % 8;  // smallest C >= cmin with (E + C) % 8 == x
      if (first > cmax) continue;
      const int n = (cmax - first) / 8 + 1;
      if (slot >= 0 && slot < count + n) { *E_out = E; *C_out = first + 8 * (slot - count); return true; }
      count += n;
    }
  }
  if (count_out) *count_out = count;
  return false;
}
// slots per XCD of the two trailing regions of step j (host: grid size; the kernel gets them as arguments)
struct StepSlots { int trail, itrail; };      // workgroup slots per XCD of the two trailing regions of a step's grid
inline StepSlots step_slots(int nbk, int j, bool with_inv) {
  StepSlots z{0, 0};
  int r = 0, c = 0;
  for (int x = 0; x < 8; ++x) {
    int n = 0;
    if (j > 0) { (void)trail_pick(nbk, j, x, -1, &r, &c, &n); z.trail = n > z.trail ? n : z.trail; }
    n = 0;
    if (with_inv && j > 0) { (void)inv_pick(nbk, j, x, -1, &r, &c, &n); z.itrail = n > z.itrail ? n : z.itrail; }
  }
  return z;
}

// Workgroup roles of column step j (with_inv = the identity rows are carried: dp = X y replaces the back substitution):
//   [0, ncol)                 column role of block rows j .. nbk (nbk = the rhs row)
//   [.., + j)                 (with_inv) column role of the identity rows e = 0 .. j-1:  X[e][j] = (E[e][j] - X[e][j-1] L[j][j-1]^T) L_d^-T
//   padding up to a multiple of 8, then, XCD-affine (trail_pick / inv_pick: workgroup 8 slot + x):
//   [.., + 8 slots.trail)     trailing super-tiles of the blocks right of column j
//   [.., + 8 slots.itrail)    trailing super-tiles of the identity rows
// (The column roles stay first in the grid, in row order: placing them on the XCD that last updated their block as well was
//  measured slower -- solve 369 vs 360 us at the C4 share, 85.5 vs 83.6 at C3.)
constexpr int kCholSq = NB * (NB + 1);
constexpr int kCholArena = 2 * kCholSq + (NB / 2) * 64 * 2 + 2;      // doubles of LDS a column-role workgroup needs

__device__ __forceinline__ void chol_step_body(const BaDev& d, int j, double lambda, int with_inv, int role_in, double* arena,
                                               StepSlots slots) {
  constexpr int kSq = kCholSq;
  const int P = d.P, nbk = d.nbk;
  const int ncol = nbk - j + 1;
  const int ncolumn_roles = ncol + (with_inv ? j : 0);
  int role = role_in;
  bool inv_row = false;
  if (role >= ncol) {
    if (role >= ncolumn_roles) {
      const int base = (ncolumn_roles + 7) & ~7;
      int t = role - base;
      if (t < 0) return;                                   // padding
      int a = 0, b = 0;
      if (t < 8 * slots.trail) {
        if (trail_pick(nbk, j, t & 7, t >> 3, &a, &b, nullptr)) chol_trailing_supertile<false>(d, j, a, b);
        return;
      }
      t -= 8 * slots.trail;
      if (t < 8 * slots.itrail && inv_pick(nbk, j, t & 7, t >> 3, &a, &b, nullptr)) chol_trailing_supertile<true>(d, j, a, b);
      return;
    }
    role -= ncol;
    inv_row = true;                        // identity rows, column role: role = e
  }
  // column roles are the launch's critical path (the dependent pivot chain): when a trailing workgroup shares the
  // CU, its MFMA waves wait
  __builtin_amdgcn_s_setprio(2);
  double(*Tm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(arena);
  double(*Dm)[NB + 1] = reinterpret_cast<double(*)[NB + 1]>(arena + kSq);
  f64x2(*xy)[64] = reinterpret_cast<f64x2(*)[64]>(arena + 2 * kSq);     // [pair-step][lane] = (x, y)
  int* flag = reinterpret_cast<int*>(arena + 2 * kSq + (NB / 2) * 64 * 2);      // pair-steps published (multi-wave elimination)
  static_assert((2 * kSq) % 2 == 0, "xy must be 16-byte aligned");
  const int r = inv_row ? nbk + 1 + role : j + role;      // nbk + 1 + e marks an identity row (never equal to j or nbk)
  const int e_row = role;
  const bool is_rhs = r == nbk;
  const bool fresh = inv_row && e_row == j - 1;           // E[e][j] is still all zero: nothing to load
  // diagnostic stamps (SFM_OPT_DEBUG bit 8): shader-clock reads of one column workgroup's phases
  unsigned long long* stamp = (d.stamps && !inv_row && role == 1 && threadIdx.x == 0) ? d.stamps + 8 * j : nullptr;
  if (stamp) stamp[0] = __builtin_amdgcn_s_memtime();
  double* red = d.red;
  double* rhs = d.red + red_rhs_off(nbk);
  double* Tblk = inv_row ? d.xinv + red_blk_base(j, e_row) : red + red_blk_base(is_rhs ? nbk - 1 : r, j);     // never dereferenced for the rhs row
  const double* Dblk = red + red_blk_base(j, j);
  const int r0 = inv_row ? 0 : r * NB, c0 = j * NB, j0 = j * NB;      // identity rows: every row is a real one
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool need_d = r != j;

  // This wave's 16x16 part of the 32x32 block, in the C/D layout of v_mfma_f64_16x16x4_f64:
  // element reg of lane l is (row = 16 sx + (l >> 4) + 4 reg, col = 16 sy + (l & 15)).
  const int sx = wave >> 1, sy = wave & 1;
  const int lr = lane & 15, lk = lane >> 4;
  const int ocol = 16 * sy + lr;
  const bool col_ok = c0 + ocol < P;

  // Every global load of the step is issued before the first wait: the old block values in the C/D layout and, from
  // the second step on, the three operand tiles of the previous panel in the A / B layout.  Rows and columns
  // beyond P read as zero (the buffer is cleared once per iteration and never written there).
  double aT[4], aD[4] = {0, 0, 0, 0};
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = 16 * sx + lk + 4 * g;
    aT[g] = is_rhs ? (i == 0 ? rhs[c0 + ocol] : 0.0) : (fresh ? 0.0 : Tblk[red_blk_off(i, ocol)]);
    if (need_d) aD[g] = Dblk[red_blk_off(i, ocol)];
  }
  if (j > 0) {
    double la[8], lb[8], lja[8];
    const double* Lj = red + red_blk_base(j, j - 1);
    load_operand(Lj, sy, lr, lk, lb);
    if (is_rhs) load_operand_rhs(rhs + (j - 1) * NB, sx, lr, lk, la);
    else load_operand(inv_row ? d.xinv + red_blk_base(j - 1, e_row) : red + red_blk_base(r, j - 1), sx, lr, lk, la);
    if (need_d) load_operand(Lj, sx, lr, lk, lja);
    if (stamp) stamp[1] = __builtin_amdgcn_s_memtime();
    // previous-panel update on the matrix pipe: T -= L[r][j-1] L[j][j-1]^T, D -= L[j][j-1] L[j][j-1]^T.
    // A operand: lane l holds A[row = l&15][k = l>>4]; B operand: B[k = l>>4][col = l&15] = L[j][j-1][col][k].
    f64x4 pT = {0, 0, 0, 0}, pD = {0, 0, 0, 0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      pT = __builtin_amdgcn_mfma_f64_16x16x4f64(la[ks], lb[ks], pT, 0, 0, 0);
      if (need_d) pD = __builtin_amdgcn_mfma_f64_16x16x4f64(lja[ks], lb[ks], pD, 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) { aT[g] -= pT[g]; aD[g] -= pD[g]; }
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int i = 16 * sx + lk + 4 * g;
    const bool row_ok = is_rhs ? (i == 0) : (inv_row || r0 + i < P);
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
  if (tid == 0) *flag = 0;
  if (stamp) stamp[2] = __builtin_amdgcn_s_memtime();
  __syncthreads();
  const double(*src)[NB + 1] = lane < NB ? Dm : Tm;
  const int row = lane & (NB - 1);
  {
    // four-wave elimination: this wave's eight columns of all 64 rows
    unsigned long long* stamp3 = (d.stamps && !inv_row && role == 1 && tid == 192) ? d.stamps + 8 * j : nullptr;
    double c[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) c[u] = src[row][8 * wave + u];
    if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
    chol_trsm_cols(c, xy, flag, lane, wave);
    if (stamp3) { asm volatile("" :: "v"(c[7])); stamp3[4] = __builtin_amdgcn_s_memtime(); }
    if (lane >= NB) {
      if (r == j) {
        // row `row` of X = L_d^-T (upper triangular), stored k-major: ldiag[j][k][i] = X[i][k]
        double* out = d.ldiag + (size_t)j * NB * NB + row;
#pragma unroll
        for (int u = 0; u < 8; ++u) { const int k = 8 * wave + u; c[u] = (k >= row) ? c[u] : 0.0; out[k * NB] = c[u]; }
        if (with_inv) {                  // ... and as the identity row's first block X[j][j], in the operand layout
          double* out2 = d.xinv + red_blk_base(j, j) + wave * 256 + row * 2;
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f64x2*>(out2 + q * 64) = f64x2{c[q], c[q + 4]};
        }
      } else if (is_rhs) {
        if (row == 0) {
#pragma unroll
          for (int u = 0; u < 8; ++u) rhs[c0 + 8 * wave + u] = c[u];
        }
      } else {
        // columns 8 wave .. 8 wave + 7 of row `row` of L[r][j]: four (k, k + 4) pairs of the k-interleaved block
        double* out = Tblk + wave * 256 + row * 2;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<f64x2*>(out + q * 64) = f64x2{c[q], c[q + 4]};
      }
    }
    if (stamp3) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp3[5] = __builtin_amdgcn_s_memtime(); }
    return;
  }
}

__global__ __launch_bounds__(256) void ba_chol_step_kernel(BaDev d, int j, double lambda, int with_inv, StepSlots slots) {
  __shared__ __attribute__((aligned(16))) double arena[kCholArena];
  chol_step_body(d, j, lambda, with_inv, blockIdx.x, arena, slots);
}

// ---------------------------------------------------------------------------------------------
// L^T dp = y (y sits in rhs after the column steps): blocked back substitution.
//
// ba_back_solve: one 8-wave workgroup handles a GROUP of up to 28 block rows [b_lo, b_hi), last block first.  Per
// block b, wave 0 computes x_b = L_d^-T y_b as a 32x32 mat-vec with the inverse factor the factorisation left in
// ldiag (both halves of the wave take half of the k range each; the next block's factor is prefetched) while the
// other seven waves hold the L values of block row b in registers -- thread (block bi above, pair q) owns columns c
// and c + 4 of block (b, bi), 32 x 16 contiguous bytes in the block layout, 32 independent loads issued before x_b
// exists -- and fold x_b into their y.  448 update threads cover the 28 block rows of a group (896 unknowns) from
// registers.  The group that ends at row 0 also does the camera update of ba:383-392 and prepares the next
// iteration's cameras.
// ba_back_update: between two groups, y[0 : 32 g0) -= L[g0 : g1, 0 : g0)^T x[g0 : g1) -- the part of the sweep that
// is a plain (transposed) matrix-vector product, one workgroup per block column so that it streams L at the
// chip's bandwidth instead of one CU's (a single workgroup reading the 8 MB of L at V = 200 took 250 us).
// Systems of up to 28 block rows (V <= 128) are one group and one launch.
// ---------------------------------------------------------------------------------------------
constexpr int BS_THREADS = 512;
constexpr int BS_UPD = BS_THREADS - 64;        // update threads
constexpr int BS_ROWS = BS_UPD / 16;           // block rows a group covers (16 column pairs per block)

__global__ __launch_bounds__(BS_THREADS) void ba_back_solve_kernel(BaDev d, int cur, int b_hi, int b_lo) {
  __shared__ double ylds[BS_ROWS * NB];     // y of the group's rows
  __shared__ double xb[NB];
  __shared__ double yb[NB];
  const int P = d.P, nbk = d.nbk;
  const double* red = d.red;
  const double* yg = d.red + red_rhs_off(nbk);
  const int tid = threadIdx.x;
  const int lane = tid & (NB - 1);
  const int y0 = b_lo * NB;
  for (int i = tid; i < (b_hi - b_lo) * NB; i += BS_THREADS) ylds[i] = yg[y0 + i];
  // wave 0: lane l holds half a row of L_d^-T: row l & 31, k in [16 (l >> 5), 16 (l >> 5) + 16).
  // One register array for both roles: wave 0 keeps that half row in v[0..15].x across the block loop, the update
  // waves reload v for every block.
  const int khalf = (tid >> 5) & 1;
  f64x2 v[NB];
  if (tid < 64) {
    const double* Xd = d.ldiag + (size_t)(b_hi - 1) * NB * NB + (size_t)khalf * 16 * NB;
#pragma unroll
    for (int k = 0; k < NB / 2; ++k) v[k].x = Xd[k * NB + lane];
  }
  __syncthreads();
  unsigned long long* stamp = (d.stamps && tid == 0) ? d.stamps + 128 : nullptr;
  const int utid = tid - 64;            // 0 .. BS_UPD-1 for the update waves
  const int ubi = b_lo + (utid >> 4), uq = utid & 15;     // block row this thread updates
  const int ucol = (uq & 3) + 8 * (uq >> 2);              // columns ucol and ucol + 4 of a block
  const int uoff = (uq >> 2) * 256 + (uq & 3) * 64;       // + 2 k: that column pair in row k of the block
  for (int b = b_hi - 1; b >= b_lo; --b) {
    const int c0 = b * NB;
    if (stamp && b < 16) stamp[4 * b + 0] = __builtin_amdgcn_s_memtime();
    if (tid >= 64) {
      if (ubi < b) {
        const double* blk = red + red_blk_base(b, ubi) + uoff;
#pragma unroll
        for (int k = 0; k < NB; ++k) v[k] = *reinterpret_cast<const f64x2*>(blk + 2 * k);     // L[c0 + k][32 ubi + ucol (+4)]
      }
    } else {
      if (tid < NB) yb[lane] = (c0 + lane < P) ? ylds[c0 - y0 + lane] : 0.0;      // single wave: LDS in order
      double x0 = 0.0, x1 = 0.0, x2 = 0.0, x3 = 0.0;
#pragma unroll
      for (int k = 0; k < NB / 2; k += 4) {
        const double* yy = yb + 16 * khalf + k;
        x0 += v[k].x * yy[0]; x1 += v[k + 1].x * yy[1]; x2 += v[k + 2].x * yy[2]; x3 += v[k + 3].x * yy[3];
      }
      double xi = (x0 + x1) + (x2 + x3);
      xi += __shfl_xor(xi, 32, 64);
      if (tid < NB) {
        xb[lane] = (c0 + lane < P) ? xi : 0.0;
        d.delta[c0 + lane] = (c0 + lane < P) ? xi : 0.0;
      }
      if (stamp && b < 16) stamp[4 * b + 1] = __builtin_amdgcn_s_memtime();
      if (b > b_lo) {                    // next diagonal factor; lands during the update below
        const double* Xd = d.ldiag + (size_t)(b - 1) * NB * NB + (size_t)khalf * 16 * NB;
#pragma unroll
        for (int k = 0; k < NB / 2; ++k) v[k].x = Xd[k * NB + lane];
      }
    }
    __syncthreads();
    if (stamp && b < 16) stamp[4 * b + 2] = __builtin_amdgcn_s_memtime();
    if (tid >= 64 && ubi < b) {
      double s0 = 0, s1 = 0;
#pragma unroll
      for (int k = 0; k < NB; ++k) { s0 += v[k].x * xb[k]; s1 += v[k].y * xb[k]; }
      ylds[NB * (ubi - b_lo) + ucol] -= s0;
      ylds[NB * (ubi - b_lo) + ucol + 4] -= s1;
    }
    __syncthreads();
    if (stamp && b < 16) stamp[4 * b + 3] = __builtin_amdgcn_s_memtime();
  }
  if (b_lo > 0) return;
  if (tid == 0) *d.iter_count += 1;      // the next linearisation's cost goes to the next slot (sfm_ba_get_stats)
  for (int c = tid; c < d.V; c += BS_THREADS) {
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

// y[32 bi ..] -= sum_{r = g0}^{g1 - 1} L[r][bi]^T x_r for block column bi = blockIdx.x < g0 (x in d.delta).
// Thread (slice, q): block rows r = g0 + 16 blockIdx.y + slice, + 16 gridDim.y, ...; the column pair q of every one
// of them.  blockIdx.y spreads a block column over several workgroups (one f64 atomic per output and workgroup).
__global__ __launch_bounds__(256) void ba_back_update_kernel(BaDev d, int g0, int g1) {
  __shared__ double part[16][NB + 1];
  const int bi = blockIdx.x;
  const int tid = threadIdx.x, uq = tid & 15, slice = tid >> 4;
  const int ucol = (uq & 3) + 8 * (uq >> 2);
  const int uoff = (uq >> 2) * 256 + (uq & 3) * 64;
  double s0 = 0, s1 = 0;
  for (int r = g0 + 16 * blockIdx.y + slice; r < g1; r += 16 * gridDim.y) {
    const double* blk = d.red + red_blk_base(r, bi) + uoff;
    const double* x = d.delta + r * NB;
    f64x2 w[NB];
#pragma unroll
    for (int k = 0; k < NB; ++k) w[k] = *reinterpret_cast<const f64x2*>(blk + 2 * k);
#pragma unroll
    for (int k = 0; k < NB; ++k) { const double xk = x[k]; s0 += w[k].x * xk; s1 += w[k].y * xk; }
  }
  part[slice][ucol] = s0;
  part[slice][ucol + 4] = s1;
  __syncthreads();
  if (tid < NB) {
    double s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += part[q][tid];
    if (gridDim.y == 1) d.red[red_rhs_off(d.nbk) + bi * NB + tid] -= s;
    else if (s != 0.0) atomicAdd(&d.red[red_rhs_off(d.nbk) + bi * NB + tid], -s);
  }
}

// Packed S (lower blocks) -> dense symmetric copy for the parity hook.
__global__ void ba_symmetrize_kernel(const double* __restrict__ red, int P, double lambda, double* __restrict__ out) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= P * P) return;
  const int i = idx / P, j = idx % P;
  out[idx] = red[red_index(max(i, j), min(i, j))] + (i == j ? lambda : 0.0);
}

// ---------------------------------------------------------------------------------------------
// dp = X y with X = L^-T from the identity rows of the column steps and y = L^-1 rhs from the rhs row: the whole
// back substitution as ONE launch of independent block rows (it was 11 dependent block steps of ~2 us at V = 50,
// 44 in four groups at V = 200).  Workgroup e: dp_e = sum_{c >= e} X[e][c] y_c; thread (row i, slice) takes every
// 32nd block column, a block row as sixteen 16-byte loads per thread (32 x 16 contiguous bytes across the lanes).
// The workgroup that finishes last (release / acquire through a device counter) does the camera update of
// ba:383-392 and prepares the next iteration's cameras: 8.0 us in all at V = 50 against 4.8 + 5.0 us with the camera
// update as its own launch (12.4 against 7.4 + 4.2 at V = 200).
constexpr int IA_THREADS = 1024;      // 32 rows x 32 slices of block columns
__global__ __launch_bounds__(IA_THREADS) void ba_inv_apply_kernel(BaDev d, int cur) {
  __shared__ double part[IA_THREADS / 32][NB + 1];
  __shared__ int is_last;
  const int nbk = d.nbk, e = blockIdx.x;
  const int tid = threadIdx.x, i = tid & 31, slice = tid >> 5;
  const double* __restrict__ y = d.red + red_rhs_off(nbk);
  double s = 0;
  for (int c = e + slice; c < nbk; c += IA_THREADS / 32) {
    const double* blk = d.xinv + red_blk_base(c, e) + 2 * i;
    const double* yc = y + c * NB;
    f64x2 w[16];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) w[4 * m + q] = *reinterpret_cast<const f64x2*>(blk + m * 256 + q * 64);      // columns 8m+q, 8m+q+4
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int q = 0; q < 4; ++q) s = __builtin_fma(w[4 * m + q].y, yc[8 * m + q + 4], __builtin_fma(w[4 * m + q].x, yc[8 * m + q], s));
  }
  part[slice][i] = s;
  __syncthreads();
  if (tid < NB) {
    double t = 0;
#pragma unroll
    for (int q = 0; q < IA_THREADS / 32; ++q) t += part[q][tid];
    d.delta[e * NB + tid] = t;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  }
  __syncthreads();
  if (tid == 0) {
    const int done = __hip_atomic_fetch_add(d.sync_ctr, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    is_last = done == nbk - 1;
    if (is_last) __hip_atomic_store(d.sync_ctr, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next solve
  }
  __syncthreads();
  if (!is_last) return;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  if (tid == 0) *d.iter_count += 1;      // the next linearisation's cost goes to the next slot (sfm_ba_get_stats)
  for (int c = tid; c < d.V; c += IA_THREADS) {
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

// ---------------------------------------------------------------------------------------------
// Small systems (P <= 64: the nine cameras or fewer of the reference's usual scenes): the whole reduced solve --
// factorisation, forward and back substitution, camera update -- in ONE single-workgroup launch.
//
// The multi-wave elimination of chol_trsm_cols taken over the whole matrix: lane = row (all P rows at once, no block
// steps), wave w < 8 owns columns 8w .. 8w+7, two columns per step with the 2x2 pivot recurrence above.  The wave
// that owns the current pair (the chain, raised priority: it shares its SIMD with another wave) publishes
// (x, y) = (L[row][j], L[row][j+1]) of every row to LDS; later waves fold it into their columns.  The published
// values ARE L (xy[s][row] = L[row][2s .. 2s+1]) and stay in LDS for the back substitution.
// The right-hand side is one more COLUMN, owned by wave 8: it follows the published steps with the forward
// substitution y_j = b_j / l11, y_j+1 = (b_j+1 - l21 y_j) / l22, b_row -= x y_j + y y_j+1 -- off the chain's critical
// path, no 65th row.
// Back substitution: one wave, lane = row, last pair first; row j of L is read from LDS across the lanes (leading
// dimension 65 so that the column walk touches every bank twice, not thirty-two times), two steps ahead of its use.
// No block boundary, no global round trip for L, no second launch.  Measured (tools/time_small.py, tools/stamps_small.py):
// 6 cameras (P = 42) 33.1 us per iteration against 37.3 with block steps; 9 cameras (P = 63) 45.5 against 44.9 -- the
// serial pair-steps (~680 cycles each here, hand-overs included) catch up with the two block steps that run their
// rows on separate CUs, so the kernel is used up to P = 56 (eight cameras).
constexpr int SM_SEG = 8;                        // columns per wave
constexpr int SM_COLWAVES = 8;
constexpr int SM_THREADS = 64 * (SM_COLWAVES + 1), SM_STEPS = 32, SM_LD = 65;
constexpr int kSmallMaxP = 64;                 // what the kernel can hold
constexpr int kSmallUseP = 56;                 // what it is used for

__global__ __launch_bounds__(SM_THREADS) void ba_small_solve_kernel(BaDev d, int cur, double lambda) {
  __shared__ f64x2 xy[SM_STEPS][SM_LD];      // [pair-step][row] = (L[row][2s], L[row][2s+1])
  __shared__ f64x4 piv[SM_STEPS];            // (1/l11, l21, 1/l22, -) of the pair's 2x2 pivot
  __shared__ f64x2 ysol[SM_STEPS];           // y = L^-1 rhs, two entries per pair-step
  __shared__ double dp[kSmallMaxP];          // the solution, for the camera update
  __shared__ int flag;                       // pair-steps published
  const int P = d.P;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int Pe = (P + 1) & ~1, S = Pe >> 1;  // an odd P gets one identity row / column
  const int nseg = (Pe + SM_SEG - 1) / SM_SEG;
  const double* __restrict__ red = d.red;
  const double* __restrict__ rhs = d.red + red_rhs_off(d.nbk);
  if (tid == 0) flag = 0;
  // diagnostic stamps (SFM_OPT_DEBUG bit 8): [0] start, [1] loaded, [2] factorised, [3] back-substituted, [4] end,
  // [8 + w] wave w takes the chain over
  unsigned long long* stamp = (d.stamps && lane == 0) ? d.stamps : nullptr;
  if (stamp && wave == 0) stamp[0] = __builtin_amdgcn_s_memtime();
  double c[SM_SEG], b = 0.0;
  if (wave < nseg) {
#pragma unroll
    for (int u = 0; u < SM_SEG; ++u) {
      const int col = SM_SEG * wave + u;
      double v = (lane < P && col <= lane) ? red[red_index(lane, col)] : 0.0;      // lower part; the upper is never used
      if (col == lane) v = lane < P ? v + lambda : 1.0;
      c[u] = v;
    }
  } else if (wave == SM_COLWAVES) {
    b = lane < P ? rhs[lane] : 0.0;
  }
  // the camera update's operands, long before they are needed
  const int cam_i = SM_THREADS - 1 - tid;
  double cam[7];
  if (cam_i < d.V) {
#pragma unroll
    for (int k = 0; k < 7; ++k) cam[k] = d.cams[7 * cam_i + k];
  }
  __syncthreads();
  if (stamp && wave == 0) stamp[1] = __builtin_amdgcn_s_memtime();
  if (wave == SM_COLWAVES) {
    for (int s = 0; s < S; ++s) {
      while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s) __builtin_amdgcn_s_sleep(1);
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      const f64x4 pv = piv[s];
      const f64x2 own = xy[s][lane];
      const double bj = lane_bcast(b, 2 * s), bj1 = lane_bcast(b, 2 * s + 1);
      const double yj = bj * pv.x, yj1 = (bj1 - pv.y * yj) * pv.z;
      b = __builtin_fma(-own.y, yj1, __builtin_fma(-own.x, yj, b));
      if (lane == 0) ysol[s] = f64x2{yj, yj1};
    }
  } else {
    for (int seg = 0; seg < nseg; ++seg) {
      if (wave == seg) {
        if (stamp) stamp[8 + seg] = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int t = 0; t < SM_SEG / 2; ++t) {
          const int j = 2 * t, s = (SM_SEG / 2) * seg + t, l0 = SM_SEG * seg + j;       // l0: the lane (= row) of column l0
          if (s >= S) break;
          const double pa = lane_bcast(c[j], l0), pb = lane_bcast(c[j], l0 + 1), pc = lane_bcast(c[j + 1], l0 + 1);
          const double det = __builtin_fma(pa, pc, -(pb * pb));
          const double r1 = rsqrt_nr(pa), r2 = rsqrt_nr(det);
          const double l11 = pa * r1, l21 = pb * r1, i22 = r2 * l11;
          const double x = c[j] * r1;
          const double y = (c[j + 1] - x * l21) * i22;
          c[j] = x; c[j + 1] = y;
          xy[s][lane] = f64x2{x, y};
          if (lane == 0) piv[s] = f64x4{r1, l21, i22, 0.0};
          if (t < SM_SEG / 2 - 1) {
            const double x2 = lane_bcast(x, l0 + 2), y2 = lane_bcast(y, l0 + 2);
            const double x3 = lane_bcast(x, l0 + 3), y3 = lane_bcast(y, l0 + 3);
            c[j + 2] = __builtin_fma(-y, y2, __builtin_fma(-x, x2, c[j + 2]));
            c[j + 3] = __builtin_fma(-y, y3, __builtin_fma(-x, x3, c[j + 3]));
#pragma unroll
            for (int u = j + 4; u < SM_SEG; ++u) { const f64x2 q = xy[s][SM_SEG * seg + u]; c[u] = __builtin_fma(-y, q.y, __builtin_fma(-x, q.x, c[u])); }
          }
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
          if (lane == 0) __hip_atomic_store(&flag, s + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __builtin_amdgcn_s_setprio(0);
      } else if (wave > seg && wave < nseg) {
#pragma unroll
        for (int t = 0; t < SM_SEG / 2; ++t) {
          const int s = (SM_SEG / 2) * seg + t;
          if (s >= S) break;
          while (__hip_atomic_load(&flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= s) __builtin_amdgcn_s_sleep(1);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
          const f64x2 own = xy[s][lane];
#pragma unroll
          for (int u = 0; u < SM_SEG; ++u) { const f64x2 q = xy[s][SM_SEG * wave + u]; c[u] = __builtin_fma(-own.y, q.y, __builtin_fma(-own.x, q.x, c[u])); }
        }
      }
    }
  }
  __syncthreads();
  if (stamp && wave == 0) stamp[2] = __builtin_amdgcn_s_memtime();
  if (wave == 0) {
    // L^T dp = y, pair by pair from the bottom: lane i carries y_i until its pair is solved, dp_i afterwards.
    // Step operands (the pivot and L[j][lane], L[j+1][lane], used by lanes < j only) are loaded two steps ahead.
    double v = 0.0;
    if (lane < Pe) { const f64x2 t = ysol[lane >> 1]; v = (lane & 1) ? t.y : t.x; }
    const double* lrow = reinterpret_cast<const double*>(&xy[lane >> 1][0]) + (lane & 1);    // L[.][lane]
    auto step = [&](int s, const f64x4& pv, double a0, double a1) {
      const int j = 2 * s;
      const double yj = lane_bcast(v, j), yj1 = lane_bcast(v, j + 1);
      const double x1 = yj1 * pv.z;
      const double x0 = (yj - pv.y * x1) * pv.x;
      const double upd = __builtin_fma(-a1, x1, __builtin_fma(-a0, x0, v));
      v = lane < j ? upd : (lane == j ? x0 : (lane == j + 1 ? x1 : v));
    };
    int sa = S - 1, sb = S > 1 ? S - 2 : 0;
    f64x4 pvA = piv[sa], pvB = piv[sb];
    double aA0 = lrow[4 * sa], aA1 = lrow[4 * sa + 2], aB0 = lrow[4 * sb], aB1 = lrow[4 * sb + 2];
    for (int s = S - 1; s >= 0; s -= 2) {
      const int na = s >= 2 ? s - 2 : 0, nb = s >= 3 ? s - 3 : 0;
      const f64x4 pvA2 = piv[na];
      const double nA0 = lrow[4 * na], nA1 = lrow[4 * na + 2];
      step(s, pvA, aA0, aA1);
      const f64x4 pvB2 = piv[nb];
      const double nB0 = lrow[4 * nb], nB1 = lrow[4 * nb + 2];
      if (s >= 1) step(s - 1, pvB, aB0, aB1);
      pvA = pvA2; aA0 = nA0; aA1 = nA1; pvB = pvB2; aB0 = nB0; aB1 = nB1;
    }
    if (lane < P) { d.delta[lane] = v; dp[lane] = v; }
    if (stamp) stamp[3] = __builtin_amdgcn_s_memtime();
  }
  __syncthreads();
  if (tid == 0) *d.iter_count += 1;      // the next linearisation's cost goes to the next slot (sfm_ba_get_stats)
  if (cam_i < d.V) {
#pragma unroll
    for (int k = 0; k < 7; ++k) cam[k] += dp[7 * cam_i + k];                                  // ba:383
    const double nq = sqrt(cam[3] * cam[3] + cam[4] * cam[4] + cam[5] * cam[5] + cam[6] * cam[6]);   // ba:388-392
#pragma unroll
    for (int k = 3; k < 7; ++k) cam[k] /= nq;
#pragma unroll
    for (int k = 0; k < 7; ++k) d.cams[7 * cam_i + k] = cam[k];
    CamPrep out;
    const int st = cam_prepare(cam, &out);      // ba:323 of the next iteration / ba:412 after the last one
    d.prep[cur ^ 1][cam_i] = out;
    report_status(d.status, st, cam_i);
  }
  if (stamp && wave == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp[4] = __builtin_amdgcn_s_memtime(); }
}

}  // namespace sfm
#include "sfm_ba_flow.h"
namespace sfm {

// One device copy of a task table per (device, key) for the life of the process, so that the per-view sfm_ba_append of the drop-in
// classes (a new problem object every time) neither allocates nor copies it again.  key: nbk (S reduced by its own launch) or
// 1000 + V (the deferred reduce: one CAMSUM task per camera and part).
static int flow_task_table(int key, int nbk, int V, const void** table, int* ntasks) {
  static std::mutex mu;
  static std::map<long long, std::pair<void*, int>> tables;
  std::lock_guard<std::mutex> lock(mu);
  const long long k = (long long)ctx().device * 100000 + key;
  auto it = tables.find(k);
  if (it == tables.end()) {
    const std::vector<FlowTask> tasks = flow_build_tasks(nbk, V);
    void* tk = nullptr;
    SFM_HIP(hipMalloc(&tk, sizeof(FlowTask) * std::max<size_t>(1, tasks.size())));
    if (!tasks.empty()) SFM_HIP(hipMemcpy(tk, tasks.data(), sizeof(FlowTask) * tasks.size(), hipMemcpyHostToDevice));
    it = tables.emplace(k, std::make_pair(tk, (int)tasks.size())).first;
  }
  *table = it->second.first;
  *ntasks = it->second.second;
  return SFM_OK;
}

int ba_flow_setup(sfm_ba_problem* p) {
  BaDev& d = p->dev;
  if (d.nbk < 2 || d.nbk > kFlowMaxNbk) return SFM_OK;
  // field switch: SFM_FLOW_SOLVE=0 keeps every problem on the column-step launches (what SFM_OPT_DEBUG bit 1024 does per handle)
  static const bool enabled = [] { const char* e = getenv("SFM_FLOW_SOLVE"); return !(e && atoi(e) == 0); }();
  if (!enabled) return SFM_OK;
  SFM_TRY(flow_task_table(d.nbk, d.nbk, 0, &d.flow_tasks, &d.flow_ntasks));
  SFM_TRY(flow_task_table(1000 + d.V, d.nbk, d.V, &p->flow_tasks_red, &p->flow_ntasks_red));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&d.flow), sizeof(unsigned) * flow_words(d.nbk)));
  SFM_HIP(hipMemsetAsync(d.flow, 0, sizeof(unsigned) * flow_words(d.nbk), p->stream));
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->flow_camsum), sizeof(double) * 4 * 35 * (size_t)d.V));
  static int attr_device = -1;
  if (attr_device != ctx().device) {
    SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_chol_flow_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFlowLdsBytes));
    attr_device = ctx().device;
  }
  return SFM_OK;
}

// which of the three solve paths ba_enqueue_reduced_solve takes
static bool solve_uses_small(const BaDev& d) { return d.P <= ((d.debug & 256) ? kSmallMaxP : kSmallUseP) && !(d.debug & 64); }
static bool solve_uses_flow(const BaDev& d) {
  // the identity rows ride along (SFM_OPT_DEBUG bit 512: leave them out and back-substitute block row by block row);
  // SFM_OPT_DEBUG bit 1024: column steps as separate launches where the data-flow launch would run
  return !solve_uses_small(d) && !(d.debug & 512) && d.nbk <= kInvRowsMaxNbk && d.flow != nullptr && !(d.debug & 1024);
}

// The split-K reduce of the dense product can be left to the data-flow launch (FlowRed in sfm_ba_flow.h) when nothing but that launch
// reads S: iterations enqueued by sfm_ba_iterate on one GPU (no all-reduce of [S | rhs] in between), per-workgroup camera accumulators
// from ba_linearize, not the deterministic mode (which keeps its own summation order).  SFM_OPT_DEBUG bit 16384: never.
bool ba_solve_can_defer_reduce(const sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  if (!(solve_uses_flow(d) && p->flow_tasks_red != nullptr && p->flow_camsum != nullptr && !(d.debug & 16384) && !p->deterministic &&
        p->comm == nullptr && p->lin_rows > 0)) return false;
  // a task sums ALL slabs of its elements (one workgroup per 8 rows of a block): with three tiles (19-36 cameras) a tile has up to 192
  // slabs, a long dependent chain of loads, and the reduce kernel's many short ones win; with one tile (9-18 cameras, capped at 80
  // slabs) the two paths measure within 1 us of each other (44.4 / 43.5 us at 9 x 2 000, 67.8 / 66.9 at 18 x 3 000: the own launch wins)
  const SchurPlan pl = ba_schur_dense_plan(p);
  int most = 0;
  for (int c = 0; c < 4; ++c)
    if (plan_tiles_in_class(pl, c) > 0) most = std::max(most, pl.chunks[c]);
  return pl.nblk >= 3 && most <= 96;
}

int ba_enqueue_reduced_solve(sfm_ba_problem* p, double lambda) {
  hipStream_t s = p->stream;
  const BaDev& d = p->dev;
  const int nbk = d.nbk;
  if (p->reduce_deferred && !solve_uses_flow(d)) return SFM_E_HIP;      // (ba_solve_can_defer_reduce said otherwise)
  if (solve_uses_small(d)) {      // SFM_OPT_DEBUG bit 64: block steps for every size; 256: the small kernel up to P = 64
    ba_small_solve_kernel<<<1, SM_THREADS, 0, s>>>(d, p->cur, lambda);
    SFM_HIP(hipGetLastError());
    return SFM_OK;
  }
  // the identity rows ride along (SFM_OPT_DEBUG bit 512: leave them out and back-substitute block row by block row).
  // Their trailing updates grow with nbk^3 while the back substitution they replace grows with nbk: measured
  // (tools/time_solve_paths.py) 153 vs 198 us at nbk = 20, 293 vs 340 at 35, 422 vs 448 at 44, 643 vs 640 at 57,
  // 932 vs 910 at 75 -- used up to 52 block columns (V <= 237)
  const bool with_inv = !(d.debug & 512) && nbk <= kInvRowsMaxNbk;
  if (solve_uses_flow(d)) {
    FlowRed fr{};
    const FlowTask* tasks = static_cast<const FlowTask*>(d.flow_tasks);
    int ntasks = d.flow_ntasks;
    if (p->reduce_deferred) {
      fr.ws = static_cast<const double*>(p->schur_ws); fr.camsum = p->flow_camsum; fr.plan = ba_schur_dense_plan(p);
      fr.lin_rows = p->lin_rows; fr.lin_grid = p->lin_grid;
      tasks = static_cast<const FlowTask*>(p->flow_tasks_red); ntasks = p->flow_ntasks_red;
      p->reduce_deferred = false;
    }
    const int grid = 1 + std::min(ntasks, ctx().num_cus - 32);
    // SFM_OPT_DEBUG bit 2048: dp = X y and the camera update as their own launch behind the data-flow launch
    const bool fused_dp = !(d.debug & 2048);
    ba_chol_flow_kernel<<<grid, 512, kFlowLdsBytes, s>>>(d, d.flow, tasks, ntasks, lambda, fused_dp ? p->cur : -1, fr);
    if (!fused_dp) ba_inv_apply_kernel<<<nbk, IA_THREADS, 0, s>>>(d, p->cur);
    SFM_HIP(hipGetLastError());
    return SFM_OK;
  }
  for (int j = 0; j < nbk; ++j) {
    const int ncol = nbk - j + 1;                        // column role: block rows j .. nbk (nbk = the rhs row)
    // behind the column roles (+ the j column roles of the identity rows), padded to a multiple of 8: the trailing super-tiles
    // of S and of the identity rows, 8 x slots workgroups each, XCD-affine (trail_pick / inv_pick)
    const StepSlots z = step_slots(nbk, j, with_inv);
    const int ncolumn_roles = ncol + (with_inv ? j : 0);
    const int trailing = 8 * (z.trail + z.itrail);
    const int grid = trailing > 0 ? ((ncolumn_roles + 7) & ~7) + trailing : ncolumn_roles;
    ba_chol_step_kernel<<<grid, 256, 0, s>>>(d, j, lambda, with_inv ? 1 : 0, z);
  }
  if (with_inv) {
    ba_inv_apply_kernel<<<nbk, IA_THREADS, 0, s>>>(d, p->cur);
    SFM_HIP(hipGetLastError());
    return SFM_OK;
  }
  // back substitution in groups of at most 28 block rows, from the bottom; between two groups one multi-workgroup
  // launch folds the finished group into everything above it
  // (one workgroup streams a group's L blocks at one CU's bandwidth, ~100 GB/s: groups of 11-14 block rows keep a block
  // step near its latency floor; up to 28 block rows -- V <= 128 -- stay one group and one launch)
  const int gmax = nbk <= BS_ROWS ? BS_ROWS : 12;
  const int ngroups = (nbk + gmax - 1) / gmax;
  const int gsize = (nbk + ngroups - 1) / ngroups;
  for (int hi = nbk; hi > 0;) {
    const int lo = std::max(0, hi - gsize);
    ba_back_solve_kernel<<<1, BS_THREADS, 0, s>>>(d, p->cur, hi, lo);
    if (lo > 0) {
      // one slice = one writer per output, no atomics: also whenever a caller's reduced buffer is bound, i.e. the
      // multi-GPU mode, where every rank repeats this solve and all of them must end on bit-identical cameras
      const bool replicated = p->dev.red != p->own_red || p->comm != nullptr;
      const int slices = (p->deterministic || replicated || hi - lo <= 16) ? 1 : std::min(4, (hi - lo + 15) / 16 + 1);
      ba_back_update_kernel<<<dim3(lo, slices), 256, 0, s>>>(d, lo, hi);
    }
    hi = lo;
  }
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

}  // namespace sfm

// Diagnostic (CPU only, no device needed): the task table of the data-flow solve for nbk block columns, in the order the
// workgroups take it -- int[4] per task: {type (0 block of L, 1 closer, 2 hand-over (i, i-1), 3 rhs, 4 identity row), row, column, key}.
extern "C" int sfm_ba_flow_tasks_deferred(int n_cams, int* out, int capacity) {
  const int nbk = (7 * n_cams + sfm::kNB - 1) / sfm::kNB;
  if (n_cams < 1 || nbk < 2 || nbk > sfm::kFlowMaxNbk) return 0;
  const std::vector<sfm::FlowTask> t = sfm::flow_build_tasks(nbk, n_cams);
  if (out != nullptr)
    for (size_t q = 0; q < t.size() && (int)q < capacity; ++q) { out[4 * q] = t[q].type; out[4 * q + 1] = t[q].i; out[4 * q + 2] = t[q].k; out[4 * q + 3] = t[q].key; }
  return (int)t.size();
}

extern "C" int sfm_ba_flow_tasks(int nbk, int* out, int capacity) {
  if (nbk < 2 || nbk > sfm::kFlowMaxNbk) return 0;
  const std::vector<sfm::FlowTask> t = sfm::flow_build_tasks(nbk);
  if (out != nullptr)
    for (size_t q = 0; q < t.size() && (int)q < capacity; ++q) { out[4 * q] = t[q].type; out[4 * q + 1] = t[q].i; out[4 * q + 2] = t[q].k; out[4 * q + 3] = t[q].key; }
  return (int)t.size();
}

namespace sfm {

void ba_enqueue_symmetrize(sfm_ba_problem* p, double lambda, double* S_out, double* rhs_out) {
  const BaDev& d = p->dev;
  ba_symmetrize_kernel<<<(d.P * d.P + 255) / 256, 256, 0, p->stream>>>(d.red, d.P, lambda, S_out);
  (void)hipMemcpyAsync(rhs_out, d.red + red_rhs_off(d.nbk), sizeof(double) * d.P, hipMemcpyDeviceToDevice, p->stream);
}

}  // namespace sfm
