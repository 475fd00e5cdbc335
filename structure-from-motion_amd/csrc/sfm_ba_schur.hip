// sfm_ba_schur.hip — Schur-complement product  S -= sum_p Z_p Z_p^T  (= B D^-1 B^T of
// ba_processor.py:382, in the symmetric form Z_o = W_o L_p^-T).  Only the lower triangle of S
// (row >= column) is produced; block (c_a, c_b) with c_a > c_b of point p is Z_a Z_b^T.
//
//   ba_schur_pairs_kernel   sparse: (18-camera tile) x (point chunk) workgroups, the tile accumulated in LDS
//                           with ds_add_f64, only camera pairs that share a point are multiplied.  Work is
//                           proportional to sum_p k_p (k_p+1)/2 (sparse-optimal).  Used at low visibility.
//   ba_schur_mfma_kernel    dense  v_mfma_f64_16x16x4_f64  SYRK  S -= Zd^T Zd  over the dense matrix Zd
//                           that ba_linearize fills (zeros where a point is not seen); panels are DMA'd
//                           HBM -> LDS (global_load_lds); output-stationary 128x128 tiles x split-K row
//                           chunks; no atomics.  Used when visibility is high enough that dense wins.
#include <algorithm>
#include <cmath>
#include <type_traits>
#include <vector>

#include "sfm_ba.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
// Dense MFMA product  S(lower) -= Zd^T Zd  over the materialised Z.
//
// ba_linearize writes every observation's 7x3 block Z_o into the dense matrix Zd[3N][zp] (row 3p + j, column
// 7 cam + i = the row of S; zeros where the point is not seen; zp = 7V rounded up to the 16-row MFMA strip).  The
// columns are cut into blocks of RB = 128 (8 strips; the last block may hold fewer), a block's panel row is 1 KiB of
// contiguous memory.  (Rounds 1-3 grouped 18 whole cameras = 126 columns into every 128: at 50 cameras 276 MFMA tiles
// per k-step instead of the 253 of the 22 strips that 350 rows need.  Only the reduce has to know that a block
// boundary cuts through a camera: it maps elements, not blocks.)  The product is then a plain split-K SYRK:
// grid = (lower-triangular 128x128 output tiles) x (row chunks of Zd); one workgroup = 8 waves.
//   * staging: one  global_load_lds_dwordx4  per wave copies one 1 KiB panel row straight into the
//     [k][ZLD] LDS image (no VGPRs, no FP64-pipe work); 16 rows x (1 or 2) panels per slab in a ring of 4
//     stages, the DMA running two slabs ahead of the MFMAs.
//   * v_mfma_f64_16x16x4_f64: off-diagonal tiles give each wave 64x32 (4x2 MFMA tiles, 6 ds_read_b64 per 8
//     MFMAs); diagonal tiles deal their 36 lower MFMA tiles round-robin (5,5,5,5,4,4,4,4: every SIMD gets 9),
//     the 28 upper ones are never computed.
// One barrier per slab, placed mid-slab (schur_k_loop).  A operand: lane l holds A[row = l&15][k = l>>4]; B operand: B[k = l>>4][col = l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg.  Partial tiles go to per-(chunk, tile) slabs (plain stores),
// summed into S by ba_schur_reduce_kernel.
// (r01..r02f fused the re-linearisation into producer waves of this kernel instead of reading Z; the
// producers' FP64 VALU work interleaved badly with the MFMA stream -- 217 us vs the 170 us of the same
// kernel with the producer math ablated -- see DESIGN.md section 8.)
// ---------------------------------------------------------------------------------------------
typedef double double4_ __attribute__((ext_vector_type(4)));

constexpr int CB = kSchurCB;     // cameras per column block
constexpr int RB = kSchurRB;     // padded columns per block
constexpr int KSL = kSchurKSL;   // Zd rows (k) per LDS slab
constexpr int ZLD = RB + 16;     // LDS row pitch = 16 (mod 32) doubles: the k-rows of one ds_read_b64 hit disjoint banks
constexpr int N_WAVES = 8;
constexpr int SCHUR_THREADS = 64 * N_WAVES;
constexpr int STAGE = KSL * ZLD;   // doubles per LDS panel image
constexpr int NSTG = 4;            // ring of LDS stages (A panel + B panel each): DMA runs two slabs ahead
constexpr size_t kSchurLdsBytes = sizeof(double) * NSTG * 2 * STAGE;   // 147456 B
static_assert(KSL == 16 && N_WAVES == 8, "schur_k_loop is written for 4 k-steps per slab and 2 DMA rows per wave and panel");

// The k loop of one tile: a ring of NSTG LDS stages of KSL = 16 rows (4 MFMA k-steps), two operand register
// sets.  Per slab s:
//   k-step 0, 1      reads of the next k-step are issued BEFORE the MFMAs of the current one
//   mid-slab         s_waitcnt vmcnt(batch)  -> this wave's DMA of slab s+1 (issued two slabs ago) has landed
//                    s_barrier               -> ... and everybody's; every wave is past slab s-1
//                    DMA of slab s+3 into the stage slab s-1 occupied
//   k-step 2, 3      the reads issued during k-step 3 are k-step 0 of slab s+1
// so neither the barrier nor the slab hand-over ever leaves the MFMA pipe without a ready operand set (with
// the barrier at the slab boundary both waves of a SIMD stalled there together: ~1.3k cycles per slab).
// s_waitcnt vmcnt(N) alone (expcnt / lgkmcnt fields left at "no wait"); the builtin, unlike inline asm, keeps the
// compiler's own counter bookkeeping exact, so it does not fall back to lgkmcnt(0) before the next MFMA
template <int N>
__device__ __forceinline__ void wait_vmcnt() { __builtin_amdgcn_s_waitcnt(0x0F70 | N); }

template <bool DIAG, int NS, bool MMA, bool LOADS>
__device__ __forceinline__ void schur_k_loop(const double* __restrict__ pa, const double* __restrict__ pb, size_t zp,
                                             int k_beg, int nslab, double* __restrict__ img, const int (&oa)[5],
                                             const int (&ob)[5], int wr, int wc, double4_ (&acc)[8]) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: LDS-DMA destinations stay in SGPRs
  // NS: diagonal tile = this wave's lower 16x16 sub-tiles (0..5); off-diagonal tile = its 16-row strips of the
  // row block (0..4; fewer than 4 only in the last, partly filled block), each against its two column strips
  constexpr int NA = NS > 0 ? NS : 1, NB = DIAG ? NA : 2, NM = DIAG ? NS : 2 * NS;
  constexpr int BATCH = (DIAG ? 1 : 2) * (KSL / N_WAVES);      // DMA instructions per wave and slab
  double a[2][NA], b[2][NB];
  const double* z0 = img + (lane >> 4) * ZLD + (lane & 15);
  auto load_ops = [&](int set, int s, int krow) {
    const double* z = z0 + (s & (NSTG - 1)) * 2 * STAGE + krow * ZLD;
    if (DIAG) {
#pragma unroll
      for (int t = 0; t < NS; ++t) { a[set][t] = z[oa[t]]; b[set][t] = z[ob[t]]; }
    } else if (NS > 0) {
#pragma unroll
      for (int x = 0; x < NS; ++x) a[set][x] = z[wr + 16 * x];
#pragma unroll
      for (int y = 0; y < 2; ++y) b[set][y] = z[STAGE + wc + 16 * y];
    }
    __builtin_amdgcn_sched_barrier(0);      // keep the reads ahead of the MFMAs (the scheduler sinks them otherwise)
  };
  // Staging: one wave-instruction = 64 lanes x 16 B = one 1 KiB panel row, DMA'd to a wave-uniform LDS address.
  // This wave owns rows (wave, wave + 8) of each panel; its global read pointers run along the k dimension.
  const double* ga = pa + (size_t)(k_beg + wave) * zp + 2 * lane;
  const double* gb = pb + (size_t)(k_beg + wave) * zp + 2 * lane;
  const size_t row8 = 8 * zp, slab_step = (size_t)KSL * zp;
  auto dma_piece = [&](int s, int j) {       // piece j of slab s: panel j >> 1, row wave + 8 * (j & 1)
    if (!LOADS) return;
    const double* g = ((j >> 1) ? gb : ga) + (j & 1) * row8;
    double* l = img + ((s & (NSTG - 1)) * 2 + (j >> 1)) * STAGE + (wave + 8 * (j & 1)) * ZLD;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)l, 16, 0, 0);
    if (j == BATCH - 1) { ga += slab_step; gb += slab_step; }
  };
  auto issue = [&](int s) {                  // slabs must be issued in order (running pointers)
#pragma unroll
    for (int j = 0; j < BATCH; ++j) dma_piece(s, j);
  };
  // MFMAs of one k-step; dma_slab >= 0: the DMA pieces of that slab are issued one behind each of the first
  // MFMAs, in the shadow of the matrix pipe, instead of ahead of them (where both waves of a SIMD spent the
  // post-barrier cycles on address arithmetic with the pipe idle)
  auto mma = [&](int set, int dma_slab) {
#pragma unroll
    for (int i = 0; i < NM; ++i) {
      if (MMA) {
        if (DIAG) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][i], b[set][i], acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[set][i >> 1], b[set][i & 1], acc[i], 0, 0, 0);
      }
      if (i < BATCH && dma_slab >= 0) {
        __builtin_amdgcn_sched_barrier(0);
        dma_piece(dma_slab, i);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    if (dma_slab >= 0) {                     // a wave with fewer MFMAs than DMA pieces still stages its rows
#pragma unroll
      for (int j = NM; j < BATCH; ++j) dma_piece(dma_slab, j);
    }
    // The operand reads issued before these MFMAs completed long ago (the wave spent >= 4 x 64 cycles issuing
    // them): an explicit lgkmcnt(0) here is free and leaves no pending LDS read at the next load_ops, so the
    // compiler's own wait before the following MFMAs covers nothing newer than it has to (without it it fell
    // back to lgkmcnt(0) AFTER the next reads on the loop back-edge and behind the DMA block).
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_sched_barrier(0);
  };
  // one slab; the flags say whether slabs s+1, s+2, s+3 exist (compile-time, so the steady-state loop body is a
  // single basic block and the compiler's s_waitcnt placement stays exact)
  auto slab = [&](int s, auto has1, auto has2, auto has3) {
    load_ops(1, s, 4);
    mma(0, -1);
    load_ops(0, s, 8);
    mma(1, -1);
    if constexpr (decltype(has1)::value) {
      if constexpr (decltype(has2)::value) wait_vmcnt<BATCH>(); else wait_vmcnt<0>();
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    }
    load_ops(1, s, 12);
    if constexpr (decltype(has3)::value) mma(0, s + 3); else mma(0, -1);
    if constexpr (decltype(has1)::value) load_ops(0, s + 1, 0);
    mma(1, -1);
  };
  using T = std::true_type;
  using F = std::false_type;
  issue(0);
  if (nslab > 1) issue(1);
  if (nslab > 1) wait_vmcnt<BATCH>(); else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (nslab > 2) issue(2);
  load_ops(0, 0, 0);
  __builtin_amdgcn_s_waitcnt(0xC07F);      // once per kernel: no LDS read is pending on either edge into the loop
  __builtin_amdgcn_sched_barrier(0);
  int s = 0;
  for (; s + 3 < nslab; ++s) slab(s, T{}, T{}, T{});
  if (s + 2 < nslab) { slab(s, T{}, T{}, F{}); ++s; }
  if (s + 1 < nslab) { slab(s, T{}, F{}, F{}); ++s; }
  if (s < nslab) slab(s, F{}, F{}, F{});
}

template <bool DIAG>
__device__ __forceinline__ void schur_tile_body(const double* __restrict__ Zd, size_t zp, double* __restrict__ slab, int ti,
                                                int tj, int k_beg, int k_end, double* __restrict__ img /*[NSTG][2][STAGE]*/,
                                                int ra /*16-row strips of block ti that hold cameras*/, int dbg) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const double* pa = Zd + (size_t)RB * ti;
  const double* pb = Zd + (size_t)RB * tj;

  double4_ acc[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) acc[s] = double4_{0, 0, 0, 0};
  int sx[5] = {0, 0, 0, 0, 0}, sy[5] = {0, 0, 0, 0, 0};
  int nsub = 0;
  if (DIAG) {
    for (int s = 0; s < 5; ++s) {
      const int idx = wave + N_WAVES * s;
      if (idx < ra * (ra + 1) / 2) {
        int x = 0;
        while ((x + 1) * (x + 2) / 2 <= idx) ++x;
        sx[s] = __builtin_amdgcn_readfirstlane(x);
        sy[s] = __builtin_amdgcn_readfirstlane(idx - x * (x + 1) / 2);
        nsub = s + 1;
      }
    }
    nsub = __builtin_amdgcn_readfirstlane(nsub);
  }
  int oa[5], ob[5];
#pragma unroll
  for (int s = 0; s < 5; ++s) { oa[s] = 16 * sx[s]; ob[s] = 16 * sy[s]; }
  const int wr = (wave >> 2) * 64, wc = (wave & 3) * 32;     // off-diagonal tiles: this wave's 64x32 part
  const int nslab = (k_end - k_beg) / KSL;                   // chunk bounds are multiples of KSL
  // off-diagonal tiles: strips of the row block in this wave's 64-row half
  const int nstrip = __builtin_amdgcn_readfirstlane(max(0, min(4, ra - 4 * (wave >> 2))));
#define SCHUR_LOOP(D, N, M, L) schur_k_loop<D, N, M, L>(pa, pb, zp, k_beg, nslab, img, oa, ob, wr, wc, acc)
  if (dbg & 5) {      // profiling ablations (1: no MFMA, 4: no staging DMA) on full tiles; results are wrong by design
    const bool mm = !(dbg & 1), ld = !(dbg & 4);
    if (DIAG) {
      if (mm) SCHUR_LOOP(true, 5, true, false); else if (ld) SCHUR_LOOP(true, 5, false, true); else SCHUR_LOOP(true, 5, false, false);
    } else {
      if (mm) SCHUR_LOOP(false, 4, true, false); else if (ld) SCHUR_LOOP(false, 4, false, true); else SCHUR_LOOP(false, 4, false, false);
    }
  } else if (DIAG) {
    switch (nsub) {
      case 5: SCHUR_LOOP(true, 5, true, true); break;
      case 4: SCHUR_LOOP(true, 4, true, true); break;
      case 3: SCHUR_LOOP(true, 3, true, true); break;
      case 2: SCHUR_LOOP(true, 2, true, true); break;
      case 1: SCHUR_LOOP(true, 1, true, true); break;
      default: SCHUR_LOOP(true, 0, true, true); break;
    }
  } else {
    switch (nstrip) {
      case 4: SCHUR_LOOP(false, 4, true, true); break;
      case 3: SCHUR_LOOP(false, 3, true, true); break;
      case 2: SCHUR_LOOP(false, 2, true, true); break;
      case 1: SCHUR_LOOP(false, 1, true, true); break;
      default: SCHUR_LOOP(false, 0, true, true); break;
    }
  }
#undef SCHUR_LOOP
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
      if (x < nstrip) {                      // strips beyond the last camera are never read by ba_schur_reduce
#pragma unroll
        for (int y = 0; y < 2; ++y)
#pragma unroll
          for (int r = 0; r < 4; ++r) slab[(wr + 16 * x + lk + 4 * r) * RB + wc + 16 * y + lr] = acc[2 * x + y][r];
      }
  }
}

// (SchurPlan and the tile <-> workgroup mapping live in sfm_ba.h: the data-flow solve sums a tile's slabs itself when the reduce is deferred)

// ---------------------------------------------------------------------------------------------
// Sparse product (low visibility): the same 18-camera blocks and split-K slabs as the dense path, but a tile's
// 126x126 accumulator lives in LDS and only the camera pairs that actually share a point are multiplied.
// One workgroup = (tile (A, B), chunk of points), 16 waves; a wave takes a point, finds its observations in
// block A and in block B through the per-point block offsets (blk_ptr, built at plan time; the observations
// of a point are sorted by camera, so a block is a contiguous sub-range) and multiplies their Z blocks (AoS,
// 168 B each, read straight from L2 / the scalar cache); every product is one ds_add_f64 into the tile.  No global atomics: the
// tile goes to its slab and ba_schur_reduce sums the
// slabs exactly as for the dense path.  Work is proportional to sum_p k_p (k_p + 1) / 2 (sparse-optimal).
// ---------------------------------------------------------------------------------------------
constexpr int PAIR_WAVES = 16;                      // the per-visit work is tiny at low visibility: occupancy hides it
constexpr int PAIR_THREADS = 64 * PAIR_WAVES;
constexpr int TP = 7 * CB + 1;                      // LDS tile pitch (127 doubles)
constexpr size_t kPairLdsBytes = sizeof(double) * (size_t)7 * CB * TP;

// One (point, tile) visit by one wave.  Lanes = (observation b of block B, column j): each keeps its three
// Z values and its tile column in registers.  The A side is wave-uniform: the wave walks the observations a of
// block A, their 21 values arrive through the scalar data cache (s_load from the constant address space) and
// feed the FMAs as scalar operands; one ds_add_f64 per lane and row.  The visit's B side and camera slots are
// loaded one visit ahead and the block offsets two visits ahead (software pipeline over the wave's points).
template <bool DIAG>
__device__ __forceinline__ void pairs_tile_body(const BaDev& d, const int* __restrict__ blk_ptr, int nblk, int ti, int tj,
                                                int p_beg, int p_end, double* __restrict__ slab, double* __restrict__ tile) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nr = 7 * min(CB, d.V - ti * CB), nc = 7 * min(CB, d.V - tj * CB);      // used part of the tile
  for (int t = tid; t < nr * TP; t += PAIR_THREADS) tile[t] = 0.0;
  __syncthreads();
  const double* __restrict__ Z = d.Z;
  typedef const __attribute__((address_space(4))) double ConstF64;
  ConstF64* Zc = (ConstF64*)d.Z;
  const int* __restrict__ cam_idx = d.cam_idx;
  // lanes 0..3 fetch bp[ti], bp[ti+1], bp[tj], bp[tj+1] of a point
  auto fetch_bp = [&](int p) -> int {
    if (p >= p_end) return 0;
    const int which = lane & 3;
    return blk_ptr[(size_t)p * (nblk + 1) + ((which & 2) ? tj : ti) + (which & 1)];
  };
  // Three-deep software pipeline over this wave's points: block offsets two visits ahead, the visit's Z blocks
  // and camera slots one visit ahead, products now -- the memory latency of a visit hides behind the previous one.
  struct Visit {
    int a0, kA, b0, kB;
    int ca_lane;               // lanes 0..kA-1: tile row offset of camera slot a
    double zb0, zb1, zb2;      // first lane round of the B side
    int col;                   // its tile column (or -1)
  };
  auto load_visit = [&](int bp, Visit& v) {
    v.a0 = __builtin_amdgcn_readlane(bp, 0); v.kA = __builtin_amdgcn_readlane(bp, 1) - v.a0;
    v.b0 = __builtin_amdgcn_readlane(bp, 2); v.kB = __builtin_amdgcn_readlane(bp, 3) - v.b0;
    if (v.kA <= 0 || v.kB <= 0) { v.kA = 0; return; }
    v.ca_lane = 7 * TP * (cam_idx[v.a0 + (lane < v.kA ? lane : 0)] - ti * CB);
    const bool on = lane < 7 * v.kB;
    // idle lanes read element 0 of the block's first observation: an offset built from their lane number ran up to
    // 1.5 KB (3 KB in the second lane round) past the END of Z for the last observations of the list -- harmless
    // inside a pooled block, a memory fault when a tiny scene's Z is the last thing on its page
    const int b = on ? lane / 7 : 0, j = on ? lane - 7 * b : 0;
    const double* zb = Z + (size_t)(v.b0 + b) * 21 + 3 * j;
    v.zb0 = zb[0]; v.zb1 = zb[1]; v.zb2 = zb[2];
    v.col = on ? 7 * (cam_idx[v.b0 + b] - tj * CB) + j : -1;
  };
  auto products = [&](const Visit& v, int b, double zb0, double zb1, double zb2, int col) {
    double* tcol = tile + col;
#pragma unroll
    for (int a = 0; a < CB; ++a) {
      if (a < v.kA) {                                      // wave-uniform
        double* trow = tcol + __builtin_amdgcn_readlane(v.ca_lane, a);
        const bool mine = col >= 0 && (!DIAG || b <= a);   // lower part of a diagonal tile only
        // the A side is wave-uniform: its 21 values come through the scalar data cache straight into SGPRs
        // (constant address space: Z is not written while this kernel runs) and feed the FMAs as scalar operands
        const ConstF64* za = Zc + (size_t)(v.a0 + a) * 21;
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const double val = za[3 * i] * zb0 + za[3 * i + 1] * zb1 + za[3 * i + 2] * zb2;
          if (mine) atomicAdd(trow + i * TP, val);
        }
      }
    }
  };
  Visit cur, nxt;
  cur.kA = 0; nxt.kA = 0;
  int p = p_beg + wave;
  int bp1 = fetch_bp(p);                        // offsets of visit p
  if (p < p_end) load_visit(bp1, cur);
  int bp2 = fetch_bp(p + PAIR_WAVES);           // offsets of visit p + 16
  for (; p < p_end; p += PAIR_WAVES) {
    const int bp3 = fetch_bp(p + 2 * PAIR_WAVES);
    nxt.kA = 0;
    if (p + PAIR_WAVES < p_end) load_visit(bp2, nxt);
    if (cur.kA > 0) {
      products(cur, lane / 7, cur.zb0, cur.zb1, cur.zb2, cur.col);
      if (7 * cur.kB > 64) {                    // second lane round (more than 9 observations in block B)
        const int l = 64 + lane;
        const bool on = l < 7 * cur.kB;
        const int b = on ? l / 7 : 0, j = on ? l - 7 * b : 0;
        const double* zb = Z + (size_t)(cur.b0 + b) * 21 + 3 * j;
        products(cur, b, zb[0], zb[1], zb[2], on ? 7 * (cam_idx[cur.b0 + b] - tj * CB) + j : -1);
      }
    }
    cur = nxt;
    bp2 = bp3;
  }
  __syncthreads();
  for (int t = tid; t < nr * nc; t += PAIR_THREADS) {        // ba_schur_reduce never reads beyond the last camera
    const int r = t / nc, c = t - r * nc;
    slab[r * RB + c] = tile[r * TP + c];
  }
}

__global__ __launch_bounds__(PAIR_THREADS) void ba_schur_pairs_kernel(BaDev d, const int* __restrict__ blk_ptr,
                                                                     double* __restrict__ ws, SchurPlan plan) {
  extern __shared__ double lds_pairs[];
  double* tile = lds_pairs;
  const int w = blockIdx.x;
  double* slab = ws + (size_t)w * (RB * RB);
  const SchurTileRef t = plan_locate(plan, w);
  const int p_beg = t.chunk * plan.rpc[t.cls];
  const int p_end = min(d.N, p_beg + plan.rpc[t.cls]);
  if (t.ti != t.tj) pairs_tile_body<false>(d, blk_ptr, plan.nblk, t.ti, t.tj, p_beg, p_end, slab, tile);
  else pairs_tile_body<true>(d, blk_ptr, plan.nblk, t.ti, t.ti, p_beg, p_end, slab, tile);
}

__global__ __launch_bounds__(SCHUR_THREADS) void ba_schur_mfma_kernel(BaDev d, double* __restrict__ ws, SchurPlan plan) {
  extern __shared__ double img[];               // [NSTG stages][A panel, B panel][KSL][ZLD]
  const int w = blockIdx.x;
  double* slab = ws + (size_t)w * (RB * RB);
  const SchurTileRef t = plan_locate(plan, w);
  const int k_beg = t.chunk * plan.rpc[t.cls];
  const int k_end = min(d.zrows, k_beg + plan.rpc[t.cls]);
  const int ra = (t.cls & 1) ? plan.ra_last : 8;          // strips of the row block that hold cameras
  if (t.ti != t.tj) schur_tile_body<false>(d.Zd, (size_t)d.zp, slab, t.ti, t.tj, k_beg, k_end, img, ra, plan.dbg);
  else schur_tile_body<true>(d.Zd, (size_t)d.zp, slab, t.ti, t.ti, k_beg, k_end, img, ra, plan.dbg);
}

// S(lower) -= sum over the tile's chunk slabs, un-padding block coordinates (block b, row r) -> camera
// b*CB + r/7, parameter r%7.  One thread per padded tile element; blockIdx.y slices the chunk range
// (more loads in flight), one f64 atomic per slice and element.
__global__ __launch_bounds__(256) void ba_schur_reduce_kernel(BaDev d, const double* __restrict__ ws, SchurPlan plan, int tile_blocks,
                                                              int lin_rows, int lin_grid) {
  if (blockIdx.x == gridDim.x - 1) {           // last block: the per-workgroup partial costs of ba_linearize
    if (blockIdx.y == 0 && threadIdx.x < 64) cost_reduce(d, lin_grid);
    return;
  }
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
  const SchurTileRef tr = plan_tile(plan, tile);
  const int ti = tr.ti, tj = tr.tj, first = tr.first, chunks = plan.chunks[tr.cls];
  const int r = e / RB, c = e - r * RB;
  if (plan.cam_blocks && (r >= 7 * CB || c >= 7 * CB)) return;
  const int row = (plan.cam_blocks ? 7 * CB : RB) * ti + r, col = (plan.cam_blocks ? 7 * CB : RB) * tj + c;
  if (row >= d.P || col > row) return;
  const int per = (chunks + gridDim.y - 1) / gridDim.y;
  const int k0 = blockIdx.y * per, k1 = min(chunks, k0 + per);
  double s = 0;
#pragma unroll 4
  for (int k = k0; k < k1; ++k) s += ws[(size_t)(first + k) * (RB * RB) + e];
  if (s != 0.0) atomicAdd(&d.red[red_index(row, col)], -s);
}

// Deterministic variant (SFM_OPT_DETERMINISTIC): ONE thread owns an element of S: it sums the tile's split-K slabs
// in chunk order, adds -- for an element of a camera's diagonal 7x7 block -- the per-workgroup camera accumulators
// of ba_linearize in row order, and stores the result; no atomics, so the summation order is fixed.  Extra blocks
// do the same for rhs.
__global__ __launch_bounds__(256) void ba_schur_reduce_det_kernel(BaDev d, const double* __restrict__ ws, SchurPlan plan,
                                                                  int tile_blocks, int lin_rows, int lin_grid) {
  if (blockIdx.x == gridDim.x - 1) {
    if (threadIdx.x < 64) cost_reduce(d, lin_grid);
    return;
  }
  if ((int)blockIdx.x >= tile_blocks) {
    const int t = (blockIdx.x - tile_blocks) * 256 + threadIdx.x;      // rhs element
    if (t >= d.P) return;
    const int c = t / 7, i = t - 7 * c;
    double s = 0;
    for (int r = 0; r < lin_rows; ++r) s += d.lin_ws[(size_t)r * d.V * 35 + c * 35 + 28 + i];
    d.red[red_rhs_off(d.nbk) + t] = s;
    return;
  }
  const int ntiles = plan.n_off + plan.nblk;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= ntiles * RB * RB) return;
  const int tile = idx / (RB * RB);
  const int e = idx - tile * (RB * RB);
  const SchurTileRef tr = plan_tile(plan, tile);
  const int ti = tr.ti, tj = tr.tj, first = tr.first, chunks = plan.chunks[tr.cls];
  const int r = e / RB, c = e - r * RB;
  if (plan.cam_blocks && (r >= 7 * CB || c >= 7 * CB)) return;
  const int row = (plan.cam_blocks ? 7 * CB : RB) * ti + r, col = (plan.cam_blocks ? 7 * CB : RB) * tj + c;
  if (row >= d.P || col > row) return;
  double s = 0;
  for (int k = 0; k < chunks; ++k) s += ws[(size_t)(first + k) * (RB * RB) + e];
  double u = 0;
  if (row / 7 == col / 7) {
    const int i = row % 7, j = col % 7;
    const int t = (row / 7) * 35 + i * (i + 1) / 2 + j;
    for (int q = 0; q < lin_rows; ++q) u += d.lin_ws[(size_t)q * d.V * 35 + t];
  }
  d.red[red_index(row, col)] = u - s;
}

// MFMAs per SIMD and k-step of a diagonal tile with `ra` strips: sub-tile idx goes to wave idx % 8, SIMD s hosts
// waves s and s + 4.
static int diag_cost(int ra) {
  const int t = ra * (ra + 1) / 2;
  int worst = 0;
  for (int sd = 0; sd < 4; ++sd) {
    int n = 0;
    for (int w = sd; w < 8; w += 4) n += std::max(0, (t - w + 7) / 8);
    worst = std::max(worst, n);
  }
  return std::max(1, worst);
}

static int schur_dense_width(int V) { return std::max(16, ((7 * V + 15) / 16) * 16); }

static SchurPlan make_plan(const BaDev& d) {
  SchurPlan pl;
  pl.dbg = 0;
  // the columns of Zd are the rows of S themselves (camera c, parameter i -> 7 c + i), padded to a multiple of the
  // 16-row MFMA strip; 128-row blocks cut through cameras, which only the reduce has to know (it maps elements)
  const int width = schur_dense_width(d.V);
  pl.cam_blocks = 0;
  pl.nblk = (width + RB - 1) / RB;
  pl.n_off = pl.nblk * (pl.nblk - 1) / 2;
  pl.ra_last = (width - (pl.nblk - 1) * RB) / 16;
  const int slabs = std::max(1, d.zrows / KSL);
  // ONE workgroup per CU in total (each needs 144 KB of LDS, so a CU hosts one at a time): a single even
  // round pays the per-workgroup prologue / slab write once.  Rows per chunk are inversely proportional to the
  // class's MFMAs per k-step; shrink the common time budget until everything fits one round.
  const double cost[4] = {16.0, 2.0 * pl.ra_last, 9.0, (double)diag_cost(pl.ra_last)};
  double total = 0;
  for (int c = 0; c < 4; ++c) total += cost[c] * plan_tiles_in_class(pl, c);
  double budget = total * slabs / std::max(1, ctx().num_cus);      // cost x slabs per workgroup
  for (int attempt = 0; attempt < 200; ++attempt) {
    for (int c = 0; c < 4; ++c) {
      int slabs_per = std::max(1, std::min(slabs, (int)(budget / cost[c])));
      pl.chunks[c] = (slabs + slabs_per - 1) / slabs_per;
      pl.rpc[c] = slabs_per * KSL;
    }
    if (plan_wgs(pl) <= ctx().num_cus) break;
    budget *= 1.01;
  }
  // ONE tile (up to 18 cameras): it is split into at most kSchurMaxChunks = 128 slabs, and a workgroup gets at least three slabs of 16 rows,
  // even if CUs stay idle -- a workgroup's k loop is short against its prologue there (a slab is ~0.15 us of a 6-9 us kernel), while every
  // slab is 128 KB more for the reduce to sum per element row.  Sweep of the cap (16 ... 256, profiles/r4/sweep_schur_chunks.txt), us per
  // iteration at one per CU -> now: 6 x 1 260 32.2 -> 28.9, 8 x 2 000 36.4 -> 34.4, 10 x 3 000 51.9 -> 50.5, 14 x 3 000 63.1 -> 61.2, 18 x 3 000
  // 68.4 -> 65.2.  (With three tiles, 19-36 cameras, a cap costs the product more than the reduce gains: 20 x 3 000 71.5 -> 74.6 us at 80.)
  static const int cap_env = [] { const char* e = getenv("SFM_SCHUR_MAX_CHUNKS"); return e ? atoi(e) : 0; }();      // (the sweep)
  const int cap = cap_env > 0 ? cap_env : std::max(1, std::min(kSchurMaxChunks, slabs / 3));
  if (pl.nblk == 1)
    for (int c = 0; c < 4; ++c)
      if (pl.chunks[c] > cap) {
        const int slabs_per = (slabs + cap - 1) / cap;
        pl.chunks[c] = (slabs + slabs_per - 1) / slabs_per;
        pl.rpc[c] = slabs_per * KSL;
      }
  return pl;
}

// Work split of the sparse product: every tile gets the same number of point chunks, about two workgroups per
// CU in total (one is resident per CU: 128 KB of LDS, 16 waves), at least one point per wave
// (small scenes are latency-bound: a visit costs ~3 us per wave, so spread them over as many waves as possible).
static SchurPlan make_pairs_plan(const BaDev& d) {
  SchurPlan pl;
  pl.dbg = 0;
  pl.nblk = (d.V + CB - 1) / CB;
  pl.n_off = pl.nblk * (pl.nblk - 1) / 2;
  pl.ra_last = 8;
  pl.cam_blocks = 1;
  const int ntiles = pl.n_off + pl.nblk;
  int chunks = std::max(1, 2 * ctx().num_cus / ntiles);
  chunks = std::max(1, std::min(chunks, (d.N + PAIR_WAVES - 1) / PAIR_WAVES));
  const int ppc = std::max(1, (d.N + chunks - 1) / chunks);
  chunks = std::max(1, (d.N + ppc - 1) / ppc);      // (an empty shard -- N = 0 -- still plans one chunk per tile)
  for (int c = 0; c < 4; ++c) { pl.chunks[c] = chunks; pl.rpc[c] = std::max(1, ppc); }
  return pl;
}

// Plans of both products: block offsets of every point (sparse path), shape of Zd and chunking of its rows
// (dense path), and the split-K slab workspace both share.  Zd itself is allocated (and zero-filled) on first
// use by ba_schur_prepare_dense.
int ba_schur_plan(sfm_ba_problem* p) {
  BaDev& d = p->dev;
  d.zp = schur_dense_width(d.V);
  d.zrows = ((3 * d.N + KSL - 1) / KSL) * KSL;
  const SchurPlan pl = make_plan(d);
  const SchurPlan pp = make_pairs_plan(d);
  const int wgs = plan_wgs(pl);
  const int wgs_pairs = plan_wgs(pp);
  const size_t ws_dense = sizeof(double) * (size_t)wgs * RB * RB;
  const size_t ws_pairs = sizeof(double) * (size_t)wgs_pairs * RB * RB;
  const size_t zd_bytes = sizeof(double) * ((size_t)d.zrows * d.zp + RB);
  // the dense path is only ever chosen when it is cheaper than the pair path; do not reserve
  // tens of gigabytes for scenes that will never take it
  p->schur_mfma_ok = d.N > 0 && ws_dense + zd_bytes <= ((size_t)32 << 30);
  SFM_HIP(pool_alloc(&p->schur_ws, std::max(p->schur_mfma_ok ? ws_dense : 0, ws_pairs)));
  // per-point block offsets blk_ptr[p][b] = first observation of point p whose camera is >= 18 b (b = nblk: the
  // end of the track); filled on the device by ba_structure_kernel
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->schur_blk_ptr), sizeof(int) * (size_t)std::max(1, d.N) * (pp.nblk + 1)));
  // once per device (every sfm_ba_create / sfm_ba_append plans a problem: the per-view loop of the reference appends after every
  // registered view); a process that re-initialises the library on another device sets them again
  static int attr_device = -1;
  if (attr_device != ctx().device) {
    SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kSchurLdsBytes));
    SFM_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(ba_schur_pairs_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kPairLdsBytes));
    attr_device = ctx().device;
  }
  return SFM_OK;
}

// Zd on first use of the dense path: allocated once and cleared once -- ba_linearize rewrites every visible
// (point, camera) block each iteration, everything else stays zero.
int ba_schur_prepare_dense(sfm_ba_problem* p, hipStream_t s) {
  BaDev& d = p->dev;
  if (d.Zd != nullptr) return SFM_OK;
  // + RB: the staging loads of a panel are 128 columns wide whatever the last block holds, so the panel of the last
  // block reads up to 128 - 16 ra_last columns into the NEXT row of Zd -- values that only feed strips nobody computes --
  // and, for the last row, that far past it
  const size_t zd_bytes = sizeof(double) * ((size_t)d.zrows * d.zp + RB);
  SFM_HIP(pool_alloc(reinterpret_cast<void**>(&d.Zd), zd_bytes));
  SFM_HIP(hipMemsetAsync(d.Zd, 0, zd_bytes, s));
  return SFM_OK;
}

// Which product kernel the next iteration launches: SFM_SCHUR_MFMA (dense), SFM_SCHUR_ROWS (sparse, row panels) or
// SFM_SCHUR_PAIRS (sparse, 18-camera tiles).  AUTO compares three cost models fitted on MI355X (profiles/r2n for the large
// scenes; round 3 refitted the constants on 6-30 cameras, profiles/r3/time_small.txt, where round 2's "+15 us" for the dense
// product made AUTO pick the tiles although the dense kernel was 2-15 us faster from nine cameras on):
//   dense   ~29 T MAC/s of its s (s + 1) / 2 MFMA tiles (s = ceil(7V / 16) strips) x 256 x 3N MACs + 4 us
//   tiles   ~0.65 ns per (point, tile) visit + ~0.7 ps per LDS add + the split-K slabs' write and re-read (one used tile per
//           workgroup at ~8 TB/s) + 4 us   (C4 share: 975 k visits, 285 M adds -> 0.63 ms; 20 x 3000 @ 0.3: 510 workgroups
//           flushing 126 x 126 tiles -> 22 us against the dense kernel's 6)
//   rows    ~50 ps per camera pair + ~50 ps per observation + 18 us          (C4 share: 5.8 M pairs -> 0.33 ms; C3: 0.43 ms;
//           a 6 x 1260 scene: 26 us against the tiles' 17)
int ba_schur_choice(const sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  if (p->deterministic && p->schur_mfma_ok) return SFM_SCHUR_MFMA;       // the sparse products accumulate with unordered LDS atomics
  const bool rows_possible = !(p->debug & 128) && (size_t)7 * (((7 * d.V + 1) / 2) * 2) * sizeof(double) <= ((size_t)156 << 10) &&
                             !(p->rows_built && !p->rows_ok);
  if (p->schur_mode == SFM_SCHUR_MFMA && p->schur_mfma_ok) return SFM_SCHUR_MFMA;
  if (p->schur_mode == SFM_SCHUR_PAIRS) return SFM_SCHUR_PAIRS;
  if (p->schur_mode == SFM_SCHUR_ROWS) return rows_possible ? SFM_SCHUR_ROWS : SFM_SCHUR_PAIRS;
  if (d.N == 0 || d.M == 0) return SFM_SCHUR_PAIRS;
  const double nblk = (double)((d.V + CB - 1) / CB);
  const double kbar = (double)d.M / d.N;
  const double strips = schur_dense_width(d.V) / 16.0;                          // lower-triangular 16 x 16 MFMA tiles
  const double dense_s = 0.5 * strips * (strips + 1.0) * 256.0 * 3.0 * d.N / 29e12 + 4e-6;
  const double nocc = nblk * (1.0 - std::pow(1.0 - 1.0 / nblk, kbar));            // occupied blocks per point
  const double visits = 0.5 * nocc * (nocc + 1.0) * d.N;
  const double pairs = 0.5 * kbar * (kbar + 1) * d.N;
  const SchurPlan pp = make_pairs_plan(d);
  const double tile_side = 7.0 * std::min(CB, d.V);
  const double tiles_s = visits * 0.65e-9 + 49.0 * pairs * 0.7e-12 + plan_wgs(pp) * tile_side * tile_side * 16.0 / 8e12 + 4e-6;
  const double rows_s = rows_possible ? 50e-12 * pairs + 50e-12 * (double)d.M + 18e-6 : 1e30;
  const double mfma_s = p->schur_mfma_ok ? dense_s : 1e30;
  if (mfma_s <= tiles_s && mfma_s <= rows_s) return SFM_SCHUR_MFMA;
  return rows_s < tiles_s ? SFM_SCHUR_ROWS : SFM_SCHUR_PAIRS;
}

bool ba_schur_uses_mfma(const sfm_ba_problem* p) { return ba_schur_choice(p) == SFM_SCHUR_MFMA; }

SchurPlan ba_schur_dense_plan(const sfm_ba_problem* p) { return make_plan(p->dev); }

int ba_enqueue_schur(sfm_ba_problem* p, hipStream_t s, bool allow_defer) {
  const BaDev& d = p->dev;
  p->reduce_deferred = false;
  p->last_reduce_deferred = false;
  if (d.N == 0 || d.M == 0) return SFM_OK;
  double* ws = static_cast<double*>(p->schur_ws);
  SchurPlan pl;
  int choice = ba_schur_choice(p);
  if (choice == SFM_SCHUR_ROWS) {            // camera-major list and work split on first use (blocking once)
    if (!p->rows_built) {
      SFM_TRY(ba_rows_enqueue_build(p));
      SFM_TRY(ba_rows_plan(p));
      p->rows_built = true;
    }
    if (!p->rows_ok) choice = SFM_SCHUR_PAIRS;
  }
  if (choice == SFM_SCHUR_MFMA) {
    pl = make_plan(d);
    pl.dbg = p->debug;
    const int wgs = plan_wgs(pl);
    ba_tick(p, SFM_K_SCHUR, true, s);
    ba_schur_mfma_kernel<<<wgs, SCHUR_THREADS, kSchurLdsBytes, s>>>(d, ws, pl);
    ba_tick(p, SFM_K_SCHUR, false, s);
    if (allow_defer && ba_solve_can_defer_reduce(p)) {      // the data-flow solve sums the slabs and the camera accumulators itself
      p->reduce_deferred = true;
      p->last_reduce_deferred = true;
      SFM_HIP(hipGetLastError());
      return SFM_OK;
    }
  } else if (choice == SFM_SCHUR_ROWS) {
    // row-panel sparse product and its own reduce, which also adds the camera accumulators and sums the cost
    SFM_TRY(ba_rows_enqueue(p, s));
    ba_tick(p, SFM_K_REDUCE, false, s);
    SFM_HIP(hipGetLastError());
    return SFM_OK;
  } else {
    pl = make_pairs_plan(d);
    const int wgs = plan_wgs(pl);
    ba_tick(p, SFM_K_SCHUR, true, s);
    // a scene with fewer than 18 cameras uses only part of the tile: a smaller LDS footprint lets several
    // workgroups share a CU (small scenes are latency-bound)
    const size_t lds = sizeof(double) * (size_t)7 * std::min(CB, d.V) * TP;
    ba_schur_pairs_kernel<<<wgs, PAIR_THREADS, lds, s>>>(d, p->schur_blk_ptr, ws, pl);
    ba_tick(p, SFM_K_SCHUR, false, s);
  }
  const int ntiles = pl.n_off + pl.nblk;
  const int tile_blocks = (ntiles * RB * RB + 255) / 256;
  const int cam_blocks = p->lin_rows > 0 ? 12 * ((d.V * 35 + 255) / 256) : 0;
  ba_tick(p, SFM_K_REDUCE, true, s);
  if (p->deterministic && p->lin_rows > 0) ba_schur_reduce_det_kernel<<<tile_blocks + (d.P + 255) / 256 + 1, 256, 0, s>>>(d, ws, pl, tile_blocks, p->lin_rows, p->lin_grid);
  else ba_schur_reduce_kernel<<<dim3(tile_blocks + cam_blocks + 1, 4), 256, 0, s>>>(d, ws, pl, tile_blocks, p->lin_rows, p->lin_grid);
  ba_tick(p, SFM_K_REDUCE, false, s);
  SFM_HIP(hipGetLastError());
  return SFM_OK;
}

}  // namespace sfm
