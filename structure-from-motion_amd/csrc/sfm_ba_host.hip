// sfm_ba_host.hip — the device-resident bundle-adjustment problem object behind the C-ABI (gfx950).
//
// Host side of BaProcessor.__execute_bundle_adjustment (ba_processor.py:274-439) and of its caller, the
// per-view loop of BaProcessor.process (ba_processor.py:137-267): create / destroy, state upload and
// download, growing a resident scene in place (sfm_ba_append), the multi-GPU split, parity hooks.
//
// Everything that scales with the scene is built ON THE DEVICE: the host uploads the caller's CSR as it is
// (one copy per array) and ba_structure_kernel validates it, derives obs_pt, the per-point 18-camera block
// offsets of the sparse Schur product and the longest track; sfm_ba_append uploads only the NEW cameras,
// points and observations and merges them into the (point, camera)-sorted list with a count / scan /
// bucket / merge kernel chain.  The host never loops over observations.
#include <algorithm>
#include <vector>

#include "sfm_ba.h"

namespace sfm {

// ---------------------------------------------------------------------------------------------
// Structure check + derived index arrays, one thread per point.
// sinfo[0] = first failure code (1 pt_ptr not monotone, 2 camera out of range, 3 cameras of a track not strictly
// increasing), sinfo[1] = its index (point, observation, point), sinfo[2] = longest track.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void structure_fail(int* sinfo, int code, int index) {
  if (atomicCAS(&sinfo[0], 0, code) == 0) sinfo[1] = index;
}

__global__ void ba_structure_kernel(int V, int N, long long M, const int* __restrict__ pt_ptr,
                                    const int* __restrict__ cam_idx, int* __restrict__ obs_pt,
                                    int* __restrict__ blk_ptr, int nblk, int* __restrict__ sinfo) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= N) return;
  const int beg = pt_ptr[p], end = pt_ptr[p + 1];
  int* row = blk_ptr + (size_t)p * (nblk + 1);
  if (beg < 0 || end < beg || end > M) {
    structure_fail(sinfo, 1, p);
    for (int b = 0; b <= nblk; ++b) row[b] = 0;
    return;
  }
  int prev = -1, b = 0;
  for (int o = beg; o < end; ++o) {
    const int c = cam_idx[o];
    if (c < 0 || c >= V) { structure_fail(sinfo, 2, o); continue; }
    if (c <= prev) structure_fail(sinfo, 3, p);
    prev = c;
    obs_pt[o] = p;
    while (b < nblk && c >= b * kSchurCB) row[b++] = o;      // first observation whose camera is >= 18 b
  }
  while (b <= nblk) row[b++] = end;
  atomicMax(&sinfo[2], end - beg);
}

// ---------------------------------------------------------------------------------------------
// sfm_ba_append on the device.  New observation k = (obs_cam[k], obs_pt[k], u_new[k], v_new[k]).
//   count   per new observation: range check, cnt[point]++
//   scan    one workgroup: new_ptr = exclusive scan of (old track length + cnt), nstart = exclusive scan of cnt
//   bucket  per new observation: its slot in its point's bucket (order inside a bucket is fixed by the merge's sort)
//   merge   per point: sort the bucket by camera, merge with the old (sorted) track -> cam2 / u2 / v2
// Duplicates (a pair already present) survive the merge as equal neighbours and are caught by
// ba_structure_kernel's strictly-increasing check.
// ---------------------------------------------------------------------------------------------
__global__ void ba_append_count_kernel(long long n, int V2, int N2, const int* __restrict__ obs_cam,
                                       const int* __restrict__ obs_pt, int* __restrict__ cnt, int* __restrict__ sinfo) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int c = obs_cam[k], q = obs_pt[k];
  if (c < 0 || c >= V2 || q < 0 || q >= N2) { structure_fail(sinfo, 4, (int)k); return; }
  atomicAdd(&cnt[q], 1);
}

// exclusive prefix sums of two integer sequences in one pass, one 1024-thread workgroup
__global__ __launch_bounds__(1024) void ba_append_scan_kernel(int N, int N2, const int* __restrict__ old_ptr,
                                                              const int* __restrict__ cnt, int* __restrict__ new_ptr,
                                                              int* __restrict__ nstart) {
  __shared__ int wsum[2][16];
  __shared__ int carry[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < 2) carry[tid] = 0;
  __syncthreads();
  for (int base = 0; base < N2; base += 1024) {
    const int q = base + tid;
    int a = 0, b = 0;
    if (q < N2) {
      b = cnt[q];
      a = b + (q < N ? old_ptr[q + 1] - old_ptr[q] : 0);
    }
    int sa = a, sb = b;                              // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int ta = __shfl_up(sa, off, 64), tb = __shfl_up(sb, off, 64);
      if (lane >= off) { sa += ta; sb += tb; }
    }
    if (lane == 63) { wsum[0][wave] = sa; wsum[1][wave] = sb; }
    __syncthreads();
    int oa = carry[0], ob = carry[1];
    for (int w = 0; w < wave; ++w) { oa += wsum[0][w]; ob += wsum[1][w]; }
    if (q < N2) { new_ptr[q] = oa + sa - a; nstart[q] = ob + sb - b; }
    __syncthreads();
    if (tid == 1023) { carry[0] = oa + sa; carry[1] = ob + sb; }
    __syncthreads();
  }
  if (tid == 0) { new_ptr[N2] = carry[0]; nstart[N2] = carry[1]; }
}

__global__ void ba_append_bucket_kernel(long long n, const int* __restrict__ obs_pt, const int* __restrict__ nstart,
                                        int* __restrict__ fill, int* __restrict__ norder) {
  const long long k = blockIdx.x * (long long)blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int q = obs_pt[k];
  norder[nstart[q] + atomicAdd(&fill[q], 1)] = (int)k;
}

__global__ void ba_append_merge_kernel(int N, int N2, const int* __restrict__ old_ptr, const int* __restrict__ old_cam,
                                       const double* __restrict__ old_u, const double* __restrict__ old_v,
                                       const int* __restrict__ new_ptr, const int* __restrict__ nstart,
                                       int* __restrict__ norder, const int* __restrict__ obs_cam,
                                       const double* __restrict__ u_new, const double* __restrict__ v_new,
                                       int* __restrict__ cam2, double* __restrict__ u2, double* __restrict__ v2) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= N2) return;
  const int nb = nstart[q], ne = nstart[q + 1];
  for (int i = nb + 1; i < ne; ++i) {                 // insertion sort of the (short) bucket by camera
    const int k = norder[i], c = obs_cam[k];
    int j = i - 1;
    while (j >= nb && obs_cam[norder[j]] > c) { norder[j + 1] = norder[j]; --j; }
    norder[j + 1] = k;
  }
  int o = q < N ? old_ptr[q] : 0;
  const int oe = q < N ? old_ptr[q + 1] : 0;
  int i = nb, w = new_ptr[q];
  while (o < oe || i < ne) {
    const int co = o < oe ? old_cam[o] : 0x7fffffff;
    const int cn = i < ne ? obs_cam[norder[i]] : 0x7fffffff;
    if (co <= cn) { cam2[w] = co; u2[w] = old_u[o]; v2[w] = old_v[o]; ++o; }
    else { const int k = norder[i]; cam2[w] = cn; u2[w] = u_new[k]; v2[w] = v_new[k]; ++i; }
    ++w;
  }
}

// The reference packs every camera anew at the start of each BA call: q = convert_rotation_to_quaternion(view.rot)
// (ba_processor.py:285-288), and view.rot is R(q) of the previous call's result (ba:412) -- so the quaternion a call starts
// from is q(R(q_prev)), not q_prev.  For cameras the caller did not touch, that round trip is done here, on the device, from
// the R(q) and canonical q the camera expansion holds: nothing crosses PCIe.
__global__ void ba_rederive_quat_kernel(int first, int count, const CamPrep* __restrict__ prep, double* __restrict__ cams) {
  const int c = first + blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= first + count) return;
  for (int k = 0; k < 4; ++k) cams[7 * c + 3 + k] = prep[c].q[k];
}

__global__ void ba_gather_rot_kernel(int V, const CamPrep* __restrict__ prep, double* __restrict__ rots) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9 * V) return;
  rots[i] = prep[i / 9].R[i % 9];
}

static int check_problem(const sfm_ba_problem* p) {
  if (p == nullptr || p->magic != kBaMagic) {
    set_error("invalid bundle-adjustment problem handle");
    return SFM_E_HANDLE;
  }
  return SFM_OK;
}

static const char* status_name(int st) {
  switch (st) {
    case SFM_E_BAD_ROTATION: return "invalid rotation matrix";
    case SFM_E_QW_ZERO: return "quaternion qw ~ 0";
    case SFM_E_SQRT_DOMAIN: return "1 + trace(R) < 0";
    default: return "unknown";
  }
}

// Enqueue the structure kernel (all index arrays of p->dev must be on the device already).
int ba_enqueue_structure(sfm_ba_problem* p) {
  const BaDev& d = p->dev;
  hipStream_t s = p->stream;
  SFM_HIP(hipMemsetAsync(d.sinfo, 0, 4 * sizeof(int), s));
  if (d.N > 0) {
    const int nblk = (d.V + kSchurCB - 1) / kSchurCB;
    ba_structure_kernel<<<(d.N + 255) / 256, 256, 0, s>>>(d.V, d.N, d.M, d.pt_ptr, d.cam_idx, d.obs_pt, p->schur_blk_ptr, nblk, d.sinfo);
    SFM_HIP(hipGetLastError());
  }
  return SFM_OK;
}

// Wait for the structure kernel and turn its verdict into a status / message.  `who` names the entry point.
static int ba_finish_structure(sfm_ba_problem* p, const char* who) {
  int info[4] = {0, 0, 0, 0};
  SFM_HIP(hipMemcpyAsync(info, p->dev.sinfo, sizeof(info), hipMemcpyDeviceToHost, p->stream));
  SFM_TRY(stream_sync(p->stream));
  p->max_track = info[2];
  switch (info[0]) {
    case 0: return SFM_OK;
    case 1: set_error("%s: pt_ptr not monotone at point %d", who, info[1]); break;
    case 2: set_error("%s: cam_idx[%d] out of range", who, info[1]); break;
    case 3: set_error("%s: observations of point %d are not sorted by strictly increasing camera "
                      "(a camera observes the point twice, or the point is already observed by it)", who, info[1]); break;
    default: set_error("%s: observation %d (camera, point) out of range", who, info[1]); break;
  }
  return SFM_E_SHAPE;
}

// Allocate a problem of the given sizes on `stream` (nothing uploaded, no structure yet).
static int ba_alloc_problem(int V, int N, long long M, hipStream_t stream, sfm_ba_problem** out) {
  sfm_ba_problem* p = new sfm_ba_problem();
  p->stream = stream;
  BaDev& d = p->dev;
  d.V = V; d.N = N; d.M = M; d.P = 7 * V;
  d.nbk = (d.P + kNB - 1) / kNB;
  auto fail = [&](int st) { sfm_ba_destroy(p); return st; };
#define BA_ALLOC(ptr, count) do { hipError_t e_ = pool_alloc(reinterpret_cast<void**>(&(ptr)), sizeof(*(ptr)) * std::max<size_t>(1, (count))); \
    if (e_ != hipSuccess) return fail(hip_fail(e_, "hipMalloc " #ptr, __LINE__)); } while (0)
  BA_ALLOC(d.pt_ptr, (size_t)N + 1);
  BA_ALLOC(d.cam_idx, (size_t)M);
  BA_ALLOC(d.obs_pt, (size_t)M);
  BA_ALLOC(d.u, (size_t)M);
  BA_ALLOC(d.v, (size_t)M);
  BA_ALLOC(d.cams, (size_t)V * 7);
  BA_ALLOC(d.px, (size_t)N); BA_ALLOC(d.py, (size_t)N); BA_ALLOC(d.pz, (size_t)N);
  BA_ALLOC(d.prep[0], (size_t)V); BA_ALLOC(d.prep[1], (size_t)V);
  BA_ALLOC(d.lin_ws, (sizeof(double) * V * 35 <= 64 * 1024) ? (size_t)kLinGridPerCu * ctx().num_cus * V * 35 : 1);
  BA_ALLOC(p->own_red, red_size(d.nbk));
  BA_ALLOC(d.delta, (size_t)d.nbk * kNB);
  BA_ALLOC(d.ldiag, (size_t)((d.P + 31) / 32) * 32 * 32);
  BA_ALLOC(d.xinv, d.nbk <= kInvRowsMaxNbk ? red_rhs_off(d.nbk) : 1);      // beyond that the back substitution runs block row by block row
  BA_ALLOC(d.sync_ctr, 1);
  BA_ALLOC(d.status, 2);
  BA_ALLOC(d.sinfo, 4);
  BA_ALLOC(d.cost, kStatSlots);
  BA_ALLOC(d.cost_ws, (size_t)kLinGridPerCu * ctx().num_cus);
  BA_ALLOC(d.iter_count, 1);
#undef BA_ALLOC
  d.red = p->own_red;
  if (hipMemsetAsync(d.status, 0, 2 * sizeof(int), stream) != hipSuccess) return fail(SFM_E_HIP);
  if (hipMemsetAsync(d.cost, 0, kStatSlots * sizeof(double), stream) != hipSuccess) return fail(SFM_E_HIP);
  if (hipMemsetAsync(d.iter_count, 0, sizeof(int), stream) != hipSuccess) return fail(SFM_E_HIP);
  if (hipMemsetAsync(d.sync_ctr, 0, sizeof(int), stream) != hipSuccess) return fail(SFM_E_HIP);
  if (hipMemsetAsync(d.delta, 0, sizeof(double) * d.nbk * kNB, stream) != hipSuccess) return fail(SFM_E_HIP);
  { const int st_plan = ba_schur_plan(p); if (st_plan != SFM_OK) return fail(st_plan); }
  { const int st_flow = ba_flow_setup(p); if (st_flow != SFM_OK) return fail(st_flow); }
  *out = p;
  return SFM_OK;
}

// A new state starts a new cost history (sfm_ba_get_stats).
static int ba_reset_stats(sfm_ba_problem* p) {
  SFM_HIP(hipMemsetAsync(p->dev.cost, 0, kStatSlots * sizeof(double), p->stream));
  SFM_HIP(hipMemsetAsync(p->dev.iter_count, 0, sizeof(int), p->stream));
  return SFM_OK;
}

static int ba_upload(sfm_ba_problem* p, void* dst, const void* src, size_t bytes) {
  if (bytes == 0) return SFM_OK;
  SFM_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, p->stream));
  p->upload_bytes += (long long)bytes;
  return SFM_OK;
}

}  // namespace sfm

using namespace sfm;

extern "C" {

int sfm_ba_create(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx, const double* uv_norm,
                  sfm_ba_problem** out) {
  SFM_TRY(ensure_init());
  if (out == nullptr) { set_error("sfm_ba_create: out is null"); return SFM_E_SHAPE; }
  *out = nullptr;
  if (V < 1 || N < 0 || M < 0 || M > 0x7fffffffLL) {
    set_error("sfm_ba_create: bad sizes V=%d N=%d M=%lld", V, N, (long long)M);
    return SFM_E_SHAPE;
  }
  if (N == 0 && M > 0) { set_error("sfm_ba_create: %lld observations but no point", (long long)M); return SFM_E_SHAPE; }
  if (M > 0 && uv_norm == nullptr) { set_error("sfm_ba_create: uv_norm is null"); return SFM_E_SHAPE; }
  if (N > 0 && (pt_ptr[0] != 0 || pt_ptr[N] != M)) { set_error("sfm_ba_create: pt_ptr must span [0, M]"); return SFM_E_SHAPE; }
  sfm_ba_problem* p = nullptr;
  SFM_TRY(ba_alloc_problem(V, N, M, ctx().stream, &p));
  BaDev& d = p->dev;
  const int zero_ptr = 0;
  int st = N > 0 ? ba_upload(p, d.pt_ptr, pt_ptr, sizeof(int) * ((size_t)N + 1)) : ba_upload(p, d.pt_ptr, &zero_ptr, sizeof(int));
  if (st == SFM_OK) st = ba_upload(p, d.cam_idx, cam_idx, sizeof(int) * (size_t)M);
  if (st == SFM_OK) st = ba_upload(p, d.u, uv_norm, sizeof(double) * (size_t)M);
  if (st == SFM_OK) st = ba_upload(p, d.v, uv_norm + M, sizeof(double) * (size_t)M);
  if (st == SFM_OK) st = ba_enqueue_structure(p);
  if (st == SFM_OK) st = ba_finish_structure(p, "sfm_ba_create");      // synchronises: the caller's arrays are free again
  if (st != SFM_OK) { (void)hipStreamSynchronize(p->stream); sfm_ba_destroy(p); return st; }
  *out = p;
  return SFM_OK;
}

int sfm_ba_destroy(sfm_ba_problem* p) {
  if (p == nullptr) return SFM_OK;
  if (p->magic != kBaMagic) { set_error("sfm_ba_destroy: invalid handle"); return SFM_E_HANDLE; }
  if (ctx().inited && p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->comm) { (void)comm_attach(p->comm, -1); p->comm = nullptr; }
  ba_graph_drop(p);
  BaDev& d = p->dev;
  void* ptrs[] = {d.pt_ptr, d.cam_idx, d.obs_pt, d.u, d.v, d.cams, d.px, d.py, d.pz, d.prep[0], d.prep[1],
                  d.Z, d.Zd, d.lin_ws, d.stamps, p->own_red, d.delta, d.ldiag, d.xinv, d.sync_ctr, d.flow, d.status, d.sinfo, d.cost, d.cost_ws, d.iter_count,
                  p->schur_ws, p->flow_camsum, p->schur_blk_ptr, p->cam_ptr, p->cam_ent, p->cam_pairs, p->rows_table, p->rows_first, p->rows_ws};
  for (void* q : ptrs) if (q) pool_free(q);
  for (auto& t : p->timers)
    for (auto& e : t.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
  p->magic = 0;
  delete p;
  return SFM_OK;
}

int sfm_ba_set_stream(sfm_ba_problem* p, void* hip_stream) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  SFM_HIP(hipStreamSynchronize(p->stream));
  ba_graph_drop(p);
  p->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : ctx().own;
  return SFM_OK;
}

int sfm_ba_set_option(sfm_ba_problem* p, int option, int value) {
  SFM_TRY(check_problem(p));
  ba_graph_drop(p);       // every option changes what an iteration launches
  switch (option) {
    case SFM_OPT_GRAPH:
      p->use_graph = value != 0;
      return SFM_OK;
    case SFM_OPT_SCHUR:
      SFM_TRY(ba_flush(p));
      if (value < SFM_SCHUR_AUTO || value > SFM_SCHUR_ROWS) { set_error("bad schur mode %d", value); return SFM_E_SHAPE; }
      p->schur_mode = value;
      return SFM_OK;
    case SFM_OPT_DEBUG:
      SFM_TRY(ba_flush(p));
      p->debug = value;
      p->dev.debug = value;
      if ((value & 8) && p->dev.stamps == nullptr) {
        SFM_HIP(pool_alloc(reinterpret_cast<void**>(&p->dev.stamps), sizeof(unsigned long long) * 1024));
        SFM_HIP(hipMemset(p->dev.stamps, 0, sizeof(unsigned long long) * 1024));
      }
      return SFM_OK;
    case SFM_OPT_TIMING:
      p->timing = value;   // bit k set = time kernel class k
      return SFM_OK;
    case SFM_OPT_TIMING_STRIDE:
      if (value < 1) { set_error("timing stride %d < 1", value); return SFM_E_SHAPE; }
      p->timing_stride = value;
      return SFM_OK;
    case SFM_OPT_DETERMINISTIC:
      SFM_TRY(ba_flush(p));
      if (value != 0) {
        // needs the atomic-free dense product (Zd resident) and the LDS camera accumulators of ba_linearize
        if (!p->schur_mfma_ok) { set_error("deterministic mode needs the dense Schur product, which does not fit this scene"); return SFM_E_SHAPE; }
        if (sizeof(double) * (size_t)p->dev.V * 35 > 64 * 1024) { set_error("deterministic mode supports at most 234 cameras"); return SFM_E_SHAPE; }
      }
      p->deterministic = value != 0;
      return SFM_OK;
    default:
      set_error("unknown option %d", option);
      return SFM_E_SHAPE;
  }
}

int sfm_ba_info(sfm_ba_problem* p, int what, int64_t* value) {
  SFM_TRY(check_problem(p));
  if (value == nullptr) { set_error("sfm_ba_info: value is null"); return SFM_E_SHAPE; }
  switch (what) {
    case SFM_INFO_SCHUR_KERNEL: {
      // the kernel the next iteration WILL launch: the row-panel product needs its camera-major list and work split,
      // which decide whether it can run at all -- build them now rather than answer "rows" and then launch "pairs"
      int choice = ba_schur_choice(p);
      if (choice == SFM_SCHUR_ROWS && p->dev.M > 0 && p->dev.N > 0) {
        if (!p->rows_built) {
          SFM_TRY(ba_flush(p));
          SFM_TRY(ba_rows_enqueue_build(p));
          SFM_TRY(ba_rows_plan(p));
          p->rows_built = true;
        }
        if (!p->rows_ok) choice = SFM_SCHUR_PAIRS;
      }
      *value = choice;
      return SFM_OK;
    }
    case SFM_INFO_UPLOAD_BYTES: *value = p->upload_bytes; return SFM_OK;
    case SFM_INFO_N_CAMS: *value = p->dev.V; return SFM_OK;
    case SFM_INFO_N_PTS: *value = p->dev.N; return SFM_OK;
    case SFM_INFO_N_OBS: *value = p->dev.M; return SFM_OK;
    case SFM_INFO_MAX_TRACK: *value = p->max_track; return SFM_OK;
    case SFM_INFO_GRAPH_REPLAYS: *value = p->graph_replays; return SFM_OK;
    case SFM_INFO_REDUCE_IN_SOLVE: *value = p->last_reduce_deferred ? 1 : 0; return SFM_OK;
    default: set_error("sfm_ba_info: unknown item %d", what); return SFM_E_SHAPE;
  }
}

int sfm_ba_flush(sfm_ba_problem* p) {
  SFM_TRY(check_problem(p));
  return ba_flush(p);
}

int sfm_ba_set_cameras(sfm_ba_problem* p, const double* cams) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  BaDev& d = p->dev;
  SFM_TRY(ba_upload(p, d.cams, cams, sizeof(double) * 7 * d.V));
  SFM_HIP(hipMemsetAsync(d.status, 0, 2 * sizeof(int), p->stream));
  SFM_TRY(ba_reset_stats(p));
  SFM_TRY(stream_sync(p->stream));
  p->prep_valid = false;
  return SFM_OK;
}

int sfm_ba_set_points(sfm_ba_problem* p, int first, int count, const double* pts) {
  SFM_TRY(check_problem(p));
  BaDev& d = p->dev;
  SFM_TRY(ba_flush(p));
  if (first < 0 || count < 0 || first + (long long)count > d.N) {
    set_error("sfm_ba_set_points: range [%d, %d) outside the %d points", first, first + count, d.N);
    return SFM_E_SHAPE;
  }
  SFM_TRY(ba_upload(p, d.px + first, pts, sizeof(double) * count));
  SFM_TRY(ba_upload(p, d.py + first, pts + count, sizeof(double) * count));
  SFM_TRY(ba_upload(p, d.pz + first, pts + 2 * (size_t)count, sizeof(double) * count));
  SFM_TRY(ba_reset_stats(p));
  SFM_TRY(stream_sync(p->stream));
  return SFM_OK;
}

int sfm_ba_set_state(sfm_ba_problem* p, const double* cams, const double* pts) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  BaDev& d = p->dev;
  SFM_TRY(ba_upload(p, d.cams, cams, sizeof(double) * 7 * d.V));
  if (d.N > 0) {
    SFM_TRY(ba_upload(p, d.px, pts, sizeof(double) * d.N));
    SFM_TRY(ba_upload(p, d.py, pts + d.N, sizeof(double) * d.N));
    SFM_TRY(ba_upload(p, d.pz, pts + 2 * (size_t)d.N, sizeof(double) * d.N));
  }
  SFM_HIP(hipMemsetAsync(d.status, 0, 2 * sizeof(int), p->stream));
  SFM_TRY(ba_reset_stats(p));
  SFM_TRY(stream_sync(p->stream));
  p->prep_valid = false;
  return SFM_OK;
}

int sfm_ba_get_stats(sfm_ba_problem* p, double* cost, int max_iters, int* n_iters) {
  SFM_TRY(check_problem(p));
  if (max_iters < 0 || (max_iters > 0 && cost == nullptr)) { set_error("sfm_ba_get_stats: bad output buffer"); return SFM_E_SHAPE; }
  int done = 0;
  SFM_HIP(hipMemcpyAsync(&done, p->dev.iter_count, sizeof(int), hipMemcpyDeviceToHost, p->stream));
  SFM_TRY(stream_sync(p->stream));
  const int n = std::min(std::min(done, kStatSlots), max_iters);
  if (n > 0) SFM_HIP(hipMemcpyAsync(cost, p->dev.cost, sizeof(double) * n, hipMemcpyDeviceToHost, p->stream));
  SFM_TRY(stream_sync(p->stream));
  if (n_iters) *n_iters = n;
  return SFM_OK;
}

int sfm_ba_linearize_reduce(sfm_ba_problem* p, double lambda, int quirks) {
  SFM_TRY(check_problem(p));
  return ba_enqueue_linearize_reduce(p, lambda, quirks);
}

int sfm_ba_solve_update(sfm_ba_problem* p, double lambda, int quirks) {
  SFM_TRY(check_problem(p));
  return ba_enqueue_solve_update(p, lambda, quirks);
}

int sfm_ba_iterate(sfm_ba_problem* p, double lambda, int iters, int quirks) {
  SFM_TRY(check_problem(p));
  if (iters < 0) { set_error("sfm_ba_iterate: iters < 0"); return SFM_E_SHAPE; }
  return ba_enqueue_iterations(p, lambda, iters, quirks);
}

int sfm_ba_get_state(sfm_ba_problem* p, double* cams, double* pts) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  hipStream_t s = p->stream;
  BaDev& d = p->dev;
  if (!p->prep_valid) SFM_TRY(ba_enqueue_prep(p));     // validates the cameras even with zero iterations (ba:412)
  int st[2] = {0, 0};
  SFM_HIP(hipMemcpyAsync(cams, d.cams, sizeof(double) * 7 * d.V, hipMemcpyDeviceToHost, s));
  if (d.N > 0) {
    SFM_HIP(hipMemcpyAsync(pts, d.px, sizeof(double) * d.N, hipMemcpyDeviceToHost, s));
    SFM_HIP(hipMemcpyAsync(pts + d.N, d.py, sizeof(double) * d.N, hipMemcpyDeviceToHost, s));
    SFM_HIP(hipMemcpyAsync(pts + 2 * (size_t)d.N, d.pz, sizeof(double) * d.N, hipMemcpyDeviceToHost, s));
  }
  SFM_HIP(hipMemcpyAsync(st, d.status, sizeof(st), hipMemcpyDeviceToHost, s));
  SFM_TRY(stream_sync(s));
  if (st[0] != SFM_OK) {
    set_error("bundle adjustment: %s for camera %d", status_name(st[0]), st[1]);
    return st[0];
  }
  return SFM_OK;
}

int sfm_ba_rederive_quaternions(sfm_ba_problem* p, int first, int count) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  BaDev& d = p->dev;
  if (first < 0 || count < 0 || first + (long long)count > d.V) {
    set_error("sfm_ba_rederive_quaternions: range [%d, %d) outside the %d cameras", first, first + count, d.V);
    return SFM_E_SHAPE;
  }
  if (!p->prep_valid) SFM_TRY(ba_enqueue_prep(p));      // R(q) and q(R(q)) of the resident cameras (validated: status)
  if (count > 0) {
    ba_rederive_quat_kernel<<<(count + 63) / 64, 64, 0, p->stream>>>(first, count, d.prep[p->cur], d.cams);
    SFM_HIP(hipGetLastError());
  }
  SFM_TRY(ba_reset_stats(p));
  p->prep_valid = false;          // R(q') differs from R(q) in the last bits: expand again before linearising
  return SFM_OK;
}

int sfm_ba_get_state_rot(sfm_ba_problem* p, double* cams, double* pts, double* rots) {
  SFM_TRY(check_problem(p));
  if (rots == nullptr) return sfm_ba_get_state(p, cams, pts);
  SFM_TRY(ba_flush(p));
  if (!p->prep_valid) SFM_TRY(ba_enqueue_prep(p));
  BaDev& d = p->dev;
  DevBuf<double> dR;
  SFM_TRY(dR.alloc(9 * (size_t)d.V, p->stream));
  ba_gather_rot_kernel<<<(9 * d.V + 255) / 256, 256, 0, p->stream>>>(d.V, d.prep[p->cur], dR.p);
  SFM_HIP(hipGetLastError());
  SFM_TRY(dR.download(rots, 9 * (size_t)d.V, p->stream));
  return sfm_ba_get_state(p, cams, pts);      // synchronises, reports the first device-side failure
}

int sfm_ba_append(sfm_ba_problem* p, int n_new_cams, const double* cams_new, int n_new_pts, const double* pts_new,
                  int64_t n_new_obs, const int* obs_cam, const int* obs_pt, const double* uv_norm) {
  SFM_TRY(check_problem(p));
  if (n_new_cams < 0 || n_new_pts < 0 || n_new_obs < 0) { set_error("sfm_ba_append: negative count"); return SFM_E_SHAPE; }
  SFM_TRY(ba_flush(p));
  BaDev& d = p->dev;
  const int V2 = d.V + n_new_cams, N2 = d.N + n_new_pts;
  const long long M2 = d.M + n_new_obs;
  if (M2 > 0x7fffffffLL) { set_error("sfm_ba_append: too many observations"); return SFM_E_SHAPE; }
  if (N2 == 0 && M2 > 0) { set_error("sfm_ba_append: observations but no point"); return SFM_E_SHAPE; }
  hipStream_t s = p->stream;
  sfm_ba_problem* q = nullptr;
  SFM_TRY(ba_alloc_problem(V2, N2, M2, s, &q));
  BaDev& e = q->dev;
  auto run = [&]() -> int {
    // only the NEW data crosses PCIe; it is accounted to the surviving handle
    DevBuf<int> dcam, dpt, cnt, nstart, fill, norder;
    DevBuf<double> duv;
    const size_t n = (size_t)n_new_obs;
    SFM_TRY(dcam.upload(obs_cam, n, s)); SFM_TRY(dpt.upload(obs_pt, n, s)); SFM_TRY(duv.upload(uv_norm, 2 * n, s));
    p->upload_bytes += (long long)(n * (2 * sizeof(int) + 2 * sizeof(double)));
    SFM_TRY(cnt.alloc((size_t)N2 + 1, s)); SFM_TRY(nstart.alloc((size_t)N2 + 1, s));
    SFM_TRY(fill.alloc((size_t)N2 + 1, s)); SFM_TRY(norder.alloc(n, s));
    SFM_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int) * ((size_t)N2 + 1), s));
    SFM_HIP(hipMemsetAsync(fill.p, 0, sizeof(int) * ((size_t)N2 + 1), s));
    SFM_HIP(hipMemsetAsync(e.sinfo, 0, 4 * sizeof(int), s));
    if (n > 0) ba_append_count_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)n, V2, N2, dcam.p, dpt.p, cnt.p, e.sinfo);
    {   // a bad (camera, point) index must stop the chain before the bucket kernel writes through it
      int info[4] = {0, 0, 0, 0};
      SFM_HIP(hipMemcpyAsync(info, e.sinfo, sizeof(info), hipMemcpyDeviceToHost, s));
      SFM_TRY(stream_sync(s));
      if (info[0] != 0) {
        set_error("sfm_ba_append: observation %d = (camera %d, point %d) out of range", info[1], obs_cam[info[1]], obs_pt[info[1]]);
        return SFM_E_SHAPE;
      }
    }
    ba_append_scan_kernel<<<1, 1024, 0, s>>>(d.N, N2, d.pt_ptr, cnt.p, e.pt_ptr, nstart.p);
    if (n > 0) ba_append_bucket_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((long long)n, dpt.p, nstart.p, fill.p, norder.p);
    if (N2 > 0) ba_append_merge_kernel<<<(N2 + 255) / 256, 256, 0, s>>>(d.N, N2, d.pt_ptr, d.cam_idx, d.u, d.v, e.pt_ptr, nstart.p, norder.p,
                                                                        dcam.p, duv.p, duv.p + n, e.cam_idx, e.u, e.v);
    SFM_HIP(hipGetLastError());
    // state: old cameras / points stay on the device, the new ones are uploaded behind them
    SFM_HIP(hipMemcpyAsync(e.cams, d.cams, sizeof(double) * 7 * d.V, hipMemcpyDeviceToDevice, s));
    SFM_TRY(ba_upload(p, e.cams + 7 * (size_t)d.V, cams_new, sizeof(double) * 7 * n_new_cams));
    double* dst[3] = {e.px, e.py, e.pz};
    const double* old[3] = {d.px, d.py, d.pz};
    for (int k = 0; k < 3; ++k) {
      if (d.N > 0) SFM_HIP(hipMemcpyAsync(dst[k], old[k], sizeof(double) * d.N, hipMemcpyDeviceToDevice, s));
      SFM_TRY(ba_upload(p, dst[k] + d.N, pts_new + (size_t)k * n_new_pts, sizeof(double) * n_new_pts));
    }
    SFM_TRY(ba_enqueue_structure(q));
    return ba_finish_structure(q, "sfm_ba_append");
  };
  const int st = run();
  if (st != SFM_OK) { (void)hipStreamSynchronize(s); sfm_ba_destroy(q); return st; }
  // the handle keeps its identity, options, stream and counters; the old buffers leave with q
  q->schur_mode = p->schur_mode; q->debug = p->debug; q->timing = p->timing; q->quirks = p->quirks;
  q->dev.debug = p->debug;
  q->deterministic = p->deterministic && q->schur_mfma_ok && sizeof(double) * (size_t)q->dev.V * 35 <= 64 * 1024;
  const bool had_external_red = p->dev.red != p->own_red;
  ba_graph_drop(p);
  std::swap(p->dev, q->dev);
  std::swap(p->own_red, q->own_red);
  std::swap(p->schur_ws, q->schur_ws);
  std::swap(p->flow_tasks_red, q->flow_tasks_red); std::swap(p->flow_ntasks_red, q->flow_ntasks_red); std::swap(p->flow_camsum, q->flow_camsum);
  p->reduce_deferred = false; p->last_reduce_deferred = false;
  std::swap(p->schur_blk_ptr, q->schur_blk_ptr);
  std::swap(p->schur_mfma_ok, q->schur_mfma_ok);
  std::swap(p->rows_built, q->rows_built); std::swap(p->rows_ok, q->rows_ok);
  std::swap(p->cam_ptr, q->cam_ptr); std::swap(p->cam_ent, q->cam_ent); std::swap(p->cam_pairs, q->cam_pairs);
  std::swap(p->rows_table, q->rows_table); std::swap(p->rows_first, q->rows_first); std::swap(p->rows_ws, q->rows_ws);
  std::swap(p->rows_R, q->rows_R); std::swap(p->rows_tpr, q->rows_tpr); std::swap(p->rows_wgs, q->rows_wgs);
  std::swap(p->rows_groups, q->rows_groups); std::swap(p->rows_tpl, q->rows_tpl); std::swap(p->rows_cp, q->rows_cp);
  std::swap(p->max_track, q->max_track);
  // deterministic mode holds for the grown scene only while the dense product fits and the camera accumulators stay in
  // LDS (V <= 234): beyond that the handle falls back to the default path instead of mixing the two reduce kernels
  p->deterministic = q->deterministic;
  // an externally bound reduced buffer has the wrong size when cameras were added: the library's own buffer takes
  // over and the caller binds a new one (sfm_ba_reduced_buffer reports the new size); with the camera count
  // unchanged the binding survives
  if (had_external_red && n_new_cams == 0) p->dev.red = q->dev.red;
  else p->dev.red = p->own_red;
  q->dev.red = q->own_red;
  p->cur = 0; p->prep_valid = false; p->red_clean = false; p->lin_rows = 0;
  std::swap(p->dev.stamps, q->dev.stamps);      // the diagnostic stamp buffer stays with the handle
  return sfm_ba_destroy(q);
}

int sfm_ba_points_ptr(sfm_ba_problem* p, void** d_px, void** d_py, void** d_pz, int* n_pts) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));           // the deferred back substitution still has to move the points
  if (d_px) *d_px = p->dev.px;
  if (d_py) *d_py = p->dev.py;
  if (d_pz) *d_pz = p->dev.pz;
  if (n_pts) *n_pts = p->dev.N;
  return SFM_OK;
}

int sfm_ba_stream(sfm_ba_problem* p, void** hip_stream) {
  SFM_TRY(check_problem(p));
  if (hip_stream) *hip_stream = p->stream;
  return SFM_OK;
}

int sfm_ba_reduced_buffer(sfm_ba_problem* p, void** device_ptr, int64_t* n_doubles, int* ld) {
  SFM_TRY(check_problem(p));
  if (device_ptr) *device_ptr = p->dev.red;
  if (n_doubles) *n_doubles = (int64_t)red_size(p->dev.nbk);
  if (ld) *ld = p->dev.nbk * kNB;
  return SFM_OK;
}

int sfm_ba_bind_reduced_buffer(sfm_ba_problem* p, void* device_ptr, int64_t n_doubles) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));           // the deferred kernel clears the buffer that is bound now
  const int64_t need = (int64_t)red_size(p->dev.nbk);
  ba_graph_drop(p);
  p->red_clean = false;
  if (device_ptr == nullptr) { p->dev.red = p->own_red; return SFM_OK; }
  if (n_doubles < need) { set_error("reduced buffer too small: %lld < %lld doubles", (long long)n_doubles, (long long)need); return SFM_E_SHAPE; }
  p->dev.red = static_cast<double*>(device_ptr);
  return SFM_OK;
}

int sfm_ba_set_comm(sfm_ba_problem* p, sfm_comm* comm) {
  SFM_TRY(check_problem(p));
  SFM_TRY(ba_flush(p));
  ba_graph_drop(p);          // a captured iteration body has no collective in it
  SFM_TRY(comm_attach(comm, +1));
  if (p->comm) { SFM_HIP(hipStreamSynchronize(p->stream)); (void)comm_attach(p->comm, -1); }      // no collective of the old one in flight
  p->comm = comm;
  return SFM_OK;
}

int sfm_ba_kernel_time(sfm_ba_problem* p, int kernel_id, double* total_ms, int* launches) {
  SFM_TRY(check_problem(p));
  if (kernel_id < 0 || kernel_id >= SFM_K_COUNT) { set_error("bad kernel id %d", kernel_id); return SFM_E_SHAPE; }
  SFM_HIP(hipStreamSynchronize(p->stream));
  KernelTimer& t = p->timers[kernel_id];
  double tot = 0;
  for (int i = 0; i < t.used; ++i) {
    float ms = 0;
    SFM_HIP(hipEventElapsedTime(&ms, t.ev[i].first, t.ev[i].second));
    tot += ms;
  }
  if (total_ms) *total_ms = tot;
  if (launches) *launches = t.used;
  return SFM_OK;
}

namespace sfm { __global__ void ba_noop_kernel() {} }

int sfm_ba_event_overhead(sfm_ba_problem* p, int n, double* avg_ms) {
  SFM_TRY(check_problem(p));
  if (n < 1 || avg_ms == nullptr) { set_error("sfm_ba_event_overhead: bad arguments"); return SFM_E_SHAPE; }
  hipStream_t s = p->stream;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> ev((size_t)n);
  for (auto& e : ev) { SFM_HIP(hipEventCreate(&e.first)); SFM_HIP(hipEventCreate(&e.second)); }
  // the same pattern ba_tick brackets a kernel class with, around a kernel that does nothing, between other work
  for (auto& e : ev) {
    ba_noop_kernel<<<1, 64, 0, s>>>();
    SFM_HIP(hipEventRecord(e.first, s));
    ba_noop_kernel<<<1, 64, 0, s>>>();
    SFM_HIP(hipEventRecord(e.second, s));
    ba_noop_kernel<<<1, 64, 0, s>>>();
  }
  SFM_HIP(hipStreamSynchronize(s));
  double tot = 0;
  for (auto& e : ev) {
    float ms = 0;
    SFM_HIP(hipEventElapsedTime(&ms, e.first, e.second));
    tot += ms;
    (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second);
  }
  *avg_ms = tot / n;
  return SFM_OK;
}

int sfm_ba_debug_stamps(sfm_ba_problem* p, unsigned long long* out, int n) {
  SFM_TRY(check_problem(p));
  if (p->dev.stamps == nullptr || n < 0 || n > 1024) { set_error("debug stamps not enabled (SFM_OPT_DEBUG bit 8)"); return SFM_E_SHAPE; }
  SFM_HIP(hipStreamSynchronize(p->stream));
  SFM_HIP(hipMemcpy(out, p->dev.stamps, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost));
  return SFM_OK;
}

int sfm_ba_reset_timing(sfm_ba_problem* p) {
  SFM_TRY(check_problem(p));
  SFM_HIP(hipStreamSynchronize(p->stream));
  for (auto& t : p->timers) { t.used = 0; t.calls = 0; t.open = false; }
  return SFM_OK;
}

int sfm_ba_solve(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx, const double* uv_norm, double* cams,
                 double* pts, double lambda, int iters, int quirks) {
  sfm_ba_problem* p = nullptr;
  SFM_TRY(sfm_ba_create(V, N, M, pt_ptr, cam_idx, uv_norm, &p));
  int st = sfm_ba_set_state(p, cams, pts);
  if (st == SFM_OK) st = sfm_ba_iterate(p, lambda, iters, quirks);
  if (st == SFM_OK) st = sfm_ba_get_state(p, cams, pts);
  sfm_ba_destroy(p);
  return st;
}

int sfm_ba_residual_jacobian(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx, const double* uv_norm,
                             const double* cams, const double* pts, int quirks, double* r, double* Jp, double* Jx) {
  sfm_ba_problem* p = nullptr;
  SFM_TRY(sfm_ba_create(V, N, M, pt_ptr, cam_idx, uv_norm, &p));
  auto run = [&]() -> int {
    SFM_TRY(sfm_ba_set_state(p, cams, pts));
    SFM_TRY(ba_enqueue_prep(p));
    if (M == 0) return SFM_OK;
    hipStream_t s = p->stream;
    DevBuf<double> dr, djp, djx;
    SFM_TRY(dr.alloc(2 * (size_t)M, s)); SFM_TRY(djp.alloc(14 * (size_t)M, s)); SFM_TRY(djx.alloc(6 * (size_t)M, s));
    ba_enqueue_residual_jacobian(p, quirks, dr.p, djp.p, djx.p);
    SFM_HIP(hipGetLastError());
    SFM_TRY(dr.download(r, 2 * (size_t)M, s)); SFM_TRY(djp.download(Jp, 14 * (size_t)M, s)); SFM_TRY(djx.download(Jx, 6 * (size_t)M, s));
    int st[2] = {0, 0};
    SFM_HIP(hipMemcpyAsync(st, p->dev.status, sizeof(st), hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    if (st[0] != SFM_OK) { set_error("bundle adjustment: %s for camera %d", status_name(st[0]), st[1]); return st[0]; }
    return SFM_OK;
  };
  const int st = run();
  sfm_ba_destroy(p);
  return st;
}

int sfm_ba_reduced_system(int V, int N, int64_t M, const int* pt_ptr, const int* cam_idx, const double* uv_norm,
                          const double* cams, const double* pts, double lambda, int quirks, int schur_mode, double* S,
                          double* rhs) {
  sfm_ba_problem* p = nullptr;
  SFM_TRY(sfm_ba_create(V, N, M, pt_ptr, cam_idx, uv_norm, &p));
  auto run = [&]() -> int {
    SFM_TRY(sfm_ba_set_option(p, SFM_OPT_SCHUR, schur_mode));
    SFM_TRY(sfm_ba_set_state(p, cams, pts));
    SFM_TRY(ba_enqueue_linearize_reduce(p, lambda, quirks));
    hipStream_t s = p->stream;
    const BaDev& d = p->dev;
    DevBuf<double> dS, drhs;
    SFM_TRY(dS.alloc((size_t)d.P * d.P, s)); SFM_TRY(drhs.alloc((size_t)d.P, s));
    ba_enqueue_symmetrize(p, lambda, dS.p, drhs.p);
    SFM_HIP(hipGetLastError());
    SFM_TRY(dS.download(S, (size_t)d.P * d.P, s));
    SFM_TRY(drhs.download(rhs, (size_t)d.P, s));
    int st[2] = {0, 0};
    SFM_HIP(hipMemcpyAsync(st, d.status, sizeof(st), hipMemcpyDeviceToHost, s));
    SFM_TRY(stream_sync(s));
    if (st[0] != SFM_OK) { set_error("bundle adjustment: %s for camera %d", status_name(st[0]), st[1]); return st[0]; }
    return SFM_OK;
  };
  const int st = run();
  sfm_ba_destroy(p);
  return st;
}

}  // extern "C"
