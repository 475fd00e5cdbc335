"""Point-range sharding of bundle adjustment across ranks (one process per GPU).

Given the cameras, points are independent: each rank owns a contiguous range of points (with all
their observations), forms its partial reduced camera system ``[S | rhs]`` locally, and ONE
all-reduce (SUM, float64) per iteration makes the full system available everywhere; every rank
then performs the identical reduced solve and back-substitutes its own points (SURVEY.md
section 8(e)).  The reference has no distributed code; this is new, specified by the algebra of
ba_processor.py:376-406: ``A``, ``B D^-1 B^T``, ``ep`` and ``B D^-1 ex`` are sums over
observations / points, so partial sums over disjoint point ranges add up to the full matrices.

The collective is issued by the caller-supplied ``all_reduce`` (``torch.distributed.all_reduce``
on RCCL for GPU ranks, gloo in the CPU tests); the engine that produces the partial system and
consumes the reduced one is pluggable so the sharding logic is testable without a GPU.

The two other nonlinear solvers of the path shard with NO exchange step at all:
``nonlinear_triangulate`` is independent per point (triangulation_processor.py:209-228) -> ``ShardedTriangulation``
deals contiguous point ranges to the ranks; ``nonlinear_estimate_cam_pose_pnp`` is independent per view
(campose_processor.py:378-459; one view's 7x7 reduction stays on one GPU) -> ``ShardedPnp`` deals contiguous view
ranges balanced by their point counts.  The only communication is the final gather of the results.
"""
import numpy as np

from . import native


# ---- layout of the reduced buffer that is all-reduced (mirror of csrc/sfm_ba.h) ------------------------------
# S is symmetric and only its lower triangle is formed, so the buffer holds the lower-triangular 32x32 BLOCKS only
# (block (br, bc), bc <= br, at ((br (br + 1)) / 2 + bc) * 1024, row-major inside the block), then rhs padded to
# 32 nbk: 8.1 MB at 200 cameras instead of the 15.9 MB of the full square.
RED_NB = 32


def reduced_blocks(n_cams):
    return (7 * n_cams + RED_NB - 1) // RED_NB


def reduced_size(n_cams):
    nbk = reduced_blocks(n_cams)
    return nbk * (nbk + 1) // 2 * RED_NB * RED_NB + nbk * RED_NB


def reduced_index(row, col):
    """Offset of S[row, col] (row >= col) in the packed buffer; vectorised."""
    row = np.asarray(row, dtype=np.int64); col = np.asarray(col, dtype=np.int64)
    br, bc, i, k = row // RED_NB, col // RED_NB, row % RED_NB, col % RED_NB
    return (br * (br + 1) // 2 + bc) * (RED_NB * RED_NB) + i * RED_NB + k


def pack_reduced(s_mat, rhs):
    """Dense symmetric (or lower-triangular) S (P, P) and rhs (P,) -> the packed buffer (numpy float64)."""
    p = rhs.shape[0]
    n_cams = p // 7
    nbk = reduced_blocks(n_cams)
    buf = np.zeros(reduced_size(n_cams))
    ii, jj = np.tril_indices(p)
    buf[reduced_index(ii, jj)] = np.asarray(s_mat)[ii, jj]
    buf[nbk * (nbk + 1) // 2 * RED_NB * RED_NB:][:p] = rhs
    return buf


def unpack_reduced(buf, n_cams):
    """The packed buffer -> (S (P, P) symmetric, rhs (P,))."""
    buf = np.asarray(buf)
    p = 7 * n_cams
    nbk = reduced_blocks(n_cams)
    s_mat = np.zeros((p, p))
    ii, jj = np.tril_indices(p)
    s_mat[ii, jj] = buf[reduced_index(ii, jj)]
    s_mat = s_mat + np.tril(s_mat, -1).T
    return s_mat, np.array(buf[nbk * (nbk + 1) // 2 * RED_NB * RED_NB:][:p])


def shard_bounds(pt_ptr, world_size, weight="pairs"):
    """Contiguous point ranges balanced by Schur-product cost: weight k(k+1)/2 per point for
    ``"pairs"`` (camera pairs), k for ``"obs"``.  Returns ``bounds`` (world_size + 1,)."""
    pt_ptr = np.asarray(pt_ptr, dtype=np.int64)
    n_pts = pt_ptr.shape[0] - 1
    k = np.diff(pt_ptr).astype(np.float64)
    w = k * (k + 1) / 2 if weight == "pairs" else k
    w = w + 1e-3                                  # empty tracks still cost a little; keeps ranges non-degenerate
    cum = np.concatenate(([0.0], np.cumsum(w)))
    targets = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, targets, side="left")
    bounds = np.concatenate(([0], inner, [n_pts])).astype(np.int64)
    return np.maximum.accumulate(np.minimum(bounds, n_pts))


def local_shard(pt_ptr, cam_idx, uv_norm, pts, bounds, rank):
    """Slice the observation CSR, keys and points of one rank.  Returns
    (pt_ptr_local, cam_idx_local, uv_local, pts_local, (p0, p1))."""
    p0, p1 = int(bounds[rank]), int(bounds[rank + 1])
    o0, o1 = int(pt_ptr[p0]), int(pt_ptr[p1])
    ptr = (np.asarray(pt_ptr[p0:p1 + 1], dtype=np.int64) - o0).astype(np.int32)
    return (ptr, np.ascontiguousarray(cam_idx[o0:o1], dtype=np.int32),
            np.ascontiguousarray(np.asarray(uv_norm)[:, o0:o1], dtype=np.float64),
            np.ascontiguousarray(np.asarray(pts)[:, p0:p1], dtype=np.float64), (p0, p1))


class HipShardEngine:
    """One rank's device-resident shard.  The reduced buffer is a torch tensor so that
    ``torch.distributed.all_reduce`` (RCCL) can run on it in place; kernels and the collective
    share torch's current stream."""

    def __init__(self, n_cams, pt_ptr_local, cam_idx_local, uv_local, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        native.init(self.device.index or 0)
        # a dedicated (non-null) torch stream OF THIS ENGINE: its problem launches on it (sfm_ba_set_stream: the
        # stream belongs to the problem, not to the library, so several engines in one process do not interfere)
        # and the collective is issued under it, so RCCL's internal stream waits for the kernels and vice versa
        self.stream = torch.cuda.Stream(self.device)
        self.prob = native.BaProblem(n_cams, pt_ptr_local, cam_idx_local, uv_local)
        self.prob.set_stream(self.stream.cuda_stream)
        self._bind_reduced()

    def _bind_reduced(self):
        """(Re)allocate the all-reduce tensor for the problem's current camera count and bind it.  The tensor is
        created under the engine's stream, so its zero-fill is ordered before the first kernel that uses it."""
        torch = self.torch
        _ptr, n, self.ld = self.prob.reduced_buffer()
        with torch.cuda.stream(self.stream):
            self.reduced = torch.zeros(n, dtype=torch.float64, device=self.device)
        self.reduced.record_stream(self.stream)
        self.prob.bind_reduced_buffer(self.reduced.data_ptr(), n)

    def append(self, cams_new, pts_new, obs_cam, obs_pt, uv_norm):
        """Grow this rank's shard (new cameras go to EVERY rank, new points / observations to their owner).  The
        reduced buffer follows the camera count: a new tensor is bound before the next linearisation, so the
        all-reduce never runs on a stale one."""
        n_before = self.prob.n_cams
        self.prob.append(cams_new, pts_new, obs_cam, obs_pt, uv_norm)
        if self.prob.n_cams != n_before:
            self._bind_reduced()

    def stream_context(self):
        return self.torch.cuda.stream(self.stream)

    def set_state(self, cams, pts_local):
        self.prob.set_state(cams, pts_local)

    def linearize_reduce(self, lam, quirks=native.QUIRKS_REFERENCE):
        self.prob.linearize_reduce(lam, quirks)
        return self.reduced

    def solve_update(self, lam, quirks=native.QUIRKS_REFERENCE):
        self.prob.solve_update(lam, quirks)

    def flush(self):
        self.prob.flush()

    def iterate_local(self, lam, iters, quirks=native.QUIRKS_REFERENCE):
        """``iters`` complete iterations in ONE C-ABI call (``sfm_ba_iterate``): with no exchange step when one rank owns
        every point (the path the drop-in classes take), or -- after ``attach_comm`` -- with the library's own RCCL
        all-reduce of [S | rhs] between the partial reduce and the replicated solve of every iteration."""
        self.prob.iterate(lam, iters, quirks)

    def attach_comm(self, comm):
        """Hand the exchange step to the library (``native.Comm``; None detaches): no Python between the iterations."""
        self.prob.set_comm(comm)
        self.comm = comm

    def get_state(self):
        return self.prob.get_state()

    def close(self):
        if getattr(self, "comm", None) is not None:
            self.prob.set_comm(None)
            self.comm = None
        self.prob.bind_reduced_buffer(0, 0)
        self.prob.close()


class ShardedBa:
    """Damped Gauss-Newton BA over ``world_size`` ranks: engine.linearize_reduce -> all_reduce ->
    engine.solve_update, ``iters`` times.  With world_size == 1 no collective is issued."""

    def __init__(self, engine, all_reduce=None, world_size=1):
        self.engine = engine
        self.all_reduce = all_reduce
        self.world_size = world_size

    def iterate(self, lam, iters, quirks=native.QUIRKS_REFERENCE):
        with self.engine.stream_context():
            local = getattr(self.engine, "iterate_local", None)
            if self.all_reduce is None and local is not None:      # nothing to exchange: the whole loop is one library call
                local(lam, iters, quirks)
                return
            for _ in range(iters):
                buf = self.engine.linearize_reduce(lam, quirks)
                if self.all_reduce is not None:
                    self.all_reduce(buf)
                self.engine.solve_update(lam, quirks)
            flush = getattr(self.engine, "flush", None)       # the device engine may have deferred the last
            if flush is not None:                              # back substitution to a linearisation that never comes
                flush()


# ================================================================================================================
# Nonlinear triangulation by point, nonlinear PnP by view: independent units, no data-path collective.
# ================================================================================================================
def shard_points(n_pts, world_size):
    """Contiguous near-equal point ranges: ``bounds`` (world_size + 1,); every point costs the same
    (triangulation_processor.py:209-228: one fixed-count loop per point)."""
    return (np.arange(world_size + 1, dtype=np.int64) * int(n_pts)) // world_size


def shard_views(offsets, world_size):
    """Contiguous view ranges balanced by work: a view of n points costs ~ceil(n / 256) + 1 per iteration on its one
    workgroup (one linearisation round per 256 points + the 7x7 reduction and solve).  ``offsets`` is the
    (n_views + 1,) column CSR of sfm_pnp_nonlinear_batch; returns view ``bounds`` (world_size + 1,)."""
    offsets = np.asarray(offsets, dtype=np.int64)
    n_views = offsets.shape[0] - 1
    w = (np.diff(offsets) + 255) // 256 + 1.0
    cum = np.concatenate(([0.0], np.cumsum(w)))
    targets = cum[-1] * np.arange(1, world_size) / world_size
    inner = np.searchsorted(cum, targets, side="left")
    bounds = np.concatenate(([0], inner, [n_views])).astype(np.int64)
    return np.maximum.accumulate(np.minimum(bounds, n_views))


def gather_columns(local, widths, group=None, device=None):
    """All-gather of per-rank column blocks ``local`` (rows, widths[rank]) float64 -> list of the ``world`` blocks, over
    torch.distributed (``device`` = None: gloo / host tensors; a cuda device: RCCL).  Ragged and empty blocks are
    padded to the widest one for the collective and trimmed afterwards."""
    import torch
    import torch.distributed as dist
    widths = [int(w) for w in widths]
    rows, wmax = local.shape[0], max(max(widths), 1)
    pad = np.zeros((rows, wmax))
    pad[:, :local.shape[1]] = local
    mine = torch.from_numpy(pad)
    if device is not None:
        mine = mine.to(device)
    parts = [torch.empty_like(mine) for _ in widths]
    dist.all_gather(parts, mine, group=group)
    return [np.ascontiguousarray(t.cpu().numpy()[:, :w]) for t, w in zip(parts, widths)]


class ShardedTriangulation:
    """``TriangulationProcessor.nonlinear_triangulate`` (triangulation_processor.py:160-234) over ``world_size``
    ranks: rank r refines the points ``shard_points(m, world)[r : r + 2]`` of the batch through ``refine`` (default: the
    HIP kernel behind ``native.tri_nonlinear``) and ``gather`` (``gather_columns`` bound to the process group) joins the
    slices, so every rank returns the full ``(4, m)`` array.  With ``world_size == 1`` nothing is gathered."""

    def __init__(self, rank=0, world_size=1, gather=None, refine=None):
        self.rank, self.world_size = int(rank), int(world_size)
        self.gather = gather
        self.refine = refine or native.tri_nonlinear
        if self.world_size > 1 and gather is None:
            raise ValueError("ShardedTriangulation: world_size > 1 needs a gather callable")

    def local_range(self, n_pts):
        b = shard_points(n_pts, self.world_size)
        return int(b[self.rank]), int(b[self.rank + 1])

    def nonlinear_triangulate(self, projs, uv, x_in, lam, iters):
        """projs (V,3,4); uv (V,2,m); x_in (4,m) -> (4,m)."""
        projs = np.asarray(projs, dtype=np.float64); uv = np.asarray(uv, dtype=np.float64); x_in = np.asarray(x_in, dtype=np.float64)
        m = x_in.shape[1]
        p0, p1 = self.local_range(m)
        if p1 > p0:
            mine = self.refine(projs, np.ascontiguousarray(uv[:, :, p0:p1]), np.ascontiguousarray(x_in[:, p0:p1]), lam, iters)
        else:
            mine = np.empty((4, 0))
        if self.world_size == 1:
            return mine
        return np.hstack(self.gather(mine, np.diff(shard_points(m, self.world_size))))


class ShardedPnp:
    """``CamposeProcessor.nonlinear_estimate_cam_pose_pnp`` (campose_processor.py:308-459) for a batch of independent
    views over ``world_size`` ranks: rank r refines the views ``shard_views(offsets, world)[r : r + 2]`` in one launch
    (``native.pnp_nonlinear_batch``: one workgroup per view) and the poses are gathered.  Returns
    ``(rot (n_views,3,3), loc (n_views,3), status (n_views,))`` on every rank."""

    def __init__(self, rank=0, world_size=1, gather=None, refine=None):
        self.rank, self.world_size = int(rank), int(world_size)
        self.gather = gather
        self.refine = refine or native.pnp_nonlinear_batch
        if self.world_size > 1 and gather is None:
            raise ValueError("ShardedPnp: world_size > 1 needs a gather callable")

    def local_range(self, offsets):
        b = shard_views(offsets, self.world_size)
        return int(b[self.rank]), int(b[self.rank + 1])

    def nonlinear_estimate(self, offsets, uv_pix, pts_h, intrinsics, rot0, loc0, lam, iters, quirks=native.QUIRKS_REFERENCE):
        offsets = np.asarray(offsets, dtype=np.int64)
        n_views = offsets.shape[0] - 1
        intrinsics = np.asarray(intrinsics, dtype=np.float64).reshape(n_views, 3, 3)
        rot0 = np.asarray(rot0, dtype=np.float64).reshape(n_views, 3, 3)
        loc0 = np.asarray(loc0, dtype=np.float64).reshape(n_views, 3)
        v0, v1 = self.local_range(offsets)
        packed = np.empty((13, v1 - v0))               # rows: R (9) | C (3) | status (1)
        if v1 > v0:
            c0, c1 = int(offsets[v0]), int(offsets[v1])
            rot, loc, st = self.refine((offsets[v0:v1 + 1] - c0).astype(np.int32), np.ascontiguousarray(np.asarray(uv_pix)[:, c0:c1]),
                                       np.ascontiguousarray(np.asarray(pts_h)[:, c0:c1]), intrinsics[v0:v1], rot0[v0:v1], loc0[v0:v1],
                                       lam, iters, quirks)
            packed[0:9] = np.asarray(rot).reshape(-1, 9).T
            packed[9:12] = np.asarray(loc).reshape(-1, 3).T
            packed[12] = st
        if self.world_size > 1:
            packed = np.hstack(self.gather(packed, np.diff(shard_views(offsets, self.world_size))))
        return (np.ascontiguousarray(packed[0:9].T).reshape(n_views, 3, 3), np.ascontiguousarray(packed[9:12].T),
                packed[12].astype(np.int32))


class HipTriShard:
    """One rank's device-resident slice of a triangulation batch (what ``bench.py --config TRI`` times): projections,
    keys and points live in HBM as torch tensors, ``run`` only enqueues ``sfm_tri_nonlinear_dev`` on the shard's
    stream."""

    def __init__(self, projs, uv_local, x_local, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        native.init(self.device.index or 0)
        self.stream = torch.cuda.Stream(self.device)
        self.n_views, self.m = int(projs.shape[0]), int(x_local.shape[1])
        with torch.cuda.stream(self.stream):
            self.projs = torch.from_numpy(np.ascontiguousarray(projs, dtype=np.float64)).to(self.device)
            self.uv = torch.from_numpy(np.ascontiguousarray(uv_local, dtype=np.float64)).to(self.device)
            self.x_in = torch.from_numpy(np.ascontiguousarray(x_local, dtype=np.float64)).to(self.device)
            self.x_out = torch.empty_like(self.x_in)
        self.stream.synchronize()

    def run(self, lam, iters):
        native.tri_nonlinear_dev(self.m, self.n_views, self.projs.data_ptr(), self.uv.data_ptr(), self.x_in.data_ptr(), lam, iters,
                                 self.x_out.data_ptr(), self.stream.cuda_stream)

    def result(self):
        self.stream.synchronize()
        return self.x_out.cpu().numpy()


class HipPnpShard:
    """One rank's device-resident slice of a PnP batch (``bench.py --config PNP``)."""

    def __init__(self, offsets_local, uv_local, pts_local, intrinsics, rot0, loc0, device):
        import torch
        self.torch = torch
        self.device = torch.device(device)
        native.init(self.device.index or 0)
        self.stream = torch.cuda.Stream(self.device)
        self.n_views, self.total = int(np.asarray(offsets_local).shape[0]) - 1, int(np.asarray(uv_local).shape[1])
        self.widest = int(np.max(np.diff(np.asarray(offsets_local)))) if self.n_views > 0 else 0
        up = lambda a, dt: torch.from_numpy(np.ascontiguousarray(a, dtype=dt)).to(self.device)      # noqa: E731
        with torch.cuda.stream(self.stream):
            self.offsets = up(offsets_local, np.int32)
            self.uv, self.x = up(uv_local, np.float64), up(pts_local, np.float64)
            self.k = up(np.asarray(intrinsics).reshape(-1, 9), np.float64)
            self.r0, self.c0 = up(np.asarray(rot0).reshape(-1, 9), np.float64), up(np.asarray(loc0).reshape(-1, 3), np.float64)
            self.r_out, self.c_out = torch.empty_like(self.r0), torch.empty_like(self.c0)
            self.status = torch.zeros(max(1, self.n_views), dtype=torch.int32, device=self.device)
        self.stream.synchronize()

    def run(self, lam, iters, quirks=native.QUIRKS_REFERENCE):
        native.pnp_nonlinear_batch_dev(self.n_views, self.offsets.data_ptr(), self.total, self.uv.data_ptr(), self.x.data_ptr(),
                                       self.k.data_ptr(), self.r0.data_ptr(), self.c0.data_ptr(), lam, iters, quirks,
                                       self.r_out.data_ptr(), self.c_out.data_ptr(), self.status.data_ptr(), self.stream.cuda_stream,
                                       self.widest)

    def result(self):
        self.stream.synchronize()
        return (self.r_out.cpu().numpy().reshape(-1, 3, 3), self.c_out.cpu().numpy(), self.status.cpu().numpy()[:self.n_views])
