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

    def get_state(self):
        return self.prob.get_state()

    def close(self):
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
            for _ in range(iters):
                buf = self.engine.linearize_reduce(lam, quirks)
                if self.all_reduce is not None:
                    self.all_reduce(buf)
                self.engine.solve_update(lam, quirks)
            flush = getattr(self.engine, "flush", None)       # the device engine may have deferred the last
            if flush is not None:                              # back substitution to a linearisation that never comes
                flush()
