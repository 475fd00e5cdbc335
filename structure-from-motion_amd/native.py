"""ctypes binding of the C-ABI in include/sfm_hip.h (libsfm_hip.so, gfx950).

There is no CPU fallback: ``load()`` raises if the library is missing, and every compute call
raises if no MI355X is visible.  Status codes map back to the exceptions the reference raises
(``ValueError`` for bad shapes / invalid rotations, utils.py:43-51, 93-95).
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SFM_HIP_LIBRARY selects another build of the same library (tools/asan_host_check.sh points it at the
# -fsanitize=address,undefined host build); the default is the in-tree gfx950 build next to this file.
LIB_PATH = os.environ.get("SFM_HIP_LIBRARY") or os.path.join(_HERE, "libsfm_hip.so")

OK = 0
E_SHAPE, E_BAD_ROTATION, E_QW_ZERO, E_SQRT_DOMAIN, E_HIP, E_NO_DEVICE, E_HANDLE, E_RANK, E_RCCL = -1, -2, -3, -4, -5, -6, -7, -8, -9
Q1_PNP_ROW_OVERLAP, Q2_LOC_JAC_SIGN, QUIRKS_REFERENCE = 1, 2, 3
SCHUR_AUTO, SCHUR_PAIRS, SCHUR_MFMA, SCHUR_ROWS = 0, 1, 2, 3
OPT_SCHUR, OPT_TIMING, OPT_DEBUG, OPT_DETERMINISTIC, OPT_GRAPH, OPT_TIMING_STRIDE = 1, 2, 3, 4, 5, 6
K_PREP, K_LINEARIZE, K_SCHUR, K_SOLVE, K_BACKSUB, K_REDUCE, K_COUNT = 0, 1, 2, 3, 4, 5, 6
KERNEL_NAMES = ("prep", "linearize", "schur", "solve", "backsub", "reduce")
INFO_SCHUR_KERNEL, INFO_UPLOAD_BYTES, INFO_N_CAMS, INFO_N_PTS, INFO_N_OBS, INFO_MAX_TRACK, INFO_GRAPH_REPLAYS = 1, 2, 3, 4, 5, 6, 7
INFO_REDUCE_IN_SOLVE = 8

# every symbol include/sfm_hip.h declares (checked by tests/test_abi.py)
EXPORTS = (
    "sfm_version", "sfm_init", "sfm_shutdown", "sfm_set_stream", "sfm_synchronize", "sfm_last_error",
    "sfm_quat_to_rot", "sfm_rot_to_quat", "sfm_jac_cam", "sfm_jac_pt",
    "sfm_tri_nonlinear", "sfm_tri_linear", "sfm_triangulate", "sfm_pnp_nonlinear", "sfm_pnp_nonlinear_batch",
    "sfm_pnp_linear_ransac", "sfm_pnp_six_point_hypotheses", "sfm_pnp_ransac_evaluate", "sfm_pnp_inlier_mask",
    "sfm_pnp_ransac_begin", "sfm_pnp_ransac_finish", "sfm_pnp_session_destroy",
    "sfm_comm_available", "sfm_comm_unique_id", "sfm_comm_create", "sfm_comm_destroy", "sfm_ba_set_comm",
    "sfm_fundamental_ransac", "sfm_fundamental_eight_point", "sfm_essential_from_fundamental", "sfm_pose_candidates",
    "sfm_cheirality",
    "sfm_ba_solve", "sfm_ba_create", "sfm_ba_destroy", "sfm_ba_set_option", "sfm_ba_set_state",
    "sfm_ba_set_stream", "sfm_ba_info", "sfm_ba_set_cameras", "sfm_ba_set_points", "sfm_ba_get_stats", "sfm_ba_flush",
    "sfm_pool_redzone_active", "sfm_ba_iterate", "sfm_ba_get_state", "sfm_ba_append", "sfm_ba_kernel_time", "sfm_ba_reset_timing", "sfm_ba_debug_stamps",
    "sfm_ba_linearize_reduce", "sfm_ba_solve_update", "sfm_ba_reduced_buffer",
    "sfm_ba_bind_reduced_buffer", "sfm_ba_residual_jacobian", "sfm_ba_reduced_system",
    "sfm_pool_mode", "sfm_tri_nonlinear_dev", "sfm_tri_linear_dev", "sfm_triangulate_dev", "sfm_pnp_nonlinear_batch_dev",
    "sfm_gather_points_dev", "sfm_ba_points_ptr", "sfm_ba_stream", "sfm_ba_event_overhead",
    "sfm_ba_get_state_rot", "sfm_ba_rederive_quaternions", "sfm_ba_flow_tasks", "sfm_ba_flow_tasks_deferred",
)

_lib = None
_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


class SfmHipError(RuntimeError):
    """HIP / device / handle failures (no reference counterpart)."""


def load():
    """Load libsfm_hip.so (built in-tree by ``__graft_entry__.build()`` / csrc/Makefile)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SfmHipError(
            "libsfm_hip.so not found at %s — build it with `make -C %s` (there is no CPU fallback)"
            % (LIB_PATH, os.path.join(_HERE, "csrc")))
    # When PyTorch is in the process it must load its ROCm runtime FIRST: torch bundles its own
    # libamdhip64/libhsa-runtime64, and two HSA runtimes in one process leave the second one without
    # devices ("no ROCm-capable device is detected").  With torch imported first, libsfm_hip.so binds to
    # the already-loaded runtime and both share streams and device memory.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = ctypes.CDLL(LIB_PATH)
    lib.sfm_last_error.restype = ctypes.c_char_p
    lib.sfm_set_stream.argtypes = [ctypes.c_void_p]
    for name in EXPORTS:
        fn = getattr(lib, name)
        if name != "sfm_last_error":
            fn.restype = ctypes.c_int
    lib.sfm_ba_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _ip, _ip, _dp,
                                  ctypes.POINTER(ctypes.c_void_p)]
    for name in ("sfm_ba_destroy",):
        getattr(lib, name).argtypes = [ctypes.c_void_p]
    lib.sfm_ba_set_option.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
    lib.sfm_ba_set_state.argtypes = [ctypes.c_void_p, _dp, _dp]
    lib.sfm_ba_set_stream.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    lib.sfm_ba_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(ctypes.c_int64)]
    lib.sfm_ba_set_cameras.argtypes = [ctypes.c_void_p, _dp]
    lib.sfm_ba_set_points.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, _dp]
    lib.sfm_ba_append.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, ctypes.c_int, _dp, ctypes.c_int64, _ip, _ip, _dp]
    lib.sfm_ba_get_state.argtypes = [ctypes.c_void_p, _dp, _dp]
    lib.sfm_ba_iterate.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int, ctypes.c_int]
    lib.sfm_ba_linearize_reduce.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    lib.sfm_ba_solve_update.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_int]
    lib.sfm_ba_flush.argtypes = [ctypes.c_void_p]
    lib.sfm_ba_kernel_time.argtypes = [ctypes.c_void_p, ctypes.c_int, _dp, _ip]
    lib.sfm_ba_reset_timing.argtypes = [ctypes.c_void_p]
    lib.sfm_ba_reduced_buffer.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_void_p),
                                          ctypes.POINTER(ctypes.c_int64), _ip]
    lib.sfm_ba_bind_reduced_buffer.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64]
    lib.sfm_ba_solve.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _ip, _ip, _dp, _dp, _dp,
                                 ctypes.c_double, ctypes.c_int, ctypes.c_int]
    lib.sfm_ba_residual_jacobian.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _ip, _ip, _dp, _dp, _dp,
                                             ctypes.c_int, _dp, _dp, _dp]
    lib.sfm_ba_reduced_system.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int64, _ip, _ip, _dp, _dp, _dp,
                                          ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp, _dp]
    lib.sfm_tri_nonlinear.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, ctypes.c_double, ctypes.c_int, _dp]
    lib.sfm_tri_linear.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
    lib.sfm_triangulate.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, ctypes.c_double, ctypes.c_int, _dp]
    lib.sfm_pnp_nonlinear.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_double, ctypes.c_int,
                                      ctypes.c_int, _dp, _dp]
    lib.sfm_pnp_nonlinear_batch.argtypes = [ctypes.c_int, _ip, ctypes.c_int, _dp, _dp, _dp, _dp, _dp,
                                            ctypes.c_double, ctypes.c_int, ctypes.c_int, _dp, _dp, _ip]
    lib.sfm_quat_to_rot.argtypes = [ctypes.c_int, _dp, _dp, _ip]
    lib.sfm_rot_to_quat.argtypes = [ctypes.c_int, _dp, _dp, _ip]
    lib.sfm_jac_cam.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _dp, _ip]
    lib.sfm_jac_pt.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp]
    lib.sfm_fundamental_eight_point.argtypes = [ctypes.c_int, _dp, ctypes.c_int, _ip, _dp, _ip]
    lib.sfm_essential_from_fundamental.argtypes = [_dp, _dp, _dp, _dp]
    lib.sfm_pose_candidates.argtypes = [_dp, _dp, _dp]
    lib.sfm_cheirality.argtypes = [ctypes.c_int, ctypes.c_int, _dp, _dp, _dp, _ip, _ip, _ip]
    vp = ctypes.c_void_p
    lib.sfm_pool_mode.argtypes = [ctypes.c_int64, ctypes.POINTER(ctypes.c_int64), ctypes.POINTER(ctypes.c_int64)]
    lib.sfm_tri_nonlinear_dev.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, ctypes.c_double, ctypes.c_int, vp, vp]
    lib.sfm_tri_linear_dev.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, vp, vp]
    lib.sfm_triangulate_dev.argtypes = [ctypes.c_int, ctypes.c_int, vp, vp, ctypes.c_double, ctypes.c_int, vp, vp]
    lib.sfm_pnp_nonlinear_batch_dev.argtypes = [ctypes.c_int, vp, ctypes.c_int, vp, vp, vp, vp, vp, ctypes.c_double, ctypes.c_int,
                                                ctypes.c_int, vp, vp, vp, ctypes.c_int, vp]
    lib.sfm_gather_points_dev.argtypes = [ctypes.c_int, vp, vp, vp, vp, vp, vp]
    lib.sfm_ba_points_ptr.argtypes = [vp, ctypes.POINTER(vp), ctypes.POINTER(vp), ctypes.POINTER(vp), _ip]
    lib.sfm_ba_stream.argtypes = [vp, ctypes.POINTER(vp)]
    lib.sfm_ba_event_overhead.argtypes = [vp, ctypes.c_int, _dp]
    lib.sfm_ba_get_state_rot.argtypes = [vp, _dp, _dp, _dp]
    lib.sfm_ba_rederive_quaternions.argtypes = [vp, ctypes.c_int, ctypes.c_int]
    _lib = lib
    return lib


def last_error():
    return load().sfm_last_error().decode("utf-8", "replace")


def check(status):
    """Map a C-ABI status to the exception the reference would raise."""
    if status == OK:
        return
    msg = last_error()
    if status == E_BAD_ROTATION:
        raise ValueError("convert_quaternion_to_rotation : Invalid output rotation matrix (%s)" % msg)
    if status == E_QW_ZERO:
        raise ValueError("convert_rotation_to_quaternion : Invalid output qw (%s)" % msg)
    if status == E_SQRT_DOMAIN:
        raise ValueError("math domain error (%s)" % msg)
    if status == E_SHAPE:
        raise ValueError(msg)
    if status == E_RANK:
        raise ValueError(msg)
    raise SfmHipError("libsfm_hip status %d: %s" % (status, msg))


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def dptr(a):
    return a.ctypes.data_as(_dp)


def iptr(a):
    return a.ctypes.data_as(_ip)


def init(device=0):
    check(load().sfm_init(int(device)))


def set_stream(stream_ptr):
    check(load().sfm_set_stream(ctypes.c_void_p(stream_ptr) if stream_ptr else None))


def synchronize():
    check(load().sfm_synchronize())


def flow_tasks(nbk):
    """Task table of the data-flow reduced solve for nbk block columns (host only; no device call): (n, 4) int32 rows
    {type, row, column, sort key} in the order the workgroups take them."""
    lib = load()
    lib.sfm_ba_flow_tasks.argtypes = [ctypes.c_int, _ip, ctypes.c_int]
    n = lib.sfm_ba_flow_tasks(int(nbk), None, 0)
    out = np.zeros((max(n, 1), 4), dtype=np.int32)
    if n > 0:
        lib.sfm_ba_flow_tasks(int(nbk), iptr(out), n)
    return out[:n]


def flow_tasks_deferred(n_cams):
    """Task table of the data-flow solve for n_cams cameras when the split-K reduce rides in its launch (host only): rows
    {type, row, column, key}; type 5 = camera sums (camera, part), 6 = rows 8q..8q+7 (q = key & 3) of block (row, column) of S."""
    lib = load()
    lib.sfm_ba_flow_tasks_deferred.argtypes = [ctypes.c_int, _ip, ctypes.c_int]
    n = lib.sfm_ba_flow_tasks_deferred(int(n_cams), None, 0)
    out = np.zeros((max(n, 1), 4), dtype=np.int32)
    if n > 0:
        lib.sfm_ba_flow_tasks_deferred(int(n_cams), iptr(out), n)
    return out[:n]


def pool_redzone_active():
    """True when the process runs with SFM_POOL_REDZONE=1 (device buffers between checked guard zones; tests only)."""
    return bool(load().sfm_pool_redzone_active())


def pool_mode(probe_bytes=1000):
    """(mode bits, mapped bytes behind a probe buffer, guard-mode allocations so far): bit 0 = SFM_POOL_REDZONE,
    bit 1 = SFM_POOL_GUARD (every buffer ends at the end of its own mapping, the next page is unmapped)."""
    slack = ctypes.c_int64(); allocs = ctypes.c_int64()
    mode = load().sfm_pool_mode(int(probe_bytes), ctypes.byref(slack), ctypes.byref(allocs))
    return mode, int(slack.value), int(allocs.value)


# ---- device-pointer, stream-ordered forms (pointers as integers, e.g. torch.Tensor.data_ptr()) ------------------
def _vp(ptr):
    return ctypes.c_void_p(int(ptr)) if ptr else None


def tri_nonlinear_dev(m, n_views, d_projs, d_uv, d_x_in, lam, iters, d_x_out, stream=0):
    check(load().sfm_tri_nonlinear_dev(int(m), int(n_views), _vp(d_projs), _vp(d_uv), _vp(d_x_in), float(lam), int(iters),
                                       _vp(d_x_out), _vp(stream)))


def tri_linear_dev(m, n_views, d_projs, d_uv, d_x_out, stream=0):
    check(load().sfm_tri_linear_dev(int(m), int(n_views), _vp(d_projs), _vp(d_uv), _vp(d_x_out), _vp(stream)))


def triangulate_dev(m, n_views, d_projs, d_uv, lam, iters, d_x_out, stream=0):
    check(load().sfm_triangulate_dev(int(m), int(n_views), _vp(d_projs), _vp(d_uv), float(lam), int(iters), _vp(d_x_out), _vp(stream)))


def pnp_nonlinear_batch_dev(n_views, d_offsets, total, d_uv_pix, d_x, d_k, d_r0, d_c0, lam, iters, quirks, d_r_out, d_c_out,
                            d_status, stream=0, max_view_points=0):
    check(load().sfm_pnp_nonlinear_batch_dev(int(n_views), _vp(d_offsets), int(total), _vp(d_uv_pix), _vp(d_x), _vp(d_k), _vp(d_r0),
                                             _vp(d_c0), float(lam), int(iters), int(quirks), _vp(d_r_out), _vp(d_c_out),
                                             _vp(d_status), int(max_view_points), _vp(stream)))


def gather_points_dev(n, d_index, d_px, d_py, d_pz, d_x_out, stream=0):
    check(load().sfm_gather_points_dev(int(n), _vp(d_index), _vp(d_px), _vp(d_py), _vp(d_pz), _vp(d_x_out), _vp(stream)))


# ------------------------------------------------------------------------------------------------
def quat_to_rot(q):
    q = f64(q).reshape(-1, 4)
    n = q.shape[0]
    rot = np.empty((n, 3, 3)); st = np.empty(n, dtype=np.int32)
    check(load().sfm_quat_to_rot(n, dptr(q), dptr(rot), iptr(st)))
    return rot, st


def rot_to_quat(rot):
    rot = f64(rot).reshape(-1, 3, 3)
    n = rot.shape[0]
    q = np.empty((n, 4)); st = np.empty(n, dtype=np.int32)
    check(load().sfm_rot_to_quat(n, dptr(rot), dptr(q), iptr(st)))
    return q, st


def jac_cam(rot, loc, pts_h, quirks=QUIRKS_REFERENCE):
    rot = f64(rot).reshape(-1, 3, 3); loc = f64(loc).reshape(-1, 3); pts_h = f64(pts_h).reshape(-1, 4)
    n = rot.shape[0]
    jp = np.empty((n, 2, 7)); st = np.empty(n, dtype=np.int32)
    check(load().sfm_jac_cam(n, dptr(rot), dptr(loc), dptr(pts_h), quirks, dptr(jp), iptr(st)))
    return jp, st


def jac_pt(projs, pts_h):
    projs = f64(projs); pts_h = f64(pts_h).reshape(-1, 4)
    n, nv = projs.shape[0], projs.shape[1]
    jx = np.empty((n, 2 * nv, 3))
    check(load().sfm_jac_pt(n, nv, dptr(projs), dptr(pts_h), dptr(jx)))
    return jx


def tri_nonlinear(projs, uv, x_in, lam, iters):
    """projs (V,3,4); uv (V,2,m); x_in (4,m) -> (4,m)."""
    projs = f64(projs); uv = f64(uv); x_in = f64(x_in)
    nv, m = projs.shape[0], x_in.shape[1]
    out = np.empty((4, m))
    check(load().sfm_tri_nonlinear(m, nv, dptr(projs), dptr(uv), dptr(x_in), float(lam), int(iters), dptr(out)))
    return out


def tri_linear(projs, uv):
    """DLT triangulation: projs (V,3,4); uv (V,2,m) -> (4,m) with W = 1."""
    projs = f64(projs); uv = f64(uv)
    nv, m = projs.shape[0], uv.shape[2]
    out = np.empty((4, m))
    check(load().sfm_tri_linear(m, nv, dptr(projs), dptr(uv), dptr(out)))
    return out


def triangulate(projs, uv, lam, iters):
    """Linear then nonlinear triangulation in one call (initial points stay on the device)."""
    projs = f64(projs); uv = f64(uv)
    nv, m = projs.shape[0], uv.shape[2]
    out = np.empty((4, m))
    check(load().sfm_triangulate(m, nv, dptr(projs), dptr(uv), float(lam), int(iters), dptr(out)))
    return out


def pnp_nonlinear(uv_pix, pts_h, intrinsic, rot0, loc0, lam, iters, quirks=QUIRKS_REFERENCE):
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic); rot0 = f64(rot0); loc0 = f64(loc0).reshape(3)
    n = uv_pix.shape[1]
    rot = np.empty((3, 3)); loc = np.empty(3)
    check(load().sfm_pnp_nonlinear(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), dptr(rot0), dptr(loc0),
                                   float(lam), int(iters), int(quirks), dptr(rot), dptr(loc)))
    return rot, loc.reshape(3, 1)


def pnp_linear_ransac(uv_pix, pts_h, intrinsic, samples, threshold, as_array=False):
    """Evaluate six-point DLT hypotheses (samples: (n_hyp, 6) indices drawn by the caller) and return
    (rot (3,3), loc (3,1), inlier index list (or int array with ``as_array``), best hypothesis index or -1)."""
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic)
    samples = i32(samples).reshape(-1, 6)
    n, n_hyp = uv_pix.shape[1], samples.shape[0]
    rot = np.empty((3, 3)); loc = np.empty(3)
    mask = np.empty(n, dtype=np.int32)
    cnt = ctypes.c_int(); best = ctypes.c_int()
    lib = load()
    lib.sfm_pnp_linear_ransac.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_double, _dp, _dp,
                                          _ip, _ip, _ip]
    check(lib.sfm_pnp_linear_ransac(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), n_hyp, iptr(samples),
                                    float(threshold), dptr(rot), dptr(loc), iptr(mask), ctypes.byref(cnt),
                                    ctypes.byref(best)))
    idx = np.flatnonzero(mask)
    return rot, loc.reshape(3, 1), (idx if as_array else idx.tolist()), best.value


def comm_available():
    """True when the library can load RCCL in this process (sfm_comm_available)."""
    return load().sfm_comm_available() == OK


def comm_unique_id():
    """128 opaque bytes identifying a new RCCL communicator (rank 0 calls this and hands them to the other ranks)."""
    buf = ctypes.create_string_buffer(128)
    lib = load()
    lib.sfm_comm_unique_id.argtypes = [ctypes.c_char_p]
    check(lib.sfm_comm_unique_id(buf))
    return buf.raw


class Comm:
    """A library-owned RCCL communicator of this process' GPU (sfm_comm_create / sfm_comm_destroy)."""

    def __init__(self, world_size, rank, unique_id):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be the 128 bytes of comm_unique_id()")
        self._lib = load()
        self._h = ctypes.c_void_p()
        self._lib.sfm_comm_create.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.POINTER(ctypes.c_void_p)]
        check(self._lib.sfm_comm_create(int(world_size), int(rank), bytes(unique_id), ctypes.byref(self._h)))
        self.world_size, self.rank = int(world_size), int(rank)

    def close(self):
        if self._h:
            self._lib.sfm_comm_destroy.argtypes = [ctypes.c_void_p]
            check(self._lib.sfm_comm_destroy(self._h))      # refused (SfmHipError) while a problem still holds it
            self._h = ctypes.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def pnp_ransac_evaluate(uv_pix, pts_h, intrinsic, samples, threshold):
    """Every six-point hypothesis: (rot (n_hyp,3,3), loc (n_hyp,3), inlier counts under (R, C), inlier counts under (R, -C))."""
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic)
    samples = i32(samples).reshape(-1, 6)
    n, n_hyp = uv_pix.shape[1], samples.shape[0]
    rot = np.empty((n_hyp, 3, 3)); loc = np.empty((n_hyp, 3))
    cnt = np.empty(n_hyp, dtype=np.int32); cnt_neg = np.empty(n_hyp, dtype=np.int32)
    lib = load()
    lib.sfm_pnp_ransac_evaluate.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_double, _dp, _dp, _ip, _ip]
    check(lib.sfm_pnp_ransac_evaluate(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), n_hyp, iptr(samples),
                                      float(threshold), dptr(rot), dptr(loc), iptr(cnt), iptr(cnt_neg)))
    return rot, loc, cnt, cnt_neg


def pnp_ransac_begin(uv_pix, pts_h, intrinsic, samples, threshold):
    """``pnp_ransac_evaluate`` that keeps the view's keys and points on the device: returns (session handle, rot (n_hyp,3,3),
    loc (n_hyp,3), counts, counts_neg).  The handle goes to ``pnp_ransac_finish`` (or ``pnp_session_destroy``)."""
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic)
    samples = i32(samples).reshape(-1, 6)
    n, n_hyp = uv_pix.shape[1], samples.shape[0]
    rot = np.empty((n_hyp, 3, 3)); loc = np.empty((n_hyp, 3))
    cnt = np.empty(n_hyp, dtype=np.int32); cnt_neg = np.empty(n_hyp, dtype=np.int32)
    handle = ctypes.c_void_p()
    lib = load()
    lib.sfm_pnp_ransac_begin.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_double, _dp, _dp, _ip, _ip,
                                         ctypes.POINTER(ctypes.c_void_p)]
    check(lib.sfm_pnp_ransac_begin(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), n_hyp, iptr(samples), float(threshold),
                                   dptr(rot), dptr(loc), iptr(cnt), iptr(cnt_neg), ctypes.byref(handle)))
    return (handle, n), rot, loc, cnt, cnt_neg


def pnp_ransac_finish(session, rot, loc, threshold, lam, iters, quirks=QUIRKS_REFERENCE):
    """Inlier mask of the pose (rot, loc) on the session's resident view, the inlier columns compacted on the device, ``iters``
    nonlinear PnP iterations on them: returns (inlier indices (int array, ascending), R (3,3), C (3,1)).  Releases the session."""
    handle, n = session
    rot = f64(rot); loc = f64(loc).reshape(3)
    mask = np.empty(n, dtype=np.int32)
    cnt = ctypes.c_int()
    r_out = np.empty((3, 3)); c_out = np.empty(3)
    lib = load()
    lib.sfm_pnp_ransac_finish.argtypes = [ctypes.c_void_p, _dp, _dp, ctypes.c_double, ctypes.c_double, ctypes.c_int, ctypes.c_int,
                                          _ip, _ip, _dp, _dp]
    check(lib.sfm_pnp_ransac_finish(handle, dptr(rot), dptr(loc), float(threshold), float(lam), int(iters), int(quirks),
                                    iptr(mask), ctypes.byref(cnt), dptr(r_out), dptr(c_out)))
    return np.flatnonzero(mask), r_out, c_out.reshape(3, 1)


def pnp_session_destroy(session):
    handle, _n = session
    lib = load()
    lib.sfm_pnp_session_destroy.argtypes = [ctypes.c_void_p]
    check(lib.sfm_pnp_session_destroy(handle))


def pnp_inlier_mask(uv_pix, pts_h, intrinsic, rot, loc, threshold, as_array=False):
    """Inlier index list (or int array) of one pose (pixel reprojection error below `threshold`, campose_processor.py:544-554)."""
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic); rot = f64(rot); loc = f64(loc).reshape(3)
    n = uv_pix.shape[1]
    mask = np.empty(n, dtype=np.int32)
    cnt = ctypes.c_int()
    lib = load()
    lib.sfm_pnp_inlier_mask.argtypes = [ctypes.c_int, _dp, _dp, _dp, _dp, _dp, ctypes.c_double, _ip, _ip]
    check(lib.sfm_pnp_inlier_mask(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), dptr(rot), dptr(loc), float(threshold),
                                  iptr(mask), ctypes.byref(cnt)))
    idx = np.flatnonzero(mask)
    return idx if as_array else idx.tolist()


def pnp_six_point_hypotheses(uv_pix, pts_h, intrinsic, samples, threshold):
    """Parity hook: (rot (n_hyp,3,3), loc (n_hyp,3), inlier counts (n_hyp,)) of every six-point hypothesis."""
    uv_pix = f64(uv_pix); pts_h = f64(pts_h); intrinsic = f64(intrinsic)
    samples = i32(samples).reshape(-1, 6)
    n, n_hyp = uv_pix.shape[1], samples.shape[0]
    rot = np.empty((n_hyp, 3, 3)); loc = np.empty((n_hyp, 3)); cnt = np.empty(n_hyp, dtype=np.int32)
    lib = load()
    lib.sfm_pnp_six_point_hypotheses.argtypes = [ctypes.c_int, _dp, _dp, _dp, ctypes.c_int, _ip, ctypes.c_double, _dp, _dp, _ip]
    check(lib.sfm_pnp_six_point_hypotheses(n, dptr(uv_pix), dptr(pts_h), dptr(intrinsic), n_hyp, iptr(samples),
                                           float(threshold), dptr(rot), dptr(loc), iptr(cnt)))
    return rot, loc, cnt


def fundamental_ransac(left, right, samples, threshold):
    """Eight-point RANSAC (epipolar_processor.py:22-57).  left/right: (>=2, n) pixel rows; samples: (n_hyp, 8)
    indices drawn by the caller (ignored when n == 8).  Returns (F (3,3), inlier index list or None, best
    hypothesis index or -1)."""
    left = f64(np.asarray(left)[0:2]); right = f64(np.asarray(right)[0:2])
    n = left.shape[1]
    samples = i32(samples if samples is not None and n != 8 else np.arange(8)).reshape(-1, 8)
    fund = np.empty((3, 3)); mask = np.empty(max(n, 1), dtype=np.int32)
    cnt = ctypes.c_int(); best = ctypes.c_int()
    lib = load()
    lib.sfm_fundamental_ransac.argtypes = [ctypes.c_int, _dp, _dp, ctypes.c_int, _ip, ctypes.c_double, _dp, _ip, _ip, _ip]
    check(lib.sfm_fundamental_ransac(n, dptr(left), dptr(right), samples.shape[0], iptr(samples), float(threshold),
                                     dptr(fund), iptr(mask), ctypes.byref(cnt), ctypes.byref(best)))
    inliers = None if best.value < 0 else np.flatnonzero(mask[:n]).tolist()
    return fund, inliers, best.value


def fundamental_eight_point(pairs, samples):
    """epipolar_processor.py:140-193 for every 8-index sample of the normalised pairs (n, 4) -> (F (n_hyp,3,3), status)."""
    pairs = f64(pairs); samples = i32(samples).reshape(-1, 8)
    out = np.empty((samples.shape[0], 3, 3)); st = np.empty(samples.shape[0], dtype=np.int32)
    check(load().sfm_fundamental_eight_point(pairs.shape[0], dptr(pairs), samples.shape[0], iptr(samples), dptr(out), iptr(st)))
    return out, st


def essential_from_fundamental(fund, left_k, right_k):
    """epipolar_processor.py:60-95."""
    fund = f64(fund); left_k = f64(left_k); right_k = f64(right_k)
    out = np.empty((3, 3))
    check(load().sfm_essential_from_fundamental(dptr(fund), dptr(left_k), dptr(right_k), dptr(out)))
    return out


def pose_candidates(esse):
    """campose_processor.py:29-100 -> (r1, r2, c1 (3,1), c2 = -c1)."""
    esse = f64(esse)
    rots = np.empty((2, 3, 3)); c1 = np.empty(3)
    check(load().sfm_pose_candidates(dptr(esse), dptr(rots), dptr(c1)))
    return rots[0].copy(), rots[1].copy(), c1.reshape(3, 1).copy(), -c1.reshape(3, 1)


def cheirality(ref_proj, projs, pts_sets):
    """campose_processor.py:102-189 for k candidates: projs (k,3,4), pts_sets (k,4,n) -> (masks (k,n), counts (k,), best)."""
    ref_proj = f64(ref_proj); projs = f64(projs).reshape(-1, 3, 4)
    pts_sets = f64(pts_sets).reshape(projs.shape[0], 4, -1)
    k, n = projs.shape[0], pts_sets.shape[2]
    mask = np.zeros((k, max(n, 1)), dtype=np.int32); counts = np.zeros(k, dtype=np.int32)
    best = ctypes.c_int()
    check(load().sfm_cheirality(k, n, dptr(ref_proj), dptr(projs), dptr(pts_sets), iptr(mask), iptr(counts), ctypes.byref(best)))
    return mask[:, :n], counts, best.value


def pnp_nonlinear_batch(offsets, uv_pix, pts_h, intrinsics, rot0, loc0, lam, iters, quirks=QUIRKS_REFERENCE):
    offsets = i32(offsets); uv_pix = f64(uv_pix); pts_h = f64(pts_h)
    intrinsics = f64(intrinsics).reshape(-1, 3, 3); rot0 = f64(rot0).reshape(-1, 3, 3); loc0 = f64(loc0).reshape(-1, 3)
    nv = offsets.shape[0] - 1
    total = uv_pix.shape[1]
    rot = np.empty((nv, 3, 3)); loc = np.empty((nv, 3)); st = np.empty(nv, dtype=np.int32)
    check(load().sfm_pnp_nonlinear_batch(nv, iptr(offsets), total, dptr(uv_pix), dptr(pts_h), dptr(intrinsics),
                                         dptr(rot0), dptr(loc0), float(lam), int(iters), int(quirks),
                                         dptr(rot), dptr(loc), iptr(st)))
    return rot, loc, st


def ba_residual_jacobian(n_cams, pt_ptr, cam_idx, uv_norm, cams, pts, quirks=QUIRKS_REFERENCE):
    pt_ptr = i32(pt_ptr); cam_idx = i32(cam_idx); uv_norm = f64(uv_norm); cams = f64(cams); pts = f64(pts)
    n, m = pt_ptr.shape[0] - 1, cam_idx.shape[0]
    r = np.empty((m, 2)); jp = np.empty((m, 2, 7)); jx = np.empty((m, 2, 3))
    check(load().sfm_ba_residual_jacobian(n_cams, n, m, iptr(pt_ptr), iptr(cam_idx), dptr(uv_norm), dptr(cams),
                                          dptr(pts), quirks, dptr(r), dptr(jp), dptr(jx)))
    return r, jp, jx


def ba_reduced_system(n_cams, pt_ptr, cam_idx, uv_norm, cams, pts, lam, quirks=QUIRKS_REFERENCE,
                      schur_mode=SCHUR_AUTO):
    pt_ptr = i32(pt_ptr); cam_idx = i32(cam_idx); uv_norm = f64(uv_norm); cams = f64(cams); pts = f64(pts)
    n, m = pt_ptr.shape[0] - 1, cam_idx.shape[0]
    s = np.empty((7 * n_cams, 7 * n_cams)); rhs = np.empty(7 * n_cams)
    check(load().sfm_ba_reduced_system(n_cams, n, m, iptr(pt_ptr), iptr(cam_idx), dptr(uv_norm), dptr(cams),
                                       dptr(pts), float(lam), quirks, schur_mode, dptr(s), dptr(rhs)))
    return s, rhs


def ba_solve(n_cams, pt_ptr, cam_idx, uv_norm, cams, pts, lam, iters, quirks=QUIRKS_REFERENCE):
    pt_ptr = i32(pt_ptr); cam_idx = i32(cam_idx); uv_norm = f64(uv_norm)
    cams = np.array(cams, dtype=np.float64, order="C", copy=True).reshape(-1, 7)
    pts = np.array(pts, dtype=np.float64, order="C", copy=True)
    n, m = pt_ptr.shape[0] - 1, cam_idx.shape[0]
    check(load().sfm_ba_solve(n_cams, n, m, iptr(pt_ptr), iptr(cam_idx), dptr(uv_norm), dptr(cams), dptr(pts),
                              float(lam), int(iters), int(quirks)))
    return cams, pts


class BaProblem:
    """Device-resident BA problem (sfm_ba_create ... sfm_ba_destroy)."""

    def __init__(self, n_cams, pt_ptr, cam_idx, uv_norm):
        self._lib = load()
        pt_ptr = i32(pt_ptr); cam_idx = i32(cam_idx); uv_norm = f64(uv_norm)
        self.n_cams = int(n_cams)
        self.n_pts = pt_ptr.shape[0] - 1
        self.n_obs = cam_idx.shape[0]
        if uv_norm.shape != (2, self.n_obs):
            raise ValueError("uv_norm must be (2, M)")
        h = ctypes.c_void_p()
        check(self._lib.sfm_ba_create(self.n_cams, self.n_pts, self.n_obs, iptr(pt_ptr), iptr(cam_idx),
                                      dptr(uv_norm), ctypes.byref(h)))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            self._lib.sfm_ba_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_option(self, option, value):
        check(self._lib.sfm_ba_set_option(self._h, option, value))

    def set_stream(self, stream_ptr):
        """Run this problem on an existing HIP stream (``torch.cuda.Stream.cuda_stream``); 0 / None = the library's own."""
        check(self._lib.sfm_ba_set_stream(self._h, ctypes.c_void_p(stream_ptr) if stream_ptr else None))

    def info(self, what):
        v = ctypes.c_int64()
        check(self._lib.sfm_ba_info(self._h, int(what), ctypes.byref(v)))
        return int(v.value)

    @property
    def upload_bytes(self):
        return self.info(INFO_UPLOAD_BYTES)

    def set_state(self, cams, pts):
        cams = f64(cams).reshape(-1, 7); pts = f64(pts)
        if cams.shape[0] != self.n_cams or pts.shape != (3, self.n_pts):
            raise ValueError("state shapes do not match the problem")
        check(self._lib.sfm_ba_set_state(self._h, dptr(cams), dptr(pts)))

    def set_cameras(self, cams):
        cams = f64(cams).reshape(-1, 7)
        if cams.shape[0] != self.n_cams:
            raise ValueError("camera count does not match the problem")
        check(self._lib.sfm_ba_set_cameras(self._h, dptr(cams)))

    def set_points(self, first, pts):
        pts = f64(pts).reshape(3, -1)
        check(self._lib.sfm_ba_set_points(self._h, int(first), pts.shape[1], dptr(pts)))

    def iterate(self, lam, iters, quirks=QUIRKS_REFERENCE):
        check(self._lib.sfm_ba_iterate(self._h, float(lam), int(iters), int(quirks)))

    def linearize_reduce(self, lam, quirks=QUIRKS_REFERENCE):
        check(self._lib.sfm_ba_linearize_reduce(self._h, float(lam), int(quirks)))

    def solve_update(self, lam, quirks=QUIRKS_REFERENCE):
        check(self._lib.sfm_ba_solve_update(self._h, float(lam), int(quirks)))

    def flush(self):
        """Complete the back substitution ``solve_update`` may have left to the next linearisation (sfm_ba_flush)."""
        check(self._lib.sfm_ba_flush(self._h))

    def set_comm(self, comm):
        """Attach a library-owned RCCL communicator (``Comm``) or detach it (None): ``iterate`` then all-reduces the
        packed [S | rhs] buffer itself, once per iteration, on the problem's stream (sfm_ba_set_comm)."""
        self._lib.sfm_ba_set_comm.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        check(self._lib.sfm_ba_set_comm(self._h, comm._h if comm is not None else None))
        self._comm = comm                      # keep it alive as long as it is attached

    def get_stats(self, max_iters=256):
        """Per-iteration cost sum |b - f|^2 (normalised image coordinates) at the start of every iteration run since
        the state was last uploaded -- no state download needed (sfm_ba_get_stats)."""
        out = np.empty(max_iters); n = ctypes.c_int()
        self._lib.sfm_ba_get_stats.argtypes = [ctypes.c_void_p, _dp, ctypes.c_int, _ip]
        check(self._lib.sfm_ba_get_stats(self._h, dptr(out), int(max_iters), ctypes.byref(n)))
        return out[:n.value].copy()

    def get_state(self):
        cams = np.empty((self.n_cams, 7)); pts = np.empty((3, self.n_pts))
        check(self._lib.sfm_ba_get_state(self._h, dptr(cams), dptr(pts)))
        return cams, pts

    def get_state_rot(self):
        """(cams (V,7), pts (3,N), rots (V,3,3)): the state plus R(q) of every camera, validated on the device."""
        cams = np.empty((self.n_cams, 7)); pts = np.empty((3, self.n_pts)); rots = np.empty((self.n_cams, 3, 3))
        check(self._lib.sfm_ba_get_state_rot(self._h, dptr(cams), dptr(pts), dptr(rots)))
        return cams, pts, rots

    def rederive_quaternions(self, first, count):
        """q <- q(R(q)) on the device for cameras [first, first + count) (sfm_ba_rederive_quaternions)."""
        check(self._lib.sfm_ba_rederive_quaternions(self._h, int(first), int(count)))

    def append(self, cams_new, pts_new, obs_cam, obs_pt, uv_norm):
        """Grow the resident scene (sfm_ba_append): new cameras (k,7), new points (3,k), new observations
        (camera index, point index, normalised key (2,m)) of pairs not yet present; nothing already on the
        device is uploaded again."""
        cams_new = f64(cams_new).reshape(-1, 7); pts_new = f64(pts_new).reshape(3, -1)
        obs_cam = i32(obs_cam).ravel(); obs_pt = i32(obs_pt).ravel(); uv_norm = f64(uv_norm).reshape(2, -1)
        if not (obs_cam.shape[0] == obs_pt.shape[0] == uv_norm.shape[1]):
            raise ValueError("append: obs_cam, obs_pt and uv_norm disagree in length")
        check(self._lib.sfm_ba_append(self._h, cams_new.shape[0], dptr(cams_new), pts_new.shape[1], dptr(pts_new),
                                      obs_cam.shape[0], iptr(obs_cam), iptr(obs_pt), dptr(uv_norm)))
        self.n_cams += cams_new.shape[0]
        self.n_pts += pts_new.shape[1]
        self.n_obs += obs_cam.shape[0]

    def points_ptr(self):
        """Device pointers (px, py, pz) of the resident points and their count (sfm_ba_points_ptr)."""
        a, b, c = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
        n = ctypes.c_int()
        check(self._lib.sfm_ba_points_ptr(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(n)))
        return a.value, b.value, c.value, n.value

    def stream_ptr(self):
        """The HIP stream this problem runs on (hipStream_t as an integer; sfm_ba_stream)."""
        s = ctypes.c_void_p()
        check(self._lib.sfm_ba_stream(self._h, ctypes.byref(s)))
        return s.value or 0

    def reduced_buffer(self):
        ptr = ctypes.c_void_p(); n = ctypes.c_int64(); ld = ctypes.c_int()
        check(self._lib.sfm_ba_reduced_buffer(self._h, ctypes.byref(ptr), ctypes.byref(n), ctypes.byref(ld)))
        return ptr.value, n.value, ld.value

    def bind_reduced_buffer(self, device_ptr, n_doubles):
        check(self._lib.sfm_ba_bind_reduced_buffer(self._h, ctypes.c_void_p(device_ptr), int(n_doubles)))

    def kernel_time(self, kernel_id):
        ms = ctypes.c_double(); n = ctypes.c_int()
        check(self._lib.sfm_ba_kernel_time(self._h, kernel_id, ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def event_overhead(self, n=20):
        """Average hipEvent bracket (ms) around an empty kernel on this problem's stream (sfm_ba_event_overhead)."""
        ms = ctypes.c_double()
        check(self._lib.sfm_ba_event_overhead(self._h, int(n), ctypes.byref(ms)))
        return ms.value

    def reset_timing(self):
        check(self._lib.sfm_ba_reset_timing(self._h))

    def debug_stamps(self, n=1024):
        out = np.zeros(n, dtype=np.uint64)
        self._lib.sfm_ba_debug_stamps.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
        check(self._lib.sfm_ba_debug_stamps(self._h, out.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), n))
        return out
