"""Host-side rotation / quaternion helpers of the drop-in layer.

These mirror the conventions of the reference's ``utils`` module (quaternion is a
``(4, 1)`` column ``[qw, qx, qy, qz]``, w first; ``rot`` is the camera rotation R with
world->camera ``p = R^T (X - C)``):

* ``rotation_to_quaternion``  <-> reference utils.py:28-60
* ``quaternion_to_rotation``  <-> reference utils.py:64-97
* ``is_rotation``             <-> reference utils.py:101-105

They are used only to pack/unpack the 7-double camera blocks that cross the C-ABI and to
raise the reference's ``ValueError`` on the host before any device work is queued.  The
per-observation arithmetic itself lives in the HIP kernels (csrc/).
"""
import math

import numpy as np

ROT_TOL = 1e-8      # utils.py:102 (one-sided tests)
QW_MIN = 1e-6       # utils.py:49


def is_rotation(rot):
    """One-sided validity predicate of the reference (utils.py:101-105)."""
    rot = np.asarray(rot)
    if rot.shape != (3, 3):
        return False
    if np.linalg.det(rot) - 1 >= ROT_TOL:
        return False
    if np.any((np.linalg.inv(rot) - rot.T) > ROT_TOL):
        return False
    return True


def rotation_to_quaternion(rot):
    """R -> canonical (qw >= 0) quaternion column, raising like utils.py:43-51."""
    if not is_rotation(rot):
        raise ValueError('{} : Invalid input rotation matrix \n {}'.format(
            "convert_rotation_to_quaternion", rot))
    qw = math.sqrt(1 + rot[0][0] + rot[1][1] + rot[2][2]) / 2.0
    if abs(qw - 0) < QW_MIN:
        raise ValueError('{} : Invalid output qw \n {}'.format(
            "convert_rotation_to_quaternion", qw))
    qx = (rot[2][1] - rot[1][2]) / (4 * qw)
    qy = (rot[0][2] - rot[2][0]) / (4 * qw)
    qz = (rot[1][0] - rot[0][1]) / (4 * qw)
    return np.array([[qw], [qx], [qy], [qz]], dtype=np.float64)


def quaternion_to_rotation_unchecked(quat):
    w, x, y, z = (float(v) for v in np.asarray(quat, dtype=np.float64).reshape(4))
    return np.array([
        [1 - 2 * z * z - 2 * y * y, -2 * z * w + 2 * y * x, 2 * y * w + 2 * z * x],
        [2 * x * y + 2 * w * z, 1 - 2 * z * z - 2 * x * x, 2 * z * y - 2 * x * w],
        [2 * x * z - 2 * w * y, 2 * y * z + 2 * w * x, 1 - 2 * y * y - 2 * x * x]],
        dtype=np.float64)


def quaternion_to_rotation(quat):
    """q -> R (not normalised inside), raising like utils.py:93-95."""
    rot = quaternion_to_rotation_unchecked(quat)
    if not is_rotation(rot):
        raise ValueError('{} : Invalid output rotation matrix \n {}'.format(
            "convert_quaternion_to_rotation", rot))
    return rot


def pack_camera(rot, loc):
    """(R, C) -> the 7-double parameter block [Cx,Cy,Cz,qw,qx,qy,qz] (ba_processor.py:285-288)."""
    quat = rotation_to_quaternion(np.asarray(rot, dtype=np.float64))
    return np.concatenate((np.asarray(loc, dtype=np.float64).reshape(3), quat.reshape(4)))


def pack_cameras(rots, locs):
    """Batched ``pack_camera``: (V,3,3), (V,3[,1]) -> (V,7), the same arithmetic and the same ``ValueError`` (for the
    first offending view) as V calls of ``pack_camera``; one vectorised pass instead of V interpreter round trips."""
    rots = np.asarray(rots, dtype=np.float64).reshape(-1, 3, 3)
    n = rots.shape[0]
    locs = np.asarray(locs, dtype=np.float64).reshape(n, 3)
    if n == 0:
        return np.zeros((0, 7))
    bad = (np.linalg.det(rots) - 1 >= ROT_TOL) | np.any((np.linalg.inv(rots) - np.transpose(rots, (0, 2, 1))) > ROT_TOL, axis=(1, 2))
    if np.any(bad):
        rotation_to_quaternion(rots[int(np.flatnonzero(bad)[0])])     # raises the reference's message
    tr = 1 + rots[:, 0, 0] + rots[:, 1, 1] + rots[:, 2, 2]
    if np.any(tr < 0):
        raise ValueError("math domain error")
    qw = np.sqrt(tr) / 2.0
    if np.any(np.abs(qw - 0) < QW_MIN):
        rotation_to_quaternion(rots[int(np.flatnonzero(np.abs(qw) < QW_MIN)[0])])
    out = np.empty((n, 7))
    out[:, 0:3] = locs
    out[:, 3] = qw
    out[:, 4] = (rots[:, 2, 1] - rots[:, 1, 2]) / (4 * qw)
    out[:, 5] = (rots[:, 0, 2] - rots[:, 2, 0]) / (4 * qw)
    out[:, 6] = (rots[:, 1, 0] - rots[:, 0, 1]) / (4 * qw)
    return out


def quaternions_to_rotations(quats):
    """Batched ``quaternion_to_rotation``: (V,4) -> (V,3,3), raising like utils.py:93-95 for the first invalid one."""
    q = np.asarray(quats, dtype=np.float64).reshape(-1, 4)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    rot = np.empty((q.shape[0], 3, 3))
    rot[:, 0, 0] = 1 - 2 * z * z - 2 * y * y; rot[:, 0, 1] = -2 * z * w + 2 * y * x; rot[:, 0, 2] = 2 * y * w + 2 * z * x
    rot[:, 1, 0] = 2 * x * y + 2 * w * z; rot[:, 1, 1] = 1 - 2 * z * z - 2 * x * x; rot[:, 1, 2] = 2 * z * y - 2 * x * w
    rot[:, 2, 0] = 2 * x * z - 2 * w * y; rot[:, 2, 1] = 2 * y * z + 2 * w * x; rot[:, 2, 2] = 1 - 2 * y * y - 2 * x * x
    if q.shape[0]:
        bad = (np.linalg.det(rot) - 1 >= ROT_TOL) | np.any((np.linalg.inv(rot) - np.transpose(rot, (0, 2, 1))) > ROT_TOL, axis=(1, 2))
        if np.any(bad):
            quaternion_to_rotation(q[int(np.flatnonzero(bad)[0])])
    return rot


def normalise_pixels(uv_pix, intrinsic):
    """Pixel keys -> normalised camera coordinates exactly as the reference does it:
    ``inv(K) @ [u, v, 1]`` divided by its third component (ba_processor.py:339-342,
    campose_processor.py:393-394).  ``uv_pix`` is (2, m) or (3, m); returns (2, m)."""
    uv_pix = np.asarray(uv_pix, dtype=np.float64)
    m = uv_pix.shape[1]
    if uv_pix.shape[0] == 2:
        hom = np.vstack((uv_pix, np.ones((1, m))))
    else:
        hom = uv_pix
    cam = np.linalg.inv(np.asarray(intrinsic, dtype=np.float64)) @ hom
    cam = cam / cam[2:3, :]
    return np.ascontiguousarray(cam[0:2, :])
