"""Seeded synthetic bundle-adjustment scenes.

The reference has no scene generator (its demo needs OpenCV SIFT on the upenn BMPs,
ba_processor.py:443-546); BASELINE.json's configs 2-4 are "synthetic V cams x N points at
p% visibility".  This module is the single definition of that workload: it is used by
bench.py, by the GPU parity tests and by tools/capture_goldens.py (which additionally
stores the generated arrays inside the golden fixtures so the tests never depend on the
generator staying bit-stable).

Conventions follow the reference: ``rot`` = R, ``loc`` = C (camera centre), world->camera
``p = R^T (X - C)`` (view_processor.py:53-57); the intrinsics are the demo's upenn K
(ba_processor.py:457-459); the camera block is ``[Cx,Cy,Cz,qw,qx,qy,qz]``
(ba_processor.py:285-288).
"""
from dataclasses import dataclass

import numpy as np
from scipy.spatial.transform import Rotation

from .geometry import rotation_to_quaternion

UPENN_K = np.array([[568.996140852, 0.0, 643.21055941],
                    [0.0, 568.988362396, 477.982801038],
                    [0.0, 0.0, 1.0]])

# BASELINE.json configs (name -> cams, points, visibility)
CONFIGS = {
    "C2": dict(n_cams=5, n_pts=2000, visibility=1.0),
    "C3": dict(n_cams=50, n_pts=20000, visibility=0.6),
    "C4": dict(n_cams=200, n_pts=100000, visibility=0.15),
}


@dataclass
class Scene:
    """Observation-list form of a BA problem (what crosses the C-ABI)."""
    intrinsic: np.ndarray      # (3,3)
    cams_true: np.ndarray      # (V,7)
    cams_init: np.ndarray      # (V,7)  [C, q]
    pts_true: np.ndarray       # (3,N)
    pts_init: np.ndarray       # (3,N)
    pt_ptr: np.ndarray         # (N+1,) int32 CSR over observations sorted by (point, cam)
    cam_idx: np.ndarray        # (M,) int32
    pt_idx: np.ndarray         # (M,) int32
    uv_pix: np.ndarray         # (2,M) pixel observations

    @property
    def n_cams(self):
        return self.cams_init.shape[0]

    @property
    def n_pts(self):
        return self.pts_init.shape[1]

    @property
    def n_obs(self):
        return self.cam_idx.shape[0]


def _project(rot, loc, pts, intrinsic):
    cam = rot.T @ (pts - loc.reshape(3, 1))
    pix = intrinsic @ cam
    return pix[0:2] / pix[2:3], cam[2]


def visibility_mask(rng, n_cams, n_pts, visibility):
    """Bernoulli(p) per (cam, point); every point is forced into at least two cameras."""
    vis = rng.random((n_cams, n_pts)) < visibility
    need = np.flatnonzero(vis.sum(axis=0) < 2)
    for p in need:
        extra = rng.choice(n_cams, size=2, replace=False)
        vis[extra, p] = True
    return vis


def make_scene(n_cams, n_pts, visibility=1.0, seed=0, pixel_noise=0.5,
               rot_noise=0.01, loc_noise=0.05, pt_noise=0.05):
    rng = np.random.default_rng(seed)
    intrinsic = UPENN_K.copy()

    pts_true = np.vstack((rng.uniform(-4, 4, n_pts),
                          rng.uniform(-3, 3, n_pts),
                          rng.uniform(8, 16, n_pts)))

    rots, locs = [np.eye(3)], [np.zeros(3)]
    for _ in range(1, n_cams):
        ang = rng.uniform(-0.15, 0.15, 3)
        rots.append(Rotation.from_euler('zyx', ang).as_matrix())
        locs.append(rng.uniform(-1.5, 1.5, 3) * np.array([1.0, 0.3, 0.5]))

    vis = visibility_mask(rng, n_cams, n_pts, visibility)

    # observation list sorted by (point, cam): the reference's BA loop order (ba_processor.py:304-306)
    pt_idx, cam_idx = np.nonzero(vis.T)
    pt_idx = pt_idx.astype(np.int32)
    cam_idx = cam_idx.astype(np.int32)
    m = pt_idx.shape[0]
    uv = np.empty((2, m))
    for c in range(n_cams):
        sel = np.flatnonzero(cam_idx == c)
        pix, _depth = _project(rots[c], locs[c], pts_true[:, pt_idx[sel]], intrinsic)
        uv[:, sel] = pix
    uv += rng.normal(0.0, pixel_noise, uv.shape)

    pt_ptr = np.zeros(n_pts + 1, dtype=np.int32)
    np.cumsum(np.bincount(pt_idx, minlength=n_pts), out=pt_ptr[1:])

    cams_true = np.empty((n_cams, 7))
    cams_init = np.empty((n_cams, 7))
    for c in range(n_cams):
        cams_true[c, 0:3] = locs[c]
        cams_true[c, 3:7] = rotation_to_quaternion(rots[c]).reshape(4)
        if c == 0:
            r0, c0 = rots[c], locs[c]
        else:
            r0 = rots[c] @ Rotation.from_rotvec(rng.normal(0.0, rot_noise, 3)).as_matrix()
            c0 = locs[c] + rng.normal(0.0, loc_noise, 3)
        cams_init[c, 0:3] = c0
        cams_init[c, 3:7] = rotation_to_quaternion(r0).reshape(4)
    pts_init = pts_true + rng.normal(0.0, pt_noise, pts_true.shape)

    return Scene(intrinsic, cams_true, cams_init, pts_true, pts_init,
                 pt_ptr, cam_idx, pt_idx, uv)


def make_config(name, seed=0, n_pts=None):
    cfg = dict(CONFIGS[name])
    if n_pts is not None:
        cfg["n_pts"] = n_pts
    return make_scene(seed=seed, **cfg)


def reprojection_rmse(cams, pts, scene):
    """Pixel RMSE ``sqrt(mean ||K pi(R^T (X - C)) - uv||^2)`` over all observations
    (the metric definition of BASELINE.md section 4)."""
    cams = np.asarray(cams, dtype=np.float64).reshape(-1, 7)
    pts = np.asarray(pts, dtype=np.float64)
    c = cams[scene.cam_idx]
    w, x, y, z = c[:, 3], c[:, 4], c[:, 5], c[:, 6]
    d = pts[:, scene.pt_idx].T - c[:, 0:3]
    # rows of R^T applied to d
    r00 = 1 - 2 * z * z - 2 * y * y; r01 = -2 * z * w + 2 * y * x; r02 = 2 * y * w + 2 * z * x
    r10 = 2 * x * y + 2 * w * z; r11 = 1 - 2 * z * z - 2 * x * x; r12 = 2 * z * y - 2 * x * w
    r20 = 2 * x * z - 2 * w * y; r21 = 2 * y * z + 2 * w * x; r22 = 1 - 2 * y * y - 2 * x * x
    px = r00 * d[:, 0] + r10 * d[:, 1] + r20 * d[:, 2]
    py = r01 * d[:, 0] + r11 * d[:, 1] + r21 * d[:, 2]
    pz = r02 * d[:, 0] + r12 * d[:, 1] + r22 * d[:, 2]
    k = scene.intrinsic
    u = k[0, 0] * px / pz + k[0, 1] * py / pz + k[0, 2]
    v = k[1, 1] * py / pz + k[1, 2]
    err = (u - scene.uv_pix[0]) ** 2 + (v - scene.uv_pix[1]) ** 2
    return float(np.sqrt(err.mean()))
