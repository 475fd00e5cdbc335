#!/usr/bin/env python3
"""Capture golden input/output vectors from the REAL reference (build container only).

Run:  PYTHONDONTWRITEBYTECODE=1 python tools/capture_goldens.py [--skip-slow]

Imports willSapgreen/structure-from-motion from /root/reference (read-only; never copied),
drives its nonlinear-refinement hot path on seeded inputs and on the data files its own
tests hold (test_dataset/opencv/*.npy), and writes small .npz fixtures to tests/golden/.
The fixtures contain data only (inputs + the reference's outputs); the GPU box never sees
the reference.  Recipe follows SURVEY.md Appendix B: ``ba_processor`` imports
``view_processor``/``key_tracker`` which import ``cv2`` at module scope; cv2 is not
installed, the BA core never calls it, so an empty placeholder module is registered and the
BA method is driven with duck-typed View/KeyTracker objects that carry real
``KeyTrack`` tables and the real ``KeyTracker.is_visible``.
"""
import argparse
import contextlib
import importlib
import io
import os
import random
import sys
import time
import types
import warnings

import numpy as np

REF = "/root/reference"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")

warnings.filterwarnings("ignore", category=DeprecationWarning)
sys.path.insert(0, REPO)
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

import utils as ref_utils                                   # noqa: E402
import triangulation_processor as ref_tri                   # noqa: E402
import campose_processor as ref_cam                         # noqa: E402

sys.modules.setdefault("cv2", types.ModuleType("cv2"))      # placeholder; never called
import key_tracker as ref_kt                                # noqa: E402
import ba_processor as ref_ba                               # noqa: E402

sfm = importlib.import_module("structure-from-motion_amd")
scenes = sfm.scenes


def quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def random_rotation(rng, max_angle=None):
    from scipy.spatial.transform import Rotation
    if max_angle is None:
        return Rotation.random(random_state=rng.integers(1 << 31)).as_matrix()
    axis = rng.normal(size=3)
    axis /= np.linalg.norm(axis)
    return Rotation.from_rotvec(axis * rng.uniform(0, max_angle)).as_matrix()


# ---------------------------------------------------------------------------------------
def g1_jac_cam(rng):
    cp = ref_cam.CamposeProcessor(None, 5, 10)
    n = 1000
    rs, cs, xs, js = [], [], [], []
    while len(rs) < n:
        k = len(rs)
        if k < 600:
            r = random_rotation(rng, 0.6)
        elif k < 800:
            r = random_rotation(rng, 3.0)          # large angles (qw small but > 1e-6)
        elif k < 900:
            r = random_rotation(rng, 1e-6)         # near identity
        else:
            ax = np.eye(3)[k % 3]
            from scipy.spatial.transform import Rotation
            r = Rotation.from_rotvec(ax * rng.uniform(-2.5, 2.5)).as_matrix()   # axis aligned
        c = rng.uniform(-2, 2, (3, 1))
        x = np.vstack((r @ (rng.uniform(-1, 1, (3, 1)) * np.array([[3], [3], [0]])
                            + np.array([[0], [0], [rng.uniform(2, 20)]])) + c, [[1.0]]))
        try:
            j = cp.construct_jacobian_matrix(r, c, x)
        except ValueError:
            continue
        rs.append(r); cs.append(c[:, 0]); xs.append(x[:, 0]); js.append(j)
    np.savez_compressed(os.path.join(OUT, "g1_jac_cam.npz"),
                        R=np.array(rs), C=np.array(cs), X=np.array(xs), Jp=np.array(js))


def g2_jac_pt(rng):
    tp = ref_tri.TriangulationProcessor()
    n = 300
    xs, ps, js = [], [], []
    k = scenes.UPENN_K
    for i in range(n):
        projs = []
        for v in range(3):
            r = random_rotation(rng, 0.4)
            c = rng.uniform(-1, 1, (3, 1))
            p = np.hstack((r.T, r.T @ -c))
            if i % 2 == 0:
                p = k @ p                                  # pixel projection; odd = K-free
            projs.append(p)
        x = np.array([[rng.uniform(-3, 3)], [rng.uniform(-3, 3)], [rng.uniform(6, 15)], [1.0]])
        xs.append(x[:, 0]); ps.append(np.array(projs))
        js.append(tp.construct_jacobian_matrix(x, projs, 3))
    np.savez_compressed(os.path.join(OUT, "g2_jac_pt.npz"),
                        X=np.array(xs), projs=np.array(ps), Jx=np.array(js))


def g3_quat(rng):
    qs, rs, q_back = [], [], []
    for i in range(200):
        r = random_rotation(rng, 2.8)
        q = ref_utils.convert_rotation_to_quaternion(r)
        r2 = ref_utils.convert_quaternion_to_rotation(q)
        qs.append(q[:, 0]); rs.append(r); q_back.append(r2)
    # accept / reject set of verify_rotation_mat around its 1e-8 one-sided thresholds
    cases, verdict = [], []
    base = random_rotation(rng, 1.0)
    for eps in (0.0, 1e-10, 9e-9, 1.1e-8, 1e-7, -1e-7, 1e-6, -1e-6, 1e-3, -1e-3):
        for kind in range(4):
            m = base.copy()
            if kind == 0:
                m = m * (1.0 + eps)                        # scale: det and inverse both move
            elif kind == 1:
                m[0, 1] += eps                             # shear one entry
            elif kind == 2:
                m[2, 2] += eps
            else:
                m = -m if eps > 5e-4 else m * (1.0 - eps)  # reflection / shrink
            cases.append(m); verdict.append(bool(ref_utils.verify_rotation_mat(m)))
    # unnormalised quaternions through convert_quaternion_to_rotation (raise or not)
    qn, ok = [], []
    for s in (1.0, 1.0 + 1e-10, 1.0 + 2e-9, 1.0 + 1e-8, 1.0 - 1e-8, 1.0 + 1e-6, 1.0 - 1e-6, 1.01, 0.99):
        q = ref_utils.convert_rotation_to_quaternion(random_rotation(rng, 1.0)) * s
        try:
            ref_utils.convert_quaternion_to_rotation(q); good = True
        except ValueError:
            good = False
        qn.append(q[:, 0]); ok.append(good)
    np.savez_compressed(os.path.join(OUT, "g3_quat.npz"), q=np.array(qs), R=np.array(rs),
                        R_back=np.array(q_back), verify_cases=np.array(cases),
                        verify_verdict=np.array(verdict), q_scaled=np.array(qn), q_scaled_ok=np.array(ok))


def opencv_two_view():
    d = os.path.join(REF, "test_dataset", "opencv")
    k = np.load(d + "/ess_intrinsic_mat.npy")
    ref_r = np.load(d + "/ess_self_r.npy").T
    ref_c = np.load(d + "/ess_self_c.npy")
    p1 = np.load(d + "/ess_pixel_pt1.npy").T
    p2 = np.load(d + "/ess_pixel_pt2.npy").T
    r1 = np.load(d + "/ess_r1.npy").T
    c2 = np.load(d + "/ess_c2.npy")
    proj_ref = k @ np.hstack((ref_r.T, ref_r.T @ -ref_c))
    proj_que = k @ np.hstack((r1.T, -r1.T @ c2))            # best candidate idx 1 (campose:907,942)
    m = p1.shape[1]
    kp1 = ref_utils.KeyPt(m); kp2 = ref_utils.KeyPt(m)
    kp1[0:2] = p1[0:2]; kp2[0:2] = p2[0:2]
    return k, proj_ref, proj_que, kp1, kp2


def g4_tri(rng, slow):
    tp = ref_tri.TriangulationProcessor()
    out = {}
    # (a) the literal known-answer case of the reference's own test (tri:415-473)
    p1 = np.array([[5.010e+03, 0.000e+00, 3.600e+02, 0.000e+00],
                   [0.000e+00, 5.010e+03, 6.400e+02, 0.000e+00],
                   [0.000e+00, 0.000e+00, 1.000e+00, 0.000e+00]])
    p2 = np.array([[5.037e+03, -9.611e+01, -1.756e+03, 4.284e+03],
                   [2.148e+02, 5.354e+03, 1.918e+02, 8.945e+02],
                   [3.925e-01, 7.092e-02, 9.169e-01, 4.930e-01]])
    p3 = np.array([[5.217e+03, 2.246e+02, 2.366e+03, -3.799e+03],
                   [-5.734e+02, 5.669e+03, 8.233e+02, -2.567e+02],
                   [-3.522e-01, -5.839e-02, 9.340e-01, 6.459e-01]])
    x1 = ref_utils.KeyPt(1); x1[:, 0] = [274.128, 624.409, 1.0]
    x2 = ref_utils.KeyPt(1); x2[:, 0] = [239.571, 533.568, 1.0]
    x3 = ref_utils.KeyPt(1); x3[:, 0] = [297.574, 549.260, 1.0]
    lin = tp.linear_triangulate([p1, p2], [x1, x2])
    whole = tp.triangulate([p1, p2], [x1, x2], 0.5, 300)
    out.update(lit_projs=np.array([p1, p2, p3]), lit_uv=np.array([x1, x2, x3]),
               lit_linear=np.asarray(lin), lit_whole=np.asarray(whole),
               lit_three_view=tp.nonlinear_triangulate(np.asarray(lin), [p1, p2, p3], [x1, x2, x3], 0.5, 50))
    # (b) the reference's two-view data files (1538 pairs)
    k, pr, pq, kp1, kp2 = opencv_two_view()
    init = np.asarray(tp.linear_triangulate([pr, pq], [kp1, kp2]))
    out.update(cv_K=k, cv_projs=np.array([pr, pq]), cv_uv=np.array([np.asarray(kp1), np.asarray(kp2)]),
               cv_init=init)
    for its in ((1, 10, 100) if slow else (1, 10)):
        out["cv_its%d" % its] = tp.nonlinear_triangulate(init, [pr, pq], [kp1, kp2], 0.5, its)
    out["cv_lam10_its5"] = tp.nonlinear_triangulate(init, [pr, pq], [kp1, kp2], 10, 5)
    # (c) synthetic 2/3/5 views
    for nv in (2, 3, 5):
        sc = scenes.make_scene(nv, 40, 1.0, seed=100 + nv)
        projs, mps = [], []
        for c in range(nv):
            r = ref_utils.convert_quaternion_to_rotation(sc.cams_true[c, 3:7].reshape(4, 1))
            loc = sc.cams_true[c, 0:3].reshape(3, 1)
            projs.append(sc.intrinsic @ np.hstack((r.T, r.T @ -loc)))
            kp = ref_utils.KeyPt(40)
            kp[0:2] = sc.uv_pix[:, sc.cam_idx == c]
            mps.append(kp)
        init = ref_utils.TriPt(40)
        init[0:3] = sc.pts_init
        res = tp.nonlinear_triangulate(init, projs, mps, 0.5, 20)
        out["syn%d_projs" % nv] = np.array(projs)
        out["syn%d_uv" % nv] = np.array([np.asarray(m) for m in mps])
        out["syn%d_init" % nv] = np.asarray(init)
        out["syn%d_out" % nv] = np.asarray(res)
    np.savez_compressed(os.path.join(OUT, "g4_tri.npz"), **out)


def g5_pnp(rng, slow):
    d = os.path.join(REF, "test_dataset", "opencv")
    k = np.load(d + "/ess_intrinsic_mat.npy")
    p3 = np.load(d + "/pnp_points_3d.npy").T
    p2 = np.load(d + "/pnp_points_2d.npy").T
    ones = np.ones((1, p3.shape[1]))
    p3h = np.vstack((p3, ones)); p2h = np.vstack((p2, ones))
    rot_truth = np.load(d + "/pnp_rotation.npy").T
    loc_truth = rot_truth @ -np.load(d + "/pnp_translation.npy")
    cfg = quiet(ref_utils.RansacConfig, 8.0, 0.99, 0.75, 6, 300)       # seeds python random with -1
    cp = ref_cam.CamposeProcessor(cfg, 5, 200)
    t0 = time.time()
    inl, r0, c0 = cp.linear_estimate_cam_pose_pnp(p2h, p3h, k, cfg)
    print("  linear pnp ransac: %d inliers (%.1fs)" % (len(inl), time.time() - t0))
    inl = np.asarray(inl)
    out = dict(K=k, pts2d=p2h, pts3d=p3h, inliers=inl, R0=r0, C0=c0,
               rot_truth=rot_truth, loc_truth=loc_truth)
    for its in ((1, 2, 10, 200) if slow else (1, 2, 10)):
        t0 = time.time()
        r, c = cp.nonlinear_estimate_cam_pose_pnp(p2h[:, inl], p3h[:, inl], k, r0, c0, 5, its)
        out["R_its%d" % its] = r; out["C_its%d" % its] = np.array(c)
        print("  nonlinear pnp its=%d (%.1fs) C=%s" % (its, time.time() - t0, np.array(c).T))
    # synthetic views: perturbed pose, exact + noisy projections
    sc = scenes.make_scene(4, 120, 1.0, seed=7)
    for c in range(1, 4):
        sel = sc.cam_idx == c
        x = np.vstack((sc.pts_true[:, sc.pt_idx[sel]], np.ones((1, sel.sum()))))
        uv = np.vstack((sc.uv_pix[:, sel], np.ones((1, sel.sum()))))
        ri = ref_utils.convert_quaternion_to_rotation(sc.cams_init[c, 3:7].reshape(4, 1))
        ci = sc.cams_init[c, 0:3].reshape(3, 1)
        r, cc = cp.nonlinear_estimate_cam_pose_pnp(uv, x, sc.intrinsic, ri, ci, 5, 25)
        out["syn%d_uv" % c] = uv; out["syn%d_X" % c] = x
        out["syn%d_R0" % c] = ri; out["syn%d_C0" % c] = ci
        out["syn%d_R" % c] = r; out["syn%d_C" % c] = np.array(cc)
    out["syn_K"] = sc.intrinsic
    np.savez_compressed(os.path.join(OUT, "g5_pnp.npz"), **out)


# ---------------------------------------------------------------------------------------
class _KP:                       # stands in for cv2.KeyPoint (only .pt is read, ba:339)
    def __init__(self, x, y):
        self.pt = (x, y)


class _View:                     # fields BA touches: ba:287-288, 318, 340, 413
    def __init__(self, rot, loc, k, kps):
        self.rot, self.loc, self.k, self.key_pts = rot, loc, k, kps

    def update_cam_pose(self, rot, loc):
        self.rot, self.loc = rot, loc


class _VP:
    pass


class _KT:
    is_visible = ref_kt.KeyTracker.is_visible          # the real visibility oracle

    def __init__(self, tracks):
        self.track_list = tracks


def run_reference_ba(sc, iteration, damping):
    """Drive BaProcessor.__execute_bundle_adjustment on a Scene; returns (cams (V,7), pts (3,N))."""
    nv = sc.n_cams
    views, tracks = [], []
    for c in range(nv):
        sel = np.flatnonzero(sc.cam_idx == c)
        kps = [_KP(-1.0, -1.0)] + [_KP(float(sc.uv_pix[0, o]), float(sc.uv_pix[1, o])) for o in sel]
        tr = ref_kt.KeyTrack(nv, len(kps), c)
        tr.table[c, 1:] = sc.pt_idx[sel]                # key 0 is a dummy (Q3: index 0 is never visible)
        rot = ref_utils.convert_quaternion_to_rotation(sc.cams_init[c, 3:7].reshape(4, 1))
        views.append(_View(rot, sc.cams_init[c, 0:3].reshape(3, 1).copy(), sc.intrinsic.copy(), kps))
        tracks.append(tr)
    vp = _VP(); vp.view_list = views
    tp = ref_tri.TriangulationProcessor()
    tp.tri_pts = np.vstack((sc.pts_init, np.ones((1, sc.n_pts))))
    cp = ref_cam.CamposeProcessor(None, 5, 300)
    bp = ref_ba.BaProcessor(vp, _KT(tracks), None, tp, cp, iteration=iteration, damping_factor=damping)
    quiet(bp._BaProcessor__execute_bundle_adjustment)
    cams = np.empty((nv, 7))
    for c in range(nv):
        cams[c, 0:3] = views[c].loc[:, 0]
        cams[c, 3:7] = ref_utils.convert_rotation_to_quaternion(views[c].rot)[:, 0]
    return cams, tp.tri_pts[0:3].copy(), views


def reference_linearisation(sc):
    """Per-observation r/Jp/Jx and the reduced system at the initial estimate, assembled from
    the reference's PUBLIC construct_jacobian_matrix functions exactly as ba:317-382 does."""
    cp = ref_cam.CamposeProcessor(None, 5, 300)
    tp = ref_tri.TriangulationProcessor()
    nv, npt, m = sc.n_cams, sc.n_pts, sc.n_obs
    lam = 5
    j_p = np.zeros((2 * m, 7 * nv)); j_x = np.zeros((2 * m, 3 * npt))
    bf = np.zeros((2 * m, 1))
    jps, jxs = np.zeros((m, 2, 7)), np.zeros((m, 2, 3))
    for o in range(m):
        c, p = sc.cam_idx[o], sc.pt_idx[o]
        rot = ref_utils.convert_quaternion_to_rotation(sc.cams_init[c, 3:7].reshape(4, 1))
        loc = sc.cams_init[c, 0:3].reshape(3, 1)
        x4 = np.append(sc.pts_init[:, p:p + 1], [[1.0]], axis=0)
        proj = np.hstack((rot.T, rot.T @ -loc))
        jp = cp.construct_jacobian_matrix(rot, loc, x4)
        jx = tp.construct_jacobian_matrix(x4, [proj], 1)
        key = np.array([[sc.uv_pix[0, o], sc.uv_pix[1, o], 1.0]]).T
        key = np.linalg.inv(sc.intrinsic) @ key
        key /= key[2]
        f = proj @ x4
        f /= f[2]
        j_p[2 * o:2 * o + 2, 7 * c:7 * c + 7] = jp
        j_x[2 * o:2 * o + 2, 3 * p:3 * p + 3] = jx
        bf[2 * o:2 * o + 2, 0] = key[0:2, 0] - f[0:2, 0]
        jps[o], jxs[o] = jp, jx
    d_inv = np.zeros((3 * npt, 3 * npt))
    for p in range(npt):
        blk = j_x[:, 3 * p:3 * p + 3]
        d_inv[3 * p:3 * p + 3, 3 * p:3 * p + 3] = np.linalg.inv(blk.T @ blk + lam * np.eye(3))
    ep = j_p.T @ bf; ex = j_x.T @ bf
    a = j_p.T @ j_p + lam * np.eye(7 * nv)
    b = j_p.T @ j_x
    s = a - b @ d_inv @ b.T
    rhs = ep - b @ d_inv @ ex
    delta_p = np.linalg.inv(s) @ rhs
    delta_x = d_inv @ (ex - b.T @ delta_p)
    return dict(lin_r=bf.reshape(m, 2), lin_Jp=jps, lin_Jx=jxs, lin_S=s, lin_rhs=rhs[:, 0],
                lin_delta_p=delta_p[:, 0], lin_delta_x=delta_x[:, 0])


def g6_ba(slow):
    cases = [("3x50", 3, 50, 1.0, 11), ("5x200v80", 5, 200, 0.8, 12), ("6x120v60", 6, 120, 0.6, 13)]
    if slow:
        cases.append(("8x300v50", 8, 300, 0.5, 14))
    for name, nv, npt, vis, seed in cases:
        sc = scenes.make_scene(nv, npt, vis, seed=seed)
        out = dict(K=sc.intrinsic, cams_init=sc.cams_init, pts_init=sc.pts_init, pt_ptr=sc.pt_ptr,
                   cam_idx=sc.cam_idx, pt_idx=sc.pt_idx, uv_pix=sc.uv_pix)
        for its in (1, 2, 3):
            t0 = time.time()
            cams, pts, _ = run_reference_ba(sc, its, 5)
            out["cams_it%d" % its] = cams; out["pts_it%d" % its] = pts
            print("  BA %s its=%d: %.1fs rmse %.6f" % (name, its, time.time() - t0,
                                                      scenes.reprojection_rmse(cams, pts, sc)))
        out["rmse_init"] = scenes.reprojection_rmse(sc.cams_init, sc.pts_init, sc)
        out["rmse_it3"] = scenes.reprojection_rmse(out["cams_it3"], out["pts_it3"], sc)
        if npt <= 200:
            out.update(reference_linearisation(sc))
        np.savez_compressed(os.path.join(OUT, "g6_ba_%s.npz" % name), **out)


def g6_ba_c2():
    """BASELINE config 2 (5 x 2000 dense): ~5 min and ~6 GB in the reference."""
    sc = scenes.make_config("C2", seed=0)
    t0 = time.time()
    cams, pts, _ = run_reference_ba(sc, 3, 5)
    dt = time.time() - t0
    print("  BA C2 its=3: %.1fs" % dt)
    np.savez_compressed(os.path.join(OUT, "g6_ba_C2.npz"), K=sc.intrinsic, cams_init=sc.cams_init,
                        pts_init=sc.pts_init, pt_ptr=sc.pt_ptr, cam_idx=sc.cam_idx, pt_idx=sc.pt_idx,
                        uv_pix=sc.uv_pix, cams_it3=cams, pts_it3=pts, ref_seconds=dt,
                        rmse_init=scenes.reprojection_rmse(sc.cams_init, sc.pts_init, sc),
                        rmse_it3=scenes.reprojection_rmse(cams, pts, sc))


def g7_visible(rng):
    """KeyTrack tables with the Q3 corner cases -> the triples the reference's loop visits."""
    nv, npt = 4, 12
    tables, triples = [], []
    tracks = []
    for c in range(nv):
        nk = 10 + c
        tr = ref_kt.KeyTrack(nv, nk, c)
        ids = rng.permutation(npt)[:6]
        pos = rng.permutation(nk)[:6]
        tr.table[c, pos] = ids
        if c == 1:
            tr.table[c, :] = -1
            tr.table[c, 0] = 3            # only match is key index 0 -> invisible (Q3)
            tr.table[c, 4] = 5
        if c == 2:
            tr.table[c, :] = -1
            tr.table[c, 0] = 7            # key 0 AND key 6 match point 7 -> visible, returns 0
            tr.table[c, 6] = 7
            tr.table[c, 2] = 9            # duplicates: first (smallest) index wins
            tr.table[c, 8] = 9
        tracks.append(tr)
        tables.append(tr.table[c, :].copy())
    kt = _KT(tracks)
    for t in range(npt):
        for c in range(nv):
            k = kt.is_visible(c, t)
            if k != -1:
                triples.append((c, t, int(k)))
    maxk = max(len(t) for t in tables)
    pad = np.full((nv, maxk), -2, dtype=np.int64)
    for c, t in enumerate(tables):
        pad[c, :len(t)] = t
    np.savez_compressed(os.path.join(OUT, "g7_visible.npz"), rows=pad,
                        row_len=np.array([len(t) for t in tables]), n_pts=npt,
                        triples=np.array(triples, dtype=np.int64))


def _text_points(path):
    """The (x, y) rows of the reference's epipolar_set text files (first line = count)."""
    with open(path) as f:
        rows = [ln.split() for ln in f.read().strip().splitlines()]
    rows = [r for r in rows if len(r) >= 2]
    return np.array([[float(r[0]), float(r[1])] for r in rows])


def g8_fundamental():
    """Eight-point fundamental matrix + RANSAC (epipolar_processor.py:22-267)."""
    import epipolar_processor as ref_epi
    out = {}
    # (i) the literal 8 pairs of the reference's unit test I (epipolar:283-290)
    lit = np.array([[580, 2362, 492, 1803], [2050, 2097, 1381, 1956], [2558, 2174, 1544, 2115],
                    [1395, 1970, 1166, 1752], [2490, 3003, 466, 2440], [3368, 1622, 3320, 2011],
                    [2183, 1500, 2471, 1621], [1972, 1775, 1674, 1736]], dtype=float)
    lp, rp = ref_utils.KeyPt(8), ref_utils.KeyPt(8)
    lp[0:2, :] = lit[:, 0:2].T
    rp[0:2, :] = lit[:, 2:4].T
    cfg = quiet(ref_utils.RansacConfig, 1e-3, 0.99, 0.75, 8, 200)
    ep = ref_epi.EpipolarProcessor(cfg)
    inl = ep.determine_fundamental_mat([lp, rp], cfg)
    out["lit_pairs"], out["lit_fund"], out["lit_inliers"] = lit, ep.fund_mat.copy(), np.array(inl)

    # (ii) RANSAC runs: epipolar_set text points (unit test II: threshold 1, 300 its) and the opencv
    # two-view pixel pairs with the demo's configuration (ba_processor.py:470-475: 1e-3, 300 its)
    d = os.path.join(REF, "test_dataset")
    p1 = _text_points(d + "/epipolar_set/pt_2D_1.txt")
    p2 = _text_points(d + "/epipolar_set/pt_2D_2.txt")
    k, _, _, kp1, kp2 = opencv_two_view()
    for tag, left, right, thr, its in (("eps", p1.T, p2.T, 1.0, 300), ("ocv", np.asarray(kp1), np.asarray(kp2), 1e-3, 300)):
        cfg = quiet(ref_utils.RansacConfig, thr, 0.99, 0.75, 8, its)
        n = left.shape[1]
        random.seed(1)
        samples = np.array([random.sample(range(n), 8) for _ in range(cfg.iteration)])
        ep = ref_epi.EpipolarProcessor(cfg)
        pairs_norm, tl, tr = ep._EpipolarProcessor__normalize([left, right])
        f_hyp = np.array([ep._EpipolarProcessor__estimate_eight_pts(pairs_norm[list(s), :]) for s in samples[:24]])
        random.seed(1)
        inl = ep.determine_fundamental_mat([left, right], cfg)
        out.update({tag + "_left": left[0:2].copy(), tag + "_right": right[0:2].copy(), tag + "_threshold": thr,
                    tag + "_samples": samples, tag + "_pairs_norm": pairs_norm, tag + "_tl": tl, tag + "_tr": tr,
                    tag + "_f_hyp": f_hyp, tag + "_inliers": np.array(inl), tag + "_fund": ep.fund_mat.copy()})
        if tag == "ocv":
            ep.extract_essential_mat(k, k)
            out["ocv_K"], out["ocv_esse"] = k, ep.esse_mat.copy()
    np.savez_compressed(os.path.join(OUT, "g8_fundamental.npz"), **out)


def g9_two_view_pose():
    """Pose candidates from E, cheirality, disambiguation on the reference's own data files
    (campose_processor.py:29-189; its unit test at campose:822-947)."""
    d = os.path.join(REF, "test_dataset", "opencv")
    esse = np.load(d + "/ess_ess_mat.npy")
    k = np.load(d + "/ess_intrinsic_mat.npy")
    ref_r = np.load(d + "/ess_self_r.npy").T
    ref_c = np.load(d + "/ess_self_c.npy")
    ref_proj = k @ np.hstack((ref_r.T, ref_r.T @ -ref_c))
    cfg = quiet(ref_utils.RansacConfig, 8.0, 0.99, 0.75, 6, 300)
    cp = ref_cam.CamposeProcessor(cfg, 5, 200)
    r1, r2, c1, c2 = cp.extract_cam_pose_from_essential_mat(esse)
    projs = [k @ np.hstack((r.T, -r.T @ c)) for r, c in ((r1, c1), (r1, c2), (r2, c1), (r2, c2))]
    pts = [np.load(d + "/ess_points_3d_%s_result.npy" % t).T[0] for t in ("r1t1", "r1t2", "r2t1", "r2t2")]
    valid = [cp.evalulate_cam_pose_cheirality(ref_proj, projs[i], pts[i]) for i in range(4)]
    best, best_valid = cp.disambiguate_cam_pose_four(ref_proj, projs, pts)
    # the reference's own linear triangulation of the four candidates on the matched pixel pairs (ba:89-93)
    _, _, _, kp1, kp2 = opencv_two_view()
    tp = ref_tri.TriangulationProcessor()
    lin = [np.asarray(quiet(tp.linear_triangulate, [ref_proj, projs[i]], [kp1, kp2])) for i in range(4)]
    best_lin, best_lin_valid = cp.disambiguate_cam_pose_four(ref_proj, projs, lin)
    np.savez_compressed(os.path.join(OUT, "g9_two_view_pose.npz"), esse=esse, K=k, ref_proj=ref_proj,
                        r1=r1, r2=r2, c1=c1, c2=c2,
                        r1_truth=np.load(d + "/ess_r1.npy").T, r2_truth=np.load(d + "/ess_r2.npy").T,
                        c1_truth=np.load(d + "/ess_c1.npy"), c2_truth=np.load(d + "/ess_c2.npy"),
                        projs=np.array(projs), pts=np.array(pts), best=best, best_valid=np.array(best_valid),
                        valid_counts=np.array([len(v) for v in valid]),
                        valid1=np.array(valid[1]), left=np.asarray(kp1)[0:2], right=np.asarray(kp2)[0:2],
                        lin_counts=np.array([len(cp.evalulate_cam_pose_cheirality(ref_proj, projs[i], lin[i])) for i in range(4)]),
                        best_lin=best_lin, best_lin_valid=np.array(best_lin_valid), lin_best_pts=lin[best_lin])


def g5h_pnp_hypotheses():
    """Every hypothesis of the reference's seeded six-point RANSAC on its own PnP fixture (campose_processor.py:
    524-560 loop body): the six-point samples as Python's `random` draws them after RansacConfig seeds it, the
    pose `__estimate_six_pts` returns (campose:565-633), the inlier count of that pose, and whether the
    det(rot) < 0 branch (campose:629-631) fired -- quirk Q13: in that branch loc is negated although it does not
    depend on the sign of the null vector, so those hypotheses carry -C."""
    d = os.path.join(REF, "test_dataset", "opencv")
    k = np.load(d + "/ess_intrinsic_mat.npy")
    p3 = np.load(d + "/pnp_points_3d.npy").T
    p2 = np.load(d + "/pnp_points_2d.npy").T
    ones = np.ones((1, p3.shape[1]))
    p3h = np.vstack((p3, ones)); p2h = np.vstack((p2, ones))
    n = p2h.shape[1]
    cfg = quiet(ref_utils.RansacConfig, 8.0, 0.99, 0.75, 6, 300)       # seeds python random with -1
    cp = ref_cam.CamposeProcessor(cfg, 5, 200)
    six = getattr(cp, "_CamposeProcessor__estimate_six_pts")
    kinv = np.linalg.inv(k)
    samples, rots, locs, cnts, branch = [], [], [], [], []
    for _ in range(cfg.iteration):
        idx = random.sample(range(n), 6)                                  # campose:531
        k6 = kinv @ p2h[:, idx]
        rot, loc = six(k6, p3h[:, idx])
        # which way did campose:629-631 go?  Re-derive the un-negated rotation from the same LAPACK calls.
        w = np.zeros((12, 12))
        for i in range(6):
            x, y, z = k6[:, i]; xx = p3h[:, idx[i]]
            w[2 * i, 0:4] = z * xx; w[2 * i, 8:12] = -x * xx
            w[2 * i + 1, 4:8] = z * xx; w[2 * i + 1, 8:12] = -y * xx
        cam_mat = np.linalg.svd(w)[2].T[:, -1].reshape(3, 4)
        uu, ss, vvh = np.linalg.svd(cam_mat[:, 0:3])
        raw = (uu @ vvh).T
        fired = bool(np.linalg.det(raw) < 0)
        assert np.array_equal(rot, -raw if fired else raw)
        proj = k @ np.hstack((rot.T, rot.T @ -loc))
        q = proj @ p3h
        q = q / q[2]
        err = np.sqrt(np.sum((p2h - q) ** 2, axis=0))                     # campose:546-549
        samples.append(idx); rots.append(rot); locs.append(np.array(loc).reshape(3))
        cnts.append(int(np.sum(err < cfg.inlier_threshold))); branch.append(fired)
    best = int(np.argmax(cnts))     # first maximum = the reference's strict '>' update
    print("  %d hypotheses, %d took the det<0 branch (max %d inliers among them), best = #%d with %d inliers" % (
        len(cnts), int(np.sum(branch)), max([c for c, b in zip(cnts, branch) if b] or [0]), best, cnts[best]))
    np.savez_compressed(os.path.join(OUT, "g5_pnp_hypotheses.npz"), K=k, pts2d=p2h, pts3d=p3h,
                        samples=np.array(samples, dtype=np.int32), R=np.array(rots), C=np.array(locs),
                        counts=np.array(cnts, dtype=np.int32), branch=np.array(branch), threshold=cfg.inlier_threshold)


# ---------------------------------------------------------------------------------------
class _ChainView:                # fields the per-view chain touches: ba:191-267 (cam_proj for triangulate, update_cam_pose)
    def __init__(self, rot, loc, k, kps):
        self.k, self.key_pts = k, kps
        self.update_cam_pose(rot, loc)

    def update_cam_pose(self, rot, loc):     # view_processor.py:61-69
        self.rot, self.loc = rot, loc
        self.cam_pose = np.hstack((rot, loc))
        self.cam_proj = self.k @ np.hstack((rot.T, rot.T @ -loc))


def _rng_digest():
    import hashlib
    return hashlib.sha256(repr(random.getstate()).encode()).hexdigest()


def g10_incremental():
    """BASELINE config 5's call chain pinned by the REAL classes: per registered view
    CamposeProcessor.estimate_cam_pose_pnp (seeded RANSAC + nonlinear PnP, campose:192-246) ->
    View.update_cam_pose -> TriangulationProcessor.triangulate (tri:31-88) -> add_tri_pt ->
    BaProcessor.__execute_bundle_adjustment (ba:274-439), view after view on ONE stream of Python's global RNG
    (seeded by RansacConfig, utils.py:172-174).  Synthetic 6-view sequences (the upenn BMPs need SIFT), three seeds;
    8 % of every view's keys of already-known points are gross outliers so that the inlier lists are not trivial.
    Stored per view: the inlier list, the PnP pose, the new points, the state after BA, the RNG digest, and -- from an
    instrumented replay of the RANSAC loop with the RNG state restored afterwards -- every hypothesis' six-point
    sample, whether campose:629-631 (quirk Q13) fired for it, and its inlier count."""
    for seed in (61, 62, 63):
        n_views, n_pts = 6, 150
        sc = scenes.make_scene(n_views, n_pts, 1.0, seed=seed, pixel_noise=0.3)
        K = sc.intrinsic
        rng = np.random.default_rng(1000 + seed)
        per = (n_pts + n_views - 2) // (n_views - 1)
        birth = 1 + np.arange(n_pts) // per
        uv = []
        for c in range(n_views):
            u = np.vstack((sc.uv_pix[:, sc.cam_idx == c], np.ones((1, n_pts))))
            known_before = np.flatnonzero(birth < c)          # the PnP input set of view c; never used to triangulate from c
            if known_before.size:
                bad = rng.choice(known_before, size=max(1, int(0.08 * known_before.size)), replace=False)
                u[0:2, bad] += rng.normal(0, 30.0, (2, bad.size))
            uv.append(u)
        from scipy.spatial.transform import Rotation
        rots = [ref_utils.convert_quaternion_to_rotation(sc.cams_true[c, 3:7].reshape(4, 1)) for c in range(n_views)]
        locs = [sc.cams_true[c, 0:3].reshape(3, 1).copy() for c in range(n_views)]
        rot1 = rots[1] @ Rotation.from_rotvec(rng.normal(0, 0.002, 3)).as_matrix()      # second view: "two-view initialisation"
        loc1 = locs[1] + rng.normal(0, 0.01, (3, 1))

        cfg = quiet(ref_utils.RansacConfig, 8.0, 0.99, 0.75, 6, 300)                     # seeds Python's RNG with -1
        cp = ref_cam.CamposeProcessor(cfg, 5, 300)                                       # ba_processor.py:486
        tp = ref_tri.TriangulationProcessor()                                            # 0.5, 100
        vp = _VP(); vp.view_list = []
        tracks = []
        kt = _KT(tracks)
        bp = ref_ba.BaProcessor(vp, kt, None, tp, cp, iteration=3, damping_factor=5)
        six = getattr(cp, "_CamposeProcessor__estimate_six_pts")

        def add_view(c, rot, loc):
            kps = [_KP(-1.0, -1.0)] + [_KP(float(uv[c][0, j]), float(uv[c][1, j])) for j in range(n_pts)]
            vp.view_list.append(_ChainView(rot, loc, K.copy(), kps))
            tracks.append(ref_kt.KeyTrack(n_views, n_pts + 1, c))

        out = dict(K=K, n_views=n_views, n_pts=n_pts, birth=birth, uv=np.array(uv), rot0=rots[0], loc0=locs[0], rot1=rot1, loc1=loc1,
                   ransac=np.array([8.0, 0.99, 0.75, 6, 300]), pnp=np.array([5.0, 300.0]), tri=np.array([0.5, 100.0]), ba=np.array([5.0, 3.0]),
                   rng_digest_start=np.array(_rng_digest()))
        add_view(0, rots[0].copy(), locs[0].copy())
        known = np.zeros(n_pts, dtype=bool)
        t_seed = time.time()
        for c in range(1, n_views):
            pre = "v%d_" % c
            if known.any():
                idx = np.flatnonzero(known)
                key2d, tri3d = uv[c][:, idx], tp.tri_pts[:, idx]
                # instrumented replay of the RANSAC loop (campose:524-560), RNG state restored afterwards
                state = random.getstate()
                kinv = np.linalg.inv(K)
                samples, fired, counts = [], [], []
                for _ in range(cfg.iteration):
                    s6 = random.sample(range(idx.size), 6)
                    k6 = kinv @ key2d[:, s6]
                    rot, loc = six(k6, tri3d[:, s6])
                    w = np.zeros((12, 12))
                    for i in range(6):
                        x, y, z = k6[:, i]; xx = tri3d[:, s6[i]]
                        w[2 * i, 0:4] = z * xx; w[2 * i, 8:12] = -x * xx
                        w[2 * i + 1, 4:8] = z * xx; w[2 * i + 1, 8:12] = -y * xx
                    cam_mat = np.linalg.svd(w)[2].T[:, -1].reshape(3, 4)
                    uu, ss, vvh = np.linalg.svd(cam_mat[:, 0:3])
                    raw = (uu @ vvh).T
                    f = bool(np.linalg.det(raw) < 0)
                    assert np.array_equal(rot, -raw if f else raw)
                    proj = K @ np.hstack((rot.T, rot.T @ -loc))
                    q = proj @ tri3d
                    q = q / q[2]
                    err = np.sqrt(np.sum((key2d - q) ** 2, axis=0))
                    samples.append(s6); fired.append(f); counts.append(int(np.sum(err < cfg.inlier_threshold)))
                random.setstate(state)
                out[pre + "rng_before_pnp"] = np.array(_rng_digest())
                inl, r_new, c_new = quiet(cp.estimate_cam_pose_pnp, key2d, tri3d, K)          # ba_processor.py:191
                out[pre + "hyp_samples"] = np.array(samples, dtype=np.int32)
                out[pre + "hyp_fired"] = np.array(fired)
                out[pre + "hyp_counts"] = np.array(counts, dtype=np.int32)
                out[pre + "pnp_index"] = idx.astype(np.int32)
                out[pre + "inliers"] = np.array(inl, dtype=np.int32)
                out[pre + "pnp_rot"] = np.array(r_new); out[pre + "pnp_loc"] = np.array(c_new).reshape(3, 1)
                best = int(np.argmax(counts))
                print("    seed %d view %d: %d known, %d inliers, %d hypotheses fired Q13 (best fired count %d), winner #%d with %d" % (
                    seed, c, idx.size, len(inl), int(np.sum(fired)), max([n for n, f in zip(counts, fired) if f] or [0]), best, counts[best]))
                assert counts[best] == len(inl)
            else:
                r_new, c_new = rot1, loc1
            add_view(c, r_new, c_new)
            out[pre + "rng_after_pnp"] = np.array(_rng_digest())
            new = np.flatnonzero(birth == c)
            views = vp.view_list
            pts_new = quiet(tp.triangulate, [views[c - 1].cam_proj, views[c].cam_proj], [uv[c - 1][:, new], uv[c][:, new]])      # ba:246
            out[pre + "new_pts"] = np.array(pts_new)
            tp.add_tri_pt(np.array(pts_new))                                                                               # ba:262
            known[new] = True
            ids = np.flatnonzero(known)
            for v in range(c + 1):
                tracks[v].table[v, 1 + ids] = ids            # key index = point + 1 (key 0 is a dummy: Q3)
            quiet(bp._BaProcessor__execute_bundle_adjustment)                                                               # ba:267
            out[pre + "ba_rots"] = np.array([v.rot for v in views]); out[pre + "ba_locs"] = np.array([v.loc for v in views])
            out[pre + "ba_pts"] = np.array(tp.tri_pts)
        print("  g10 seed %d: %.1fs" % (seed, time.time() - t_seed))
        np.savez_compressed(os.path.join(OUT, "g10_incremental_%d.npz" % seed), **out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-slow", action="store_true")
    ap.add_argument("--only", default="")
    ap.add_argument("--c2", action="store_true", help="also capture BASELINE config 2 (5 min, 6 GB)")
    args = ap.parse_args()
    os.makedirs(OUT, exist_ok=True)
    slow = not args.skip_slow
    rng = np.random.default_rng(20261004)
    random.seed(1)
    steps = [("g1", lambda: g1_jac_cam(rng)), ("g2", lambda: g2_jac_pt(rng)), ("g3", lambda: g3_quat(rng)),
             ("g4", lambda: g4_tri(rng, slow)), ("g5", lambda: g5_pnp(rng, slow)), ("g6", lambda: g6_ba(slow)),
             ("g5h", g5h_pnp_hypotheses),
             ("g7", lambda: g7_visible(rng)), ("g8", g8_fundamental), ("g9", g9_two_view_pose), ("g10", g10_incremental)]
    for name, fn in steps:
        if args.only and name not in args.only.split(","):
            continue
        t0 = time.time()
        fn()
        print("%s done in %.1fs" % (name, time.time() - t0))
    if args.c2:
        g6_ba_c2()


if __name__ == "__main__":
    main()
