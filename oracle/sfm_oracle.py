"""CPU ORACLE — TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's nonlinear-refinement hot path
(willSapgreen/structure-from-motion), used only as the *checker*: by ``tests/``, by
``__graft_entry__.smoke()`` and by the ``cpu_baseline`` leg of ``bench.py``.  Nothing
under ``structure-from-motion_amd/`` may import this module; the product path is the HIP
library and fails loudly without it.

Pinning: every function here is checked (tests/test_oracle_golden.py) against golden
vectors captured by importing the real reference in the build container
(tools/capture_goldens.py -> tests/golden/*.npz) and against the reference's own
known-answer values (triangulation_processor.py:415-473, campose_processor.py:1073-1090,
campose_processor.py:934; the two-view functions by g8/g9 from epipolar_processor.py and
campose_processor.py:29-189 run on the reference's own data files).
The reference has no test that pins bundle adjustment; BA is pinned by the captured
goldens only.

Each function cites the reference file:line it follows.  Conventions: rot = R, loc = C,
world->camera p = R^T (X - C); quaternion [qw,qx,qy,qz]; camera block [C, q] (7 doubles);
points homogeneous (4, m) with W carried as given; everything float64.
"""
import math

import numpy as np

# quirk bits: default = reproduce the reference
Q1_PNP_ROW_OVERLAP = 1   # campose_processor.py:404-405
Q2_LOC_JAC_SIGN = 2      # campose_processor.py:802-804
QUIRKS_REFERENCE = Q1_PNP_ROW_OVERLAP | Q2_LOC_JAC_SIGN


# --------------------------------------------------------------------------------------
# utils.py:28-105
# --------------------------------------------------------------------------------------
def verify_rotation(rot):
    """utils.py:101-105 (one-sided tests, as written)."""
    rot = np.asarray(rot)
    return not (rot.shape != (3, 3) or np.linalg.det(rot) - 1 >= 1e-8
                or np.any((np.linalg.inv(rot) - rot.T) > 1e-8))


def rot_to_quat(rot):
    """utils.py:28-60: canonical qw >= 0 quaternion, (4,) vector."""
    if not verify_rotation(rot):
        raise ValueError("convert_rotation_to_quaternion : Invalid input rotation matrix")
    qw = math.sqrt(1 + rot[0][0] + rot[1][1] + rot[2][2]) / 2.0
    if abs(qw - 0) < 1e-6:
        raise ValueError("convert_rotation_to_quaternion : Invalid output qw")
    return np.array([qw,
                     (rot[2][1] - rot[1][2]) / (4 * qw),
                     (rot[0][2] - rot[2][0]) / (4 * qw),
                     (rot[1][0] - rot[0][1]) / (4 * qw)])


def quat_to_rot_unchecked(q):
    """utils.py:83-91."""
    w, x, y, z = (float(t) for t in np.asarray(q).reshape(4))
    r = np.zeros((3, 3))
    r[0][0] = 1 - 2 * z * z - 2 * y * y
    r[0][1] = -2 * z * w + 2 * y * x
    r[0][2] = 2 * y * w + 2 * z * x
    r[1][0] = 2 * x * y + 2 * w * z
    r[1][1] = 1 - 2 * z * z - 2 * x * x
    r[1][2] = 2 * z * y - 2 * x * w
    r[2][0] = 2 * x * z - 2 * w * y
    r[2][1] = 2 * y * z + 2 * w * x
    r[2][2] = 1 - 2 * y * y - 2 * x * x
    return r


def quat_to_rot(q):
    """utils.py:64-97 (validated)."""
    r = quat_to_rot_unchecked(q)
    if not verify_rotation(r):
        raise ValueError("convert_quaternion_to_rotation : Invalid output rotation matrix")
    return r


# --------------------------------------------------------------------------------------
# campose_processor.py:462-482, 636-808  (Jp = d proj / d (C, q), 2x7)
# --------------------------------------------------------------------------------------
def jac_quat(q):
    """campose_processor.py:636-702: d R(row-major) / d (w,x,y,z), 9x4."""
    w, x, y, z = (float(t) for t in np.asarray(q).reshape(4))
    w2, x2, y2, z2 = 2 * w, 2 * x, 2 * y, 2 * z
    x4, y4, z4 = 4 * x, 4 * y, 4 * z
    return np.array([[0, 0, -y4, -z4],
                     [-z2, y2, x2, -w2],
                     [y2, z2, w2, x2],
                     [z2, y2, x2, w2],
                     [0, -x4, 0, -z4],
                     [-x2, -w2, z2, y2],
                     [-y2, z2, -w2, x2],
                     [x2, w2, z2, y2],
                     [0, -x4, -y4, 0]], dtype=np.float64)


def jac_cam(rot, loc, pt_h, quirks=QUIRKS_REFERENCE):
    """campose_processor.py:462-482: hstack(J_C, J_R @ J_q) for one (R, C, X~)."""
    rot = np.asarray(rot, dtype=np.float64)
    loc = np.asarray(loc, dtype=np.float64).reshape(3, 1)
    pt_h = np.asarray(pt_h, dtype=np.float64).reshape(4, 1)
    qhat = rot_to_quat(rot)                                  # campose:464 (re-derived from R, Q7)
    proj = np.hstack((rot.T, rot.T @ -loc))                  # campose:727
    p = (proj @ pt_h).reshape(3)
    px, py, pz = p
    d = (pt_h[0:3] - loc).reshape(3)                         # campose:735
    jr = np.zeros((2, 9))
    for i in range(3):                                       # campose:742-764
        jr[0, 3 * i] = pz * d[i]
        jr[0, 3 * i + 2] = -px * d[i]
        jr[1, 3 * i + 1] = pz * d[i]
        jr[1, 3 * i + 2] = -py * d[i]
    jr /= pz * pz
    jc = np.zeros((2, 3))
    sgn = 1.0 if (quirks & Q2_LOC_JAC_SIGN) else -1.0
    for i in range(3):                                       # campose:798-804
        jc[0, i] = pz * -rot[i][0] - px * -rot[i][2]
        jc[1, i] = pz * -rot[i][1] - py * (sgn * rot[i][2])
    jc /= pz * pz
    return np.hstack((jc, jr @ jac_quat(qhat)))


# --------------------------------------------------------------------------------------
# triangulation_processor.py:237-309
# --------------------------------------------------------------------------------------
def jac_pt(pt_h, projs):
    """triangulation_processor.py:237-271: (2*len(projs), 3)."""
    pt_h = np.asarray(pt_h, dtype=np.float64).reshape(4)
    jac = np.zeros((2 * len(projs), 3))
    for v, proj in enumerate(projs):
        s = proj @ pt_h
        jac[2 * v] = (s[2] * proj[0, 0:3] - s[0] * proj[2, 0:3]) / s[2] ** 2
        jac[2 * v + 1] = (s[2] * proj[1, 0:3] - s[1] * proj[2, 0:3]) / s[2] ** 2
    return jac


def reproj_error(pt_h, projs, uvs):
    """triangulation_processor.py:274-309: e = f - b (Q10), shape (2V,)."""
    pt_h = np.asarray(pt_h, dtype=np.float64).reshape(4)
    err = np.zeros(2 * len(projs))
    for v, proj in enumerate(projs):
        s = proj @ pt_h
        s = s / s[2]
        err[2 * v] = s[0] - uvs[v][0]
        err[2 * v + 1] = s[1] - uvs[v][1]
    return err


def nonlinear_triangulate(init_3d_pts, projs, matched_pairs, damping_factor, iteration):
    """triangulation_processor.py:160-234.  Per point, per iteration:
    delta = inv(J^T J + lambda I3) J^T e ; X[0:3] -= delta.  Row W is carried unchanged."""
    out = np.array(init_3d_pts, dtype=np.float64, copy=True)
    projs = [np.asarray(p, dtype=np.float64) for p in projs]
    m = matched_pairs[0].shape[1]
    for p in range(m):
        x = out[:, p].copy()
        uvs = [(mp[0, p], mp[1, p]) for mp in matched_pairs]
        for _ in range(iteration):
            err = reproj_error(x, projs, uvs)
            jac = jac_pt(x, projs)
            delta = np.linalg.inv(jac.T.dot(jac) + damping_factor * np.identity(3)).dot(jac.T).dot(err)
            x[0:3] -= delta
        out[:, p] = x
    return out


def nonlinear_triangulate_vec(init_3d_pts, projs, matched_pairs, damping_factor, iteration):
    """Vectorised-over-points form of the same arithmetic (used at sizes where the loop form
    is too slow); checked against nonlinear_triangulate in the tests."""
    x = np.array(init_3d_pts, dtype=np.float64, copy=True)
    projs = [np.asarray(p, dtype=np.float64) for p in projs]
    m = x.shape[1]
    for _ in range(iteration):
        jtj = np.zeros((m, 3, 3))
        jte = np.zeros((m, 3))
        for v, proj in enumerate(projs):
            s = proj @ x                                         # (3,m)
            ju = (s[2] * proj[0, 0:3, None] - s[0] * proj[2, 0:3, None]) / s[2] ** 2   # (3,m)
            jv = (s[2] * proj[1, 0:3, None] - s[1] * proj[2, 0:3, None]) / s[2] ** 2
            eu = s[0] / s[2] - matched_pairs[v][0]
            ev = s[1] / s[2] - matched_pairs[v][1]
            jtj += np.einsum('im,jm->mij', ju, ju) + np.einsum('im,jm->mij', jv, jv)
            jte += (ju * eu + jv * ev).T
        jtj += damping_factor * np.eye(3)
        delta = np.linalg.solve(jtj, jte[:, :, None])[:, :, 0]
        x[0:3] -= delta.T
    return x


# --------------------------------------------------------------------------------------
# campose_processor.py:308-459
# --------------------------------------------------------------------------------------
def nonlinear_pnp(key_2d_pts, tri_3d_pts, intrinsic, init_rot, init_loc,
                  damping_factor, iteration, quirks=QUIRKS_REFERENCE, trace=None):
    """campose_processor.py:308-459.  Returns (R (3,3), C (3,1)).

    Q1 (default on): rows are stored at [pt : pt+2] so the effective system is the u-row of
    every point, the v-row of the LAST point, and zero rows (campose:404-405)."""
    key_2d_pts = np.asarray(key_2d_pts, dtype=np.float64)
    tri_3d_pts = np.asarray(tri_3d_pts, dtype=np.float64)
    n = key_2d_pts.shape[1]
    if n != tri_3d_pts.shape[1]:
        raise ValueError("key pts num - triangulated pts num : {} - {}".format(n, tri_3d_pts.shape[1]))
    q0 = rot_to_quat(init_rot)
    q0 = q0 / math.sqrt(np.sum(np.square(q0)))                   # campose:361-363
    params = np.concatenate((np.asarray(init_loc, dtype=np.float64).reshape(3), q0)).reshape(7, 1)
    rot = np.array(init_rot, dtype=np.float64, copy=True)        # campose:367 (R0 itself, not R(q0))
    loc = np.array(init_loc, dtype=np.float64, copy=True).reshape(3, 1)
    kinv = np.linalg.inv(np.asarray(intrinsic, dtype=np.float64))
    stride = 1 if (quirks & Q1_PNP_ROW_OVERLAP) else 2
    for it in range(iteration):
        jac_all = np.zeros((2 * n, 7))
        err_all = np.zeros((2 * n, 1))
        proj = np.hstack((rot.T, rot.T @ -loc))
        for p in range(n):
            x3 = tri_3d_pts[:, p:p + 1]
            f = proj @ x3
            f = f / f[2]
            mcam = kinv @ key_2d_pts[:, p:p + 1]
            mcam = mcam / mcam[2]
            jac_all[stride * p:stride * p + 2] = jac_cam(rot, loc, x3, quirks)
            err_all[stride * p:stride * p + 2] = (mcam - f)[0:2]
        delta = np.linalg.inv(jac_all.T.dot(jac_all) + damping_factor * np.identity(7)).dot(jac_all.T).dot(err_all)
        params = params + delta
        qn = params[3:7, 0] / math.sqrt(np.sum(np.square(params[3:7, 0])))
        params[3:7, 0] = qn
        loc = params[0:3, 0:1].copy()
        rot = quat_to_rot(qn)
        if trace is not None:
            trace.append(params[:, 0].copy())
    return quat_to_rot(params[3:7, 0]), params[0:3, 0:1].copy()


# --------------------------------------------------------------------------------------
# key_tracker.py:198-204  (visibility oracle -> observation list)
# --------------------------------------------------------------------------------------
def is_visible(table_row, tri_idx):
    """key_tracker.py:198-204 incl. Q3: a point matched only by key index 0 is invisible;
    with several matches the first index is returned."""
    key_idx = np.where(np.asarray(table_row) == tri_idx)
    if np.any(key_idx):
        return int(key_idx[0][0])
    return -1


def observation_list(self_rows, n_pts):
    """All (cam, pt, key) triples the reference's BA loop would visit, in its loop order
    (point-major, then view; ba_processor.py:304-310).  ``self_rows[c]`` is
    ``track_list[c].table[c, :]``."""
    cams, pts, keys = [], [], []
    for t in range(n_pts):
        for c, row in enumerate(self_rows):
            k = is_visible(row, t)
            if k != -1:
                cams.append(c)
                pts.append(t)
                keys.append(k)
    return (np.asarray(cams, dtype=np.int32), np.asarray(pts, dtype=np.int32),
            np.asarray(keys, dtype=np.int32))


# --------------------------------------------------------------------------------------
# ba_processor.py:274-439
# --------------------------------------------------------------------------------------
def _obs_terms(cams, pts, cam_idx, pt_idx, uv_norm, quirks):
    """Per observation r (2), Jp (2x7), Jx (2x3) exactly as ba_processor.py:317-349 builds
    them (loop form; small problems)."""
    m = cam_idx.shape[0]
    r = np.zeros((m, 2))
    jp = np.zeros((m, 2, 7))
    jx = np.zeros((m, 2, 3))
    rots = [quat_to_rot(cams[c, 3:7]) for c in range(cams.shape[0])]         # ba:323 (validated)
    for o in range(m):
        c, p = cam_idx[o], pt_idx[o]
        rot = rots[c]
        loc = cams[c, 0:3].reshape(3, 1)
        x4 = np.array([pts[0, p], pts[1, p], pts[2, p], 1.0]).reshape(4, 1)  # ba:317
        proj = np.hstack((rot.T, rot.T @ -loc))                              # ba:328
        jp[o] = jac_cam(rot, loc, x4, quirks)                                # ba:332
        jx[o] = jac_pt(x4, [proj])                                           # ba:333
        f = proj @ x4
        f = f / f[2]
        r[o] = uv_norm[:, o] - f[0:2, 0]                                     # b - f (ba:376)
    return r, jp, jx


def ba_dense(cams, pts, cam_idx, pt_idx, uv_norm, damping_factor, iteration,
             quirks=QUIRKS_REFERENCE, trace=None):
    """Line-faithful dense restatement of ba_processor.py:297-406 on an observation list
    sorted by (point, cam): dense j_p (2M x 7V), j_x (2M x 3N), block-diagonal d_inv,
    explicit inverses, identical expression order.  O(M*N) memory — small scenes only.

    cams (V,7) [C,q]; pts (3,N); uv_norm (2,M) = inv(K)[u,v,1] / z.  Returns new (cams, pts).
    """
    cams = np.array(cams, dtype=np.float64, copy=True).reshape(-1, 7)
    pts = np.array(pts, dtype=np.float64, copy=True)
    nv, npt, m = cams.shape[0], pts.shape[1], cam_idx.shape[0]
    lam = damping_factor
    for _ in range(iteration):
        r, jp, jx = _obs_terms(cams, pts, cam_idx, pt_idx, uv_norm, quirks)
        j_p = np.zeros((2 * m, 7 * nv))
        j_x = np.zeros((2 * m, 3 * npt))
        d = np.zeros((npt, 3, 3))
        for o in range(m):
            c, p = cam_idx[o], pt_idx[o]
            j_p[2 * o:2 * o + 2, 7 * c:7 * c + 7] = jp[o]
            j_x[2 * o:2 * o + 2, 3 * p:3 * p + 3] = jx[o]
            d[p] += jx[o].T @ jx[o]                                           # ba:355
        d_inv = np.zeros((3 * npt, 3 * npt))
        for p in range(npt):
            d_inv[3 * p:3 * p + 3, 3 * p:3 * p + 3] = np.linalg.inv(d[p] + lam * np.eye(3))   # ba:359-363
        bf = r.reshape(2 * m, 1)
        ep = j_p.T @ bf
        ex = j_x.T @ bf
        a = j_p.T @ j_p + lam * np.eye(7 * nv)
        b = j_p.T @ j_x
        delta_p = np.linalg.inv(a - b @ d_inv @ b.T) @ (ep - b @ d_inv @ ex)  # ba:382
        cams = cams + delta_p.reshape(nv, 7)
        for c in range(nv):                                                   # ba:388-392
            cams[c, 3:7] /= math.sqrt(np.sum(np.square(cams[c, 3:7])))
        delta_x = d_inv @ (ex - b.T @ delta_p)                                # ba:405
        pts = pts + delta_x.reshape(npt, 3).T
        if trace is not None:
            trace.append((cams.copy(), pts.copy()))
    for c in range(nv):
        quat_to_rot(cams[c, 3:7])                                             # ba:412 (validation)
    return cams, pts


def obs_terms_vec(cams, pts, cam_idx, pt_idx, uv_norm, quirks=QUIRKS_REFERENCE):
    """Vectorised r (M,2), Jp (M,2,7), Jx (M,2,3) — same expressions as _obs_terms
    (SURVEY.md Appendix A.2-A.4), evaluated for all observations at once."""
    nv = cams.shape[0]
    rots = np.stack([quat_to_rot(cams[c, 3:7]) for c in range(nv)])          # validated, (V,3,3)
    qhat = np.stack([rot_to_quat(rots[c]) for c in range(nv)])               # Q7
    jq = np.stack([jac_quat(qhat[c]) for c in range(nv)])                    # (V,9,4)
    tvec = np.stack([rots[c].T @ -cams[c, 0:3] for c in range(nv)])          # (V,3)
    rc = rots[cam_idx]                                                       # (M,3,3)
    xw = pts[:, pt_idx].T                                                    # (M,3)
    p = np.einsum('mji,mj->mi', rc, xw) + tvec[cam_idx]                      # R^T X + (R^T -C) * 1
    px, py, pz = p[:, 0], p[:, 1], p[:, 2]
    pz2 = pz * pz
    d = xw - cams[cam_idx, 0:3]
    m = cam_idx.shape[0]
    r = np.stack((uv_norm[0] - px / pz, uv_norm[1] - py / pz), axis=1)
    # Jx with the K-free projection P = [R^T | t]: rows (pz*P[0,:3] - px*P[2,:3]) / pz^2 ...
    rt = np.transpose(rc, (0, 2, 1))                                         # R^T per obs
    jx = np.empty((m, 2, 3))
    jx[:, 0, :] = (pz[:, None] * rt[:, 0, :] - px[:, None] * rt[:, 2, :]) / pz2[:, None]
    jx[:, 1, :] = (pz[:, None] * rt[:, 1, :] - py[:, None] * rt[:, 2, :]) / pz2[:, None]
    jr = np.zeros((m, 2, 9))
    for i in range(3):
        jr[:, 0, 3 * i] = pz * d[:, i]
        jr[:, 0, 3 * i + 2] = -px * d[:, i]
        jr[:, 1, 3 * i + 1] = pz * d[:, i]
        jr[:, 1, 3 * i + 2] = -py * d[:, i]
    jr /= pz2[:, None, None]
    sgn = 1.0 if (quirks & Q2_LOC_JAC_SIGN) else -1.0
    jc = np.empty((m, 2, 3))
    for i in range(3):
        jc[:, 0, i] = pz * -rc[:, i, 0] - px * -rc[:, i, 2]
        jc[:, 1, i] = pz * -rc[:, i, 1] - py * (sgn * rc[:, i, 2])
    jc /= pz2[:, None, None]
    jp = np.concatenate((jc, np.einsum('mij,mjk->mik', jr, jq[cam_idx])), axis=2)
    return r, jp, jx


def ba_reduced_system(cams, pts, cam_idx, pt_idx, uv_norm, damping_factor,
                      quirks=QUIRKS_REFERENCE):
    """One linearisation: returns dict with the Schur-reduced system and the pieces the
    kernels expose for unit parity:
    S = A - B D^-1 B^T (7V x 7V), rhs = ep - B D^-1 ex (7V), plus per-point D^-1, ex and per-obs W."""
    nv, npt = cams.shape[0], pts.shape[1]
    lam = damping_factor
    r, jp, jx = obs_terms_vec(cams, pts, cam_idx, pt_idx, uv_norm, quirks)
    u = np.zeros((nv, 7, 7))
    np.add.at(u, cam_idx, np.einsum('mki,mkj->mij', jp, jp))
    ep = np.zeros((nv, 7))
    np.add.at(ep, cam_idx, np.einsum('mki,mk->mi', jp, r))
    d = np.zeros((npt, 3, 3))
    np.add.at(d, pt_idx, np.einsum('mki,mkj->mij', jx, jx))
    d += lam * np.eye(3)
    ex = np.zeros((npt, 3))
    np.add.at(ex, pt_idx, np.einsum('mki,mk->mi', jx, r))
    d_inv = np.linalg.inv(d)
    w = np.einsum('mki,mkj->mij', jp, jx)                                    # (M,7,3) block B_{c,p}
    y = np.einsum('mij,mjk->mik', w, d_inv[pt_idx])                          # W D^-1
    s = np.zeros((7 * nv, 7 * nv))
    for c in range(nv):
        s[7 * c:7 * c + 7, 7 * c:7 * c + 7] = u[c] + lam * np.eye(7)
    # B D^-1 B^T: per point outer products over the cameras that see it.  Done as a dense
    # product of the (7V x 3N) matrices Y and W when that is small, else block-wise.
    if 7 * nv * 3 * npt <= 64_000_000:
        yd = np.zeros((7 * nv, 3 * npt))
        wd = np.zeros((7 * nv, 3 * npt))
        rows = (7 * cam_idx[:, None, None] + np.arange(7)[None, :, None])
        cols = (3 * pt_idx[:, None, None] + np.arange(3)[None, None, :])
        yd[rows, cols] = y
        wd[rows, cols] = w
        s -= yd @ wd.T
    else:
        chunk = max(1, 64_000_000 // (7 * nv * 3))
        order = np.argsort(pt_idx, kind='stable')
        ci, pi = cam_idx[order], pt_idx[order]
        ys, ws = y[order], w[order]
        bounds = np.searchsorted(pi, np.arange(0, npt + chunk, chunk))
        for b0, b1, p0 in zip(bounds[:-1], bounds[1:], range(0, npt, chunk)):
            if b1 == b0:
                continue
            npc = min(chunk, npt - p0)
            yd = np.zeros((7 * nv, 3 * npc))
            wd = np.zeros((7 * nv, 3 * npc))
            rows = (7 * ci[b0:b1, None, None] + np.arange(7)[None, :, None])
            cols = (3 * (pi[b0:b1, None, None] - p0) + np.arange(3)[None, None, :])
            yd[rows, cols] = ys[b0:b1]
            wd[rows, cols] = ws[b0:b1]
            s -= yd @ wd.T
    rhs = ep.copy()
    np.subtract.at(rhs, cam_idx, np.einsum('mij,mj->mi', y, ex[pt_idx]))
    return dict(S=s, rhs=rhs.reshape(7 * nv), U=u, ep=ep, D_inv=d_inv, ex=ex, W=w, Y=y,
                r=r, Jp=jp, Jx=jx)


def ba_sparse(cams, pts, cam_idx, pt_idx, uv_norm, damping_factor, iteration,
              quirks=QUIRKS_REFERENCE, trace=None):
    """Block-sparse restatement of ba_processor.py:297-406 (identical arithmetic per
    observation and per block, different summation order): the `cpu_ref_sparse` baseline of
    BASELINE.md section 4.  Returns new (cams (V,7), pts (3,N))."""
    cams = np.array(cams, dtype=np.float64, copy=True).reshape(-1, 7)
    pts = np.array(pts, dtype=np.float64, copy=True)
    nv = cams.shape[0]
    for _ in range(iteration):
        t = ba_reduced_system(cams, pts, cam_idx, pt_idx, uv_norm, damping_factor, quirks)
        delta_p = (np.linalg.inv(t["S"]) @ t["rhs"]).reshape(nv, 7)           # ba:382
        cams = cams + delta_p
        cams[:, 3:7] /= np.sqrt(np.sum(np.square(cams[:, 3:7]), axis=1))[:, None]
        # delta_x = D^-1 (ex - B^T delta_p)  (ba:405); B^T delta_p summed per point
        btd = np.zeros_like(t["ex"])
        np.add.at(btd, pt_idx, np.einsum('mij,mi->mj', t["W"], delta_p[cam_idx]))
        delta_x = np.einsum('pij,pj->pi', t["D_inv"], t["ex"] - btd)
        pts = pts + delta_x.T
        if trace is not None:
            trace.append((cams.copy(), pts.copy()))
    for c in range(nv):
        quat_to_rot(cams[c, 3:7])
    return cams, pts


def rmse_pixels(cams, pts, cam_idx, pt_idx, uv_pix, intrinsic):
    """sqrt(mean ||K pi(R^T (X - C)) - uv||^2) over all observations."""
    nv = cams.shape[0]
    rots = np.stack([quat_to_rot_unchecked(cams[c, 3:7]) for c in range(nv)])
    d = pts[:, pt_idx].T - cams[cam_idx, 0:3]
    p = np.einsum('mji,mj->mi', rots[cam_idx], d)
    pix = (np.asarray(intrinsic) @ p.T)
    uv = pix[0:2] / pix[2:3]
    return float(np.sqrt(np.mean(np.sum((uv - uv_pix) ** 2, axis=0))))


# --------------------------------------------------------------------------------------
# Two-view initialisation: epipolar_processor.py:22-267,
# campose_processor.py:29-189
# --------------------------------------------------------------------------------------
def fund_normalize(left, right):
    """epipolar_processor.py:97-137.  left/right: (>=2, n) pixel rows -> ((n,4) pairs, T_left, T_right)."""
    n = left.shape[1]
    lp, rp = left[0:2, :].T, right[0:2, :].T
    lm, rm = np.average(lp, axis=0), np.average(rp, axis=0)
    ls = (2 * n) ** 0.5 / np.sum((np.sum((lp - lm) ** 2, axis=1)) ** 0.5)
    rs = (2 * n) ** 0.5 / np.sum((np.sum((rp - rm) ** 2, axis=1)) ** 0.5)
    tl = np.array([[ls, 0, -lm[0] * ls], [0, ls, -lm[1] * ls], [0, 0, 1]])
    tr = np.array([[rs, 0, -rm[0] * rs], [0, rs, -rm[1] * rs], [0, 0, 1]])
    one = np.ones((n, 1))
    ln = (tl @ np.column_stack((lp, one)).T).T
    rn = (tr @ np.column_stack((rp, one)).T).T
    return np.column_stack((ln[:, 0:2], rn[:, 0:2])), tl, tr


def fund_eight_point(pairs8):
    """epipolar_processor.py:140-193: null vector of the 8x9 system, rank-2 projection, / f[2][2]."""
    w = np.zeros((8, 9))
    for i in range(8):
        x1, y1, x2, y2 = pairs8[i]
        w[i] = [x1 * x2, y1 * x2, x2, x1 * y2, y1 * y2, y2, x1, y1, 1.0]
    _, _, vh = np.linalg.svd(w)
    f = np.reshape(vh.T[:, 8], (3, 3))
    u, s, vh = np.linalg.svd(f)
    f2 = u @ np.diag([s[0], s[1], 0]) @ vh
    if np.linalg.matrix_rank(f2) != 2:
        raise ValueError("f__ rank is not equal to 2")
    return f2 / f2[2][2]


def fund_ransac(pairs, samples, threshold):
    """epipolar_processor.py:196-247 with the 8-index draws of ``random.sample`` passed in.
    Returns (inlier index list or None, F (normalised coordinates), winning hypothesis or -1)."""
    rows = pairs.shape[0]
    if rows < 8:
        raise ValueError("Insufficient matched pairs : {}".format(rows))
    if rows == 8:
        return list(range(8)), fund_eight_point(pairs), 0
    xl = np.column_stack((pairs[:, 0:2], np.ones(rows)))
    xr = np.column_stack((pairs[:, 2:4], np.ones(rows)))
    best_n, best_idx, best_f, best_h = 0, None, np.zeros((3, 3)), -1
    for h, idx in enumerate(samples):
        f8 = fund_eight_point(pairs[list(idx), :])
        val = np.abs(np.einsum('ni,ij,nj->n', xr, f8, xl))
        inl = np.nonzero(val < threshold)[0]
        if len(inl) > best_n:
            best_n, best_idx, best_f, best_h = len(inl), [int(i) for i in inl], f8, h
    return best_idx, best_f, best_h


def fund_denormalize(f, tl, tr):
    """epipolar_processor.py:251-267."""
    g = tr.T @ f @ tl
    return g / g[2][2]


def determine_fundamental(left, right, samples, threshold):
    """epipolar_processor.py:22-57: (inlier indices, fundamental matrix in pixel coordinates)."""
    pairs, tl, tr = fund_normalize(left, right)
    inl, f, _ = fund_ransac(pairs, samples, threshold)
    return inl, fund_denormalize(f, tl, tr)


def essential_from_fundamental(fund, left_k, right_k):
    """epipolar_processor.py:60-95."""
    e = right_k.T @ fund @ left_k
    u, _, vh = np.linalg.svd(e)
    e = u @ np.diag([1, 1, 0]) @ vh
    if np.linalg.matrix_rank(e) != 2:
        raise ValueError("esse_mat rank is not equal to 2")
    return e / e[2][2]


def pose_candidates(esse):
    """campose_processor.py:29-100 -> (r1, r2, c1, c2).  The ORDER of (r1, r2) and the sign of c1
    follow LAPACK's singular-vector signs; the set {r1, r2} x {c1, -c1} does not."""
    w = np.array([[0, -1, 0], [1, 0, 0], [0, 0, 1]])
    u, _, vh = np.linalg.svd(esse)
    c1 = u[:, 2].reshape(-1, 1)
    r1, r2 = u @ w @ vh, u @ w.T @ vh
    if np.linalg.det(r1) < 0:
        r1 = -r1
    if np.linalg.det(r2) < 0:
        r2 = -r2
    return r1.T, r2.T, c1, -c1


def cheirality(p1, p2, pts_h):
    """campose_processor.py:133-189: indices of the points in front of both cameras."""
    z1 = (p1 @ pts_h)[2]
    z2 = (p2 @ pts_h)[2]
    return [int(i) for i in np.nonzero((z1 > 0) & (z2 > 0))[0]]


def disambiguate(ref_proj, projs_four, pts_four):
    """campose_processor.py:102-131: first candidate with the strictly largest valid count."""
    best, best_n, best_idx = 0, 0, []
    for i in range(4):
        idx = cheirality(ref_proj, projs_four[i], pts_four[i])
        if len(idx) > best_n:
            best, best_n, best_idx = i, len(idx), idx
    return best, best_idx
