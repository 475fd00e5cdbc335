"""BASELINE config 5's call chain against the REAL reference.

tests/golden/g10_incremental_{61,62,63}.npz were captured by tools/capture_goldens.py (g10_incremental) from the
reference's own classes: per registered view ``CamposeProcessor.estimate_cam_pose_pnp`` (seeded six-point RANSAC +
300 nonlinear iterations, campose_processor.py:192-246) -> ``View.update_cam_pose`` -> ``TriangulationProcessor.triangulate``
(triangulation_processor.py:31-88) -> ``add_tri_pt`` -> ``BaProcessor.__execute_bundle_adjustment`` (ba_processor.py:274-439),
view after view on ONE stream of Python's global RNG (utils.py:172-174).  Here the same sequences run FREE through the
drop-in classes -- nothing is re-synchronised with the fixture between views -- and every link must reproduce the
reference: the RANSAC winner (quirk Q13 included: q13.py), the inlier LIST, the RNG state, and to 1e-9 (relative, max
norm) the PnP pose, the new points and the state after every BA.

Sequence 62 is the reference going wrong on its own: at view 4 its RANSAC finds 10 inliers among 90 points and at view 5
the winner is a hypothesis whose det branch fired (pose (R, -C), 77 inliers).  The drop-in has to follow it there too."""
import hashlib
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def digest():
    return hashlib.sha256(repr(random.getstate()).encode()).hexdigest()


class KP:
    def __init__(self, x, y):
        self.pt = (x, y)


class View:
    def __init__(self, rot, loc, k, kps):
        self.k, self.key_pts = k, kps
        self.update_cam_pose(rot, loc)

    def update_cam_pose(self, rot, loc):      # view_processor.py:61-69
        self.rot, self.loc = rot, loc
        self.cam_proj = self.k @ np.hstack((rot.T, rot.T @ -loc))


class Holder:
    pass


def run_chain(sfm, g, reproduce_q13=True):
    """The per-view chain through the drop-in classes; returns the per-view records."""
    K = g["K"]
    n_views, n_pts = int(g["n_views"]), int(g["n_pts"])
    uv, birth = g["uv"], g["birth"]
    thr, sub, samp, k6, its = g["ransac"].tolist()
    cfg = sfm.processors.RansacConfig(thr, sub, samp, int(k6), int(its))          # seeds Python's RNG (utils.py:172-174)
    cp = sfm.processors.HipCamposeProcessor(cfg, float(g["pnp"][0]), int(g["pnp"][1]))
    cp.reproduce_q13 = reproduce_q13
    tp = sfm.processors.HipTriangulationProcessor(float(g["tri"][0]), int(g["tri"][1]))
    vp, kt = Holder(), Holder()
    vp.view_list, kt.track_list = [], []
    bp = sfm.processors.HipBaProcessor(vp, kt, None, tp, cp, iteration=int(g["ba"][1]), damping_factor=float(g["ba"][0]))
    bp.ba_verbose = False
    start = digest()

    def add_view(c, rot, loc):
        kps = [KP(-1.0, -1.0)] + [KP(float(uv[c][0, j]), float(uv[c][1, j])) for j in range(n_pts)]
        vp.view_list.append(View(rot, loc, K.copy(), kps))
        tr = Holder()
        tr.table = np.full((n_views, n_pts + 1), -1, dtype=int)
        kt.track_list.append(tr)

    add_view(0, g["rot0"].copy(), g["loc0"].copy())
    known = np.zeros(n_pts, dtype=bool)
    records = []
    for c in range(1, n_views):
        rec = {"view": c}
        if known.any():
            idx = np.flatnonzero(known)
            rec["rng_before_pnp"] = digest()
            inl, r_new, c_new = cp.estimate_cam_pose_pnp(uv[c][:, idx], tp.tri_pts[:, idx], K)          # ba_processor.py:191
            rec.update(inliers=list(inl), pnp_rot=r_new, pnp_loc=c_new, ransac=dict(cp.ransac_last or {}))
        else:
            r_new, c_new = g["rot1"].copy(), g["loc1"].copy()
        add_view(c, r_new, c_new)
        rec["rng_after_pnp"] = digest()
        new = np.flatnonzero(birth == c)
        views = vp.view_list
        pts_new = tp.triangulate([views[c - 1].cam_proj, views[c].cam_proj], [uv[c - 1][:, new], uv[c][:, new]])      # ba:246
        rec["new_pts"] = np.array(pts_new)
        tp.add_tri_pt(np.array(pts_new))                                                                              # ba:262
        known[new] = True
        ids = np.flatnonzero(known)
        for v in range(c + 1):
            kt.track_list[v].table[v, 1 + ids] = ids
        bp._BaProcessor__execute_bundle_adjustment()                                                                   # ba:267
        rec["ba_rots"] = np.array([v.rot for v in views]); rec["ba_locs"] = np.array([v.loc for v in views])
        rec["ba_pts"] = np.array(tp.tri_pts)
        records.append(rec)
    bp.ba_release()
    return start, records


@pytest.mark.parametrize("seed", [61, 62, 63])
def test_incremental_chain_matches_the_reference(hip, sfm, golden, seed):
    g = golden("g10_incremental_%d.npz" % seed)
    start, records = run_chain(sfm, g)
    assert start == str(g["rng_digest_start"])
    worst = {}
    for rec in records:
        c = rec["view"]
        pre = "v%d_" % c
        if "inliers" in rec:
            assert rec["rng_before_pnp"] == str(g[pre + "rng_before_pnp"]), (seed, c)
            want_winner = int(np.argmax(g[pre + "hyp_counts"]))
            assert rec["ransac"]["hypothesis"] == want_winner, (seed, c, rec["ransac"], want_winner)
            assert rec["ransac"]["q13_fired"] == bool(g[pre + "hyp_fired"][want_winner])
            assert rec["inliers"] == g[pre + "inliers"].tolist(), (seed, c)
            worst["pnp"] = max(worst.get("pnp", 0.0), rel(rec["pnp_rot"], g[pre + "pnp_rot"]), rel(rec["pnp_loc"], g[pre + "pnp_loc"]))
        assert rec["rng_after_pnp"] == str(g[pre + "rng_after_pnp"]), (seed, c)
        worst["tri"] = max(worst.get("tri", 0.0), rel(rec["new_pts"], g[pre + "new_pts"]))
        worst["ba"] = max(worst.get("ba", 0.0), rel(rec["ba_rots"], g[pre + "ba_rots"]), rel(rec["ba_locs"], g[pre + "ba_locs"]),
                          rel(rec["ba_pts"], g[pre + "ba_pts"]))
        print("chain seed %d view %d: ransac %s, worst so far %s" % (seed, c, rec.get("ransac"), {k: "%.2e" % v for k, v in worst.items()}))
    assert max(worst.values()) < 1e-9, (seed, worst)


def test_sane_ransac_differs_where_q13_bites(hip, sfm, golden):
    """What the chain looks like WITHOUT the Q13 reproduction (``reproduce_q13 = False``: every hypothesis keeps its
    sign-invariant centre): same RNG stream, but the winner is the first best hypothesis of all, which the reference's
    LAPACK had ruined in most views -- the measure of how often quirk Q13 decides the reference's result."""
    differs = total = 0
    for seed in (61, 63):
        g = golden("g10_incremental_%d.npz" % seed)
        _start, records = run_chain(sfm, g, reproduce_q13=True)
        for rec in records:
            if "ransac" in rec:
                total += 1
                differs += int(rec["ransac"]["sane_winner"] != rec["ransac"]["hypothesis"])
        _start, sane = run_chain(sfm, g, reproduce_q13=False)
        for rec in sane:      # the RNG stream is the same either way
            assert rec["rng_after_pnp"] == str(g["v%d_rng_after_pnp" % rec["view"]])
    print("Q13: the sane winner differs from the reference's in %d of %d RANSAC calls" % (differs, total))
    assert total == 8
