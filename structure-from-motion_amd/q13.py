"""Quirk Q13 of the reference's six-point RANSAC, reproduced (round 4).

``CamposeProcessor.__estimate_six_pts`` (campose_processor.py:565-633) takes the last right-singular vector of the
12 x 12 design matrix as the camera matrix, orthogonalises its left 3 x 3 block, ``rot = (uu @ vvh).T``, forms
``loc = rot @ -cam_mat[:, 3] / s`` and then, if ``det(rot) < 0``, negates rot AND loc (campose:629-631).  rot and the
null vector flip sign together, loc does not: whether the branch fires -- whether the hypothesis comes out with the
camera centre C or with -C -- is decided by the arbitrary sign LAPACK gives that singular vector.  About half of the
hypotheses of a RANSAC run are hit (tests/golden/g5_pnp_hypotheses.npz, g10_incremental_*.npz); a hit one usually
scores next to nothing, so the reference's winner (first strictly larger inlier count, campose:554-558) is the first
best hypothesis among the ones its LAPACK spared.

The device solves every hypothesis with the sign-invariant centre and counts its inliers under (R, C) and under
(R, -C) (``sfm_pnp_ransac_evaluate``).  What only the host can know is the branch decision; this module asks NumPy --
the library the reference itself would have asked, in the same process, on the same six points -- and only for the
hypotheses that can still win.  It is part of the drop-in's host logic, not a compute path: two small SVDs for, on
average, two hypotheses per RANSAC call.
"""
import numpy as np


def det_branch_fires(key_2d_pts_6_in_cam_coord, tri_3d_pts_6):
    """True when campose_processor.py:629 sees ``det(rot) < 0`` for this six-point sample: the reference's own
    expressions (campose:585-626) on the reference's own inputs, evaluated by the host's LAPACK."""
    p2d, p3d = np.asarray(key_2d_pts_6_in_cam_coord), np.asarray(tri_3d_pts_6)
    # campose:588-611, the same products element by element (the fourth entry of a point is taken to be 1 there)
    x, y, z = p2d[0], p2d[1], p2d[2]
    pts = p3d[0:3].T                          # (6, 3)
    w = np.zeros((12, 12))
    w[0::2, 0:3] = z[:, None] * pts
    w[0::2, 3] = z
    w[0::2, 8:11] = -x[:, None] * pts
    w[0::2, 11] = -x
    w[1::2, 4:7] = z[:, None] * pts
    w[1::2, 7] = z
    w[1::2, 8:11] = -y[:, None] * pts
    w[1::2, 11] = -y
    _u, _s, vh = np.linalg.svd(w)
    cam_mat = np.reshape(vh.transpose()[:, -1], (3, 4))
    uu, _ss, vvh = np.linalg.svd(cam_mat[:, 0:3])
    rot = (uu @ vvh).T
    return bool(np.linalg.det(rot) < 0)


def reference_winner(counts, counts_neg, fires):
    """The hypothesis the reference's loop keeps (campose:524-560): the FIRST one with the largest count, where a
    hypothesis counts ``counts_neg[h]`` if its det branch fired and ``counts[h]`` otherwise.  ``fires(h) -> bool`` is
    asked lazily, in the order of what a hypothesis could score at best.  Returns (h, fired) or (-1, False) when no
    hypothesis has an inlier (the reference then returns its initial identity pose)."""
    counts = np.asarray(counts, dtype=np.int64)
    counts_neg = np.asarray(counts_neg, dtype=np.int64)
    upper = np.maximum(counts, counts_neg)
    order = np.lexsort((np.arange(upper.shape[0]), -upper))          # best possible score first, earlier index first
    best_h, best_cnt, best_fired = -1, 0, False
    for h in order.tolist():
        if upper[h] < best_cnt or upper[h] == 0:
            break
        if upper[h] == best_cnt and best_h >= 0 and h > best_h:
            break                                                     # could only tie, and ties go to the earlier hypothesis
        if counts[h] == counts_neg[h]:
            fired, score = None, int(counts[h])                       # the decision does not matter for the score
        else:
            fired = bool(fires(h))
            score = int(counts_neg[h] if fired else counts[h])
        if score > best_cnt or (score == best_cnt and score > 0 and h < best_h):
            if fired is None:
                fired = bool(fires(h))                                # ... but it decides which centre is returned
            best_h, best_cnt, best_fired = h, score, fired
    return best_h, best_fired
