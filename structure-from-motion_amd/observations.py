"""KeyTrack tables -> observation CSR, with the reference's exact visibility semantics.

The reference's BA loop asks ``KeyTracker.is_visible(view, tri)`` for every (point, view) pair
of every iteration — an ``np.where`` over the view's whole key row each time (ba_processor.py:309,
key_tracker.py:198-204).  The device path needs that answer once, as a list of observations sorted
by (point, view).  This module reproduces the oracle's answer including quirk Q3:

* ``np.any(key_idx)`` tests the *index values*: a point whose only matching key index is 0 is
  reported invisible;
* with several matching keys the first (smallest) index is returned — 0 included, as long as some
  other matching index is non-zero.
"""
import numpy as np

from .geometry import normalise_pixels


def visible_keys(self_row, n_pts):
    """For one view: arrays (tri_idx, key_idx) of the points ``is_visible`` reports, ascending tri_idx.

    ``self_row`` is ``track_list[v].table[v, :]`` (tri-point id per key, -1 = unused)."""
    row = np.asarray(self_row).astype(np.int64, copy=False)
    keys = np.flatnonzero((row >= 0) & (row < n_pts))
    if keys.size == 0:
        return np.empty(0, dtype=np.int64), np.empty(0, dtype=np.int64)
    ids = row[keys]
    order = np.argsort(ids, kind="stable")          # keys stay ascending inside one id
    ids, keys = ids[order], keys[order]
    first = np.ones(ids.shape[0], dtype=bool)
    first[1:] = ids[1:] != ids[:-1]
    last = np.ones(ids.shape[0], dtype=bool)
    last[:-1] = first[1:]
    min_key = keys[first]                            # what key_idx[0][0] returns
    max_key = keys[last]
    seen = max_key > 0                               # np.any(key_idx): some matching index is non-zero
    return ids[first][seen], min_key[seen]


def build_observations(self_rows, n_pts):
    """All (cam, pt, key) triples of the reference's loop (ba_processor.py:304-310), sorted by
    (point, view), plus the CSR pointer over points.  Returns (pt_ptr, cam_idx, pt_idx, key_idx)."""
    cams, pts, keys = [], [], []
    for c, row in enumerate(self_rows):
        t, k = visible_keys(row, n_pts)
        cams.append(np.full(t.shape[0], c, dtype=np.int64))
        pts.append(t)
        keys.append(k)
    cam_idx = np.concatenate(cams) if cams else np.empty(0, dtype=np.int64)
    pt_idx = np.concatenate(pts) if pts else np.empty(0, dtype=np.int64)
    key_idx = np.concatenate(keys) if keys else np.empty(0, dtype=np.int64)
    order = np.lexsort((cam_idx, pt_idx))
    cam_idx, pt_idx, key_idx = cam_idx[order], pt_idx[order], key_idx[order]
    pt_ptr = np.zeros(n_pts + 1, dtype=np.int32)
    np.cumsum(np.bincount(pt_idx, minlength=n_pts), out=pt_ptr[1:])
    return pt_ptr, cam_idx.astype(np.int32), pt_idx.astype(np.int32), key_idx.astype(np.int32)


def gather_normalised_keys(views, cam_idx, key_idx):
    """uv_norm (2, M): ``inv(view.k) @ [u, v, 1]`` divided by its third component, per observation
    (ba_processor.py:339-342).  ``views[c].key_pts[k].pt`` is the pixel key."""
    m = cam_idx.shape[0]
    uv = np.empty((2, m), dtype=np.float64)
    for c, view in enumerate(views):
        sel = np.flatnonzero(cam_idx == c)
        if sel.size == 0:
            continue
        pix = np.array([view.key_pts[int(k)].pt for k in key_idx[sel]], dtype=np.float64).reshape(-1, 2).T
        uv[:, sel] = normalise_pixels(pix, view.k)
    return uv
