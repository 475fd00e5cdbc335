"""Drop-in mirrors of the reference's ``*_processor`` classes for the nonlinear-refinement hot path.

Same method names, argument meaning, defaults, return shapes, in-place semantics, prints and
exceptions as the reference; the arithmetic runs in libsfm_hip.so on an MI355X.

Two ways to use them:

* as **mixins in front of the reference classes** — the reference keeps its front end, linear
  solvers and state machine, the hot path is overridden::

      class TriangulationProcessor(HipTriangulationMixin, triangulation_processor.TriangulationProcessor): pass
      class CamposeProcessor(HipCamposeMixin, campose_processor.CamposeProcessor): pass
      class BaProcessor(HipBaMixin, ba_processor.BaProcessor): pass

* as the **standalone classes** below (``HipTriangulationProcessor`` ...), which carry the
  constructor state of the reference classes and are what the tests drive.

Reference interfaces mirrored (file:line in the reference repo):
  TriangulationProcessor.nonlinear_triangulate / construct_jacobian_matrix / triangulate
      triangulation_processor.py:160-234 / 237-271 / 31-88
  CamposeProcessor.nonlinear_estimate_cam_pose_pnp / construct_jacobian_matrix / estimate_cam_pose_pnp
      campose_processor.py:308-459 / 462-482 / 192-246
  BaProcessor.__execute_bundle_adjustment          ba_processor.py:274-439
"""
import logging
import math

import numpy as np

from . import native
from .geometry import (pack_cameras, quaternion_to_rotation_unchecked, quaternions_to_rotations)
from .observations import KeyCache, ObservationTracker, build_observations, gather_normalised_keys
from .q13 import det_branch_fires, reference_winner
from .sampling import sample_indices


# ------------------------------------------------------------------------------------------------
class HipTriangulationMixin:
    """Hot-path methods of TriangulationProcessor (triangulation_processor.py:7-309)."""

    def nonlinear_triangulate(self, init_3d_pts, projs, matched_pairs,
                              damping_factor=None, iteration=None):
        # falsy -> instance default, exactly like the reference (tri:200-203, quirk Q4)
        if not damping_factor:
            damping_factor = self.damping_factor
        if not iteration:
            iteration = self.iteration
        num_pts = matched_pairs[0].shape[1]
        num_views = len(projs)
        init = np.ascontiguousarray(np.asarray(init_3d_pts), dtype=np.float64)
        if num_pts == 0:
            return np.copy(init)
        uv = np.empty((num_views, 2, num_pts), dtype=np.float64)
        for v in range(num_views):
            uv[v] = np.asarray(matched_pairs[v])[0:2, :]
        projs_arr = np.stack([np.asarray(p, dtype=np.float64).reshape(3, 4) for p in projs])
        # the reference iterates over matched_pairs[0].shape[1] columns of a copy of init_3d_pts
        out = np.copy(init)
        out[:, :num_pts] = native.tri_nonlinear(projs_arr, uv, init[:, :num_pts], damping_factor, iteration)
        return out

    def construct_jacobian_matrix(self, tri_3d_pt, projs, num_views):
        projs_arr = np.stack([np.asarray(p, dtype=np.float64).reshape(3, 4) for p in projs[:num_views]])
        jx = native.jac_pt(projs_arr[None], np.asarray(tri_3d_pt, dtype=np.float64).reshape(1, 4))
        return jx[0]

    def triangulate(self, projs, matched_pairs, damping_factor=None, iteration=None):
        if not damping_factor:
            damping_factor = self.damping_factor
        if not iteration:
            iteration = self.iteration
        num_projs = len(projs)
        num_views = len(matched_pairs)
        if num_projs != num_views:
            logging.warning('%s : the numbers of views and of projections are different, %d and %d',
                            self.__class__.__name__, num_views, num_projs)
            raise ValueError("different numbers of views and projections : {} - {}".format(num_views, num_projs))
        if num_projs < 2:
            logging.warning('%s : insufficient projections, %d', self.__class__.__name__, num_projs)
            raise ValueError("insufficient projections : {}".format(num_projs))
        # linear (tri:85) + nonlinear (tri:86) in one device call: the DLT points never visit the host
        uv, projs_arr = self._pack_views(projs, matched_pairs)
        return native.triangulate(projs_arr, uv, damping_factor, iteration)

    @staticmethod
    def _pack_views(projs, matched_pairs):
        num_views = len(matched_pairs)
        num_pts = matched_pairs[0].shape[1]
        uv = np.empty((num_views, 2, num_pts), dtype=np.float64)
        for v in range(num_views):
            uv[v] = np.asarray(matched_pairs[v])[0:2, :]
        projs_arr = np.stack([np.asarray(p, dtype=np.float64).reshape(3, 4) for p in projs])
        return uv, projs_arr

    def linear_triangulate(self, projs, matched_pairs):
        """DLT triangulation (triangulation_processor.py:91-157) on the device: per point the null
        vector of the (2V x 4) system, divided by W.  Same sanity checks / prints / ``None`` returns."""
        if len(matched_pairs) != len(projs) != 2:       # sic: chained comparison of the reference (Q12)
            print('{}:{} - num of projs {} and matched pairs {} need to be 2'.format(
                self.__class__.__name__, 'linear_triangulate', len(projs), len(matched_pairs)))
            return None
        if matched_pairs[0].shape[1] != matched_pairs[1].shape[1]:
            print('{}:{} - matched pairs number does not match {} vs {}'.format(
                self.__class__.__name__, 'linear_triangulate',
                matched_pairs[0].shape[1], matched_pairs[1].shape[1]))
            return None
        uv, projs_arr = self._pack_views(projs, matched_pairs)
        return native.tri_linear(projs_arr, uv)


class HipTriangulationProcessor(HipTriangulationMixin):
    """Standalone TriangulationProcessor: constructor state of triangulation_processor.py:12-28."""

    def __init__(self, damping_factor=0.5, iteration=100):
        self.damping_factor = damping_factor
        self.iteration = iteration
        self.tri_pts = None

    def add_tri_pt(self, tri_pt):
        if not np.any(self.tri_pts):
            self.tri_pts = tri_pt
        else:
            self.tri_pts = np.hstack((self.tri_pts, tri_pt))


# ------------------------------------------------------------------------------------------------
class HipCamposeMixin:
    """Hot-path methods of CamposeProcessor (campose_processor.py:11-808)."""

    quirk_flags = native.QUIRKS_REFERENCE        # bug-compatible by default
    reproduce_q13 = True                         # the RANSAC winner the reference's LAPACK would have left standing (q13.py);
                                                 # False: the sane RANSAC (every hypothesis with its sign-invariant centre)
    ransac_last = None                           # diagnostics of the last RANSAC call

    def nonlinear_estimate_cam_pose_pnp(self, key_2d_pts, tri_3d_pts, intrinsic_mat,
                                        init_rot, init_loc, damping_factor=None, iteration=None):
        if not damping_factor:
            damping_factor = self.damping_factor
        if not iteration:
            iteration = self.iteration
        if key_2d_pts.shape[1] != tri_3d_pts.shape[1]:
            logging.warning('%s : different numbers of key points and of triangulated points',
                            self.__class__.__name__)
            raise ValueError("key pts num - triangulated pts num : {} - {}"
                             .format(key_2d_pts.shape[1], tri_3d_pts.shape[1]))
        rot, loc = native.pnp_nonlinear(key_2d_pts, tri_3d_pts, intrinsic_mat, init_rot, init_loc,
                                        damping_factor, iteration, self.quirk_flags)
        return rot, loc

    def construct_jacobian_matrix(self, rot, loc, pt_3d):
        jp, st = native.jac_cam(np.asarray(rot, dtype=np.float64).reshape(1, 3, 3),
                                np.asarray(loc, dtype=np.float64).reshape(1, 3),
                                np.asarray(pt_3d, dtype=np.float64).reshape(1, 4), self.quirk_flags)
        native.check(int(st[0]))
        return jp[0]

    def linear_estimate_cam_pose_pnp(self, key_2d_pts, tri_3d_pts, intrinsic_mat, ransac_config=None):
        """RANSAC 6-point DLT PnP (campose_processor.py:249-305, 485-633).  The six-point samples are drawn
        here with ``random.sample`` exactly as the reference does (same consumption of Python's global RNG
        stream, campose:531; ``sampling.sample_indices`` draws them in bulk); every hypothesis is solved and scored on the device."""
        inliers, rot, loc = self._linear_pnp(key_2d_pts, tri_3d_pts, intrinsic_mat, ransac_config)
        return inliers.tolist(), rot, loc

    def _check_pnp_input(self, key_2d_pts, tri_3d_pts):
        if key_2d_pts.shape[1] != tri_3d_pts.shape[1]:
            logging.warning('%s : different numbers of key points and of triangulated points',
                            self.__class__.__name__)
            raise ValueError("key pts num - triangulated pts num : {} - {}"
                             .format(key_2d_pts.shape[1], tri_3d_pts.shape[1]))
        num_pts = key_2d_pts.shape[1]
        if num_pts < 6:
            logging.warning('%s : required equal or more than six points %d', self.__class__.__name__, num_pts)
            raise ValueError("required equal or more than six points {}".format(num_pts))
        return num_pts

    def _q13_winner(self, key_2d_pts, tri_3d_pts, intrinsic_mat, samples, rots, locs, counts, counts_neg):
        """Quirk Q13 (q13.py): the device has solved and scored every hypothesis, under its centre C and under -C; which of
        the two the reference would have scored is its LAPACK's decision, asked of NumPy for the hypotheses that can still
        win.  Returns (rot, loc) of the reference's winner, or None when no hypothesis has an inlier."""
        kinv = np.linalg.inv(intrinsic_mat)

        def fires(h):
            idx = samples[h].tolist()
            return det_branch_fires(kinv @ key_2d_pts[:, idx], tri_3d_pts[:, idx])        # campose:532-535

        best, fired = reference_winner(counts, counts_neg, fires)
        self.ransac_last = {"hypothesis": best, "q13_fired": fired,
                            "sane_winner": int(np.argmax(counts)) if counts.max() > 0 else -1}
        if best < 0:
            return None
        return rots[best].copy(), (-locs[best] if fired else locs[best]).reshape(3, 1).copy()

    def _linear_pnp(self, key_2d_pts, tri_3d_pts, intrinsic_mat, ransac_config):
        """``linear_estimate_cam_pose_pnp`` with the inlier indices as an int array (the list the reference returns costs
        0.1 ms to build and 0.25 ms to turn back into an index at 5 000 inliers: ``estimate_cam_pose_pnp`` builds it once)."""
        if not ransac_config:
            ransac_config = self.ransac_config
        num_pts = self._check_pnp_input(key_2d_pts, tri_3d_pts)
        samples = sample_indices(num_pts, 6, ransac_config.iteration, as_array=True)       # = [random.sample(range(num_pts), 6) ...], campose:531
        if not self.reproduce_q13:
            rot, loc, inlier_indices, _best = native.pnp_linear_ransac(
                key_2d_pts, tri_3d_pts, intrinsic_mat, samples, ransac_config.inlier_threshold, as_array=True)
            return inlier_indices, rot, loc
        rots, locs, counts, counts_neg = native.pnp_ransac_evaluate(
            key_2d_pts, tri_3d_pts, intrinsic_mat, samples, ransac_config.inlier_threshold)
        won = self._q13_winner(key_2d_pts, tri_3d_pts, intrinsic_mat, samples, rots, locs, counts, counts_neg)
        if won is None:                                # no hypothesis has an inlier: the initial pose (campose:519-522)
            return np.empty(0, dtype=np.intp), np.identity(3), np.zeros((3, 1))
        rot, loc = won
        inlier_indices = native.pnp_inlier_mask(key_2d_pts, tri_3d_pts, intrinsic_mat, rot, loc, ransac_config.inlier_threshold,
                                                as_array=True)
        return inlier_indices, rot, loc

    def estimate_cam_pose_pnp(self, key_2d_pts, tri_3d_pts, intrinsic_mat,
                              ransac_config=None, damping_factor=None, iteration=None):
        if not ransac_config:
            ransac_config = self.ransac_config
        if not damping_factor:
            damping_factor = self.damping_factor
        if not iteration:
            iteration = self.iteration
        if self.reproduce_q13:
            # RANSAC evaluation and refinement around the host's choice of the winner, the view resident on the device in
            # between (sfm_pnp_ransac_begin / _finish): one upload of the keys and points, no host-side gather of the inliers
            num_pts = self._check_pnp_input(key_2d_pts, tri_3d_pts)
            samples = sample_indices(num_pts, 6, ransac_config.iteration, as_array=True)   # campose:531
            session, rots, locs, counts, counts_neg = native.pnp_ransac_begin(
                key_2d_pts, tri_3d_pts, intrinsic_mat, samples, ransac_config.inlier_threshold)
            try:
                won = self._q13_winner(key_2d_pts, tri_3d_pts, intrinsic_mat, samples, rots, locs, counts, counts_neg)
            except BaseException:
                native.pnp_session_destroy(session)
                raise
            if won is not None:
                sel, ref_rot, ref_loc = native.pnp_ransac_finish(session, won[0], won[1], ransac_config.inlier_threshold,
                                                                 damping_factor, iteration, self.quirk_flags)
                return sel.tolist(), ref_rot, ref_loc      # a list is the reference's return type (campose:246)
            native.pnp_session_destroy(session)
            sel, ini_rot, ini_loc = np.empty(0, dtype=np.intp), np.identity(3), np.zeros((3, 1))      # campose:519-522
        else:
            sel, ini_rot, ini_loc = self._linear_pnp(key_2d_pts, tri_3d_pts, intrinsic_mat, ransac_config)
        ref_rot, ref_loc = self.nonlinear_estimate_cam_pose_pnp(
            key_2d_pts[:, sel], tri_3d_pts[:, sel], intrinsic_mat,
            ini_rot, ini_loc, damping_factor, iteration)
        return sel.tolist(), ref_rot, ref_loc          # a list is the reference's return type (campose:246)

    # ---- two-view initialisation (campose_processor.py:29-189) --------------------------------------------
    def extract_cam_pose_from_essential_mat(self, esse_mat):
        """(r1, r2, c1, c2) of campose_processor.py:29-100.  The set {r1, r2} x {c1, c2} is the reference's;
        which rotation is called r1 and which sign c1 carries follows LAPACK's singular-vector signs there and
        the device's Jacobi sweep here (the caller tries all four combinations, ba_processor.py:81-97)."""
        return native.pose_candidates(esse_mat)

    def evalulate_cam_pose_cheirality(self, proj_1, proj_2, tri_3d_pts):
        """Indices of the points in front of both cameras (campose_processor.py:133-189)."""
        mask, _counts, _best = native.cheirality(proj_1, np.asarray(proj_2)[np.newaxis], np.asarray(tri_3d_pts)[np.newaxis])
        return np.flatnonzero(mask[0]).tolist()

    def disambiguate_cam_pose_four(self, ref_proj, projs_four, tri_3d_pts_four):
        """(best_idx, most_valid_indices) of campose_processor.py:102-131: one device call for the four candidates."""
        mask, counts, best = native.cheirality(ref_proj, np.array(projs_four), np.array(tri_3d_pts_four))
        if counts[best] == 0:
            return 0, []
        return best, np.flatnonzero(mask[best]).tolist()


class HipEpipolarMixin:
    """Eight-point RANSAC and essential-matrix extraction of ``EpipolarProcessor`` (epipolar_processor.py:22-95)
    on the device.  Expects ``self.ransac``; sets ``self.fund_mat`` / ``self.esse_mat`` as the reference does."""

    def determine_fundamental_mat(self, matched_pairs, ransac_config=None):
        cfg = self.ransac if ransac_config is None else ransac_config
        left, right = np.asarray(matched_pairs[0]), np.asarray(matched_pairs[1])
        rows = left.shape[1]
        if rows < 8:
            logging.error('%s : number of matched pairs needs equal or more than eight')
            raise ValueError("Insufficient matched pairs : {}".format(rows))
        # same consumption of Python's global RNG stream as epipolar:225 (no draw when rows == 8)
        samples = None if rows == 8 else sample_indices(rows, 8, cfg.iteration, as_array=True)
        fund, inliers, _best = native.fundamental_ransac(left, right, samples, cfg.inlier_threshold)
        self.fund_mat = fund
        return inliers

    def extract_essential_mat(self, left_intrinsic_mat, right_intrinsic_mat):
        self.esse_mat = native.essential_from_fundamental(self.fund_mat, left_intrinsic_mat, right_intrinsic_mat)


class HipEpipolarProcessor(HipEpipolarMixin):
    """Standalone EpipolarProcessor (constructor of epipolar_processor.py:13-20)."""

    def __init__(self, ransac_config):
        self.fund_mat = np.identity(3)
        self.esse_mat = np.identity(3)
        self.ransac = ransac_config


class RansacConfig:
    """Mirror of utils.RansacConfig (utils.py:129-174): iteration count raised to the confidence bound and
    Python's global RNG seeded with -1 on construction."""

    def __init__(self, inlier_threshold, subset_confidence, sample_confidence, sample_num, iteration,
                 is_use_seed=True):
        self.inlier_threshold = inlier_threshold
        self.subset_confidence = subset_confidence
        self.sample_confidence = sample_confidence
        self.sample_num = int(sample_num)
        self.iteration = int(iteration)
        self.random_seed = -1
        calc_iteration = math.log(1.0 - subset_confidence) / math.log(1.0 - math.pow(sample_confidence, sample_num))
        if calc_iteration > iteration:
            print('RANSAC : iteration increases from {} to {}'.format(iteration, calc_iteration))
            self.iteration = int(calc_iteration)
        if is_use_seed:
            import random
            random.seed(self.random_seed)


class HipCamposeProcessor(HipCamposeMixin):
    """Standalone CamposeProcessor (constructor of campose_processor.py:12-26)."""

    def __init__(self, ransac_config, damping_factor, iteration):
        self.ransac_config = ransac_config
        self.damping_factor = damping_factor
        self.iteration = iteration


# ------------------------------------------------------------------------------------------------
class _ResidentScene:
    """What the drop-in keeps between two BA calls of one processor: the device-resident problem and a host picture of
    what it holds (the track-table rows as of the last call, the key coordinates of every view, intrinsics, the poses and
    points of the last write-back), enough to decide what is NEW in the next call."""

    def __init__(self):
        self.prob = None
        self.tracker = ObservationTracker()      # self rows + per-view visibility as of the last call
        self.keys = KeyCache()                   # (n_keys, 2) pixel coordinates per view, converted once
        self.ks = []              # copies of view.k
        self.n_views = 0
        self.n_pts = 0
        self.pts_written = None   # (3, N) the points of the last write-back
        self.rots_written = None  # (V, 3, 3) / (V, 3): the poses of the last write-back (what view.rot / .loc hold if untouched)
        self.locs_written = None
        self.n_views_before = 0
        self.retired_bytes = 0    # upload bytes of problems this scene has replaced

    def close(self):
        if self.prob is not None:
            self.retired_bytes += self.prob.upload_bytes
            self.prob.close()
            self.prob = None

    @property
    def upload_bytes(self):
        return self.retired_bytes + (self.prob.upload_bytes if self.prob is not None else 0)


class HipBaMixin:
    """``BaProcessor.__execute_bundle_adjustment`` (ba_processor.py:274-439) on the device.

    Reads ``self.view_processor.view_list`` (``.rot/.loc/.k/.key_pts[i].pt``),
    ``self.key_tracker.track_list[i].table``, ``self.tri_processor.tri_pts``, ``self.iteration``,
    ``self.damping_factor``; writes the refined poses back through ``view.update_cam_pose`` and the
    refined points into ``tri_pts[0:3, :]`` in place; prints the reference's DEBUG lines.

    The reference runs this after EVERY registered view, over all views and all points
    (ba_processor.py:267).  The scene therefore stays resident on the device between calls
    (``ba_resident``): each call diffs the observation list against what the device holds and uploads
    only what is new -- the new view's pose, the new points, the new observations (``sfm_ba_append``) --
    plus whatever pose or old point the caller changed since the last write-back.  (The reference re-derives every
    quaternion from ``view.rot`` at the start of a call, ba:285-288; for views whose ``rot`` / ``loc`` still hold what the
    last call wrote, that round trip q -> R(q) -> q(R) runs on the device, ``sfm_ba_rederive_quaternions``.)  If anything was REMOVED (an observation,
    a point, a view), an existing key -> point entry of a track table changed, or a view's key LIST or intrinsic matrix was
    replaced, the problem is rebuilt from scratch.  The pixel coordinates of a key are read ONCE, when its view's key list is
    first seen (``observations.KeyCache``): ``view.key_pts`` is treated as immutable -- the reference's front end never edits a
    ``cv2.KeyPoint`` after detection (view_processor.py) -- and a caller that does edit ``key_pts[i].pt`` in place has to call
    ``ba_release()`` (or hand the view a new list) for the edit to reach the device.
    ``ba_upload_bytes`` reports the PCIe bytes spent so far; ``ba_release()`` frees the device copy."""

    ba_quirk_flags = native.QUIRKS_REFERENCE
    ba_verbose = True          # the reference prints unconditionally (ba:418-439)
    ba_resident = True         # keep the problem on the device between calls (False: one sfm_ba_solve per call)
    ba_last_action = None      # "create" | "append" | "reuse" | "solve": what the last call did (diagnostics / tests)

    def ba_release(self):
        scene = self.__dict__.pop("_hip_scene", None)
        if scene is not None:
            scene.close()

    @property
    def ba_upload_bytes(self):
        scene = self.__dict__.get("_hip_scene")
        return scene.upload_bytes if scene is not None else 0

    # ---- resident problem ---------------------------------------------------------------------------
    def _ba_sync_structure(self, views, tri_num, n_same, new_cams, init_tri_pts):
        """Bring the resident problem to the current observation list; returns it.  ``new_cams`` are the packed cameras of
        the views from ``n_same`` on (the first ``n_same`` are unchanged since the last write-back).

        The track tables are diffed incrementally (``observations.ObservationTracker``: only the entries that changed
        since the last call are looked at, ba:309's semantics preserved); pure growth becomes one ``sfm_ba_append``,
        anything else a rebuild."""
        scene = self.__dict__.get("_hip_scene")
        if scene is None:
            scene = self.__dict__["_hip_scene"] = _ResidentScene()
        view_num = len(views)
        n_old = scene.n_views_before = scene.n_views
        rows = [self.key_tracker.track_list[v].table[v, :] for v in range(view_num)]
        same_intrinsics = scene.prob is not None and view_num >= n_old and all(
            np.array_equal(views[v].k, scene.ks[v]) for v in range(n_old))
        grown = scene.tracker.diff(rows, tri_num) if same_intrinsics else None
        if grown is not None:
            cam_new, pt_new, key_new = grown
            if cam_new.shape[0] == 0 and view_num == n_old and tri_num == scene.n_pts:
                self.ba_last_action = "reuse"
                return scene
            try:
                uv_new = scene.keys.gather_normalised(views, cam_new, key_new)                  # ba:339-342, new keys only
                cams_app = new_cams[n_old - n_same:] if n_same <= n_old else pack_cameras(
                    np.stack([np.asarray(v.rot, dtype=np.float64) for v in views[n_old:]]),
                    np.stack([np.asarray(v.loc, dtype=np.float64).reshape(3) for v in views[n_old:]]))
                scene.prob.append(cams_app, init_tri_pts[:, scene.n_pts:tri_num], cam_new, pt_new, uv_new)
            except Exception:
                self.ba_release()          # the tracker has moved on, the device has not: start over on the next call
                raise
            self.ba_last_action = "append"
        else:
            scene.close()
            pt_ptr, cam_idx, _pt_idx, key_idx = scene.tracker.reset(rows, tri_num)             # ba:309
            uv_norm = scene.keys.gather_normalised(views, cam_idx, key_idx)
            scene.prob = native.BaProblem(view_num, pt_ptr, cam_idx, uv_norm)
            scene.pts_written = None
            scene.rots_written = scene.locs_written = None
            self.ba_last_action = "create"
        scene.ks = [np.array(v.k, dtype=np.float64, copy=True) for v in views]
        scene.n_views = view_num
        scene.n_pts = tri_num
        return scene

    def execute_bundle_adjustment(self):
        views = self.view_processor.view_list
        tri_pts = self.tri_processor.tri_pts
        view_num = len(views)
        tri_num = tri_pts.shape[1]
        init_rots = np.stack([np.asarray(v.rot, dtype=np.float64) for v in views])                    # ba:287
        init_locs = np.stack([np.asarray(v.loc, dtype=np.float64).reshape(3) for v in views])         # ba:288
        init_tri_pts = np.ascontiguousarray(tri_pts[0:3, :], dtype=np.float64)                          # ba:292-294

        if self.ba_resident:
            scene = self.__dict__.get("_hip_scene")
            # cameras the caller did not touch since the last write-back: their quaternion q(R(q)) (ba:285-288 after
            # ba:412) is re-derived on the device; only changed or new views are packed on the host and uploaded
            n_same = 0
            if scene is not None and scene.prob is not None and scene.rots_written is not None:
                n_old = min(scene.rots_written.shape[0], view_num)
                if np.array_equal(init_rots[:n_old], scene.rots_written[:n_old]) and np.array_equal(init_locs[:n_old], scene.locs_written[:n_old]):
                    n_same = n_old
            new_cams = pack_cameras(init_rots[n_same:], init_locs[n_same:]) if n_same < view_num else np.zeros((0, 7))
            scene = self._ba_sync_structure(views, tri_num, n_same, new_cams, init_tri_pts)
            prob = scene.prob
            try:
                if self.ba_last_action == "create":
                    prob.set_cameras(pack_cameras(init_rots, init_locs) if n_same else new_cams)
                    prob.set_points(0, init_tri_pts)
                else:
                    if n_same == view_num or (self.ba_last_action == "append" and n_same == scene.n_views_before):
                        prob.rederive_quaternions(0, n_same)          # appended cameras went up with sfm_ba_append
                    else:
                        cams_all = pack_cameras(init_rots, init_locs) if n_same else new_cams
                        prob.set_cameras(cams_all)
                    # points the caller did not touch since the last write-back are already on the device (bit for
                    # bit what get_state returned); appended points went up with sfm_ba_append
                    done = 0 if scene.pts_written is None else min(scene.pts_written.shape[1], tri_num)
                    if scene.pts_written is None:
                        prob.set_points(0, init_tri_pts)          # no record of what the device holds: upload everything
                    elif done and not np.array_equal(init_tri_pts[:, :done], scene.pts_written[:, :done]):
                        prob.set_points(0, init_tri_pts[:, :done])
                prob.iterate(self.damping_factor, self.iteration, self.ba_quirk_flags)
                cams, pts, rots = prob.get_state_rot()                                             # ba:412 (validated on the device)
            except Exception:
                # the device state has advanced (or is invalid: a bad rotation) while tri_pts / the views keep the old
                # values; the reference re-reads them on every call (ba:292-294), so the next call must start from the
                # host state again: drop the resident copy rather than continue from diverged device points
                self.ba_release()
                raise
            scene.pts_written = pts
            scene.rots_written, scene.locs_written = rots, cams[:, 0:3].copy()
        else:
            init_cam_poses = pack_cameras(init_rots, init_locs)
            rows = [self.key_tracker.track_list[v].table[v, :] for v in range(view_num)]
            pt_ptr, cam_idx, _pt_idx, key_idx = build_observations(rows, tri_num)                  # ba:309
            uv_norm = gather_normalised_keys(views, cam_idx, key_idx)                              # ba:339-342
            cams, pts = native.ba_solve(view_num, pt_ptr, cam_idx, uv_norm, init_cam_poses, init_tri_pts,
                                        self.damping_factor, self.iteration, self.ba_quirk_flags)
            rots = quaternions_to_rotations(cams[:, 3:7])                                          # ba:412 (validated)
            self.ba_last_action = "solve"

        for view_idx in range(view_num):                                                       # ba:409-413
            views[view_idx].update_cam_pose(rots[view_idx].copy(), cams[view_idx, 0:3].reshape(3, 1).copy())
        tri_pts[0:3, :] = pts                                                                  # ba:415-416

        if self.ba_verbose:                                                                    # ba:418-439
            from scipy.spatial.transform import Rotation
            # (the reference converts its packed quaternions back: R(q(R)) == R to rounding; all views in two calls --
            # one Rotation object per view costs ~0.1 ms each, more than the device spends on a small scene)
            init_angles = Rotation.from_matrix(init_rots).as_euler('zyx', degrees=True).reshape(view_num, 3)
            refi_angles = Rotation.from_matrix(np.stack([quaternion_to_rotation_unchecked(cams[v, 3:7]) for v in range(view_num)])
                                               ).as_euler('zyx', degrees=True).reshape(view_num, 3)
            for view_idx in range(view_num):
                diff_loc = math.sqrt(np.sum(np.square(init_locs[view_idx] - cams[view_idx, 0:3])))
                print('DEBUG: {}-th view loc distance changes {} unit'.format(view_idx, diff_loc))
                print('DEBUG: {}-th view angles changes {} degree'.format(view_idx, np.abs(init_angles[view_idx] - refi_angles[view_idx])))
            moved = np.sqrt(np.sum(np.square(init_tri_pts - pts), axis=0))
            for tri_idx in np.flatnonzero(moved >= 5):
                print('DEBUG: {}-th pt loc changes more than 5 unit, {} unit'.format(tri_idx, moved[tri_idx]))

    # the reference calls the name-mangled private method (ba_processor.py:267)
    _BaProcessor__execute_bundle_adjustment = execute_bundle_adjustment


class HipBaProcessor(HipBaMixin):
    """Standalone holder of the BaProcessor constructor state (ba_processor.py:22-40).  The
    incremental state machine ``process`` (ba:43-270) needs the OpenCV front end and stays in the
    reference; use ``class BaProcessor(HipBaMixin, ba_processor.BaProcessor)`` for the full pipeline."""

    def __init__(self, view_processor, key_tracker, epi_processor, tri_processor, campose_processor,
                 filter_size=10, iteration=3, damping_factor=5):
        self.curr_data_idx = 0
        self.filter_size = filter_size
        self.iteration = iteration
        self.damping_factor = damping_factor
        self.view_processor = view_processor
        self.key_tracker = key_tracker
        self.epi_processor = epi_processor
        self.tri_processor = tri_processor
        self.campose_processor = campose_processor
