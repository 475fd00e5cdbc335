#!/usr/bin/env python3
"""Where the HOST side of the drop-in BA call spends its time (no GPU: native.BaProblem is replaced by the recorder of
tests/test_host_dropin_logic.py).  The C5 sequence of bench.py: 10 views, 5000 points, one view appended per call.
  python tools/profile_dropin_host.py [--pts 5000] [--profile]"""
import argparse
import cProfile
import importlib
import os
import pstats
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pts", type=int, default=5000)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--repeats", type=int, default=5)
    args = ap.parse_args()
    sfm = importlib.import_module("structure-from-motion_amd")
    import bench
    from test_host_dropin_logic import RecordingProblem, View, Holder, KP
    sfm.native.BaProblem = RecordingProblem
    inside = [0.0]

    def timed(fn):
        def w(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                inside[0] += time.perf_counter() - t0
        return w
    for name in ("__init__", "set_cameras", "rederive_quaternions", "set_points", "append", "iterate", "get_state_rot"):
        setattr(RecordingProblem, name, timed(getattr(RecordingProblem, name)))      # the recorder's own time is not host logic
    n_views, n_pts = 10, args.pts
    sc, rots, locs, uv, birth = bench.c5_sequence(sfm, n_views, n_pts)
    keypoints = [[KP(-1.0, -1.0)] + [KP(float(uv[c][0, j]), float(uv[c][1, j])) for j in range(n_pts)] for c in range(n_views)]
    prof = cProfile.Profile() if args.profile else None
    best = {}
    for rep in range(args.repeats):
        tp = sfm.processors.HipTriangulationProcessor()
        vp, kt = Holder(), Holder()
        vp.view_list, kt.track_list = [], []
        bp = sfm.processors.HipBaProcessor(vp, kt, None, tp, None, iteration=3, damping_factor=5)
        bp.ba_verbose = False
        known = np.zeros(n_pts, dtype=bool)
        full = np.vstack((sc.pts_init, np.ones((1, n_pts))))
        for c in range(n_views):
            vp.view_list.append(View(rots[c].copy(), locs[c].copy(), sc.intrinsic.copy(), keypoints[c]))
            tr = Holder(); tr.table = np.full((n_views, n_pts + 1), -1, dtype=int); kt.track_list.append(tr)
            if c == 0:
                continue
            known[birth == c] = True
            for v in range(c + 1):
                kt.track_list[v].table[v, 1:][known] = np.flatnonzero(known)
            last = int(np.flatnonzero(known).max()) + 1
            tp.tri_pts = full[:, :last].copy()
            inside[0] = 0.0
            t0 = time.perf_counter()
            if prof: prof.enable()
            bp._BaProcessor__execute_bundle_adjustment()
            if prof: prof.disable()
            t1 = time.perf_counter() - inside[0]
            action, inside[0] = bp.ba_last_action, 0.0
            t1b = time.perf_counter()
            bp._BaProcessor__execute_bundle_adjustment()
            t2 = time.perf_counter() - inside[0]
            a, b = best.get(c, (1e9, 1e9))
            best[c] = (min(a, t1 - t0), min(b, t2 - t1b))
            if rep == args.repeats - 1:
                print("views %2d  points %5d  %-7s host %.3f ms   repeat call host %.3f ms   (best of %d)" % (c + 1, last, action, best[c][0] * 1e3, best[c][1] * 1e3, args.repeats))
    if prof:
        pstats.Stats(prof).sort_stats("tottime").print_stats(25)


if __name__ == "__main__":
    main()
