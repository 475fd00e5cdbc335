"""Latency of small scenes (the reference's own demo: 6 views x 1260 points): resident iterate(3) + sync, per-iteration
wall time and per-class kernel time."""
import importlib, sys, time
import numpy as np
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for name, args in (("demo 6x1260", (6, 1260, 1.0)), ("8x2000@0.7", (8, 2000, 0.7)), ("9x2000@0.8", (9, 2000, 0.8)), ("10x3000@0.6", (10, 3000, 0.6)), ("14x3000@0.5", (14, 3000, 0.5)), ("18x3000@0.4", (18, 3000, 0.4)), ("20x3000@0.3", (20, 3000, 0.3)), ("30x3000@0.3", (30, 3000, 0.3))):
    sc = sfm.scenes.make_scene(*args, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    for mode, dbg, graph in ((native.SCHUR_AUTO, 0, 0), (native.SCHUR_AUTO, 16384, 0), (native.SCHUR_AUTO, 0, 1), (native.SCHUR_AUTO, 64, 0), (native.SCHUR_PAIRS, 0, 0), (native.SCHUR_MFMA, 0, 0), (native.SCHUR_ROWS, 0, 0)):
        with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(native.OPT_SCHUR, mode); prob.set_option(native.OPT_DEBUG, dbg); prob.set_option(native.OPT_GRAPH, graph)
            prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); native.synchronize()
            t0 = time.perf_counter()
            for _ in range(200):
                prob.iterate(5.0, 3)
            prob.get_state()
            dt = (time.perf_counter() - t0) / 600
            t0 = time.perf_counter()
            prob.iterate(5.0, 600)
            prob.get_state()
            dt_long = (time.perf_counter() - t0) / 600
            prob.set_option(native.OPT_TIMING, 63); prob.reset_timing(); prob.iterate(5.0, 30)
            parts = {n: round(1e3 * prob.kernel_time(k)[0] / 30, 1) for k, n in enumerate(native.KERNEL_NAMES)}
            print(name, "mode", mode, "debug", dbg, "graph", graph, "kernel", prob.info(native.INFO_SCHUR_KERNEL), "us/iteration wall: 200 calls of 3 iterations %.1f, one call of 600 %.1f" % (dt * 1e6, dt_long * 1e6), parts)
