import importlib, sys, time, os
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for name, args in (("6x1260", (6, 1260, 1.0)), ("8x2000", (8, 2000, 0.7)), ("10x3000", (10, 3000, 0.6)), ("10x5000", (10, 5000, 1.0)), ("14x3000", (14, 3000, 0.5)), ("18x3000", (18, 3000, 0.4))):
    sc = sfm.scenes.make_scene(*args, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); native.synchronize()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter(); prob.iterate(5.0, 600); prob.get_state(); best = min(best, (time.perf_counter() - t0) / 600)
        print("cap", os.environ.get("SFM_SCHUR_MAX_CHUNKS"), name, "kernel", prob.info(native.INFO_SCHUR_KERNEL), "%.1f us per iteration" % (best * 1e6), flush=True)
