"""Iteration wall time with the split-K reduce inside the data-flow launch (default) and as its own launch (SFM_OPT_DEBUG bit 16384),
dense product forced, V cameras x 4000 points."""
import importlib, sys, time
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
v = int(sys.argv[1])
sc = sfm.scenes.make_scene(v, 4000, 0.5, seed=1); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
for dbg in (0, 16384, 0, 16384):
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(native.OPT_SCHUR, native.SCHUR_MFMA); prob.set_option(native.OPT_DEBUG, dbg)
        prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 5); native.synchronize()
        t0 = time.perf_counter(); prob.iterate(5.0, 300); prob.get_state(); dt = (time.perf_counter() - t0) / 300
        print("V %d debug %5d: %.1f us per iteration" % (v, dbg, 1e6 * dt), flush=True)
