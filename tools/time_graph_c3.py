"""C3 through sfm_ba_iterate: eager launches against hipGraph replays (SFM_OPT_GRAPH), wall us per iteration."""
import importlib, sys, time
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
sc = sfm.scenes.make_scene(50, 20000, 0.6, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
for rep in range(3):
    for graph in (0, 1):
        with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(native.OPT_GRAPH, graph)
            prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 20); native.synchronize()
            t0 = time.perf_counter(); prob.iterate(5.0, 200); native.synchronize()
            print("graph", graph, "us/iteration %.1f" % ((time.perf_counter() - t0) / 200 * 1e6), "replays", prob.info(native.INFO_GRAPH_REPLAYS), flush=True)
