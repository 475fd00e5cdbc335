"""C3, wall us per iteration of the same kernels through three host paths: sfm_ba_iterate on the library's stream,
the split calls (linearize_reduce / solve_update) on the library's stream, and bench.py's path (HipShardEngine on a
torch stream + ShardedBa)."""
import importlib, sys, time
import torch
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
sc = sfm.scenes.make_scene(50, 20000, 0.6, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
dev = torch.device("cuda", 0)
for rep in range(3):
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 20); native.synchronize()
        t0 = time.perf_counter(); prob.iterate(5.0, 200); native.synchronize()
        a = (time.perf_counter() - t0) / 200 * 1e6
        t0 = time.perf_counter()
        for _ in range(200):
            prob.linearize_reduce(5.0); prob.solve_update(5.0)
        prob.flush(); native.synchronize()
        b = (time.perf_counter() - t0) / 200 * 1e6
    eng = sfm.sharding.HipShardEngine(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn, dev)
    ba = sfm.sharding.ShardedBa(eng, None, 1)
    eng.set_state(sc.cams_init, sc.pts_init); ba.iterate(5.0, 20); torch.cuda.synchronize(dev)
    t0 = time.perf_counter(); ba.iterate(5.0, 200); torch.cuda.synchronize(dev)
    c = (time.perf_counter() - t0) / 200 * 1e6
    eng.close()
    print("iterate %.1f   split calls %.1f   engine on a torch stream %.1f" % (a, b, c), flush=True)
