"""Shader-clock phases of the single-launch small-system solve (SFM_OPT_DEBUG bit 8; 100 MHz constant clock)."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for v in (6, 9):
    sc = sfm.scenes.make_scene(v, 1500, 0.9, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(native.OPT_DEBUG, 8 | 256)
        prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); prob.get_state()
        st = prob.debug_stamps(32).astype(np.int64)
        t0 = st[0]
        print("V", v, "cycles: loaded %d factorised %d backsolved %d end %d" % tuple(st[1:5] - t0), "chain take-overs", (st[8:8 + (7 * v + 8) // 8] - t0).tolist())
