"""Reduced-solve time by camera count: the data-flow launch (default up to 52 block columns), the column steps as separate
launches (SFM_OPT_DEBUG bit 1024), and the block-row back substitution in place of the identity rows (bit 512).  Per-class
hipEvent times, us per iteration."""
import importlib, sys
import numpy as np
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for v, n, vis in ((10, 3000, 0.6), (14, 3000, 0.5), (20, 3000, 0.3), (30, 4000, 0.5), (50, 4000, 0.5), (90, 4000, 0.3), (120, 4000, 0.2), (160, 4000, 0.15), (200, 4000, 0.15), (260, 4000, 0.1), (340, 4000, 0.08)):
    sc = sfm.scenes.make_scene(v, n, vis, seed=1); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    out = []
    for dbg in (0, 1024, 512):
        with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(native.OPT_DEBUG, dbg)
            prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); native.synchronize()
            prob.set_option(native.OPT_TIMING, 1 << native.KERNEL_NAMES.index("solve")); prob.reset_timing(); prob.iterate(5.0, 20)
            ms, cnt = prob.kernel_time(native.KERNEL_NAMES.index("solve"))
            out.append(1e3 * ms / cnt)
    print("V %4d nbk %3d  solve us: data-flow launch %8.1f   column steps + identity rows %8.1f   back substitution %8.1f" % (v, (7 * v + 31) // 32, out[0], out[1], out[2]), flush=True)
