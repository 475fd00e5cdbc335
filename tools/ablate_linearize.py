"""Ablation helper for ba_linearize at C3: per-class hipEvent time of the kernel with SFM_OPT_DEBUG = 0 against another
debug value given on the command line (an experiment bit of a working build; none is defined in the committed tree)."""
import importlib, sys
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
sc = sfm.scenes.make_scene(50, 20000, 0.6, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
other = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for dbg in (0, other, 0, other):
    with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(native.OPT_DEBUG, dbg)
        prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); native.synchronize()
        prob.set_option(native.OPT_TIMING, 1 << native.KERNEL_NAMES.index("linearize")); prob.reset_timing(); prob.iterate(5.0, 30)
        ms, cnt = prob.kernel_time(native.KERNEL_NAMES.index("linearize"))
        print("debug", dbg, "linearize us %.1f" % (1e3 * ms / cnt), flush=True)
