import importlib, sys
sys.path.insert(0, ".")
sfm = importlib.import_module("structure-from-motion_amd"); native = sfm.native; native.init(0)
for name, args in (("C3", (50, 20000, 0.6)), ("C4shard", (200, 12500, 0.15)), ("demo", (6, 1260, 1.0)), ("mid", (60, 8000, 0.3))):
    sc = sfm.scenes.make_scene(*args, seed=0); uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    for mode in (native.SCHUR_ROWS, native.SCHUR_PAIRS, native.SCHUR_MFMA):
        with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(native.OPT_SCHUR, mode); prob.set_option(native.OPT_TIMING, (1 << native.K_SCHUR) | (1 << native.K_REDUCE))
            prob.set_state(sc.cams_init, sc.pts_init); prob.iterate(5.0, 3); prob.reset_timing(); prob.iterate(5.0, 10)
            ms, n = prob.kernel_time(native.K_SCHUR); ms2, n2 = prob.kernel_time(native.K_REDUCE); print(name, "mode", mode, "kernel us", round(1e3 * ms / n, 1), "reduce us", round(1e3 * ms2 / max(n2, 1), 1), "(0.0: the reduce rides in the solve launch)" if n2 == 0 else "")
