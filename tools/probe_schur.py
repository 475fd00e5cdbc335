#!/usr/bin/env python3
"""Ablation timing of the Schur MFMA kernel (diagnostic; results of the ablated runs are wrong by design).
Prints the SCHUR-class kernel time per variant at config C3."""
import importlib, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sfm = importlib.import_module("structure-from-motion_amd")
native = sfm.native
native.init(0)
sc = sfm.scenes.make_config("C3", seed=0)
uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
out = {}
with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
    prob.set_option(native.OPT_SCHUR, native.SCHUR_MFMA)
    prob.set_option(native.OPT_TIMING, 1 << native.K_SCHUR)
    for name, dbg in (("full", 0), ("no_mfma", 1), ("no_staging_loads", 4), ("barriers_only", 5)):
        prob.set_option(native.OPT_DEBUG, dbg)
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.linearize_reduce(5.0)
        native.synchronize()
        prob.reset_timing()
        for _ in range(10):
            prob.linearize_reduce(5.0)
        ms, n = prob.kernel_time(native.K_SCHUR)
        out[name] = ms / n * 1e3
    prob.set_option(native.OPT_DEBUG, 0)
print(json.dumps({"schur_class_us": out}))
