#!/usr/bin/env python3
"""Clock stamps of workgroup 0 / thread 0 of ba_linearize at C3 (diagnostic, SFM_OPT_DEBUG bit 8)."""
import importlib, json, os, sys
import numpy as np
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sfm = importlib.import_module("structure-from-motion_amd")
native = sfm.native
native.init(0)
sc = sfm.scenes.make_config("C3", seed=0)
uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
with native.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
    prob.set_option(native.OPT_DEBUG, 8)
    prob.set_state(sc.cams_init, sc.pts_init)
    prob.iterate(5.0, 2)
    native.synchronize()
    raw = prob.debug_stamps(256).astype(np.int64)[192:256]
n = int(raw[63])
t = raw[:n]
print(json.dumps({"n": n, "deltas_cycles": np.diff(t).tolist(), "total": int(t[-1] - t[0]),
                  "legend": "entry, setup done, then per batch: [pass1 done, reduce+chol done] ..., loop done, flush done"}))
