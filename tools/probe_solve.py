#!/usr/bin/env python3
"""Phase timing of one column workgroup of ba_chol_step (diagnostic clock stamps, SFM_OPT_DEBUG bit 8), C3."""
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
    prob.set_option(native.OPT_DEBUG, 8 | 1024)      # the column steps as separate launches (the data-flow launch: tools/flow_check.py stamps)
    prob.set_state(sc.cams_init, sc.pts_init)
    prob.iterate(5.0, 3)
    native.synchronize()
    raw = prob.debug_stamps(256).astype(np.int64)
    st = raw[:128].reshape(16, 8)
    bs = raw[128:128 + 64].reshape(16, 4)
names = ["entry->tiles_in_lds", "gemm+combine", "sync+load_a", "elimination", "stores"]
rows = []
for j in range(11):
    d = np.diff(st[j, :6])
    rows.append({"j": j, **{n: int(v) for n, v in zip(names, d)}, "total_cycles": int(st[j, 5] - st[j, 0])})
back = [{"b": b, "solve": int(bs[b, 1] - bs[b, 0]), "to_sync": int(bs[b, 2] - bs[b, 1]), "update+sync": int(bs[b, 3] - bs[b, 2])} for b in range(11)]
print(json.dumps({"back_solve_blocks": back}))
print(json.dumps({"clock": "s_memtime ticks = shader-clock cycles on this part (a 7.8 us column step reads as ~16.6 k ticks, i.e. ~2.1 GHz)", "steps": rows}))
