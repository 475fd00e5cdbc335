#!/usr/bin/env python3
"""Timing of the two small hot-path kernels next to the CPU oracle, on the reference's own data files
(tests/golden/g4_tri.npz: 1538 two-view pairs; g5_pnp.npz: 882 RANSAC inliers) and on larger
synthetic batches.  Host-buffer (PCIe-inclusive) wall times through the C-ABI.  Prints one JSON line.

    python tools/bench_tri_pnp.py
"""
import importlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "oracle"))


def best(fn, reps=5):
    fn()
    t = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        t.append(time.perf_counter() - t0)
    return min(t)


def main():
    sfm = importlib.import_module("structure-from-motion_amd")
    oracle = importlib.import_module("sfm_oracle")
    native = sfm.native
    native.init(0)
    out = {}

    g = np.load(os.path.join(REPO, "tests", "golden", "g4_tri.npz"))
    uv = np.ascontiguousarray(g["cv_uv"][:, 0:2, :])
    t_gpu = best(lambda: native.tri_nonlinear(g["cv_projs"], uv, g["cv_init"], 0.5, 100))
    t0 = time.perf_counter()
    oracle.nonlinear_triangulate_vec(g["cv_init"], list(g["cv_projs"]), list(g["cv_uv"]), 0.5, 100)
    t_cpu = time.perf_counter() - t0
    out["tri_1538pts_100its"] = {"gpu_s": t_gpu, "oracle_vectorised_s": t_cpu, "reference_python_s": 11.9,
                                 "point_iterations_per_s": 1538 * 100 / t_gpu}

    # large synthetic triangulation: 1M points x 3 views x 100 iterations
    sc = sfm.scenes.make_scene(3, 1_000_000, 1.0, seed=1)
    projs = []
    for c in range(3):
        rot = sfm.geometry.quaternion_to_rotation(sc.cams_true[c, 3:7])
        loc = sc.cams_true[c, 0:3].reshape(3, 1)
        projs.append(sc.intrinsic @ np.hstack((rot.T, rot.T @ -loc)))
    uvs = np.stack([sc.uv_pix[:, sc.cam_idx == c] for c in range(3)])
    init = np.vstack((sc.pts_init, np.ones((1, sc.n_pts))))
    t_gpu = best(lambda: native.tri_nonlinear(np.stack(projs), uvs, init, 0.5, 100), reps=3)
    out["tri_1Mpts_3views_100its"] = {"gpu_s_incl_pcie": t_gpu, "point_iterations_per_s": 1e8 / t_gpu}

    g = np.load(os.path.join(REPO, "tests", "golden", "g5_pnp.npz"))
    inl = g["inliers"]
    uv, x = np.ascontiguousarray(g["pts2d"][:, inl]), np.ascontiguousarray(g["pts3d"][:, inl])
    t_gpu = best(lambda: native.pnp_nonlinear(uv, x, g["K"], g["R0"], g["C0"], 5, 200))
    out["pnp_882pts_200its"] = {"gpu_s": t_gpu, "reference_python_s": 40.5, "point_iterations_per_s": 882 * 200 / t_gpu}

    # batched PnP: 256 views x 1000 points x 200 iterations in one launch
    nv, n = 256, 1000
    offsets = (np.arange(nv + 1) * n).astype(np.int32)
    uvb = np.tile(uv[:, :n] if uv.shape[1] >= n else np.resize(uv, (3, n)), (1, nv))
    xb = np.tile(x[:, :n] if x.shape[1] >= n else np.resize(x, (4, n)), (1, nv))
    t_gpu = best(lambda: native.pnp_nonlinear_batch(offsets, uvb, xb, np.stack([g["K"]] * nv), np.stack([g["R0"]] * nv),
                                                    np.stack([g["C0"][:, 0]] * nv), 5, 200), reps=3)
    out["pnp_batch_256views_1000pts_200its"] = {"gpu_s_incl_pcie": t_gpu, "views_per_s": nv / t_gpu,
                                                "point_iterations_per_s": nv * n * 200 / t_gpu}
    # linear PnP RANSAC on the reference's fixture (300 hypotheses x 1639 points)
    import random
    random.seed(-1)
    samples = [random.sample(range(g["pts2d"].shape[1]), 6) for _ in range(300)]
    t_gpu = best(lambda: native.pnp_linear_ransac(g["pts2d"], g["pts3d"], g["K"], samples, 8.0))
    out["pnp_linear_ransac_300hyp_1639pts"] = {"gpu_s": t_gpu, "reference_python_s": 3.5}

    # DLT triangulation, 1538 pairs (reference: 0.05 s) and 1M points x 3 views
    g4 = np.load(os.path.join(REPO, "tests", "golden", "g4_tri.npz"))
    uv4 = np.ascontiguousarray(g4["cv_uv"][:, 0:2, :])
    out["tri_linear_1538pts"] = {"gpu_s": best(lambda: native.tri_linear(g4["cv_projs"], uv4)), "reference_python_s": 0.05}
    out["tri_linear_1Mpts_3views"] = {"gpu_s_incl_pcie": best(lambda: native.tri_linear(np.stack(projs), uvs), reps=3)}

    # small-scene BA (the size of the reference's stored demo result: 6 views x 1260 points), host-buffer path
    scs = sfm.scenes.make_scene(6, 1260, 0.7, seed=2)
    uvn = sfm.geometry.normalise_pixels(scs.uv_pix, scs.intrinsic)
    t_call = best(lambda: native.ba_solve(scs.n_cams, scs.pt_ptr, scs.cam_idx, uvn, scs.cams_init, scs.pts_init, 5.0, 3))
    with native.BaProblem(scs.n_cams, scs.pt_ptr, scs.cam_idx, uvn) as prob:
        prob.set_state(scs.cams_init, scs.pts_init)
        def run():
            prob.iterate(5.0, 30)
            native.synchronize()
        t_res = best(run) / 30
    out["ba_small_6x1260"] = {"ba_solve_3its_incl_setup_s": t_call, "resident_s_per_iteration": t_res}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
