"""GPU parity tests: the HIP path (through the C-ABI) against the CPU oracle on the same seeded
inputs and against the golden fixtures captured from the reference.

Tolerances: the contract (BASELINE.json north_star) is 1e-6 relative reprojection error; float64
kernels with a different summation order agree far tighter, so vectors are held to 1e-9 relative
(max-norm) and RMSE to 1e-9 relative, with the 1e-6 contract asserted separately.
"""
import numpy as np
import pytest

from conftest import load_golden

pytestmark = pytest.mark.gpu

TOL = 1e-9


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def uv_norm_of(g):
    kinv = np.linalg.inv(g["K"])
    hom = np.vstack((g["uv_pix"], np.ones((1, g["uv_pix"].shape[1]))))
    cam = kinv @ hom
    return cam[0:2] / cam[2:3]


# ---- unit hooks -------------------------------------------------------------------------------
def test_jac_cam_golden(hip):
    g = load_golden("g1_jac_cam.npz")
    jp, st = hip.jac_cam(g["R"], g["C"], g["X"])
    assert np.all(st == 0)
    assert rel(jp, g["Jp"]) < 1e-11
    worst = max(rel(a, b) for a, b in zip(jp, g["Jp"]))
    assert worst < 1e-9


def test_jac_pt_golden(hip):
    g = load_golden("g2_jac_pt.npz")
    jx = hip.jac_pt(g["projs"], g["X"])
    assert rel(jx, g["Jx"]) < 1e-12


def test_quaternion_helpers_golden(hip):
    g = load_golden("g3_quat.npz")
    q, st = hip.rot_to_quat(g["R"])
    assert np.all(st == 0) and rel(q, g["q"]) < 1e-14
    rot, st = hip.quat_to_rot(g["q"])
    assert np.all(st == 0) and rel(rot, g["R_back"]) < 1e-14
    # accept / reject set of verify_rotation_mat at its 1e-8 one-sided thresholds
    _, st = hip.rot_to_quat(g["verify_cases"])
    assert np.array_equal(st != hip.E_BAD_ROTATION, g["verify_verdict"])
    _, st = hip.quat_to_rot(g["q_scaled"])
    assert np.array_equal(st == 0, g["q_scaled_ok"])


def test_rotation_error_statuses(hip):
    bad = np.diag([1.0, 1.0, 1.0 + 1e-3])
    _, st = hip.rot_to_quat(bad)
    assert st[0] == hip.E_BAD_ROTATION
    half_turn = np.diag([1.0, -1.0, -1.0])          # trace = -1 -> qw = 0
    _, st = hip.rot_to_quat(half_turn)
    assert st[0] in (hip.E_QW_ZERO, hip.E_SQRT_DOMAIN)


# ---- nonlinear triangulation ---------------------------------------------------------------------
def test_tri_known_answer(hip):
    """triangulation_processor.py:415-473 literal case."""
    g = load_golden("g4_tri.npz")
    out = hip.tri_nonlinear(g["lit_projs"][0:2], g["lit_uv"][0:2, 0:2, :], g["lit_linear"], 0.5, 300)
    assert rel(out, g["lit_whole"]) < TOL
    assert np.allclose(out[:, 0], [-0.034700141239, -0.005983101498, 2.021981140992, 1.0], atol=1e-10)
    out3 = hip.tri_nonlinear(g["lit_projs"], g["lit_uv"][:, 0:2, :], g["lit_linear"], 0.5, 50)
    assert rel(out3, g["lit_three_view"]) < TOL


def test_tri_opencv_fixture(hip):
    g = load_golden("g4_tri.npz")
    uv = g["cv_uv"][:, 0:2, :]
    for key, lam, its in (("cv_its1", 0.5, 1), ("cv_its10", 0.5, 10), ("cv_lam10_its5", 10, 5), ("cv_its100", 0.5, 100)):
        if key not in g.files:
            continue
        out = hip.tri_nonlinear(g["cv_projs"], uv, g["cv_init"], lam, its)
        assert rel(out, g[key]) < TOL, key
        assert np.array_equal(out[3], g["cv_init"][3])          # W row untouched


@pytest.mark.parametrize("nv", [2, 3, 5])
def test_tri_synthetic(hip, nv):
    g = load_golden("g4_tri.npz")
    out = hip.tri_nonlinear(g["syn%d_projs" % nv], g["syn%d_uv" % nv][:, 0:2, :], g["syn%d_init" % nv], 0.5, 20)
    assert rel(out, g["syn%d_out" % nv]) < TOL


def test_tri_large_vs_oracle_and_ragged_sizes(hip, oracle, sfm):
    sc = sfm.scenes.make_scene(4, 5003, 1.0, seed=21)          # not a multiple of the block size
    projs, uv = [], []
    for c in range(4):
        rot = sfm.geometry.quaternion_to_rotation(sc.cams_true[c, 3:7])
        loc = sc.cams_true[c, 0:3].reshape(3, 1)
        projs.append(sc.intrinsic @ np.hstack((rot.T, rot.T @ -loc)))
        uv.append(sc.uv_pix[:, sc.cam_idx == c])
    init = np.vstack((sc.pts_init, np.ones((1, sc.n_pts))))
    got = hip.tri_nonlinear(np.stack(projs), np.stack(uv), init, 0.5, 30)
    want = oracle.nonlinear_triangulate_vec(init, projs, uv, 0.5, 30)
    assert rel(got, want) < TOL
    # idempotence-style property at size: 0 iterations returns the input bit-exactly
    same = hip.tri_nonlinear(np.stack(projs), np.stack(uv), init, 0.5, 0)
    assert np.array_equal(same, init)
    # empty input
    assert hip.tri_nonlinear(np.stack(projs), np.zeros((4, 2, 0)), np.zeros((4, 0)), 0.5, 5).shape == (4, 0)


# ---- nonlinear PnP -----------------------------------------------------------------------------
def test_pnp_opencv_fixture(hip):
    g = load_golden("g5_pnp.npz")
    inl = g["inliers"]
    uv, x = g["pts2d"][:, inl], g["pts3d"][:, inl]
    for its in (1, 2, 10, 200):
        if "C_its%d" % its not in g.files:
            continue
        r, c = hip.pnp_nonlinear(uv, x, g["K"], g["R0"], g["C0"], 5, its)
        assert rel(r, g["R_its%d" % its]) < TOL, its
        assert rel(c, g["C_its%d" % its]) < TOL, its
    r, c = hip.pnp_nonlinear(uv, x, g["K"], g["R0"], g["C0"], 5, 200)
    assert np.allclose(c[:, 0], [-1.690676720621, 0.054300873096, 0.658193978912], atol=1e-9)
    assert np.linalg.norm(c - g["loc_truth"]) < 0.1          # the reference's own acceptance bound


def test_pnp_quirk_modes_vs_oracle(hip, oracle):
    g = load_golden("g5_pnp.npz")
    uv, x = g["syn1_uv"], g["syn1_X"]
    for quirks in (0, 1, 2, 3):
        r, c = hip.pnp_nonlinear(uv, x, g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 15, quirks)
        ro, co = oracle.nonlinear_pnp(uv, x, g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 15, quirks)
        assert rel(r, ro) < TOL and rel(c, co) < TOL, quirks


def test_pnp_batch_matches_single(hip):
    g = load_golden("g5_pnp.npz")
    uvs = [g["syn%d_uv" % c] for c in (1, 2, 3)]
    xs = [g["syn%d_X" % c] for c in (1, 2, 3)]
    offsets = np.cumsum([0] + [u.shape[1] for u in uvs]).astype(np.int32)
    rot, loc, st = hip.pnp_nonlinear_batch(offsets, np.hstack(uvs), np.hstack(xs), np.stack([g["syn_K"]] * 3),
                                           np.stack([g["syn%d_R0" % c] for c in (1, 2, 3)]),
                                           np.stack([g["syn%d_C0" % c][:, 0] for c in (1, 2, 3)]), 5, 25)
    assert np.all(st == 0)
    for i, c in enumerate((1, 2, 3)):
        assert rel(rot[i], g["syn%d_R" % c]) < TOL
        assert rel(loc[i], g["syn%d_C" % c][:, 0]) < TOL


def test_pnp_invalid_rotation_raises(hip):
    g = load_golden("g5_pnp.npz")
    bad = g["syn1_R0"] * 1.01
    with pytest.raises(ValueError):
        hip.pnp_nonlinear(g["syn1_uv"], g["syn1_X"], g["syn_K"], bad, g["syn1_C0"], 5, 3)


# ---- bundle adjustment -----------------------------------------------------------------------------
BA_CASES = ["3x50", "5x200v80", "6x120v60", "8x300v50"]


@pytest.mark.parametrize("name", ["3x50", "5x200v80", "6x120v60"])
def test_ba_residual_jacobian_golden(hip, name):
    g = load_golden("g6_ba_%s.npz" % name)
    r, jp, jx = hip.ba_residual_jacobian(g["cams_init"].shape[0], g["pt_ptr"], g["cam_idx"], uv_norm_of(g),
                                         g["cams_init"], g["pts_init"])
    assert rel(r, g["lin_r"]) < 1e-11
    assert rel(jp, g["lin_Jp"]) < 1e-11
    assert rel(jx, g["lin_Jx"]) < 1e-11


@pytest.mark.parametrize("name", ["3x50", "5x200v80", "6x120v60"])
@pytest.mark.parametrize("mode", ["pairs", "mfma", "rows"])
def test_ba_reduced_system_golden(hip, name, mode):
    g = load_golden("g6_ba_%s.npz" % name)
    mode_id = {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "rows": hip.SCHUR_ROWS}[mode]
    s, rhs = hip.ba_reduced_system(g["cams_init"].shape[0], g["pt_ptr"], g["cam_idx"], uv_norm_of(g),
                                   g["cams_init"], g["pts_init"], 5.0, schur_mode=mode_id)
    assert rel(s, g["lin_S"]) < 1e-11
    assert rel(rhs, g["lin_rhs"]) < 1e-10
    assert np.array_equal(s, s.T)


@pytest.mark.parametrize("name", BA_CASES)
@pytest.mark.parametrize("mode", ["pairs", "mfma", "rows"])
def test_ba_iterations_golden(hip, name, mode):
    """1, 2 and 3 iterations against the reference's own results."""
    g = load_golden("g6_ba_%s.npz" % name)
    uvn = uv_norm_of(g)
    with hip.BaProblem(g["cams_init"].shape[0], g["pt_ptr"], g["cam_idx"], uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "rows": hip.SCHUR_ROWS}[mode])
        prob.set_state(g["cams_init"], g["pts_init"])
        for it in (1, 2, 3):
            prob.iterate(5.0, 1)
            cams, pts = prob.get_state()
            assert rel(cams, g["cams_it%d" % it]) < TOL, (name, it)
            assert rel(pts, g["pts_it%d" % it]) < TOL, (name, it)


def test_ba_config2_reference_golden(hip, sfm):
    """BASELINE config 2 (5 x 2000 dense), 3 iterations, against the reference run (329 s on CPU)."""
    g = load_golden("g6_ba_C2.npz")
    cams, pts = hip.ba_solve(5, g["pt_ptr"], g["cam_idx"], uv_norm_of(g), g["cams_init"], g["pts_init"], 5.0, 3)
    assert rel(cams, g["cams_it3"]) < TOL and rel(pts, g["pts_it3"]) < TOL
    sc = sfm.scenes.make_config("C2", seed=0)
    assert np.array_equal(sc.cam_idx, g["cam_idx"])
    rm = sfm.scenes.reprojection_rmse(cams, pts, sc)
    assert abs(rm - float(g["rmse_it3"])) / float(g["rmse_it3"]) < 1e-6       # the north-star contract
    assert abs(rm - float(g["rmse_it3"])) / float(g["rmse_it3"]) < 1e-9


@pytest.mark.parametrize("mode", ["pairs", "mfma", "rows"])
def test_ba_vs_oracle_midsize_ragged(hip, oracle, sfm, mode):
    """20 cams x 3000 points at 30 % visibility: ragged tracks (2..~12), some longer than the lane group."""
    sc = sfm.scenes.make_scene(20, 3001, 0.3, seed=8)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "rows": hip.SCHUR_ROWS}[mode])
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 3)
        cams, pts = prob.get_state()
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    assert rel(cams, ocams) < TOL and rel(pts, opts) < TOL
    r0 = sfm.scenes.reprojection_rmse(sc.cams_init, sc.pts_init, sc)
    r3 = sfm.scenes.reprojection_rmse(cams, pts, sc)
    ro = sfm.scenes.reprojection_rmse(ocams, opts, sc)
    assert r3 < r0 and abs(r3 - ro) / ro < 1e-9


def test_ba_config3_full_size_vs_oracle(hip, oracle, sfm):
    """BASELINE config 3 (50 x 20 000 @ 60 %): 1, 2 and 3 iterations (the reference's default count,
    ba_processor.py:24) against the block-sparse oracle for both Schur kernels, plus size-independent
    properties: RMSE decreases monotonically, quaternions stay unit, and the two Schur kernels agree."""
    sc = sfm.scenes.make_config("C3", seed=0)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    trace = []
    oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3, trace=trace)
    results = {}
    for mode in (hip.SCHUR_MFMA, hip.SCHUR_PAIRS, hip.SCHUR_ROWS):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, mode)
            assert prob.info(hip.INFO_SCHUR_KERNEL) == mode
            prob.set_state(sc.cams_init, sc.pts_init)
            rm = [sfm.scenes.reprojection_rmse(sc.cams_init, sc.pts_init, sc)]
            for it in range(3):
                prob.iterate(5.0, 1)
                c, p = prob.get_state()
                assert rel(c, trace[it][0]) < TOL and rel(p, trace[it][1]) < TOL, (mode, it)
                rm.append(sfm.scenes.reprojection_rmse(c, p, sc))
            assert all(b < a for a, b in zip(rm, rm[1:])), rm
            ro = sfm.scenes.reprojection_rmse(trace[2][0], trace[2][1], sc)
            assert abs(rm[-1] - ro) / ro < 1e-9          # north-star contract is 1e-6
            assert np.allclose(np.linalg.norm(c[:, 3:7], axis=1), 1.0, atol=1e-14)
            results[mode] = (c, p)
    assert rel(results[hip.SCHUR_MFMA][0], results[hip.SCHUR_PAIRS][0]) < TOL
    assert rel(results[hip.SCHUR_MFMA][1], results[hip.SCHUR_PAIRS][1]) < TOL
    assert rel(results[hip.SCHUR_MFMA][0], results[hip.SCHUR_ROWS][0]) < TOL
    assert rel(results[hip.SCHUR_MFMA][1], results[hip.SCHUR_ROWS][1]) < TOL


# ---- BASELINE config 4: 200 cameras x 100 000 points @ 15 %, point blocks sharded over 8 GPUs ------------
def test_ba_config4_one_gpu_share_vs_oracle(hip, oracle, sfm):
    """One GPU's share of config 4 (200 cameras x 12 500 points @ 15 %, seed 0): 1, 2 and 3 iterations against the
    block-sparse oracle, with the kernel AUTO picks at this size (the sparse row-panel product), with the 18-camera
    tile product and with the dense MFMA product forced.  Covers the 44-step Cholesky, the big back substitution (7V = 1400 > 416) and the
    global-accumulator mode of ba_linearize at the size the 8-GPU configuration runs them."""
    sc = sfm.scenes.make_scene(200, 12500, 0.15, seed=0)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    trace = []
    oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3, trace=trace)
    for mode in (hip.SCHUR_AUTO, hip.SCHUR_PAIRS, hip.SCHUR_MFMA):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, mode)
            prob.set_state(sc.cams_init, sc.pts_init)
            for it in range(3):
                prob.iterate(5.0, 1)
                c, p = prob.get_state()
                assert rel(c, trace[it][0]) < TOL and rel(p, trace[it][1]) < TOL, (mode, it)
            if mode == hip.SCHUR_AUTO:
                assert prob.info(hip.INFO_SCHUR_KERNEL) == hip.SCHUR_ROWS
    r3 = sfm.scenes.reprojection_rmse(c, p, sc)
    ro = sfm.scenes.reprojection_rmse(trace[2][0], trace[2][1], sc)
    assert abs(r3 - ro) / ro < 1e-9


def test_ba_config4_full_size_one_gpu(hip, oracle, sfm):
    """All of config 4 (200 x 100 000 @ 15 %, M = 3.0 M observations) resident on ONE GPU: one iteration against
    the oracle (~30 s of NumPy), then the size-independent properties over 3 iterations -- monotone RMSE, unit
    quaternions, dense MFMA product == sparse product to 1e-9."""
    sc = sfm.scenes.make_config("C4", seed=0)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 1)
    results = {}
    for mode in (hip.SCHUR_ROWS, hip.SCHUR_MFMA):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, mode)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 1)
            c, p = prob.get_state()
            assert rel(c, ocams) < TOL and rel(p, opts) < TOL, mode
            rm = [sfm.scenes.reprojection_rmse(sc.cams_init, sc.pts_init, sc), sfm.scenes.reprojection_rmse(c, p, sc)]
            for _ in range(2):
                prob.iterate(5.0, 1)
                c, p = prob.get_state()
                rm.append(sfm.scenes.reprojection_rmse(c, p, sc))
            assert all(b < a for a, b in zip(rm, rm[1:])), rm
            assert np.allclose(np.linalg.norm(c[:, 3:7], axis=1), 1.0, atol=1e-14)
            results[mode] = (c, p)
    assert rel(results[hip.SCHUR_MFMA][0], results[hip.SCHUR_ROWS][0]) < TOL
    assert rel(results[hip.SCHUR_MFMA][1], results[hip.SCHUR_ROWS][1]) < TOL


def test_ba_edge_cases(hip, oracle, sfm):
    sc = sfm.scenes.make_scene(3, 40, 0.8, seed=2)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    # zero iterations: state returned bit-exactly
    cams, pts = hip.ba_solve(3, sc.pt_ptr, sc.cam_idx, uvn, sc.cams_init, sc.pts_init, 5.0, 0)
    assert np.array_equal(cams, sc.cams_init) and np.array_equal(pts, sc.pts_init)
    # a point with no observation keeps its position (D_p = lambda I, ex = 0; ba:359)
    pt_ptr = np.concatenate((sc.pt_ptr, [sc.pt_ptr[-1]])).astype(np.int32)
    pts_plus = np.hstack((sc.pts_init, [[1.0], [2.0], [9.0]]))
    cams, pts = hip.ba_solve(3, pt_ptr, sc.cam_idx, uvn, sc.cams_init, pts_plus, 5.0, 2)
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    assert np.array_equal(pts[:, -1], [1.0, 2.0, 9.0])
    assert rel(cams, ocams) < TOL and rel(pts[:, :-1], opts) < TOL
    # a non-unit quaternion far from a rotation raises like convert_quaternion_to_rotation (ba:323)
    bad = sc.cams_init.copy()
    bad[1, 3:7] *= 1.05
    with pytest.raises(ValueError):
        hip.ba_solve(3, sc.pt_ptr, sc.cam_idx, uvn, bad, sc.pts_init, 5.0, 1)
    # unsorted cameras inside a track are rejected
    ci = sc.cam_idx.copy()
    ci[[0, 1]] = ci[[1, 0]]
    with pytest.raises(ValueError):
        hip.ba_solve(3, sc.pt_ptr, ci, uvn, sc.cams_init, sc.pts_init, 5.0, 1)


@pytest.mark.parametrize("n_cams,mode", [(9, "pairs"), (30, "mfma"), (120, "auto")])
def test_ba_per_iteration_cost_statistics(hip, oracle, sfm, n_cams, mode):
    """sfm_ba_get_stats: cost[i] = sum |b - f|^2 (normalised coordinates, ba_processor.py:376) at the linearisation
    point of iteration i, accumulated on the device -- against the oracle's residuals at the oracle's states.  Covers
    the fused back-substitution + linearisation launch (9 / 30 cameras) and the separate launches (120 cameras: the
    fused kernel's LDS budget is exceeded), and the history reset on a state upload."""
    sc = sfm.scenes.make_scene(n_cams, 900, 0.4, seed=61)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    trace = []
    oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 4, trace=trace)
    states = [(sc.cams_init, sc.pts_init)] + trace[:3]
    want = [float(np.sum(oracle.obs_terms_vec(c, p, sc.cam_idx, sc.pt_idx, uvn)[0] ** 2)) for c, p in states]
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "auto": hip.SCHUR_AUTO}[mode])
        prob.set_state(sc.cams_init, sc.pts_init)
        assert prob.get_stats().shape == (0,)
        prob.iterate(5.0, 3)
        prob.iterate(5.0, 1)
        got = prob.get_stats()
        assert got.shape == (4,) and rel(got, np.array(want)) < 1e-10
        assert np.all(np.diff(got) < 0)                       # the damped Gauss-Newton steps reduce the cost here
        cams, pts = prob.get_state()
        assert rel(cams, trace[3][0]) < TOL and rel(pts, trace[3][1]) < TOL
        prob.set_state(sc.cams_init, sc.pts_init)             # a new state starts a new history
        prob.iterate(5.0, 1)
        again = prob.get_stats()
        assert again.shape == (1,) and rel(again, np.array(want[:1])) < 1e-10


def test_ba_fused_and_separate_launches_agree(hip, sfm):
    """The back substitution riding in the next linearisation's launch (default) against the two separate kernels
    (SFM_OPT_DEBUG bit 16): same arithmetic, same state."""
    sc = sfm.scenes.make_scene(14, 1500, 0.5, seed=62)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    out = []
    for dbg in (0, 16):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 4)
            out.append(prob.get_state())
    assert rel(out[0][0], out[1][0]) < 1e-13 and rel(out[0][1], out[1][1]) < 1e-13


@pytest.mark.parametrize("n_cams,n_pts,vis", [(9, 900, 0.8), (14, 1500, 0.5), (37, 2500, 0.4), (131, 4000, 0.1)])
def test_ba_inverse_rows_against_block_back_substitution(hip, oracle, sfm, n_cams, n_pts, vis):
    """dp = X y with X = L^-T carried through the column steps as identity rows (default) against the block-row back
    substitution (SFM_OPT_DEBUG bit 512), and both against the oracle: 2, 4, 9 and 29 block columns."""
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=300 + n_cams)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    out = []
    for dbg in (0, 512):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            out.append(prob.get_state())
    for cams, pts in out:
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL
    assert rel(out[0][0], out[1][0]) < 1e-11 and rel(out[0][1], out[1][1]) < 1e-11


@pytest.mark.parametrize("n_cams,n_pts,vis", [(9, 600, 0.8), (10, 800, 0.6), (14, 900, 0.5), (19, 900, 0.5), (50, 1500, 0.6), (73, 1200, 0.3),
                                               (74, 1200, 0.3), (150, 2500, 0.15), (237, 2500, 0.15)])
def test_ba_data_flow_solve_against_column_steps_and_oracle(hip, oracle, sfm, n_cams, n_pts, vis):
    """The reduced solve as one persistent data-flow launch (csrc/sfm_ba_flow.h: 2 to 52 block columns, the default) against the
    column steps as separate launches (SFM_OPT_DEBUG bit 1024) and both against the oracle: 2, 3, 4, 5 block columns (no task of
    the third sub-diagonal / the first closer), 11 (C3), 16 / 17 (the 16-flag poll window), 33 and 52 (several tasks per workgroup)."""
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=500 + n_cams)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    out = []
    for dbg in (0, 1024):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            out.append(prob.get_state())
    for cams, pts in out:
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL
    assert rel(out[0][0], out[1][0]) < 1e-11 and rel(out[0][1], out[1][1]) < 1e-11


def test_ba_data_flow_solve_random_shapes(hip, oracle, sfm):
    """Sixteen seeded random shapes between 9 and 140 cameras (every number of block columns has its own task table, its own
    first closer, its own ragged last block of 7 V mod 32 rows), random visibility and damping, 1-3 iterations, against the oracle."""
    rng = np.random.default_rng(20240)
    for case in range(16):
        n_cams = int(rng.integers(9, 141))
        n_pts = int(rng.integers(300, 1500))
        vis = float(rng.uniform(max(0.08, 4.0 / n_cams), 0.9))
        lam = float(rng.choice([0.5, 5.0, 50.0]))
        iters = int(rng.integers(1, 4))
        sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=700 + case)
        uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
        want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, lam, iters)
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(lam, iters)
            cams, pts = prob.get_state()
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL, (case, n_cams, n_pts, vis, lam, iters, rel(cams, want_c), rel(pts, want_p))


@pytest.mark.parametrize("n_cams", [12, 50, 120])
def test_ba_data_flow_solve_is_bitwise_repeatable(hip, sfm, n_cams):
    """Every block of the data-flow solve is produced by one task with a fixed summation order, whatever the timing of the
    hand-overs: in deterministic mode (fixed order in the other kernels as well) ten solves of the same state agree bit for bit,
    also while a second problem runs its own data-flow solves on another stream."""
    import torch
    sc = sfm.scenes.make_scene(n_cams, 1500, 0.4, seed=900 + n_cams)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    other = sfm.scenes.make_scene(40, 1200, 0.5, seed=77)
    uvo = sfm.geometry.normalise_pixels(other.uv_pix, other.intrinsic)
    side = torch.cuda.Stream()
    ref = None
    with hip.BaProblem(other.n_cams, other.pt_ptr, other.cam_idx, uvo) as noise:
        noise.set_stream(side.cuda_stream)
        noise.set_state(other.cams_init, other.pts_init)
        for rep in range(10):
            with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
                prob.set_option(hip.OPT_DETERMINISTIC, 1)
                prob.set_state(sc.cams_init, sc.pts_init)
                if rep % 2:
                    noise.iterate(5.0, 4)
                prob.iterate(5.0, 3)
                cams, pts = prob.get_state()
            if ref is None:
                ref = (cams.copy(), pts.copy())
            assert np.array_equal(cams, ref[0]) and np.array_equal(pts, ref[1])
        noise.get_state()


def test_ba_data_flow_solve_gives_up_instead_of_hanging(hip, oracle, sfm):
    """Every wait of the data-flow launch is bounded.  With SFM_OPT_DEBUG bit 8192 the chain never announces W_1: the tasks that
    need it give up after 20 000 polls, raise the abort word, every other wait of the launch ends, the launch drains and the
    state read reports SFM_E_HIP -- no hang, no fault.  The same handle then solves correctly again (flags carry the solve's
    epoch, the abandoned solve closed its own)."""
    import time
    sc = sfm.scenes.make_scene(40, 1200, 0.5, seed=8)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_DEBUG, 8192)
        prob.set_state(sc.cams_init, sc.pts_init)
        t0 = time.time()
        prob.iterate(5.0, 1)
        with pytest.raises(hip.SfmHipError):
            prob.get_state()
        assert time.time() - t0 < 20.0
        prob.set_option(hip.OPT_DEBUG, 0)
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 2)
        cams, pts = prob.get_state()
    assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL


def test_ba_data_flow_solves_on_four_streams_share_the_chip(hip, sfm):
    """Four problems of 120 cameras enqueue their iterations on four streams at once: 4 x 225 workgroups that each want a CU of
    their own meet 256 CUs, so no launch has all its workgroups resident.  Tasks are taken by ticket in table order (a task
    waits only for tasks taken before it), so every launch advances with whatever share of the chip it holds; the results are
    those of the same problems run one after the other, bit for bit in deterministic mode."""
    import torch
    sc = [sfm.scenes.make_scene(120, 1500, 0.2, seed=40 + q) for q in range(4)]
    uv = [sfm.geometry.normalise_pixels(x.uv_pix, x.intrinsic) for x in sc]
    alone = []
    for x, u in zip(sc, uv):
        with hip.BaProblem(x.n_cams, x.pt_ptr, x.cam_idx, u) as prob:
            prob.set_option(hip.OPT_DETERMINISTIC, 1)
            prob.set_state(x.cams_init, x.pts_init)
            prob.iterate(5.0, 4)
            alone.append(prob.get_state())
    streams = [torch.cuda.Stream() for _ in range(4)]
    probs = [hip.BaProblem(x.n_cams, x.pt_ptr, x.cam_idx, u) for x, u in zip(sc, uv)]
    try:
        for prob, st, x in zip(probs, streams, sc):
            prob.set_option(hip.OPT_DETERMINISTIC, 1)
            prob.set_stream(st.cuda_stream)
            prob.set_state(x.cams_init, x.pts_init)
        for rep in range(4):                  # one iteration per problem and round: the four queues stay full together
            for prob in probs:
                prob.iterate(5.0, 1)
        for prob, want in zip(probs, alone):
            cams, pts = prob.get_state()
            assert np.array_equal(cams, want[0]) and np.array_equal(pts, want[1])
    finally:
        for prob in probs:
            prob.close()


@pytest.mark.parametrize("n_cams,n_pts,vis", [(37, 2500, 0.5), (50, 3000, 0.6), (73, 2500, 0.4), (120, 3000, 0.3), (234, 2500, 0.15)])
def test_ba_reduce_inside_the_solve_launch_against_its_own_launch_and_oracle(hip, oracle, sfm, n_cams, n_pts, vis):
    """sfm_ba_iterate on one GPU with the dense product leaves the split-K reduce to the first tasks of the data-flow launch
    (camera sums per camera, 8 rows of a block of S per task, D_0 by the chain itself: csrc/sfm_ba_flow.h, FlowRed).  Against
    SFM_OPT_DEBUG bit 16384 (ba_schur_reduce as its own launch) and the oracle, cameras, points and the per-iteration cost;
    SFM_INFO_REDUCE_IN_SOLVE says which path ran; a second run and graph replays take the same path (1e-12: outside the
    deterministic mode ba_linearize's LDS accumulation order varies from run to run)."""
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=300 + n_cams)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    res = {}
    for name, dbg, graph in (("in the solve", 0, 0), ("again", 0, 0), ("own launch", 16384, 0), ("graph", 0, 1)):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_option(hip.OPT_GRAPH, graph)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            cams, pts = prob.get_state()
            assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == (0 if dbg else 1)
            if graph:      # the first iterations captured the two graphs (the camera slots alternate); the same again is replays
                prob.set_state(sc.cams_init, sc.pts_init)
                prob.iterate(5.0, 3)
                again = prob.get_state()
                assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 1
                assert prob.info(hip.INFO_GRAPH_REPLAYS) >= (1 if n_cams <= 102 else 0)      # (graphs need the fused linearisation: up to 102 cameras)
                assert rel(again[0], cams) < 1e-12 and rel(again[1], pts) < 1e-12
            res[name] = (cams, pts, prob.get_stats()[:3].copy())
    for name, (cams, pts, cost) in res.items():
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL, name
        assert np.all(cost > 0) and rel(cost, res["own launch"][2]) < 1e-12, name
    assert rel(res["in the solve"][0], res["again"][0]) < 1e-12 and rel(res["in the solve"][1], res["again"][1]) < 1e-12
    assert rel(res["in the solve"][0], res["own launch"][0]) < 1e-11
    assert rel(res["graph"][0], res["in the solve"][0]) < 1e-12


def test_ba_reduce_inside_the_solve_launch_random_shapes(hip, oracle, sfm):
    """Fourteen seeded random shapes between 37 and 200 cameras with the dense product forced (every camera count has its own
    split of the slabs over the tiles, its own cameras straddling the 32-row blocks and its own ragged last block), random
    visibility, damping and iteration count: the reduce inside the solve's launch against the oracle and against its own launch
    (growth behind the same handle: tests/test_gpu_append.py)."""
    rng = np.random.default_rng(5150)
    for case in range(14):
        n_cams = int(rng.integers(37, 201))
        n_pts = int(rng.integers(400, 2500))
        vis = float(rng.uniform(max(0.08, 6.0 / n_cams), 0.8))
        lam = float(rng.choice([0.5, 5.0, 50.0]))
        iters = int(rng.integers(1, 4))
        sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=1200 + case)
        uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
        want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, lam, iters)
        got = []
        for dbg in (0, 16384):
            with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
                prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
                prob.set_option(hip.OPT_DEBUG, dbg)
                prob.set_state(sc.cams_init, sc.pts_init)
                prob.iterate(lam, iters)
                got.append(prob.get_state())
                assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == (0 if dbg else 1), (case, n_cams)
        info = (case, n_cams, n_pts, vis, lam, iters)
        assert rel(got[0][0], want_c) < TOL and rel(got[0][1], want_p) < TOL, info
        assert rel(got[0][0], got[1][0]) < 1e-11 and rel(got[0][1], got[1][1]) < 1e-11, info


def test_ba_reduce_stays_its_own_launch_where_the_solve_cannot_take_it(hip, oracle, sfm):
    """Up to 36 cameras (one to three tiles share the slabs: the own launch measures faster), the sparse products, the
    deterministic mode, the column-step solve, more than 234 cameras (ba_linearize then adds to S with global atomics: no per-workgroup sums to take) and the split entry points sfm_ba_linearize_reduce / sfm_ba_solve_update (whose caller all-reduces S in
    between) keep ba_schur_reduce: SFM_INFO_REDUCE_IN_SOLVE reads 0 and the results are the oracle's."""
    for n_cams, mode, dbg, det in ((8, "mfma", 0, 0), (14, "mfma", 0, 0), (20, "mfma", 0, 0), (50, "pairs", 0, 0), (50, "mfma", 1024, 0), (50, "mfma", 0, 1), (237, "mfma", 0, 0)):
        sc = sfm.scenes.make_scene(n_cams, 1500, 0.5, seed=500 + n_cams)
        uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
        want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_SCHUR, {"mfma": hip.SCHUR_MFMA, "pairs": hip.SCHUR_PAIRS}[mode])
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_option(hip.OPT_DETERMINISTIC, det)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 2)
            cams, pts = prob.get_state()
            assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 0
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL
    sc = sfm.scenes.make_scene(50, 1500, 0.5, seed=550)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
        prob.set_state(sc.cams_init, sc.pts_init)
        for _ in range(2):
            prob.linearize_reduce(5.0)
            assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 0
            prob.solve_update(5.0)
        cams, pts = prob.get_state()
    assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL


def test_ba_reduce_inside_the_solve_launch_on_four_streams(hip, sfm):
    """The same crowding with the split-K reduce inside the launches (dense product forced, default mode): 4 x 225 workgroups whose
    first tasks are the reduce's meet 256 CUs; the chain is whichever workgroup of a launch arrives first, every task waits only for
    the chain and for tasks taken before it.  Results as when the problems run alone (1e-11: the accumulation order inside
    ba_linearize varies from run to run outside the deterministic mode)."""
    import torch
    sc = [sfm.scenes.make_scene(90, 1500, 0.3, seed=60 + q) for q in range(4)]
    uv = [sfm.geometry.normalise_pixels(x.uv_pix, x.intrinsic) for x in sc]
    alone = []
    for x, u in zip(sc, uv):
        with hip.BaProblem(x.n_cams, x.pt_ptr, x.cam_idx, u) as prob:
            prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
            prob.set_state(x.cams_init, x.pts_init)
            prob.iterate(5.0, 4)
            alone.append(prob.get_state())
            assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 1
    streams = [torch.cuda.Stream() for _ in range(4)]
    probs = [hip.BaProblem(x.n_cams, x.pt_ptr, x.cam_idx, u) for x, u in zip(sc, uv)]
    try:
        for prob, st, x in zip(probs, streams, sc):
            prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
            prob.set_stream(st.cuda_stream)
            prob.set_state(x.cams_init, x.pts_init)
        for rep in range(4):
            for prob in probs:
                prob.iterate(5.0, 1)
        for prob, want in zip(probs, alone):
            cams, pts = prob.get_state()
            assert rel(cams, want[0]) < 1e-11 and rel(pts, want[1]) < 1e-11
    finally:
        for prob in probs:
            prob.close()


@pytest.mark.parametrize("n_cams", [2, 3, 5, 6, 8, 9])
def test_ba_small_system_kernel_against_block_steps_and_oracle(hip, oracle, sfm, n_cams):
    """P <= 56 (up to eight cameras; nine with SFM_OPT_DEBUG bit 256) solves in the single-launch whole-matrix kernel;
    bit 64 sends the same system through the block column steps.  Both against the oracle (odd and even P, P = 63 with
    its identity padding row, the smallest scene)."""
    sc = sfm.scenes.make_scene(n_cams, 700, 0.8, seed=70 + n_cams)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    want_c, want_p = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    out = []
    for dbg in (256, 64):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_DEBUG, dbg)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            out.append(prob.get_state())
    for cams, pts in out:
        assert rel(cams, want_c) < TOL and rel(pts, want_p) < TOL
    assert rel(out[0][0], out[1][0]) < 1e-11 and rel(out[0][1], out[1][1]) < 1e-11


@pytest.mark.parametrize("shape", [(6, 800, 1.0), (30, 3000, 0.5), (60, 4000, 0.15)])
def test_ba_graph_replay_equals_eager_launches(hip, sfm, shape):
    """SFM_OPT_GRAPH: the iteration body captured as a hipGraph and replayed == the same kernels launched eagerly
    (bitwise in deterministic mode, 1e-12 otherwise); options, state changes and appends in between are honoured."""
    sc = sfm.scenes.make_scene(*shape, seed=91)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    for det in (1, 0):
        out, stats, replays = [], [], []
        for graph in (0, 1):
            with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
                prob.set_option(hip.OPT_DETERMINISTIC, det)
                prob.set_option(hip.OPT_GRAPH, graph)
                prob.set_state(sc.cams_init, sc.pts_init)
                prob.iterate(5.0, 6)
                prob.iterate(5.0, 2)                       # a second call: starts eagerly, then replays
                prob.iterate(4.0, 3)                       # another lambda: new graphs
                stats.append(prob.get_stats())
                out.append(prob.get_state())
                replays.append(prob.info(hip.INFO_GRAPH_REPLAYS))
        assert replays[0] == 0 and replays[1] >= 4
        if det:
            assert np.array_equal(out[0][0], out[1][0]) and np.array_equal(out[0][1], out[1][1])
            assert np.array_equal(stats[0], stats[1])
        else:
            assert rel(out[0][0], out[1][0]) < 1e-12 and rel(out[0][1], out[1][1]) < 1e-12


def test_ba_split_phases_equal_iterate(hip, sfm):
    """linearize_reduce + solve_update (the multi-GPU split) == iterate on one rank."""
    sc = sfm.scenes.make_scene(7, 500, 0.5, seed=4)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as a, hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as b:
        for prob in (a, b):
            prob.set_option(hip.OPT_SCHUR, hip.SCHUR_PAIRS)
            prob.set_state(sc.cams_init, sc.pts_init)
        a.iterate(5.0, 2)
        for _ in range(2):
            b.linearize_reduce(5.0)
            b.solve_update(5.0)
        ca, pa = a.get_state()
        cb, pb = b.get_state()
    assert rel(ca, cb) < 1e-13 and rel(pa, pb) < 1e-13


# ---- drop-in processors ---------------------------------------------------------------------------
def test_processors_drop_in(hip, sfm, oracle, capsys):
    g4 = load_golden("g4_tri.npz")
    tp = sfm.processors.HipTriangulationProcessor()
    whole = tp.triangulate([g4["lit_projs"][0], g4["lit_projs"][1]], [g4["lit_uv"][0], g4["lit_uv"][1]], 0.5, 300)
    assert rel(whole, g4["lit_whole"]) < 1e-8            # linear DLT (host SVD) + device refinement
    jx = tp.construct_jacobian_matrix(g4["lit_linear"], [g4["lit_projs"][0], g4["lit_projs"][1]], 2)
    assert rel(jx, oracle.jac_pt(g4["lit_linear"][:, 0], [g4["lit_projs"][0], g4["lit_projs"][1]])) < 1e-12
    # falsy arguments select the instance defaults (quirk Q4)
    tp2 = sfm.processors.HipTriangulationProcessor(0.5, 10)
    a = tp2.nonlinear_triangulate(g4["cv_init"], list(g4["cv_projs"]), list(g4["cv_uv"]), 0, 0)
    assert rel(a, g4["cv_its10"]) < TOL
    with pytest.raises(ValueError):
        tp.triangulate([g4["lit_projs"][0]], [g4["lit_uv"][0], g4["lit_uv"][1]])

    g5 = load_golden("g5_pnp.npz")
    cp = sfm.processors.HipCamposeProcessor(None, 5, 25)
    r, c = cp.nonlinear_estimate_cam_pose_pnp(g5["syn2_uv"], g5["syn2_X"], g5["syn_K"], g5["syn2_R0"], g5["syn2_C0"])
    assert rel(r, g5["syn2_R"]) < TOL and rel(c, g5["syn2_C"]) < TOL and c.shape == (3, 1)
    with pytest.raises(ValueError):
        cp.nonlinear_estimate_cam_pose_pnp(g5["syn2_uv"][:, :5], g5["syn2_X"], g5["syn_K"], g5["syn2_R0"], g5["syn2_C0"])
    jp = cp.construct_jacobian_matrix(g5["syn2_R0"], g5["syn2_C0"], g5["syn2_X"][:, 0:1])
    assert rel(jp, oracle.jac_cam(g5["syn2_R0"], g5["syn2_C0"], g5["syn2_X"][:, 0])) < 1e-11

    # BaProcessor.__execute_bundle_adjustment on duck-typed views / tracks (the reference's data contract)
    g6 = load_golden("g6_ba_6x120v60.npz")

    class KP:
        def __init__(self, x, y):
            self.pt = (x, y)

    class View:
        def __init__(self, rot, loc, k, kps):
            self.rot, self.loc, self.k, self.key_pts = rot, loc, k, kps

        def update_cam_pose(self, rot, loc):
            self.rot, self.loc = rot, loc

    class Track:
        pass

    class Holder:
        pass

    nv, npt = g6["cams_init"].shape[0], g6["pts_init"].shape[1]
    views, tracks = [], []
    for cidx in range(nv):
        sel = np.flatnonzero(g6["cam_idx"] == cidx)
        kps = [KP(-1.0, -1.0)] + [KP(float(g6["uv_pix"][0, o]), float(g6["uv_pix"][1, o])) for o in sel]
        tr = Track()
        tr.table = np.full((nv, len(kps)), -1, dtype=int)
        tr.table[cidx, 1:] = g6["pt_idx"][sel]
        tracks.append(tr)
        views.append(View(sfm.geometry.quaternion_to_rotation(g6["cams_init"][cidx, 3:7]),
                          g6["cams_init"][cidx, 0:3].reshape(3, 1).copy(), g6["K"].copy(), kps))
    vp, kt, tpp = Holder(), Holder(), sfm.processors.HipTriangulationProcessor()
    vp.view_list, kt.track_list = views, tracks
    tpp.tri_pts = np.vstack((g6["pts_init"], np.ones((1, npt))))
    bp = sfm.processors.HipBaProcessor(vp, kt, None, tpp, cp, iteration=3, damping_factor=5)
    bp._BaProcessor__execute_bundle_adjustment()
    out = capsys.readouterr().out
    assert out.count("view loc distance changes") == nv
    assert rel(tpp.tri_pts[0:3], g6["pts_it3"]) < TOL
    assert np.all(tpp.tri_pts[3] == 1.0)
    for cidx in range(nv):
        assert rel(views[cidx].loc[:, 0], g6["cams_it3"][cidx, 0:3]) < TOL
        assert rel(sfm.geometry.rotation_to_quaternion(views[cidx].rot)[:, 0], g6["cams_it3"][cidx, 3:7]) < TOL


# ---- paths the headline config does not touch --------------------------------------------------------
@pytest.mark.parametrize("n_cams", [160, 240])
@pytest.mark.parametrize("mode", ["pairs", "mfma", "rows"])
def test_ba_many_cameras_global_accumulator_path(hip, oracle, sfm, mode, n_cams):
    """160 cameras: cameras + accumulators exceed the 64 KB LDS budget of ba_linearize, so only the
    accumulators stay in LDS (mode 1) and ba_backsub reads cameras from global memory; 240 cameras: not
    even the accumulators fit (global-atomic mode 0).  The MFMA product has 9 / 14 camera blocks and tracks
    are short and ragged (8 % visibility)."""
    sc = sfm.scenes.make_scene(n_cams, 600, 0.08, seed=31)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "rows": hip.SCHUR_ROWS}[mode])
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 2)
        cams, pts = prob.get_state()
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    assert rel(cams, ocams) < TOL and rel(pts, opts) < TOL


def test_ba_auto_mode_picks_a_correct_kernel_on_sparse_scene(hip, oracle, sfm):
    sc = sfm.scenes.make_scene(60, 2000, 0.1, seed=32)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    cams, pts = hip.ba_solve(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn, sc.cams_init, sc.pts_init, 5.0, 2)
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    assert rel(cams, ocams) < TOL and rel(pts, opts) < TOL


def test_ba_repeatable_and_state_reset(hip, sfm):
    """Same problem object, state reset between runs: the MFMA path has no atomics on its inner loop and
    the reductions it does use commute to ~1 ulp; results must agree to 1e-13 run to run."""
    sc = sfm.scenes.make_scene(12, 800, 0.5, seed=33)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    runs = []
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        for _ in range(3):
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            runs.append(prob.get_state())
    for c, p in runs[1:]:
        assert rel(c, runs[0][0]) < 1e-13 and rel(p, runs[0][1]) < 1e-13


@pytest.mark.parametrize("shape", [(40, 6000, 0.5), (170, 1500, 0.12)])
def test_ba_deterministic_mode_is_bitwise_repeatable(hip, oracle, sfm, shape):
    """SFM_OPT_DETERMINISTIC: fixed summation order (one wave per ba_linearize workgroup, atomic-free dense Schur
    product, single-writer reduce).  Fresh problems and a re-run on the same problem give bit-identical states, and
    the result is still the oracle's to 1e-9.  The second shape keeps only the camera accumulators in LDS."""
    n_cams, n_pts, vis = shape
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=71)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    runs = []
    for rep in range(2):
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_option(hip.OPT_DETERMINISTIC, 1)
            assert prob.info(hip.INFO_SCHUR_KERNEL) == hip.SCHUR_MFMA
            for _ in range(2):
                prob.set_state(sc.cams_init, sc.pts_init)
                prob.iterate(5.0, 4)
                runs.append(prob.get_state())
    for c, p in runs[1:]:
        assert np.array_equal(c, runs[0][0]) and np.array_equal(p, runs[0][1])
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 4)
    assert rel(runs[0][0], ocams) < TOL and rel(runs[0][1], opts) < TOL


def test_tri_many_views_global_projection_path(hip, oracle, sfm):
    """More views than the LDS staging of the projections holds (512): projections come from global memory."""
    rng = np.random.default_rng(3)
    nv, m = 520, 5
    sc = sfm.scenes.make_scene(3, m, 1.0, seed=34)
    projs, uv = [], []
    for v in range(nv):
        c = v % 3
        rot = sfm.geometry.quaternion_to_rotation(sc.cams_true[c, 3:7])
        loc = sc.cams_true[c, 0:3].reshape(3, 1) + rng.normal(0, 1e-3, (3, 1))
        projs.append(sc.intrinsic @ np.hstack((rot.T, rot.T @ -loc)))
        uv.append(sc.uv_pix[:, sc.cam_idx == c] + rng.normal(0, 0.1, (2, m)))
    init = np.vstack((sc.pts_init, np.ones((1, m))))
    got = hip.tri_nonlinear(np.stack(projs), np.stack(uv), init, 0.5, 5)
    want = oracle.nonlinear_triangulate_vec(init, projs, uv, 0.5, 5)
    assert rel(got, want) < TOL


def test_pnp_tiny_inputs(hip, oracle):
    g = load_golden("g5_pnp.npz")
    for n in (1, 2, 7):
        uv, x = g["syn1_uv"][:, :n], g["syn1_X"][:, :n]
        r, c = hip.pnp_nonlinear(uv, x, g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 6)
        ro, co = oracle.nonlinear_pnp(uv, x, g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 6)
        assert rel(r, ro) < TOL and rel(c, co) < TOL, n
    # zero iterations: R(q0) of the normalised initial quaternion and C0 come back (campose:458-459)
    r, c = hip.pnp_nonlinear(g["syn1_uv"], g["syn1_X"], g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 0)
    ro, co = oracle.nonlinear_pnp(g["syn1_uv"], g["syn1_X"], g["syn_K"], g["syn1_R0"], g["syn1_C0"], 5, 0)
    assert rel(r, ro) < 1e-14 and np.array_equal(c, co)
