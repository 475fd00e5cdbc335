"""GPU tests of the device-resident growing scene: sfm_ba_append merges new cameras,
points and observations into a resident problem without uploading what is already there; the result must be
indistinguishable from a problem created from scratch with the merged structure."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _subset(sc, n_cams, n_pts):
    keep = (sc.cam_idx < n_cams) & (sc.pt_idx < n_pts)
    pt_ptr = np.zeros(n_pts + 1, dtype=np.int32)
    np.add.at(pt_ptr, sc.pt_idx[keep] + 1, 1)
    return keep, np.cumsum(pt_ptr).astype(np.int32)


@pytest.mark.parametrize("schur", ["pairs", "mfma", "rows"])
def test_append_equals_fresh_problem(hip, sfm, schur):
    sc = sfm.scenes.make_scene(7, 400, 0.7, seed=17)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    mode = {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "rows": hip.SCHUR_ROWS}[schur]
    v0, n0 = 4, 250
    keep, ptr0 = _subset(sc, v0, n0)
    with hip.BaProblem(v0, ptr0, sc.cam_idx[keep], uvn[:, keep]) as prob:
        prob.set_option(hip.OPT_SCHUR, mode)
        prob.set_state(sc.cams_init[:v0], sc.pts_init[:, :n0])
        prob.iterate(5.0, 2)
        cams_a, pts_a = prob.get_state()
        # grow: 3 cameras, 150 points, and every observation that was not in the first stage (new cameras seeing
        # old points, old cameras seeing new points, new seeing new), in scrambled order
        new = np.flatnonzero(~keep)
        new = np.random.default_rng(3).permutation(new)
        before = prob.upload_bytes
        prob.append(sc.cams_init[v0:], sc.pts_init[:, n0:], sc.cam_idx[new], sc.pt_idx[new], uvn[:, new])
        assert (prob.n_cams, prob.n_pts, prob.n_obs) == (sc.n_cams, sc.n_pts, sc.cam_idx.shape[0])
        assert prob.info(hip.INFO_N_OBS) == sc.cam_idx.shape[0] and prob.info(hip.INFO_N_CAMS) == sc.n_cams
        # only the NEW data crossed PCIe: 56 B per camera, 24 B per point, 24 B per observation (cam, pt, u, v)
        assert prob.upload_bytes - before == 56 * (sc.n_cams - v0) + 24 * (sc.n_pts - n0) + 24 * new.shape[0]
        assert prob.info(hip.INFO_MAX_TRACK) == int(np.max(np.diff(sc.pt_ptr)))
        cams_b, pts_b = prob.get_state()
        assert np.array_equal(cams_b[:v0], cams_a) and np.array_equal(cams_b[v0:], sc.cams_init[v0:])
        assert np.array_equal(pts_b[:, :n0], pts_a) and np.array_equal(pts_b[:, n0:], sc.pts_init[:, n0:])
        prob.iterate(5.0, 2)
        cams_c, pts_c = prob.get_state()
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as fresh:
        fresh.set_option(hip.OPT_SCHUR, mode)
        fresh.set_state(cams_b, pts_b)
        fresh.iterate(5.0, 2)
        cams_f, pts_f = fresh.get_state()
    # same structure, same kernels, same launch shapes; only the order of the f64 atomic accumulations (LDS
    # camera accumulators, split-K reduce, pair kernel) varies from run to run
    tol = 1e-13 if schur == "mfma" else 1e-10
    assert np.max(np.abs(cams_c - cams_f)) <= tol * max(1.0, np.max(np.abs(cams_f)))
    assert np.max(np.abs(pts_c - pts_f)) <= tol * max(1.0, np.max(np.abs(pts_f)))
    assert sfm.scenes.reprojection_rmse(cams_c, pts_c, sc) < sfm.scenes.reprojection_rmse(cams_b, pts_b, sc)


def test_append_keeps_the_reduce_inside_the_solve_launch(hip, sfm):
    """A scene of 40 cameras grows to 46 behind the same handle (sfm_ba_append plans a new problem: other tiles, other tables):
    the iterations before and after the growth leave the split-K reduce to the solve's launch (SFM_INFO_REDUCE_IN_SOLVE) and end
    where a fresh problem of the grown scene ends."""
    sc = sfm.scenes.make_scene(46, 1800, 0.5, seed=171)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    v0, n0 = 40, 1500
    keep, ptr0 = _subset(sc, v0, n0)
    with hip.BaProblem(v0, ptr0, sc.cam_idx[keep], uvn[:, keep]) as prob:
        prob.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
        prob.set_state(sc.cams_init[:v0], sc.pts_init[:, :n0])
        prob.iterate(5.0, 2)
        assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 1
        new = np.flatnonzero(~keep)
        prob.append(sc.cams_init[v0:], sc.pts_init[:, n0:], sc.cam_idx[new], sc.pt_idx[new], uvn[:, new])
        cams_b, pts_b = prob.get_state()
        prob.iterate(5.0, 2)
        assert prob.info(hip.INFO_REDUCE_IN_SOLVE) == 1
        cams_c, pts_c = prob.get_state()
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as fresh:
        fresh.set_option(hip.OPT_SCHUR, hip.SCHUR_MFMA)
        fresh.set_option(hip.OPT_DEBUG, 16384)
        fresh.set_state(cams_b, pts_b)
        fresh.iterate(5.0, 2)
        cams_f, pts_f = fresh.get_state()
    assert np.max(np.abs(cams_c - cams_f)) <= 1e-11 * max(1.0, np.max(np.abs(cams_f)))
    assert np.max(np.abs(pts_c - pts_f)) <= 1e-11 * max(1.0, np.max(np.abs(pts_f)))


def test_append_rejects_bad_input(hip, sfm):
    sc = sfm.scenes.make_scene(3, 40, 1.0, seed=2)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_state(sc.cams_init, sc.pts_init)
        with pytest.raises(ValueError, match="already observed"):
            prob.append(np.zeros((0, 7)), np.zeros((3, 0)), [1], [5], np.zeros((2, 1)))
        with pytest.raises(ValueError, match="out of range"):
            prob.append(np.zeros((0, 7)), np.zeros((3, 0)), [3], [5], np.zeros((2, 1)))
        with pytest.raises(ValueError, match="out of range"):
            prob.append(np.zeros((0, 7)), np.zeros((3, 0)), [1], [40], np.zeros((2, 1)))
        # the same new pair twice in one call
        with pytest.raises(ValueError, match="already observed"):
            prob.append(np.zeros((1, 7)), np.zeros((3, 0)), [3, 3], [5, 5], np.zeros((2, 2)))
        # a failed append leaves the problem as it was
        assert (prob.n_cams, prob.n_pts, prob.n_obs) == (sc.n_cams, sc.n_pts, sc.cam_idx.shape[0])
        assert prob.info(hip.INFO_N_CAMS) == sc.n_cams
        # an empty append is a no-op that keeps the state
        prob.append(np.zeros((0, 7)), np.zeros((3, 0)), [], [], np.zeros((2, 0)))
        cams, pts = prob.get_state()
        assert np.array_equal(cams, sc.cams_init) and np.array_equal(pts, sc.pts_init)


@pytest.mark.parametrize("shape", [(50, 4000, 0.6, "mfma"), (200, 3000, 0.15, "pairs"), (40, 2500, 0.3, "auto")])
def test_two_rank_emulation_on_one_gpu_equals_single_problem(hip, sfm, shape):
    """The multi-GPU decomposition on the real HIP path: two point shards as two resident problems on one GPU,
    their partial [S | rhs] summed by hand where RCCL's all-reduce would sit (sharding.ShardedBa), every "rank"
    solving the same reduced system and back-substituting its own points -- against the unsharded problem."""
    import torch
    n_cams, n_pts, vis, schur = shape
    sh = sfm.sharding
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=31)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    mode = {"mfma": hip.SCHUR_MFMA, "pairs": hip.SCHUR_PAIRS, "auto": hip.SCHUR_AUTO}[schur]
    bounds = sh.shard_bounds(sc.pt_ptr, 2)
    assert 0 < bounds[1] < n_pts
    engines, ranges = [], []
    try:
        for r in range(2):
            ptr_l, cam_l, uv_l, pts_l, rng = sh.local_shard(sc.pt_ptr, sc.cam_idx, uvn, sc.pts_init, bounds, r)
            eng = sh.HipShardEngine(n_cams, ptr_l, cam_l, uv_l, torch.device("cuda", 0))
            eng.prob.set_option(hip.OPT_SCHUR, mode)
            eng.set_state(sc.cams_init, pts_l)
            engines.append(eng); ranges.append(rng)
        # each engine launches on ITS OWN stream (sfm_ba_set_stream); the hand-made "all-reduce" runs on torch's
        # default stream, so the three streams are joined explicitly where RCCL would order them
        assert engines[0].stream.cuda_stream != engines[1].stream.cuda_stream
        for _ in range(3):
            bufs = []
            for e in engines:
                with e.stream_context():
                    bufs.append(e.linearize_reduce(5.0))
            torch.cuda.synchronize()
            total = bufs[0] + bufs[1]               # <- all_reduce(SUM)
            for b in bufs:
                b.copy_(total)
            torch.cuda.synchronize()
            for e in engines:
                with e.stream_context():
                    e.solve_update(5.0)
        states = [e.get_state() for e in engines]
    finally:
        for e in engines:
            e.close()
    assert np.array_equal(states[0][0], states[1][0])   # identical reduced system -> identical cameras on both ranks
    pts = np.hstack((states[0][1], states[1][1]))
    with hip.BaProblem(n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_option(hip.OPT_SCHUR, mode)
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 3)
        cams_f, pts_f = prob.get_state()
    assert np.max(np.abs(states[0][0] - cams_f)) < 1e-10 * max(1.0, np.max(np.abs(cams_f)))
    assert np.max(np.abs(pts - pts_f)) < 1e-10 * np.max(np.abs(pts_f))


def test_incremental_growth_matches_oracle(hip, sfm, oracle):
    """View-by-view growth with BA after every registration (BASELINE config 5 stand-in), the scene resident on
    the device throughout, against the oracle's block-sparse BA on the same sequence of structures."""
    sc = sfm.scenes.make_scene(6, 300, 0.8, seed=29)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    first_cam = np.full(sc.n_pts, sc.n_cams, dtype=np.int64)          # view that brings the point in = 2nd observer
    order = np.lexsort((sc.cam_idx, sc.pt_idx))
    seen = {}
    for o in order:
        seen.setdefault(int(sc.pt_idx[o]), []).append(int(sc.cam_idx[o]))
    for p, cams in seen.items():
        first_cam[p] = max(1, sorted(cams)[1])                           # needs two views to be triangulated
    # relabel the points in order of appearance so that appended points get the next indices
    perm = np.argsort(first_cam, kind="stable")
    rank = np.empty_like(perm); rank[perm] = np.arange(sc.n_pts)
    pt_new = rank[sc.pt_idx]
    pts_init = sc.pts_init[:, perm]
    born = first_cam[perm]
    active = lambda v: (sc.cam_idx <= v) & (born[pt_new] <= v)           # noqa: E731
    a1 = active(1)
    n1 = int(np.sum(born <= 1))
    o1 = np.lexsort((sc.cam_idx[a1], pt_new[a1]))
    ptr = np.zeros(n1 + 1, dtype=np.int32); np.add.at(ptr, pt_new[a1] + 1, 1); ptr = np.cumsum(ptr).astype(np.int32)
    cams_o, pts_o = sc.cams_init[:2].copy(), pts_init[:, :n1].copy()
    with hip.BaProblem(2, ptr, sc.cam_idx[a1][o1], uvn[:, a1][:, o1]) as prob:
        prob.set_state(cams_o, pts_o)
        have = a1.copy()
        for v in range(1, sc.n_cams):
            if v > 1:
                act = active(v)
                add = np.flatnonzero(act & ~have)
                n_old, n_now = cams_o.shape[0], int(np.sum(born <= v))
                prob.append(sc.cams_init[n_old:v + 1], pts_init[:, pts_o.shape[1]:n_now], sc.cam_idx[add], pt_new[add], uvn[:, add])
                cams_o = np.vstack((cams_o, sc.cams_init[n_old:v + 1]))
                pts_o = np.hstack((pts_o, pts_init[:, pts_o.shape[1]:n_now]))
                have = act
            prob.iterate(5.0, 2)
            idx = np.flatnonzero(have)
            idx = idx[np.lexsort((sc.cam_idx[idx], pt_new[idx]))]
            cams_o, pts_o = oracle.ba_sparse(cams_o, pts_o, sc.cam_idx[idx], pt_new[idx], uvn[:, idx], 5.0, 2)
            cams_g, pts_g = prob.get_state()
            assert np.max(np.abs(cams_g - cams_o)) < 1e-9 * max(1.0, np.max(np.abs(cams_o))), v
            assert np.max(np.abs(pts_g - pts_o)) < 1e-9 * np.max(np.abs(pts_o)), v
        assert prob.n_cams == sc.n_cams and prob.n_pts == int(np.sum(born <= sc.n_cams - 1))


def test_sharded_engine_append_rebinds_reduced_buffer(hip, sfm, oracle):
    """A sharded engine that grows by a camera: the all-reduce tensor must follow the new camera count (a stale
    binding would all-reduce a tensor the kernels no longer write).  One rank, the collective replaced by a check
    that the tensor the engine hands out IS the buffer the problem writes."""
    import torch
    sh = sfm.sharding
    sc = sfm.scenes.make_scene(5, 300, 0.7, seed=41)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    keep = sc.cam_idx < 4
    ptr0 = np.zeros(sc.n_pts + 1, dtype=np.int32); np.add.at(ptr0, sc.pt_idx[keep] + 1, 1); ptr0 = np.cumsum(ptr0).astype(np.int32)
    eng = sh.HipShardEngine(4, ptr0, sc.cam_idx[keep], uvn[:, keep], torch.device("cuda", 0))
    try:
        eng.set_state(sc.cams_init[:4], sc.pts_init)
        n4 = eng.reduced.numel()
        new = np.flatnonzero(~keep)
        eng.append(sc.cams_init[4:], np.zeros((3, 0)), sc.cam_idx[new], sc.pt_idx[new], uvn[:, new])
        assert eng.reduced.numel() >= n4 and eng.prob.reduced_buffer()[0] == eng.reduced.data_ptr()
        seen = []
        ba = sh.ShardedBa(eng, all_reduce=lambda t: seen.append((t.data_ptr(), float(t.abs().sum()))), world_size=1)
        ba.iterate(5.0, 2)
        torch.cuda.synchronize()
        assert len(seen) == 2 and all(ptr == eng.reduced.data_ptr() and mass > 0 for ptr, mass in seen)
        cams, pts = eng.get_state()
    finally:
        eng.close()
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 2)
    assert np.max(np.abs(cams - ocams)) < 1e-9 * np.max(np.abs(ocams))
    assert np.max(np.abs(pts - opts)) < 1e-9 * np.max(np.abs(opts))


@pytest.mark.parametrize("n_cams,mode", [(6, "pairs"), (23, "mfma"), (50, "auto")])
def test_packed_reduced_buffer_matches_oracle(hip, sfm, oracle, n_cams, mode):
    """What a rank hands to the all-reduce: the packed lower-block buffer [S | rhs] (sfm_ba_reduced_buffer,
    layout in include/sfm_hip.h / sharding.unpack_reduced) equals the oracle's partial reduced system before
    lambda is added, and nothing outside the lower blocks' valid entries is written."""
    import torch
    sh = sfm.sharding
    sc = sfm.scenes.make_scene(n_cams, 700, 0.5, seed=43)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    eng = sh.HipShardEngine(n_cams, sc.pt_ptr, sc.cam_idx, uvn, torch.device("cuda", 0))
    try:
        eng.prob.set_option(hip.OPT_SCHUR, {"pairs": hip.SCHUR_PAIRS, "mfma": hip.SCHUR_MFMA, "auto": hip.SCHUR_AUTO}[mode])
        eng.set_state(sc.cams_init, sc.pts_init)
        assert eng.reduced.numel() == sh.reduced_size(n_cams)
        with eng.stream_context():
            buf = eng.linearize_reduce(5.0)
        torch.cuda.synchronize()
        host = buf.cpu().numpy()
    finally:
        eng.close()
    s_gpu, rhs_gpu = sh.unpack_reduced(host, n_cams)
    t = oracle.ba_reduced_system(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0)
    s_or = t["S"] - 5.0 * np.eye(7 * n_cams)
    assert np.max(np.abs(s_gpu - s_or)) < 1e-11 * np.max(np.abs(s_or))
    assert np.max(np.abs(rhs_gpu - t["rhs"])) < 1e-10 * np.max(np.abs(t["rhs"]))
    # the round trip through pack_reduced reproduces every byte the device wrote: padding and upper parts are zero
    assert np.array_equal(sh.pack_reduced(s_gpu, rhs_gpu), host)


def test_pool_red_zone_mode_follows_the_environment(hip):
    """tools/gpu_round.sh runs the GPU suite a second time with SFM_POOL_REDZONE=1 (every device buffer between two
    checked 4 KB guard zones: an out-of-bounds WRITE of any kernel aborts with a message); this only checks that the
    library honours the variable."""
    import os
    assert hip.pool_redzone_active() == (os.environ.get("SFM_POOL_REDZONE", "0") == "1")


@pytest.mark.parametrize("shape", [(50, 4000, 0.6), (9, 800, 1.0), (60, 1500, 0.2)])
def test_library_owned_communicator_runs_the_sharded_loop_in_one_call(hip, sfm, oracle, shape):
    """the library owns the RCCL communicator (sfm_comm_unique_id / sfm_comm_create) and,
    once it is attached (sfm_ba_set_comm), sfm_ba_iterate issues the all-reduce of [S | rhs] itself between the partial
    reduce and the replicated solve -- K iterations in ONE C call.  One GPU here, so a communicator of ONE rank (RCCL
    refuses two ranks on a device): the collective runs, its sum is the rank's own system, and the result must equal the
    loop without a communicator and the oracle.  Through the engine (caller-bound reduced buffer) and the bare problem."""
    import torch
    n_cams, n_pts, vis = shape
    sc = sfm.scenes.make_scene(n_cams, n_pts, vis, seed=41)
    uvn = sfm.geometry.normalise_pixels(sc.uv_pix, sc.intrinsic)
    ocams, opts = oracle.ba_sparse(sc.cams_init, sc.pts_init, sc.cam_idx, sc.pt_idx, uvn, 5.0, 3)
    with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
        prob.set_state(sc.cams_init, sc.pts_init)
        prob.iterate(5.0, 3)
        cams0, pts0 = prob.get_state()
    comm = hip.Comm(1, 0, hip.comm_unique_id())
    try:
        with hip.BaProblem(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn) as prob:
            prob.set_comm(comm)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            cams1, pts1 = prob.get_state()
            prob.set_comm(None)
            prob.set_state(sc.cams_init, sc.pts_init)
            prob.iterate(5.0, 3)
            cams2, pts2 = prob.get_state()
        eng = sfm.sharding.HipShardEngine(sc.n_cams, sc.pt_ptr, sc.cam_idx, uvn, torch.device("cuda", 0))
        try:
            eng.attach_comm(comm)
            with pytest.raises(hip.SfmHipError, match="still hold"):
                comm.close()                               # a communicator cannot be destroyed under a problem that holds it
            eng.set_state(sc.cams_init, sc.pts_init)
            sfm.sharding.ShardedBa(eng, None, 1).iterate(5.0, 3)
            cams3, pts3 = eng.get_state()
        finally:
            eng.close()
    finally:
        comm.close()
    scale_c, scale_p = np.max(np.abs(ocams)), np.max(np.abs(opts))
    for cams, pts in ((cams1, pts1), (cams2, pts2), (cams3, pts3)):
        assert np.max(np.abs(cams - ocams)) < 1e-9 * scale_c and np.max(np.abs(pts - opts)) < 1e-9 * scale_p
        assert np.max(np.abs(cams - cams0)) < 1e-11 * scale_c and np.max(np.abs(pts - pts0)) < 1e-11 * scale_p
    with pytest.raises(ValueError):
        hip.Comm(1, 0, b"short")
    with pytest.raises(ValueError):
        hip.Comm(2, 5, hip.comm_unique_id())          # rank outside the world: SFM_E_SHAPE
