#!/usr/bin/env python3
"""bench.py — BA LM-iterations/s on BASELINE.json's headline config (C3: 50 cams x 20 000 points,
60 % visibility, lambda = 5) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N ...        (N > 1, no launcher environment: starts the line above as a CHILD process itself,
                                         before torch is imported or a GPU touched, and relays rank 0's line + exit code)

A "step" is one damped Gauss-Newton iteration (ba_processor.py:297-406): linearise all
observations, form the Schur-reduced camera system, solve it, update cameras, back-substitute
points.  Inputs are resident in HBM before the timed region.

  --config C3 (default, the headline): for N > 1 the scene is WEAK-scaled: 50 cameras x (20 000 N) points,
      each rank owns a contiguous ~20 000-point shard and one RCCL all-reduce of [S | rhs] per iteration
      joins them; `value` counts 20 000-point shard-iterations per second over all ranks (= N x global
      iterations/s).  For N > 1 the line ALSO carries `strong_scaled`: the metric's own fixed 50 x 20 000 scene split
      over the ranks by sharding.shard_bounds, its own timed regions, global iterations/s (`--scaling strong` swaps
      the two; `--single-scaling` times only one).  `hbm.per_rank_algorithmic_frac_of_peak` at every N.
  --config C4 (BASELINE config 4): STRONG-scaled: the 200-camera x 100 000-point scene is fixed, its
      points are split over the N ranks by sharding.shard_bounds (balanced by camera pairs); `value` is
      global LM iterations per second of that one scene, "scaling": "strong".
  --config TRI / PNP (the two other solvers of the path): per rank 10^6 points x 3 views x 100
      iterations of nonlinear triangulation / 256 views x 1 000 points x 200 iterations of nonlinear PnP, sharded by
      point / by view with no data-path collective; a step is one pass over the rank's batch; `value` is
      point-iterations per second over all ranks (weak scaling).
  --config C5 (BASELINE config 5 stand-in, N = 1): the incremental loop of ba_processor.py:137-267 on a synthetic
      10-view x 5 000-point sequence through the drop-in classes: per view PnP (RANSAC + nonlinear), triangulation of
      the new points (DLT + nonlinear) and a global BA; a step is one registered view; `value` is views per second.

Timing: W warm-up steps (every kernel class bracketed: which one dominates), the event-bracket calibration, then an
untimed PRE-ROLL of the same steps until ~30 ms of GPU work have passed (`preroll_steps` in the line; `--no-preroll`
skips it): a step is 0.3 ms, so after the host-side calibration the clocks are down and K = 20 steps would be over before
they are back up.  Then R (= --repeats, 5) timed regions back to back, each barrier + synchronize, EXACTLY K steps,
synchronize + barrier, MAX over ranks; `value` / `ms_per_step` are the MEDIAN region, `value_runs` / `ms_per_step_runs`
list all of them (the error bar of a 5.6 ms measurement).
N > 1: the replicated solves must leave every rank with the same cameras; `max_camera_deviation_across_ranks` records it.
A non-zero deviation is not a lost run: the measurement is repeated with reduce(dst=0) + broadcast in place of the
all-reduce (`--collective reduce_broadcast` forces that form) and `replica_drift` keeps the first figures.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, hipEvent-timed inside the timed
region) and `cpu_baseline` (the NumPy oracle on the host cores, N = 1 only).
"""
import argparse
import importlib
import json
import os
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD spec; v_mfma_f64_16x16x4 measured 77.8, v_fma_f64 63.3: profiles/microbench_fp64_r01.txt)
FP64_FMA_MEASURED_TFLOPS = 63.3
LDS_ADD_PEAK_TADDS = 4.8    # ds_add_f64, conflict-free, all 256 CUs (profiles/microbench_fp64_r01.txt; 2.4 at random addresses)
TIMING_STRIDE = 0      # 0: bracket the dominant kernel with hipEvents on every n-th launch, n = min(10, steps // 5): at least five samples
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md
LAMBDA = 5.0                # reference default damping_factor (ba_processor.py:24)
EMPTY_KERNEL_MS = 0.0015    # what a kernel that does nothing takes by itself (rocprofv3): the part of the calibration bracket that is NOT overhead

# algorithmic flops (FMA = 2) per point-iteration of the two small solvers, counted from the kernels' arithmetic
# (csrc/sfm_core.hip tri_nonlinear_kernel / pnp_nonlinear_kernel; reciprocals counted as 8: seed + third-order step):
#   triangulation, per view: projection 21 + reciprocal 8 + 1 + Jacobian 24 + residual 4 + J^T J 24 + J^T e 12 = 94;
#                  per point: damping 3 + adjugate 18 + determinant 5 + reciprocal 8 + update 21 = 55
#   PnP under the reference's row-stacking quirk Q1 (only the u-row of every point but the last reaches the normal equations,
#   campose_processor.py:404-405): projection 21 + reciprocal 8 + d 3 + u-row of J_C 12 + its quaternion part 48 + residual 2 +
#   J^T J 56 + J^T e 14 = 164 (the reference also forms the v-rows it then overwrites: 213 with them)
TRI_FLOPS_PER_VIEW, TRI_FLOPS_PER_POINT = 94.0, 55.0
PNP_FLOPS_PER_POINT = 164.0
PREROLL_MS = 30.0        # untimed GPU work right before a timed region (clock ramp; see run_ba)


def algorithmic_costs(n_cams, pt_ptr, n_obs):
    """Per-iteration algorithmic flops / bytes of each kernel class (restated in DESIGN.md)."""
    k = np.diff(pt_ptr).astype(np.float64)
    n_pts = k.shape[0]
    p = 7 * n_cams
    return {
        "linearize": dict(bound="hbm", bytes=20.0 * n_obs + 28.0 * n_pts, flops=567.0 * n_obs + 60.0 * n_pts),
        "schur": dict(bound="mfma", flops=float(np.sum(294.0 * k * (k - 1) / 2 + 168.0 * k)), bytes=168.0 * n_obs,
                      lds_adds=float(np.sum(49.0 * k * (k - 1) / 2 + 28.0 * k))),      # one ds_add_f64 per lower-triangle entry of every camera pair
        "solve": dict(bound="mfma", flops=p ** 3 / 3.0 + 2.0 * p * p, bytes=8.0 * p * p),
        "backsub": dict(bound="hbm", bytes=20.0 * n_obs + 52.0 * n_pts, flops=300.0 * n_obs + 60.0 * n_pts),
        "prep": dict(bound="hbm", bytes=56.0 * n_cams + 152.0 * n_cams, flops=100.0 * n_cams),
        "reduce": dict(bound="hbm", bytes=8.0 * (p * p / 2.0 + p), flops=0.0),     # [S | rhs] written once
    }


def steady_state_launches(n_cams, schur_kernel, fused, debug=0, reduce_in_solve=False):
    """Kernels ONE steady-state iteration launches and how often (what `hbm.measured_bytes_per_iteration` may sum):
    one-time kernels (ba_structure, ba_cam_major_*, ba_cam_prep) and the stand-alone ba_backsub of a fused iteration
    are not among them."""
    p = 7 * n_cams
    nbk = (p + 31) // 32
    launches = {"ba_linearize": 1, schur_kernel: 1}
    if schur_kernel == "ba_schur_rows":
        launches["ba_schur_rows_reduce"] = 1    # (also adds ba_linearize's camera accumulators and sums its cost)
    elif not reduce_in_solve:      # (sfm_ba_iterate on one GPU leaves the dense product's split-K reduce to the data-flow launch: SFM_INFO_REDUCE_IN_SOLVE)
        launches["ba_schur_reduce"] = 1
    if not fused:
        launches["ba_backsub"] = 1
    if p <= 56:
        launches["ba_small_solve"] = 1
    elif nbk <= 52 and not (debug & (512 | 1024)):
        launches["ba_chol_flow"] = 1            # the data-flow launch (csrc/sfm_ba_flow.h): factorisation, dp = X y, camera update
        if debug & 2048:
            launches["ba_inv_apply"] = 1
    else:
        launches["ba_chol_step"] = nbk
        if nbk <= 52 and not (debug & 512):
            launches["ba_inv_apply"] = 1
        else:
            groups = (nbk + 11) // 12
            launches["ba_back_solve"] = groups
            launches["ba_back_update"] = groups - 1
    return launches


class Holder:
    pass


class KeyPoint:
    __slots__ = ("pt",)

    def __init__(self, x, y):
        self.pt = (x, y)


class View:
    """Duck-typed stand-in for view_processor.View (the fields the hot path touches: ba:287-288, 318, 340, 413)."""

    def __init__(self, rot, loc, k, kps):
        self.rot, self.loc, self.k, self.key_pts = rot, loc, k, kps
        self.cam_proj = k @ np.hstack((rot.T, rot.T @ -loc))

    def update_cam_pose(self, rot, loc):
        self.rot, self.loc = rot, loc
        self.cam_proj = self.k @ np.hstack((rot.T, rot.T @ -loc))


def drop_in_path(sfm, scene):
    """What BaProcessor.process sees (ba_processor.py:267): `_BaProcessor__execute_bundle_adjustment` of the drop-in
    class on duck-typed views / track tables of this scene, the reference's default 3 iterations.  First call =
    observation list from the track tables + upload + solve; repeat calls = the scene is resident, only the 7 V
    camera doubles go up.  Wall-clock on the host, PCIe and Python included; never used as `value`."""
    n_views = scene.n_cams

    class SelfRow:      # stands in for KeyTrack.table (n_views x n_keys): the BA path only ever reads row `v`
        def __init__(self, v, row):
            self.v, self.row = v, row

        def __getitem__(self, idx):
            assert idx[0] == self.v
            return self.row[idx[1]]

    views, tracks = [], []
    rots = sfm.geometry.quaternions_to_rotations(scene.cams_init[:, 3:7])
    for c in range(n_views):
        sel = np.flatnonzero(scene.cam_idx == c)
        kps = [KeyPoint(-1.0, -1.0)] + [KeyPoint(float(scene.uv_pix[0, o]), float(scene.uv_pix[1, o])) for o in sel]
        row = np.full(len(kps), -1, dtype=np.int64)
        row[1:] = scene.pt_idx[sel]
        tr = Holder(); tr.table = SelfRow(c, row)
        tracks.append(tr)
        views.append(View(rots[c].copy(), scene.cams_init[c, 0:3].reshape(3, 1).copy(), scene.intrinsic.copy(), kps))
    vp, kt, tp = Holder(), Holder(), sfm.processors.HipTriangulationProcessor()
    vp.view_list, kt.track_list = views, tracks
    tp.tri_pts = np.vstack((scene.pts_init, np.ones((1, scene.n_pts))))
    bp = sfm.processors.HipBaProcessor(vp, kt, None, tp, None, iteration=3, damping_factor=LAMBDA)
    bp.ba_verbose = False
    times, uploads, actions = [], [], []
    for _ in range(5):
        before = bp.ba_upload_bytes
        t0 = time.perf_counter()
        bp._BaProcessor__execute_bundle_adjustment()
        times.append(time.perf_counter() - t0)
        uploads.append(bp.ba_upload_bytes - before)
        actions.append(bp.ba_last_action)
    bp.ba_release()
    return {"scene": "%d views x %d points, %d observations" % (n_views, scene.n_pts, scene.n_obs),
            "first_call_s": times[0], "repeat_call_s": float(np.median(times[1:])), "actions": actions,
            "upload_bytes_first_call": uploads[0], "upload_bytes_repeat_call": uploads[1], "iterations_per_call": 3}


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return int(max([i.get("num_threads", 1) for i in threadpool_info()] or [1]))
    except Exception:
        return os.cpu_count() or 1


def cpu_leg(kind, config, n_iters, repeats, pts=None):
    """One CPU-baseline leg, run in THIS process (the single-thread legs are started as a child with
    OPENBLAS_NUM_THREADS=1 set before NumPy loads).  Returns a dict with the per-repeat seconds."""
    sfm = importlib.import_module("structure-from-motion_amd")
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    oracle = importlib.import_module("sfm_oracle")
    if kind == "dense":                      # the line-faithful dense restatement (ba_processor.py:274-439) at a size it can hold
        scene = sfm.scenes.make_scene(5, 400, 1.0, seed=0)
    else:
        cfg = sfm.scenes.CONFIGS[config]
        scene = sfm.scenes.make_scene(cfg["n_cams"], pts or cfg["n_pts"], cfg["visibility"], seed=0)
    uvn = sfm.geometry.normalise_pixels(scene.uv_pix, scene.intrinsic)
    fn = oracle.ba_dense if kind == "dense" else oracle.ba_sparse
    secs, state3 = [], None
    for _ in range(repeats):
        trace = []
        t0 = time.perf_counter()
        fn(scene.cams_init, scene.pts_init, scene.cam_idx, scene.pt_idx, uvn, LAMBDA, n_iters, trace=trace)
        secs.append(time.perf_counter() - t0)
        if len(trace) >= 3:
            state3 = trace[2]
    return {"seconds": secs, "iterations": n_iters, "threads": host_threads(), "state3": state3,
            "scene": "%d cams x %d pts, %d observations" % (scene.n_cams, scene.n_pts, scene.n_obs)}


def cpu_leg_child(kind, config, n_iters, repeats, threads, pts=None):
    """Run a CPU leg in a child process with the BLAS thread count pinned (it has to be set before NumPy loads)."""
    env = dict(os.environ)
    for var in ("OPENBLAS_NUM_THREADS", "OMP_NUM_THREADS", "MKL_NUM_THREADS"):
        env[var] = str(threads)
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-leg", kind, "--config", config, "--cpu-iters", str(n_iters),
           "--cpu-repeats", str(repeats)] + (["--pts", str(pts)] if pts else [])
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    if out.returncode != 0:
        return {"error": out.stderr[-400:]}
    return json.loads(out.stdout.strip().splitlines()[-1])


def setup_dist(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run for N > 1)" % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False (no CPU fallback)")
    # SFM_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo (through host memory) where RCCL sits -- the N > 1 control
    # flow of this file on a one-GPU box; the line it prints is marked and is not a measurement
    rehearsal = os.environ.get("SFM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # Under torch.distributed.run (RANK set) the RCCL path is taken even with one rank, so the same code
    # that runs on 8 GPUs can be exercised on a 1-GPU box; a plain `python bench.py` skips the process group.
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)
    ctx = Holder()
    ctx.torch, ctx.dist, ctx.world, ctx.rank, ctx.device, ctx.use_dist, ctx.rehearsal = torch, dist, world, rank, device, use_dist, rehearsal
    ctx.coll_device = torch.device("cpu") if rehearsal else device      # where the small bookkeeping collectives live

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()

    def max_over_ranks(x):
        if not use_dist:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=ctx.coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    ctx.sync, ctx.max_over_ranks = sync, max_over_ranks
    ctx.backend = ("gloo (REHEARSAL)" if rehearsal else "nccl (RCCL)") if use_dist else "none (plain python, no process group)"
    return ctx


def finish(ctx, out):
    if ctx.use_dist:
        ctx.dist.barrier()
        ctx.dist.destroy_process_group()
    if ctx.rank == 0:
        print(json.dumps(out))


# ================================================================================================================
# BA (C3 / C4)
# ================================================================================================================
def make_all_reduce(ctx, mode):
    """The per-iteration exchange of the packed [S | rhs] buffer.
    "allreduce": one RCCL all-reduce (SUM, f64), in place.  "reduce_broadcast": reduce to rank 0, then broadcast -- every rank
    then holds rank 0's bytes whatever algorithm RCCL picked, so the replicated solves cannot drift apart; it is the repair
    the bench falls back to (and prices) when an all-reduce turns out not to be bit-identical across ranks."""
    dist = ctx.dist
    if not ctx.use_dist:
        return None
    if ctx.rehearsal:                      # gloo through host memory where RCCL sits
        def op(t):
            host = t.cpu()
            if mode == "reduce_broadcast":
                dist.reduce(host, dst=0, op=dist.ReduceOp.SUM)
                dist.broadcast(host, src=0)
            else:
                dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
        return op
    if mode == "reduce_broadcast":
        def op(t):
            dist.reduce(t, dst=0, op=dist.ReduceOp.SUM)
            dist.broadcast(t, src=0)
        return op
    return lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)


def close_engine(engine):
    """Release a shard engine and, if the bench created one for it, the library-owned communicator."""
    comm = getattr(engine, "owned_comm", None)
    engine.close()
    if comm is not None:
        comm.close()
        engine.owned_comm = None


def ba_measure(args, ctx, sfm, strong, full, collective):
    """One BA workload on this process group: scene, shard, (parity leg), warm-up, pre-roll, R timed regions of EXACTLY
    K steps each.  `strong`: the scene is the config's own (points split over the ranks); otherwise every rank owns
    `pts` points of an N-times larger scene.  `full`: also the parity leg, the kernel breakdown and the roofline of the
    dominant kernel.  Returns (out dict, leftovers for the CPU legs)."""
    torch, dist = ctx.torch, ctx.dist
    native = sfm.native
    world, rank, device, use_dist, rehearsal = ctx.world, ctx.rank, ctx.device, ctx.use_dist, ctx.rehearsal
    cfg = dict(sfm.scenes.CONFIGS[args.config])
    pts_per_rank = args.pts or cfg["n_pts"]
    total_pts = (args.pts or cfg["n_pts"]) if strong else pts_per_rank * world
    scene = sfm.scenes.make_scene(cfg["n_cams"], total_pts, cfg["visibility"], seed=0)
    uvn = sfm.geometry.normalise_pixels(scene.uv_pix, scene.intrinsic)
    bounds = sfm.sharding.shard_bounds(scene.pt_ptr, world)
    ptr_l, cam_l, uv_l, pts_l, (p0, p1) = sfm.sharding.local_shard(scene.pt_ptr, scene.cam_idx, uvn, scene.pts_init, bounds, rank)

    engine = sfm.sharding.HipShardEngine(scene.n_cams, ptr_l, cam_l, uv_l, device)
    schur_mode = {"auto": native.SCHUR_AUTO, "pairs": native.SCHUR_PAIRS, "mfma": native.SCHUR_MFMA, "rows": native.SCHUR_ROWS}[args.schur]
    engine.prob.set_option(native.OPT_SCHUR, schur_mode)
    if args.debug:
        engine.prob.set_option(native.OPT_DEBUG, args.debug)
    if collective == "auto":
        # N > 1: the library's own communicator when every rank can have one (measured at one rank: 3 607-3 614 it/s against
        # 3 491-3 511 with torch.distributed.all_reduce between two C-ABI calls per iteration, profiles/r4/bench_torchrun_1rank_*);
        # all ranks agree before any of them enters ncclCommInitRank
        collective = "allreduce"
        if use_dist and world > 1 and not rehearsal:
            ok = torch.tensor([1.0 if native.comm_available() else 0.0], dtype=torch.float64, device=ctx.coll_device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if float(ok.item()) == 1.0:
                collective = "library"
    if collective == "library":
        # the library owns the communicator (sfm_comm_create: ncclCommInitRank on this rank's GPU) and issues the all-reduce
        # itself inside sfm_ba_iterate: K iterations = ONE C-ABI call per rank, no Python between them.  The 128-byte id
        # travels over the process group that torch.distributed already set up (any channel would do).
        if rehearsal:
            raise SystemExit("--collective library needs one GPU per rank (RCCL refuses two ranks on one device): not available under SFM_BENCH_REHEARSAL")
        ident = [native.comm_unique_id() if rank == 0 else None]
        if use_dist:
            dist.broadcast_object_list(ident, src=0)
        comm = native.Comm(world, rank, ident[0])
        engine.attach_comm(comm)
        engine.owned_comm = comm             # closed with the engine (close_engine below)
        ba = sfm.sharding.ShardedBa(engine, None, world)
    else:
        ba = sfm.sharding.ShardedBa(engine, make_all_reduce(ctx, collective), world)
    sync = ctx.sync

    def gather_state():
        cams, pts_loc = engine.get_state()
        if not use_dist:
            return cams, pts_loc
        parts = [None] * world
        dist.all_gather_object(parts, pts_loc)
        return cams, np.hstack(parts)

    rmse_init = rmse_gpu3 = None
    cams3 = pts3 = None
    if full:      # parity leg: the reference's default 3 iterations from the initial estimate
        engine.set_state(scene.cams_init, pts_l)
        ba.iterate(LAMBDA, 3)
        cams3, pts3 = gather_state()
        rmse_init = sfm.scenes.reprojection_rmse(scene.cams_init, scene.pts_init, scene)
        rmse_gpu3 = sfm.scenes.reprojection_rmse(cams3, pts3, scene)

    # ---- warmup with every kernel class bracketed: find the dominant kernel --------------------
    engine.set_state(scene.cams_init, pts_l)
    engine.prob.set_option(native.OPT_TIMING, (1 << native.K_COUNT) - 1)
    engine.prob.reset_timing()
    ba.iterate(LAMBDA, max(1, args.warmup))
    sync()
    event_overhead_ms = max(0.0, engine.prob.event_overhead(20) - EMPTY_KERNEL_MS)
    breakdown = {}
    for kid, name in enumerate(native.KERNEL_NAMES):
        ms, n = engine.prob.kernel_time(kid)
        breakdown[name] = ms / max(1, n) if name != "prep" else ms / max(1, args.warmup)
    dominant = max((n for n in breakdown if n != "prep"), key=lambda n: breakdown[n])
    dom_id = native.KERNEL_NAMES.index(dominant)

    # ---- timed regions: exactly K steps each, only the dominant kernel bracketed by hipEvents ----------
    # (sampled: a hipEvent pair puts two ~6 us bubbles into the stream, 3.6 % of a C3 iteration if every launch is bracketed)
    engine.prob.set_option(native.OPT_TIMING, 1 << dom_id)
    stride = args.timing_stride if args.timing_stride > 0 else max(1, min(10, args.steps // 5))
    engine.prob.set_option(native.OPT_TIMING_STRIDE, stride)
    # The W warm-up steps above are followed by host-side calibration, so the GPU is idle and its clocks are down when the
    # timed region would start; at 0.28 ms a step, K = 20 steps are over before they are back up (measured: 3 408 it/s with
    # W = 3 against 3 581 with W = 50, the dense product at 138.7 vs 126.3 us).  An untimed pre-roll of the same steps runs
    # until ~30 ms of GPU work have passed, immediately before the barrier that opens the first timed region; its length
    # is in the line ("preroll_steps").
    est_step_ms = ctx.max_over_ranks(sum(v for k, v in breakdown.items() if k != "prep"))
    preroll = 0 if args.no_preroll else int(max(0, min(400, PREROLL_MS / max(est_step_ms, 1e-3))))
    if preroll:
        ba.iterate(LAMBDA, preroll)
        sync()
    engine.prob.reset_timing()
    # R back-to-back regions of EXACTLY K steps (the 5.6 ms headline gets an error bar); every region is
    # barrier + synchronize | K steps | synchronize + barrier, MAX over ranks; `value` is the median region
    elapsed_runs = []
    for _ in range(max(1, args.repeats)):
        sync()
        t0 = time.perf_counter()
        ba.iterate(LAMBDA, args.steps)
        sync()
        elapsed_runs.append(ctx.max_over_ranks(time.perf_counter() - t0))
    elapsed = float(np.median(elapsed_runs))
    dom_ms, dom_n = engine.prob.kernel_time(dom_id)
    bracket_ms = dom_ms / max(1, dom_n)
    # solve = nbk + 1 launches inside one bracket: the bracket's own overhead is paid once
    dom_avg_ms = max(bracket_ms - event_overhead_ms, 0.5 * bracket_ms)
    cams_end, pts_end = gather_state()
    rmse_end = sfm.scenes.reprojection_rmse(cams_end, pts_end, scene)

    scale = 1 if strong else world
    out = {
        "metric": "BA LM-iterations/sec + final reprojection RMSE, 50 cams x 20k pts",
        "value": scale * args.steps / elapsed,
        "unit": ("LM-iterations/s of the one %dcam x %dk-pt scene (points split over the ranks)" % (scene.n_cams, scene.n_pts // 1000) if strong else
                 "LM-iterations/s (%dcam x %dk-pt shard-iterations, all ranks)" % (scene.n_cams, pts_per_rank // 1000)),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "preroll_steps": preroll,
        "ms_per_step": elapsed / args.steps * 1e3,
        "repeats": len(elapsed_runs),
        "value_runs": [scale * args.steps / e for e in elapsed_runs],
        "ms_per_step_runs": [e / args.steps * 1e3 for e in elapsed_runs],
        "value_note": "`value` / `ms_per_step` = the MEDIAN of %d back-to-back timed regions of exactly %d steps each (one pre-roll before the first)" % (len(elapsed_runs), args.steps),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
        "config": {"workload": ("%s: %d cams x %d pts in total @ %.0f%% visibility, strong-scaled: %d..%d pts per rank, lambda=5, Schur BA" % (
                                    args.config, scene.n_cams, scene.n_pts, 100 * cfg["visibility"],
                                    int(np.min(np.diff(bounds))), int(np.max(np.diff(bounds))))) if strong else
                               ("%s: %d cams x %d pts/rank @ %.0f%% visibility, lambda=5, Schur BA" % (
                                    args.config, scene.n_cams, pts_per_rank, 100 * cfg["visibility"])),
            "observations_per_rank": int(cam_l.shape[0]), "points_total": int(scene.n_pts),
            "parallelism": "points sharded x%d, cameras replicated, %s of [S|rhs] per iteration" % (world, collective) if world > 1 else "single GPU",
            "collective_backend": ctx.backend, "collective": collective,
            "schur": args.schur, **({"debug_bits": args.debug} if args.debug else {})},
        "rmse_px": {"initial": rmse_init, "after_3_iterations": rmse_gpu3, "after_timed_run": rmse_end},
    }
    # whole-iteration HBM figures per rank (north_star: achieved HBM-bandwidth fraction at every N): algorithmic
    # bytes of one iteration (20 M + 52 N + 112 V) over the measured time of one iteration
    iter_s = elapsed / args.steps
    alg_bytes = 20 * int(cam_l.shape[0]) + 52 * (int(ptr_l.shape[0]) - 1) + 112 * scene.n_cams
    out["hbm"] = {"algorithmic_bytes_per_iteration": alg_bytes, "algorithmic_GBps": alg_bytes / iter_s / 1e9,
                  "algorithmic_frac_of_peak": alg_bytes / iter_s / 1e9 / HBM_PEAK_GBS, "peak_GBps": HBM_PEAK_GBS,
                  "measured_bytes_per_iteration": None}
    if use_dist and world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, alg_bytes)
        out["hbm"]["per_rank_algorithmic_bytes_per_iteration"] = per_rank
        out["hbm"]["per_rank_algorithmic_frac_of_peak"] = [b / iter_s / 1e9 / HBM_PEAK_GBS for b in per_rank]
        out["hbm"]["whole_job_algorithmic_GBps"] = sum(per_rank) / iter_s / 1e9

    # ---- N > 1: do the ranks hold bit-identical cameras after the replicated solves?  Recorded, never fatal ------------
    if use_dist and world > 1:
        cams_dev = torch.from_numpy(np.ascontiguousarray(cams_end)).to(ctx.coll_device)
        ref = cams_dev.clone()
        dist.broadcast(ref, src=0)
        dev = (cams_dev - ref).abs().max().reshape(1)
        dist.all_reduce(dev, op=dist.ReduceOp.MAX)
        out["max_camera_deviation_across_ranks"] = float(dev.item())

    extra = Holder()
    extra.scene, extra.uvn, extra.engine, extra.cams3, extra.pts3, extra.rmse_gpu3 = scene, uvn, engine, cams3, pts3, rmse_gpu3
    if not full:
        close_engine(engine)
        return out, extra

    out["kernel_ms"] = breakdown
    out["kernel_ms_note"] = "hipEvent brackets of the warm-up, every class bracketed, each INCLUDING the bracket's own %.1f us" % (event_overhead_ms * 1e3)
    costs = algorithmic_costs(scene.n_cams, ptr_l, int(cam_l.shape[0]))
    c = costs[dominant]
    if c["bound"] == "mfma":
        achieved = c["flops"] / (dom_avg_ms * 1e-3) / 1e12
        roofline = dict(kernel="ba_" + dominant, bound="mfma", achieved=achieved, peak=FP64_PEAK_TFLOPS,
                        unit="TFLOP/s", frac=achieved / FP64_PEAK_TFLOPS)
    else:
        achieved = c["bytes"] / (dom_avg_ms * 1e-3) / 1e9
        roofline = dict(kernel="ba_" + dominant, bound="hbm", achieved=achieved, peak=HBM_PEAK_GBS,
                        unit="GB/s", frac=achieved / HBM_PEAK_GBS)
    # which Schur product the library launched is asked of the library, not guessed from the flags
    schur_kernel = {native.SCHUR_MFMA: "ba_schur_mfma", native.SCHUR_PAIRS: "ba_schur_pairs",
                    native.SCHUR_ROWS: "ba_schur_rows"}[engine.prob.info(native.INFO_SCHUR_KERNEL)]
    if dominant == "schur":      # the product kernel alone is bracketed (ba_schur_reduce is its own class)
        roofline["kernel"] = schur_kernel
        if schur_kernel != "ba_schur_mfma":      # no MFMA in the sparse products: every product is one ds_add_f64 into an LDS tile / panel
            achieved = c["lds_adds"] / (dom_avg_ms * 1e-3) / 1e12
            roofline.update(bound="lds", achieved=achieved, peak=LDS_ADD_PEAK_TADDS, unit="Tadd/s", frac=achieved / LDS_ADD_PEAK_TADDS)
    roofline["avg_launch_ms"] = dom_avg_ms
    roofline["event_bracket_ms"] = bracket_ms
    roofline["event_overhead_ms"] = event_overhead_ms
    roofline["timing_note"] = ("avg_launch_ms = hipEvent bracket of this kernel class (every %d-th launch of the timed regions, %d samples) minus the "
                               "bracket an empty kernel reads on the same stream (sfm_ba_event_overhead, less the %.1f us the empty kernel itself takes); "
                               "rocprofv3's average for the same command is committed under profiles/" % (stride, dom_n, EMPTY_KERNEL_MS * 1e3))
    roofline["launches"] = dom_n
    # HBM bytes per launch from rocprofv3 --pmc passes (tools/parse_pmc.py), keyed by workload and kernel:
    # profiles/traffic.json = {"<workload key>": {"<kernel>": bytes, ...}}; null when no record matches this run
    workload_key = "%s/%dcams_%dpts_per_rank/%s" % (args.config, scene.n_cams, int(ptr_l.shape[0]) - 1, schur_kernel)
    roofline["traffic"] = None
    roofline["traffic_key"] = workload_key
    tfile = os.path.join(REPO, "profiles", "traffic.json")
    traffic_rec = None
    if os.path.exists(tfile):
        try:
            traffic_rec = json.load(open(tfile)).get(workload_key)
        except Exception:
            traffic_rec = None
    launches = steady_state_launches(scene.n_cams, schur_kernel, fused=scene.n_cams <= 102 and not (args.debug & 16), debug=args.debug,
                                     reduce_in_solve=bool(engine.prob.info(native.INFO_REDUCE_IN_SOLVE)))
    out["config"]["reduce_in_solve_launch"] = bool(engine.prob.info(native.INFO_REDUCE_IN_SOLVE))
    if traffic_rec:
        kernels = {"solve": ("ba_chol_flow", "ba_chol_step", "ba_inv_apply", "ba_back_solve", "ba_back_update", "ba_small_solve"), "schur": (schur_kernel,),
                   "reduce": ("ba_schur_reduce", "ba_schur_rows_reduce")}.get(dominant, ("ba_" + dominant,))
        per = {k: traffic_rec[k] * launches[k] for k in kernels if k in launches and isinstance(traffic_rec.get(k), (int, float))}
        if per:
            roofline["traffic"] = sum(per.values())
            roofline["traffic_unit"] = "HBM bytes per iteration of this kernel class (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json[%r])" % workload_key
        have = {k: n for k, n in launches.items() if isinstance(traffic_rec.get(k), (int, float))}
        meas = sum(traffic_rec[k] * n for k, n in have.items())
        out["hbm"].update({"measured_bytes_per_iteration": meas, "measured_GBps": meas / iter_s / 1e9,
                           "measured_frac_of_peak": meas / iter_s / 1e9 / HBM_PEAK_GBS,
                           "measured_kernels": have, "kernels_without_a_record": sorted(set(launches) - set(have)),
                           "measured_source": "profiles/traffic.json[%r] (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per kernel; a committed "
                                              "record of an earlier run of this workload, not measured in this run; only the kernels a "
                                              "steady-state iteration launches, times their launches per iteration)" % workload_key})
    out["roofline"] = roofline
    return out, extra


def run_ba(args, ctx):
    sfm = importlib.import_module("structure-from-motion_amd")
    native = sfm.native
    world, rank = ctx.world, ctx.rank
    # C4 is the fixed 200 x 100k scene (strong); C3 weak-scales by default (every rank its own 20k-point shard) and, for N > 1,
    # ALSO times the metric's own fixed 50 x 20k scene split over the ranks (`strong_scaled` in the line); --scaling strong
    # makes that one `value`
    c4_fixed = args.config == "C4" and args.pts is None
    primary_strong = c4_fixed or args.scaling == "strong"
    collective = args.collective
    out, extra = ba_measure(args, ctx, sfm, primary_strong, True, collective)
    collective = out["config"].get("collective", collective)
    if ctx.use_dist and world > 1 and out.get("max_camera_deviation_across_ranks", 0.0) != 0.0 and collective in ("allreduce", "library"):
        # replicas drifted: the all-reduce did not hand every rank the same bytes.  Not a lost run: record it, repeat the
        # measurement with the reduce + broadcast exchange (identical bytes by construction) and report THAT as `value`
        drift = {"max_camera_deviation_across_ranks": out["max_camera_deviation_across_ranks"], "value_with_drift": out["value"],
                 "ms_per_step_with_drift": out["ms_per_step"],
                 "note": "the all-reduce (%s) left the ranks with different [S | rhs] bits; re-measured with reduce(dst=0) + broadcast" % collective}
        close_engine(extra.engine)
        collective = "reduce_broadcast"
        out, extra = ba_measure(args, ctx, sfm, primary_strong, True, collective)
        out["replica_drift"] = drift
    if ctx.use_dist and world > 1 and args.config == "C3" and not args.single_scaling:
        other, _extra2 = ba_measure(args, ctx, sfm, not primary_strong, False, collective)
        key = "weak_scaled" if primary_strong else "strong_scaled"
        out[key] = {k: other[k] for k in ("value", "unit", "ms_per_step", "value_runs", "ms_per_step_runs", "scaling", "hbm", "rmse_px") if k in other}
        out[key]["workload"] = other["config"]["workload"]
        if "max_camera_deviation_across_ranks" in other:
            out[key]["max_camera_deviation_across_ranks"] = other["max_camera_deviation_across_ranks"]
    scene, uvn, engine = extra.scene, extra.uvn, extra.engine
    cams3, pts3, rmse_gpu3 = extra.cams3, extra.pts3, extra.rmse_gpu3

    # ---- CPU baseline (rank 0, N = 1): the NumPy block-sparse oracle on the same scene --------------------------
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        close_engine(engine)
        # (ii) cpu_ref_sparse, default BLAS threads: median of three runs of 3 iterations (the state after 3 feeds the parity figures)
        leg = cpu_leg("sparse", args.config, 3, 3, args.pts)
        med = float(np.median(leg["seconds"]))
        ocams, opts = leg["state3"]
        rmse_cpu3 = sfm.scenes.reprojection_rmse(ocams, opts, scene)
        out["cpu_baseline"] = {"value": 3 / med, "unit": "LM-iterations/s", "cores": leg["threads"], "kind": "port",
                               "sample": "median of 3 runs of 3 LM iterations of the full %s scene with oracle/sfm_oracle.py ba_sparse "
                                         "(NumPy block-sparse restatement, OpenBLAS default threads): %s s" % (
                                             args.config, ", ".join("%.2f" % s for s in leg["seconds"])),
                               "seconds_per_run": leg["seconds"]}
        # the same leg on ONE BLAS thread, and (i) cpu_ref_dense: the line-faithful dense restatement at 5 x 400
        one = cpu_leg_child("sparse", args.config, 3, 1, 1, args.pts)
        if "seconds" in one:
            out["cpu_baseline"]["single_thread"] = {"value": 3 / one["seconds"][0], "unit": "LM-iterations/s", "cores": 1,
                                                    "sample": "one run of 3 iterations, OPENBLAS_NUM_THREADS=1: %.2f s" % one["seconds"][0]}
        dense = cpu_leg("dense", args.config, 3, 3)
        out["cpu_ref_dense"] = {"value": 3 / float(np.median(dense["seconds"])), "unit": "LM-iterations/s", "cores": dense["threads"],
                                "kind": "port", "sample": "median of 3 runs of 3 iterations of oracle.ba_dense (dense J, block_diag, inv: the "
                                                          "reference's own formulation) on %s: %s s; the reference itself took 8.2 s for 3 iterations at "
                                                          "this size (SURVEY.md Appendix C)" % (dense["scene"], ", ".join("%.2f" % s for s in dense["seconds"]))}
        out["rmse_px"]["cpu_after_3_iterations"] = rmse_cpu3
        out["rmse_px"]["rel_diff_gpu_vs_cpu"] = abs(rmse_gpu3 - rmse_cpu3) / rmse_cpu3
        out["max_rel_diff_vs_cpu"] = {
            "cams": float(np.max(np.abs(cams3 - ocams)) / np.max(np.abs(ocams))),
            "pts": float(np.max(np.abs(pts3 - opts)) / np.max(np.abs(opts)))}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
        # PCIe-inclusive rate of the host-buffer entry point (sfm_ba_solve: create + upload + 3 iterations +
        # download), for DESIGN.md; never used as `value`
        # (three calls: the first one follows ~10 s of CPU-only work, i.e. an idle GPU and a cold allocation pool)
        hb = []
        for _ in range(3):
            t0 = time.perf_counter()
            native.ba_solve(scene.n_cams, scene.pt_ptr, scene.cam_idx, uvn, scene.cams_init, scene.pts_init, LAMBDA, 3)
            hb.append(time.perf_counter() - t0)
        out["host_buffer_path"] = {"seconds_for_3_iterations_incl_setup_and_pcie": min(hb), "first_call_after_idle": hb[0]}
        out["drop_in_path"] = drop_in_path(sfm, scene)
        # ... and at the size the reference's own pipeline reaches (filter_size = 10 views, ba_processor.py:24, 44-46; its demo: 6 x 1260)
        out["drop_in_path_small"] = drop_in_path(sfm, sfm.scenes.make_scene(6, 1260, 1.0, seed=0))

    if world > 1 or args.no_cpu_baseline:
        close_engine(engine)
    finish(ctx, out)


# ================================================================================================================
# nonlinear triangulation by point / nonlinear PnP by view: no data-path collective
# ================================================================================================================
def tri_batch(sfm, n_views, m, seed):
    sc = sfm.scenes.make_scene(n_views, m, 1.0, seed=seed)
    projs = []
    for c in range(n_views):
        rot = sfm.geometry.quaternion_to_rotation(sc.cams_true[c, 3:7])
        loc = sc.cams_true[c, 0:3].reshape(3, 1)
        projs.append(sc.intrinsic @ np.hstack((rot.T, rot.T @ -loc)))
    uv = np.stack([sc.uv_pix[:, sc.cam_idx == c] for c in range(n_views)])
    return np.stack(projs), uv, np.vstack((sc.pts_init, np.ones((1, m))))


def pnp_batch(sfm, n_views, n, seed):
    """n_views independent views of n points each (distinct scenes, perturbed start poses)."""
    rng = np.random.default_rng(seed)
    base = sfm.scenes.make_scene(2, n, 1.0, seed=seed)
    sel = np.flatnonzero(base.cam_idx == 1)
    x = np.vstack((base.pts_true, np.ones((1, n))))
    rot_true = sfm.geometry.quaternion_to_rotation(base.cams_true[1, 3:7])
    loc_true = base.cams_true[1, 0:3]
    from scipy.spatial.transform import Rotation
    uv, xs, r0, c0 = [], [], [], []
    for _ in range(n_views):
        uv.append(np.vstack((base.uv_pix[:, sel] + rng.normal(0, 0.3, (2, n)), np.ones((1, n)))))
        xs.append(x)
        r0.append(rot_true @ Rotation.from_rotvec(rng.normal(0, 0.01, 3)).as_matrix())
        c0.append(loc_true + rng.normal(0, 0.05, 3))
    offsets = (np.arange(n_views + 1) * n).astype(np.int32)
    return offsets, np.hstack(uv), np.hstack(xs), np.stack([base.intrinsic] * n_views), np.stack(r0), np.stack(c0)


def time_steps(ctx, shard, run, steps, warmup, no_preroll=False):
    """W untimed + K timed steps on the shard's stream; every n-th timed launch bracketed by events recorded ON that
    stream.  Returns (elapsed seconds MAX over ranks, average bracketed launch ms, samples)."""
    torch = ctx.torch
    t_w = time.perf_counter()
    for _ in range(max(1, warmup)):
        run()
    ctx.sync()
    # untimed pre-roll up to ~30 ms of GPU work right before the timed region (clock ramp, as in run_ba)
    per_pass = ctx.max_over_ranks((time.perf_counter() - t_w) / max(1, warmup))
    preroll = 0 if no_preroll else int(max(0, min(200, PREROLL_MS * 1e-3 / max(per_pass, 1e-6)) - max(1, warmup)))
    for _ in range(preroll):
        run()
    ctx.sync()
    stride = max(1, min(10, steps // 5))
    pairs = []
    t0 = time.perf_counter()
    for i in range(steps):
        if i % stride == 0:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(shard.stream)
            run()
            b.record(shard.stream)
            pairs.append((a, b))
        else:
            run()
    ctx.sync()
    elapsed = ctx.max_over_ranks(time.perf_counter() - t0)
    ms = [a.elapsed_time(b) for a, b in pairs]
    return elapsed, float(np.mean(ms)), len(ms), stride, preroll


def run_tri_pnp(args, ctx):
    sfm = importlib.import_module("structure-from-motion_amd")
    sh = sfm.sharding
    world, rank = ctx.world, ctx.rank
    sys.path.insert(0, os.path.join(REPO, "oracle"))
    if args.config == "TRI":
        n_views, per_rank, lam, iters = 3, args.pts or 1_000_000, 0.5, 100          # reference defaults (triangulation_processor.py:12)
        projs, uv, x0 = tri_batch(sfm, n_views, per_rank * world, seed=1)
        tri = sh.ShardedTriangulation(rank, world, gather=(lambda a, w: sh.gather_columns(a, w, device=None if ctx.rehearsal else ctx.device)) if world > 1 else None)
        p0, p1 = tri.local_range(per_rank * world)
        shard = sh.HipTriShard(projs, uv[:, :, p0:p1], x0[:, p0:p1], ctx.device)
        run = lambda: shard.run(lam, iters)      # noqa: E731
        units = (p1 - p0) * iters
        flops_per_launch = (TRI_FLOPS_PER_VIEW * n_views + TRI_FLOPS_PER_POINT) * units
        alg_bytes = (16.0 * n_views + 64.0) * (p1 - p0)
        workload = "TRI: %d points/rank x %d views x %d iterations, lambda=%g (triangulation_processor.py:160-234), sharded by point" % (per_rank, n_views, iters, lam)
        kernel = "tri_nonlinear_kernel<%d>" % n_views
        metric, unit = "nonlinear-triangulation point-iterations/sec", "point-iterations/s (all ranks)"
    else:
        n_views_rank, n, lam, iters = 256, args.pts or 1000, 5.0, 200               # the reference's own test runs 200 iterations (campose_processor.py:1073)
        offsets, uvp, xs, ks, r0, c0 = pnp_batch(sfm, n_views_rank * world, n, seed=2)
        pnp = sh.ShardedPnp(rank, world, gather=(lambda a, w: sh.gather_columns(a, w, device=None if ctx.rehearsal else ctx.device)) if world > 1 else None)
        v0, v1 = pnp.local_range(offsets)
        a0, a1 = int(offsets[v0]), int(offsets[v1])
        shard = sh.HipPnpShard(offsets[v0:v1 + 1] - a0, uvp[:, a0:a1], xs[:, a0:a1], ks[v0:v1], r0[v0:v1], c0[v0:v1], ctx.device)
        run = lambda: shard.run(lam, iters)      # noqa: E731
        units = (a1 - a0) * iters
        flops_per_launch = PNP_FLOPS_PER_POINT * units
        alg_bytes = 56.0 * (a1 - a0) + 2 * 96.0 * (v1 - v0)
        workload = "PNP: %d views/rank x %d points x %d iterations, lambda=%g (campose_processor.py:308-459), sharded by view" % (n_views_rank, n, iters, lam)
        kernel = "pnp_nonlinear_kernel"
        metric, unit = "nonlinear-PnP point-iterations/sec", "point-iterations/s (all ranks)"

    elapsed, bracket_ms, samples, stride, preroll = time_steps(ctx, shard, run, args.steps, args.warmup, args.no_preroll)
    total_units = units
    if ctx.use_dist:
        t = ctx.torch.tensor([float(units)], dtype=ctx.torch.float64, device=ctx.coll_device)
        ctx.dist.all_reduce(t)
        total_units = float(t.item())
    achieved = flops_per_launch / (bracket_ms * 1e-3) / 1e12
    out = {
        "metric": metric, "value": total_units * args.steps / elapsed, "unit": unit, "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "preroll_steps": preroll, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f64",
        "data": "synthetic" if not ctx.rehearsal else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
        "config": {"workload": workload, "parallelism": "independent units dealt to %d rank(s), results gathered once after the timed region" % world,
                   "collective_backend": ctx.backend},
        "roofline": {"kernel": kernel, "bound": "valu_f64", "achieved": achieved, "peak": FP64_PEAK_TFLOPS, "unit": "TFLOP/s",
                     "frac": achieved / FP64_PEAK_TFLOPS, "frac_of_measured_v_fma_f64_rate": achieved / FP64_FMA_MEASURED_TFLOPS,
                     "avg_launch_ms": bracket_ms, "launches": samples,
                     "timing_note": "hipEvent bracket on the kernel's stream, every %d-th launch of the timed region; the launch is ms long, so the "
                                    "bracket's own ~10 us is left in" % stride,
                     "algorithmic_flops_per_launch": flops_per_launch, "algorithmic_bytes_per_launch": alg_bytes,
                     "traffic": None},
    }
    # ---- parity + gather: the sharded result against this rank's kernel output, and against the oracle on a sample --------
    oracle = importlib.import_module("sfm_oracle")
    if args.config == "TRI":
        mine = shard.result()
        if world > 1:
            full = np.hstack(sh.gather_columns(mine, np.diff(sh.shard_points(per_rank * world, world)), device=None if ctx.rehearsal else ctx.device))
            assert full.shape[1] == per_rank * world and np.array_equal(full[:, p0:p1], mine)
        ns = min(20000, p1 - p0)
        want = oracle.nonlinear_triangulate_vec(x0[:, p0:p0 + ns], list(projs), [np.vstack((u[:, p0:p0 + ns], np.ones((1, ns)))) for u in uv], lam, iters)
        out["parity"] = {"max_rel_diff_vs_oracle_on_%d_points" % ns: float(np.max(np.abs(mine[:, :ns] - want)) / np.max(np.abs(want)))}
    else:
        rot, loc, st = shard.result()
        if world > 1:
            packed = np.vstack((rot.reshape(-1, 9).T, loc.T, st[None, :].astype(np.float64)))
            full = np.hstack(sh.gather_columns(packed, np.diff(sh.shard_views(offsets, world)), device=None if ctx.rehearsal else ctx.device))
            assert full.shape[1] == n_views_rank * world
        r_or, c_or = oracle.nonlinear_pnp(uvp[:, a0:a0 + n], xs[:, a0:a0 + n], ks[v0], r0[v0], c0[v0].reshape(3, 1), lam, 3)
        shard.run(lam, 3)
        rot3, loc3, _ = shard.result()
        out["parity"] = {"status_nonzero": int(np.count_nonzero(st)),
                         "max_rel_diff_vs_oracle_first_view_3_iterations": float(max(np.max(np.abs(rot3[0] - r_or)), np.max(np.abs(loc3[0] - c_or.reshape(3))) / np.max(np.abs(c_or))))}
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        if args.config == "TRI":
            ns, t0 = 100000, time.perf_counter()
            oracle.nonlinear_triangulate_vec(x0[:, :ns], list(projs), [np.vstack((u[:, :ns], np.ones((1, ns)))) for u in uv], lam, 20)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns * 20 / dt, "unit": "point-iterations/s", "cores": host_threads(), "kind": "port",
                                   "sample": "oracle.nonlinear_triangulate_vec (NumPy, vectorised over points) on %d points x %d views x 20 iterations: %.1f s; "
                                             "the reference's own per-point Python loop runs 1.3e4 point-iterations/s (SURVEY.md Appendix C)" % (ns, n_views, dt)}
        else:
            ns, t0 = min(n, 400), time.perf_counter()
            oracle.nonlinear_pnp(uvp[:, :ns], xs[:, :ns], ks[0], r0[0], c0[0].reshape(3, 1), lam, 40)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {"value": ns * 40 / dt, "unit": "point-iterations/s", "cores": 1, "kind": "port",
                                   "sample": "oracle.nonlinear_pnp (the reference's per-point loop restated) on one view of %d points x 40 iterations: %.1f s; "
                                             "the reference itself runs 4.3e3 point-iterations/s (SURVEY.md Appendix C)" % (ns, dt)}
        out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
    finish(ctx, out)


# ================================================================================================================
# C5: the per-view loop of BaProcessor.process (ba_processor.py:137-267) through the drop-in classes
# ================================================================================================================
class Stopwatch:
    """Accumulates the wall time spent inside the native (C-ABI) calls of one stage."""

    def __init__(self, native):
        self.native, self.t, self._saved, self.by_name = native, 0.0, [], {}

    def wrap(self, obj, name):
        fn = getattr(obj, name)

        def timed(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                dt = time.perf_counter() - t0
                self.t += dt
                self.by_name[name] = self.by_name.get(name, 0.0) + dt
        self._saved.append((obj, name, fn))
        setattr(obj, name, timed)

    def install(self):
        n = self.native
        for name in ("pnp_linear_ransac", "pnp_ransac_evaluate", "pnp_inlier_mask", "pnp_ransac_begin", "pnp_ransac_finish", "pnp_nonlinear",
                     "triangulate", "tri_nonlinear", "tri_linear", "ba_solve"):
            self.wrap(n, name)
        for name in ("__init__", "iterate", "get_state", "get_state_rot", "rederive_quaternions", "append", "set_cameras", "set_points", "set_state"):
            self.wrap(n.BaProblem, name)

    def remove(self):
        for obj, name, fn in reversed(self._saved):
            setattr(obj, name, fn)
        self._saved = []

    def take(self):
        t, self.t = self.t, 0.0
        self.last_by_name, self.by_name = self.by_name, {}
        return t


def c5_sequence(sfm, n_views, n_pts, seed=51):
    """Synthetic incremental sequence: view c >= 1 brings the points born at c (n_pts / (n_views - 1) per view); every
    registered view observes every known point (0.3 px noise)."""
    sc = sfm.scenes.make_scene(n_views, n_pts, 1.0, seed=seed, pixel_noise=0.3)
    rots = [sfm.geometry.quaternion_to_rotation(sc.cams_true[c, 3:7]) for c in range(n_views)]
    locs = [sc.cams_true[c, 0:3].reshape(3, 1) for c in range(n_views)]
    uv = [np.vstack((sc.uv_pix[:, sc.cam_idx == c], np.ones((1, n_pts)))) for c in range(n_views)]
    per = (n_pts + n_views - 2) // (n_views - 1)
    birth = 1 + np.arange(n_pts) // per
    return sc, rots, locs, uv, birth


def run_c5(args, ctx):
    import random
    sfm = importlib.import_module("structure-from-motion_amd")
    native = sfm.native
    native.init(ctx.device.index or 0)
    n_views, n_pts = 10, args.pts or 5000            # filter_size = 10 caps the reference's sequence (ba_processor.py:24, 44-46)
    sc, rots, locs, uv, birth = c5_sequence(sfm, n_views, n_pts)
    K = sc.intrinsic
    sw = Stopwatch(native)
    sw.install()
    keypoints = [[KeyPoint(-1.0, -1.0)] + [KeyPoint(float(uv[c][0, j]), float(uv[c][1, j])) for j in range(n_pts)] for c in range(n_views)]

    def one_pass(record):
        from scipy.spatial.transform import Rotation
        rng = np.random.default_rng(7)
        tp = sfm.processors.HipTriangulationProcessor()                     # 0.5, 100 (triangulation_processor.py:12)
        cfg = sfm.processors.RansacConfig(8.0, 0.99, 0.75, 6, 300)          # ba_processor.py:470-480 / campose test values; seeds Python's RNG
        cp = sfm.processors.HipCamposeProcessor(cfg, 5, 300)                # ba_processor.py:486
        vp, kt = Holder(), Holder()
        vp.view_list, kt.track_list = [], []
        bp = sfm.processors.HipBaProcessor(vp, kt, None, tp, cp, iteration=3, damping_factor=5)
        bp.ba_verbose = False

        def add_view(c, rot, loc):      # the front end's product (key points, an empty track table) exists before the view is registered
            vp.view_list.append(View(rot, loc, K.copy(), keypoints[c]))
            tr = Holder()
            tr.table = np.full((n_views, n_pts + 1), -1, dtype=int)
            kt.track_list.append(tr)

        add_view(0, rots[0].copy(), locs[0].copy())
        known = np.zeros(n_pts, dtype=bool)
        full = np.zeros((4, n_pts)); full[3] = 1.0
        for c in range(1, n_views):
            stage = {}
            t_view = time.perf_counter()
            sw.take()
            if known.any():
                idx = np.flatnonzero(known)
                t0 = time.perf_counter()
                inl, r_new, c_new = cp.estimate_cam_pose_pnp(uv[c][:, idx], full[:, idx], K)          # ba_processor.py:191
                stage["pnp_s"] = time.perf_counter() - t0
                stage["pnp_native_s"] = sw.take()
                stage["pnp_native_calls_ms"] = {k: round(v * 1e3, 4) for k, v in sw.last_by_name.items()}
                stage["pnp_inliers"] = len(inl)
            else:
                r_new = rots[c] @ Rotation.from_rotvec(rng.normal(0, 0.002, 3)).as_matrix()     # second view: pose from the two-view initialisation
                c_new = locs[c] + rng.normal(0, 0.01, (3, 1))
            add_view(c, r_new, c_new)
            new = np.flatnonzero(birth == c)
            views = vp.view_list
            t0 = time.perf_counter()
            pts_new = tp.triangulate([views[c - 1].cam_proj, views[c].cam_proj], [uv[c - 1][:, new], uv[c][:, new]])      # ba_processor.py:246
            stage["triangulate_s"] = time.perf_counter() - t0
            stage["triangulate_native_s"] = sw.take()
            full[:, new] = pts_new
            known[new] = True
            for v in range(c + 1):
                kt.track_list[v].table[v, 1:][known] = np.flatnonzero(known)
            last = int(np.flatnonzero(known).max()) + 1
            tp.tri_pts = full[:, :last]
            before = bp.ba_upload_bytes
            t0 = time.perf_counter()
            bp._BaProcessor__execute_bundle_adjustment()                                            # ba_processor.py:267
            stage["ba_s"] = time.perf_counter() - t0
            stage["ba_native_s"] = sw.take()
            stage["ba_native_calls_ms"] = {k: round(v * 1e3, 4) for k, v in sw.last_by_name.items()}
            stage["ba_action"], stage["ba_upload_bytes"] = bp.ba_last_action, bp.ba_upload_bytes - before
            stage["view_s"] = stage.get("pnp_s", 0.0) + stage["triangulate_s"] + stage["ba_s"]      # the three drop-in calls of ba_processor.py:191, 246, 267
            stage["harness_s"] = time.perf_counter() - t_view - stage["view_s"]                     # stand-in for the front end / track bookkeeping: not timed
            stage["views"], stage["points"], stage["observations"] = c + 1, last, (c + 1) * last
            record.append(stage)
        cams = np.stack([sfm.geometry.pack_camera(v.rot, v.loc) for v in vp.view_list])
        bp.ba_release()
        return cams, full

    for _ in range(max(1, min(args.warmup, 2))):
        one_pass([])
    passes = max(1, args.steps // (n_views - 1))
    torch = ctx.torch
    torch.cuda.synchronize()
    records = []
    for _ in range(passes):
        rec = []
        cams, full = one_pass(rec)
        records.append(rec)
    torch.cuda.synchronize()
    # the timed region of a step = the three drop-in calls of that view (every one of them synchronises before it returns);
    # creating the synthetic key points and filling the track tables stands in for the out-of-scope front end
    elapsed = sum(st["view_s"] for rec in records for st in rec)
    sw.remove()
    steps = passes * (n_views - 1)
    rmse = sfm.scenes.reprojection_rmse(cams, full[0:3], sc)
    per_view = []
    for i in range(n_views - 1):
        rows = [r[i] for r in records]
        agg = {k: float(np.median([r[k] for r in rows])) for k in rows[0] if k.endswith("_s")}
        agg.update({k: rows[0][k] for k in ("views", "points", "observations", "ba_action", "ba_upload_bytes")})
        agg["ba_native_calls_ms"] = {k: float(np.median([r["ba_native_calls_ms"].get(k, 0.0) for r in rows])) for k in rows[0]["ba_native_calls_ms"]}
        if "pnp_native_calls_ms" in rows[0]:
            agg["pnp_native_calls_ms"] = {k: float(np.median([r["pnp_native_calls_ms"].get(k, 0.0) for r in rows])) for k in rows[0]["pnp_native_calls_ms"]}
        agg["host_python_s"] = agg["view_s"] - sum(agg.get(k, 0.0) for k in ("pnp_native_s", "triangulate_native_s", "ba_native_s"))
        per_view.append(agg)
    out = {
        "metric": "incremental SfM views/sec (PnP + triangulation + global BA per registered view)",
        "value": steps / elapsed, "unit": "registered views/s", "n_gpus": 1, "steps": steps, "warmup": args.warmup,
        "ms_per_step": elapsed / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "C5: %d views x %d points, every view sees every known point; per view: RANSAC PnP (300 hypotheses) + nonlinear PnP "
                               "(lambda=5, 300 iterations) + DLT and nonlinear triangulation (lambda=0.5, 100 iterations) of the new points + global BA "
                               "(lambda=5, 3 iterations) through the drop-in classes (ba_processor.py:137-267)" % (n_views, n_pts),
                   "host_buffers": "the drop-in API hands over host arrays: PCIe and Python are inside every figure of this config"},
        "rmse_px": {"final": rmse},
        "per_view": per_view,
        "roofline": None,
    }
    if not args.no_cpu_baseline:
        # CPU leg on a bounded sample: view index 3 of the sequence (4 views, the points known by then) -- the oracle's nonlinear
        # PnP (per-point Python loop, 20 of the 300 iterations, scaled), vectorised nonlinear triangulation of that view's new
        # points, block-sparse BA.  The six-point RANSAC has no CPU restatement in oracle/ and is left out of the sample.
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        oracle = importlib.import_module("sfm_oracle")
        c = 3
        idx = np.flatnonzero(birth < c)
        new = np.flatnonzero(birth == c)
        x_known = np.vstack((sc.pts_true[:, idx], np.ones((1, idx.size))))
        t0 = time.perf_counter()
        oracle.nonlinear_pnp(uv[c][:, idx[:400]], x_known[:, :400], K, rots[c], locs[c], 5, 20)
        t_pnp = (time.perf_counter() - t0) * (idx.size / 400.0) * (300 / 20.0)
        projs = [K @ np.hstack((rots[v].T, rots[v].T @ -locs[v])) for v in (c - 1, c)]
        x_new = np.vstack((sc.pts_init[:, new], np.ones((1, new.size))))
        t0 = time.perf_counter()
        oracle.nonlinear_triangulate_vec(x_new, projs, [uv[c - 1][:, new], uv[c][:, new]], 0.5, 100)
        t_tri = time.perf_counter() - t0
        kn = np.flatnonzero(birth <= c)
        cam_idx = np.tile(np.arange(c + 1), kn.size).astype(np.int32)
        pt_idx = np.repeat(np.arange(kn.size), c + 1).astype(np.int32)
        uvn = np.empty((2, cam_idx.size))
        for v in range(c + 1):
            uvn[:, cam_idx == v] = sfm.geometry.normalise_pixels(uv[v][0:2, kn], K)
        t0 = time.perf_counter()
        oracle.ba_sparse(sc.cams_init[:c + 1], sc.pts_init[:, kn], cam_idx, pt_idx, uvn, 5, 3)
        t_ba = time.perf_counter() - t0
        gpu_view = per_view[c - 1]["view_s"]
        out["cpu_baseline"] = {"value": 1.0 / (t_pnp + t_tri + t_ba), "unit": "registered views/s", "cores": host_threads(), "kind": "port",
                               "sample": "view %d of the sequence (%d known points, %d new): oracle.nonlinear_pnp %.1f s (400 points x 20 iterations measured, "
                                         "scaled to %d points x 300), nonlinear_triangulate_vec %.2f s, ba_sparse 3 iterations %.2f s; RANSAC not included" % (
                                             c + 1, idx.size, new.size, t_pnp, idx.size, t_tri, t_ba),
                               "gpu_same_view_s": gpu_view}
        out["speedup_vs_cpu_baseline_same_view"] = (t_pnp + t_tri + t_ba) / gpu_view
    finish(ctx, out)


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
        sock.bind(("127.0.0.1", 0))
        return sock.getsockname()[1]


def spawn_ranks(n, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment (what the driver's SCALE run may issue): this
    process -- which has not imported torch and never touches a GPU -- starts the N ranks as CHILD processes
    (`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>`, rendezvous on 127.0.0.1), lets
    rank 0's single JSON line through on stdout and returns the launcher's exit code.  Never exec: a parent that had
    initialised the GPU must not replace itself, and this way the rule cannot be broken by a later edit either."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // max(1, n))))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    if os.environ.get("SFM_BENCH_TRACE_LAUNCH", "0") == "1":
        print("bench.py launcher: torch imported in the parent: %s; child command: %s" % ("torch" in sys.modules, " ".join(cmd)), file=sys.stderr)
    proc = subprocess.Popen(cmd, env=env)          # stdout / stderr inherited: the ranks' output is this process's output
    try:
        return proc.wait()
    except KeyboardInterrupt:
        proc.terminate()
        return proc.wait()


def dry_rank(args):
    """SFM_BENCH_DRY=1: the rank-side control flow without a GPU (CPU tests): rendezvous over gloo, one all-reduce,
    rank 0 prints a marked JSON line."""
    import torch
    import torch.distributed as dist
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    dist.all_reduce(t)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "rank_sum": float(t.item()), "config": {"workload": args.config},
                          "data": "none: SFM_BENCH_DRY=1 (launcher / rendezvous check, no GPU work)"}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5, help="back-to-back timed regions of exactly --steps steps each; `value` is their median (BA configs)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="C3 with N > 1: which figure is `value` (auto = weak: 20 000 points per rank; strong = the fixed 50 x 20k scene split "
                         "over the ranks); the other one is timed as well and reported next to it")
    ap.add_argument("--single-scaling", action="store_true", help="C3 with N > 1: time only the figure --scaling names")
    ap.add_argument("--collective", default="auto", choices=["auto", "allreduce", "reduce_broadcast", "library"],
                    help="the per-iteration exchange of [S | rhs] (auto = library for N > 1 when every rank can load RCCL, else allreduce): allreduce = torch.distributed (RCCL) between two C-ABI calls per iteration; "
                         "reduce_broadcast = the repair the bench falls back to by itself when the ranks' cameras drift apart under allreduce; "
                         "library = the library's own RCCL communicator inside sfm_ba_iterate (one C-ABI call for all K iterations)")
    ap.add_argument("--config", default="C3", choices=["C3", "C4", "TRI", "PNP", "C5"])
    ap.add_argument("--pts", type=int, default=None, help="override points per rank / per view (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-preroll", action="store_true", help="start the timed region right after the W warm-up steps and the calibration (cold clocks)")
    ap.add_argument("--schur", default="auto", choices=["auto", "pairs", "mfma", "rows"])
    ap.add_argument("--timing-stride", type=int, default=TIMING_STRIDE, help="bracket the dominant kernel with hipEvents on every n-th launch of the timed region")
    ap.add_argument("--debug", type=int, default=0, help="SFM_OPT_DEBUG bits for same-box A/B runs of a code path (invalidates the metric)")
    ap.add_argument("--cpu-leg", default=None, choices=["sparse", "dense"], help=argparse.SUPPRESS)
    ap.add_argument("--cpu-iters", type=int, default=3, help=argparse.SUPPRESS)
    ap.add_argument("--cpu-repeats", type=int, default=1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_leg:                      # child of cpu_leg_child: CPU only, BLAS threads pinned by the parent
        leg = cpu_leg(args.cpu_leg, args.config if args.config in ("C3", "C4") else "C3", args.cpu_iters, args.cpu_repeats, args.pts)
        leg.pop("state3", None)
        print(json.dumps(leg))
        return

    # N > 1 without a launcher's environment: become the parent of N ranks (before torch is imported, before any GPU call)
    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus, sys.argv[1:]))
    if os.environ.get("SFM_BENCH_DRY", "0") == "1":
        dry_rank(args)
        return

    ctx = setup_dist(args)
    if args.config in ("C3", "C4"):
        run_ba(args, ctx)
    elif args.config in ("TRI", "PNP"):
        run_tri_pnp(args, ctx)
    else:
        if ctx.world != 1:
            raise SystemExit("--config C5 is a single-GPU latency workload (one growing scene)")
        run_c5(args, ctx)


if __name__ == "__main__":
    main()
