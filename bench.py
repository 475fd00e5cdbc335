#!/usr/bin/env python3
"""bench.py — BA LM-iterations/s on BASELINE.json's headline config (C3: 50 cams x 20 000 points,
60 % visibility, lambda = 5) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one damped Gauss-Newton iteration (ba_processor.py:297-406): linearise all
observations, form the Schur-reduced camera system, solve it, update cameras, back-substitute
points.  Inputs are resident in HBM before the timed region.

  --config C3 (default, the headline): for N > 1 the scene is WEAK-scaled: 50 cameras x (20 000 N) points,
      each rank owns a contiguous ~20 000-point shard and one RCCL all-reduce of [S | rhs] per iteration
      joins them; `value` counts 20 000-point shard-iterations per second over all ranks (= N x global
      iterations/s).
  --config C4 (BASELINE config 4): STRONG-scaled: the 200-camera x 100 000-point scene is fixed, its
      points are split over the N ranks by sharding.shard_bounds (balanced by camera pairs); `value` is
      global LM iterations per second of that one scene, "scaling": "strong".

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, hipEvent-timed inside the timed
region) and `cpu_baseline` (the NumPy block-sparse oracle on the host cores, N = 1 only).
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

FP64_PEAK_TFLOPS = 78.6     # MI355X FP64 vector = matrix peak (AMD spec; v_mfma_f64_16x16x4 measured 77.8, profiles/microbench_fp64_r01.txt)
LDS_ADD_PEAK_TADDS = 4.8    # ds_add_f64, conflict-free, all 256 CUs (profiles/microbench_fp64_r01.txt; 2.4 at random addresses)
TIMING_STRIDE = 0      # 0: bracket the dominant kernel with hipEvents on every n-th launch, n = min(10, steps // 5): at least five samples
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md
LAMBDA = 5.0                # reference default damping_factor (ba_processor.py:24)


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


def drop_in_path(sfm, scene):
    """What BaProcessor.process sees (ba_processor.py:267): `_BaProcessor__execute_bundle_adjustment` of the drop-in
    class on duck-typed views / track tables of this scene, the reference's default 3 iterations.  First call =
    observation list from the track tables + upload + solve; repeat calls = the scene is resident, only the 7 V
    camera doubles go up.  Wall-clock on the host, PCIe and Python included; never used as `value`."""
    n_views = scene.n_cams

    class KP:
        __slots__ = ("pt",)

        def __init__(self, x, y):
            self.pt = (x, y)

    class View:
        def __init__(self, rot, loc, k, kps):
            self.rot, self.loc, self.k, self.key_pts = rot, loc, k, kps

        def update_cam_pose(self, rot, loc):
            self.rot, self.loc = rot, loc

    class SelfRow:      # stands in for KeyTrack.table (n_views x n_keys): the BA path only ever reads row `v`
        def __init__(self, v, row):
            self.v, self.row = v, row

        def __getitem__(self, idx):
            assert idx[0] == self.v
            return self.row[idx[1]]

    class Holder:
        pass

    views, tracks = [], []
    rots = sfm.geometry.quaternions_to_rotations(scene.cams_init[:, 3:7])
    for c in range(n_views):
        sel = np.flatnonzero(scene.cam_idx == c)
        kps = [KP(-1.0, -1.0)] + [KP(float(scene.uv_pix[0, o]), float(scene.uv_pix[1, o])) for o in sel]
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
    return {"first_call_s": times[0], "repeat_call_s": float(np.median(times[1:])), "actions": actions,
            "upload_bytes_first_call": uploads[0], "upload_bytes_repeat_call": uploads[1], "iterations_per_call": 3}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--pts", type=int, default=None, help="override points per rank (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schur", default="auto", choices=["auto", "pairs", "mfma", "rows"])
    ap.add_argument("--timing-stride", type=int, default=TIMING_STRIDE, help="bracket the dominant kernel with hipEvents on every n-th launch of the timed region")
    ap.add_argument("--debug", type=int, default=0, help="SFM_OPT_DEBUG bits for same-box A/B runs of a code path (invalidates the metric)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    sfm = importlib.import_module("structure-from-motion_amd")
    native = sfm.native

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
    coll_device = torch.device("cpu") if rehearsal else device      # where the small bookkeeping collectives live
    # Under torch.distributed.run (RANK set) the RCCL path is taken even with one rank, so the same code
    # that runs on 8 GPUs can be exercised on a 1-GPU box; a plain `python bench.py` skips the process group.
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("gloo" if rehearsal else "nccl", rank=rank, world_size=world)

    # ---- workload: C3 weak-scaled (points per rank fixed), C4 strong-scaled (scene fixed) --------------
    cfg = dict(sfm.scenes.CONFIGS[args.config])
    strong = args.config == "C4" and args.pts is None
    pts_per_rank = args.pts or cfg["n_pts"]
    total_pts = cfg["n_pts"] if strong else pts_per_rank * world
    scene = sfm.scenes.make_scene(cfg["n_cams"], total_pts, cfg["visibility"], seed=0)
    uvn = sfm.geometry.normalise_pixels(scene.uv_pix, scene.intrinsic)
    bounds = sfm.sharding.shard_bounds(scene.pt_ptr, world)
    ptr_l, cam_l, uv_l, pts_l, (p0, p1) = sfm.sharding.local_shard(scene.pt_ptr, scene.cam_idx, uvn, scene.pts_init, bounds, rank)

    engine = sfm.sharding.HipShardEngine(scene.n_cams, ptr_l, cam_l, uv_l, device)
    schur_mode = {"auto": native.SCHUR_AUTO, "pairs": native.SCHUR_PAIRS, "mfma": native.SCHUR_MFMA, "rows": native.SCHUR_ROWS}[args.schur]
    engine.prob.set_option(native.OPT_SCHUR, schur_mode)
    if args.debug:
        engine.prob.set_option(native.OPT_DEBUG, args.debug)
    all_reduce = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)) if use_dist else None
    if use_dist and rehearsal:
        def all_reduce(t):
            host = t.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM)
            t.copy_(host)
    ba = sfm.sharding.ShardedBa(engine, all_reduce, world)

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()

    def gather_state():
        cams, pts_loc = engine.get_state()
        if not use_dist:
            return cams, pts_loc
        parts = [None] * world
        dist.all_gather_object(parts, pts_loc)
        return cams, np.hstack(parts)

    # ---- parity leg: the reference's default 3 iterations from the initial estimate -----------
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
    breakdown = {}
    for kid, name in enumerate(native.KERNEL_NAMES):
        ms, n = engine.prob.kernel_time(kid)
        breakdown[name] = ms / max(1, n) if name != "prep" else ms / max(1, args.warmup)
    dominant = max((n for n in breakdown if n != "prep"), key=lambda n: breakdown[n])
    dom_id = native.KERNEL_NAMES.index(dominant)

    # ---- timed region: exactly K steps, only the dominant kernel bracketed by hipEvents ----------
    # (sampled: a hipEvent pair puts two ~6 us bubbles into the stream, 3.6 % of a C3 iteration if every launch is bracketed)
    engine.prob.set_option(native.OPT_TIMING, 1 << dom_id)
    stride = args.timing_stride if args.timing_stride > 0 else max(1, min(10, args.steps // 5))
    engine.prob.set_option(native.OPT_TIMING_STRIDE, stride)
    engine.prob.reset_timing()
    sync()
    t0 = time.perf_counter()
    ba.iterate(LAMBDA, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom_ms, dom_n = engine.prob.kernel_time(dom_id)
    dom_avg_ms = dom_ms / max(1, dom_n)
    cams_end, pts_end = gather_state()
    rmse_end = sfm.scenes.reprojection_rmse(cams_end, pts_end, scene)

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
    roofline["launches"] = dom_n
    roofline["launches_note"] = "hipEvent-bracketed launches of this kernel class: every %d-th of the timed region" % stride
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
    if traffic_rec:
        kernels = {"solve": ("ba_solve", "ba_chol_step", "ba_back_solve"), "schur": (schur_kernel,),
                   "reduce": ("ba_schur_reduce",)}.get(dominant, ("ba_" + dominant,))
        per = {k: v for k, v in traffic_rec.items() if k.startswith(kernels) and isinstance(v, (int, float))}
        if per:
            mult = {"ba_chol_step": (7 * scene.n_cams + 31) // 32}
            roofline["traffic"] = sum(v * mult.get(k, 1) for k, v in per.items())
            roofline["traffic_unit"] = "HBM bytes per iteration of this kernel class (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json[%r])" % workload_key

    out = {
        "metric": "BA LM-iterations/sec + final reprojection RMSE, 50 cams x 20k pts",
        "value": (1 if strong else world) * args.steps / elapsed,
        "unit": ("LM-iterations/s of the one 200cam x 100k-pt scene (points split over the ranks)" if strong else
                 "LM-iterations/s (50cam x 20k-pt shard-iterations, all ranks)"),
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
        "config": {"workload": ("%s: %d cams x %d pts in total @ %.0f%% visibility, strong-scaled: %d..%d pts per rank, lambda=5, Schur BA" % (
                                    args.config, scene.n_cams, scene.n_pts, 100 * cfg["visibility"],
                                    int(np.min(np.diff(bounds))), int(np.max(np.diff(bounds))))) if strong else
                               ("%s: %d cams x %d pts/rank @ %.0f%% visibility, lambda=5, Schur BA" % (
                                    args.config, scene.n_cams, pts_per_rank, 100 * cfg["visibility"])),
            "observations_per_rank": int(cam_l.shape[0]), "points_total": int(scene.n_pts),
            "parallelism": "points sharded x%d, cameras replicated, all-reduce [S|rhs]" % world if world > 1 else "single GPU",
            "schur": args.schur, **({"debug_bits": args.debug} if args.debug else {})},
        "rmse_px": {"initial": rmse_init, "after_3_iterations": rmse_gpu3, "after_timed_run": rmse_end},
        "kernel_ms": breakdown,
        "roofline": roofline,
    }
    # whole-iteration HBM figures per rank (north_star: achieved HBM-bandwidth fraction at every N): algorithmic
    # bytes of one iteration (20 M + 52 N + 112 V) and, when profiles/traffic.json matches
    # this configuration's kernels, the bytes rocprofv3's PMC passes measured per iteration (materialised
    # intermediates such as the dense Z included), both over the measured time of one iteration
    iter_s = elapsed / args.steps
    alg_bytes = 20 * int(cam_l.shape[0]) + 52 * (int(ptr_l.shape[0]) - 1) + 112 * scene.n_cams
    out["hbm"] = {"algorithmic_bytes_per_iteration": alg_bytes, "algorithmic_GBps": alg_bytes / iter_s / 1e9,
                  "algorithmic_frac_of_peak": alg_bytes / iter_s / 1e9 / HBM_PEAK_GBS, "peak_GBps": HBM_PEAK_GBS,
                  "measured_bytes_per_iteration": None}
    if traffic_rec:
        mult = {"ba_chol_step": (7 * scene.n_cams + 31) // 32, "ba_cam_prep": 0}
        meas = sum(v * mult.get(k, 1) for k, v in traffic_rec.items() if k.startswith("ba_") and isinstance(v, (int, float)))
        out["hbm"].update({"measured_bytes_per_iteration": meas, "measured_GBps": meas / iter_s / 1e9,
                           "measured_frac_of_peak": meas / iter_s / 1e9 / HBM_PEAK_GBS,
                           "measured_source": "profiles/traffic.json[%r] (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per kernel; a committed "
                                              "record of an earlier run of this workload, not measured in this run)" % workload_key})

    # ---- N > 1: every rank must hold bit-identical cameras after the redundant solves -----------------
    if use_dist and world > 1:
        cams_dev = torch.from_numpy(np.ascontiguousarray(cams_end)).to(coll_device)
        ref = cams_dev.clone()
        dist.broadcast(ref, src=0)
        dev = (cams_dev - ref).abs().max().reshape(1)
        dist.all_reduce(dev, op=dist.ReduceOp.MAX)
        out["max_camera_deviation_across_ranks"] = float(dev.item())
        if float(dev.item()) != 0.0:
            if rank == 0:
                print(json.dumps(out))
            raise SystemExit("ranks disagree on the cameras after the replicated reduced solve: max |cams - cams(rank 0)| = %g" % float(dev.item()))

    # ---- CPU baseline (rank 0, N = 1): the NumPy block-sparse oracle, 3 iterations of the same scene
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        engine.close()
        sys.path.insert(0, os.path.join(REPO, "oracle"))
        oracle = importlib.import_module("sfm_oracle")
        n_cpu_iters = 8 if args.config == "C3" else 3      # ~10 s of CPU work at C3; the state after 3 feeds the parity figures
        trace = []
        t0 = time.perf_counter()
        oracle.ba_sparse(scene.cams_init, scene.pts_init, scene.cam_idx, scene.pt_idx, uvn, LAMBDA, n_cpu_iters, trace=trace)
        cpu_s = time.perf_counter() - t0
        ocams, opts = trace[2]
        rmse_cpu3 = sfm.scenes.reprojection_rmse(ocams, opts, scene)
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
        except Exception:
            cores = os.cpu_count() or 1
        out["cpu_baseline"] = {"value": n_cpu_iters / cpu_s, "unit": "LM-iterations/s", "cores": int(cores), "kind": "port",
                               "sample": "%d LM iterations of the full %s scene with oracle/sfm_oracle.py ba_sparse "
                                         "(NumPy block-sparse restatement, OpenBLAS threads), %.1f s" % (n_cpu_iters, args.config, cpu_s)}
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

    if world > 1 or args.no_cpu_baseline:
        engine.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
