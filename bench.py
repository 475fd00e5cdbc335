#!/usr/bin/env python3
"""bench.py — BA LM-iterations/s on BASELINE.json's headline config (C3: 50 cams x 20 000 points,
60 % visibility, lambda = 5) on N MI355X GPUs of one node.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one damped Gauss-Newton iteration (ba_processor.py:297-406): linearise all
observations, form the Schur-reduced camera system, solve it, update cameras, back-substitute
points.  Inputs are resident in HBM before the timed region.  For N > 1 the scene is weak-scaled:
50 cameras x (20 000 N) points, each rank owns a contiguous ~20 000-point shard and one RCCL
all-reduce of [S | rhs] (0.99 MB) per iteration joins them; `value` counts 20 000-point
shard-iterations per second over all ranks (= N x global iterations/s).

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
HBM_PEAK_GBS = 8000.0       # /opt/skills/guides/MI355X_MICROARCH.md
LAMBDA = 5.0                # reference default damping_factor (ba_processor.py:24)


def algorithmic_costs(n_cams, pt_ptr, n_obs):
    """Per-iteration algorithmic flops / bytes of each kernel class (restated in DESIGN.md)."""
    k = np.diff(pt_ptr).astype(np.float64)
    n_pts = k.shape[0]
    p = 7 * n_cams
    return {
        "linearize": dict(bound="hbm", bytes=20.0 * n_obs + 28.0 * n_pts, flops=567.0 * n_obs + 60.0 * n_pts),
        "schur": dict(bound="mfma", flops=float(np.sum(294.0 * k * (k - 1) / 2 + 168.0 * k)), bytes=168.0 * n_obs),
        "solve": dict(bound="mfma", flops=p ** 3 / 3.0 + 2.0 * p * p, bytes=8.0 * p * p),
        "backsub": dict(bound="hbm", bytes=20.0 * n_obs + 52.0 * n_pts, flops=300.0 * n_obs + 60.0 * n_pts),
        "prep": dict(bound="hbm", bytes=56.0 * n_cams + 152.0 * n_cams, flops=100.0 * n_cams),
        "reduce": dict(bound="hbm", bytes=8.0 * (p * p / 2.0 + p), flops=0.0),     # [S | rhs] written once
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C3")
    ap.add_argument("--pts", type=int, default=None, help="override points per rank (debug only; invalidates the metric)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--schur", default="auto", choices=["auto", "pairs", "mfma"])
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
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    # Under torch.distributed.run (RANK set) the RCCL path is taken even with one rank, so the same code
    # that runs on 8 GPUs can be exercised on a 1-GPU box; a plain `python bench.py` skips the process group.
    use_dist = world > 1 or "RANK" in os.environ
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    # ---- workload: weak-scaled C3 ------------------------------------------------------------
    cfg = dict(sfm.scenes.CONFIGS[args.config])
    pts_per_rank = args.pts or cfg["n_pts"]
    scene = sfm.scenes.make_scene(cfg["n_cams"], pts_per_rank * world, cfg["visibility"], seed=0)
    uvn = sfm.geometry.normalise_pixels(scene.uv_pix, scene.intrinsic)
    bounds = sfm.sharding.shard_bounds(scene.pt_ptr, world)
    ptr_l, cam_l, uv_l, pts_l, (p0, p1) = sfm.sharding.local_shard(scene.pt_ptr, scene.cam_idx, uvn, scene.pts_init, bounds, rank)

    engine = sfm.sharding.HipShardEngine(scene.n_cams, ptr_l, cam_l, uv_l, device)
    schur_mode = {"auto": native.SCHUR_AUTO, "pairs": native.SCHUR_PAIRS, "mfma": native.SCHUR_MFMA}[args.schur]
    engine.prob.set_option(native.OPT_SCHUR, schur_mode)
    all_reduce = (lambda t: dist.all_reduce(t, op=dist.ReduceOp.SUM)) if use_dist else None
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
    engine.prob.set_option(native.OPT_TIMING, 1 << dom_id)
    engine.prob.reset_timing()
    sync()
    t0 = time.perf_counter()
    ba.iterate(LAMBDA, args.steps)
    sync()
    elapsed = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
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
    if dominant == "schur":      # the product kernel alone is bracketed (ba_schur_reduce is its own class)
        roofline["kernel"] = "ba_schur_pairs" if args.schur == "pairs" or args.config != "C3" else "ba_schur_mfma"
    roofline["avg_launch_ms"] = dom_avg_ms
    roofline["launches"] = dom_n
    roofline["traffic"] = None
    tfile = os.path.join(REPO, "profiles", "traffic.json")     # HBM bytes per launch from rocprofv3 --pmc passes
    if os.path.exists(tfile):
        try:      # a kernel class may be several kernels (schur = schur_mfma + schur_reduce, solve = chol_step x11 + back_solve)
            tj = json.load(open(tfile))
            prefix = {"solve": ("ba_chol_step", "ba_back_solve"), "schur": ("ba_schur_mfma", "ba_schur_pairs"),
                      "reduce": ("ba_schur_reduce",)}.get(dominant, ("ba_" + dominant,))
            per = {k: v for k, v in tj.items() if k.startswith(prefix) and isinstance(v, (int, float))}
            if per:
                mult = {"ba_chol_step": (7 * scene.n_cams + 31) // 32}
                roofline["traffic"] = sum(v * mult.get(k, 1) for k, v in per.items())
                roofline["traffic_unit"] = "HBM bytes per iteration of this kernel class (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)"
        except Exception:
            pass

    out = {
        "metric": "BA LM-iterations/sec + final reprojection RMSE, 50 cams x 20k pts",
        "value": world * args.steps / elapsed,
        "unit": "LM-iterations/s (50cam x 20k-pt shard-iterations, all ranks)",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s: %d cams x %d pts/rank @ %.0f%% visibility, lambda=5, Schur BA" % (
            args.config, scene.n_cams, pts_per_rank, 100 * cfg["visibility"]),
            "observations_per_rank": int(cam_l.shape[0]), "points_total": int(scene.n_pts),
            "parallelism": "points sharded x%d, cameras replicated, all-reduce [S|rhs]" % world if world > 1 else "single GPU",
            "schur": args.schur},
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
    if os.path.exists(tfile) and args.config == "C3" and args.schur == "auto":
        try:
            tj = json.load(open(tfile))
            mult = {"ba_chol_step": (7 * scene.n_cams + 31) // 32, "ba_cam_prep": 0}
            meas = sum(v * mult.get(k, 1) for k, v in tj.items() if k.startswith("ba_") and isinstance(v, (int, float)))
            out["hbm"].update({"measured_bytes_per_iteration": meas, "measured_GBps": meas / iter_s / 1e9,
                               "measured_frac_of_peak": meas / iter_s / 1e9 / HBM_PEAK_GBS,
                               "measured_source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE per kernel, 1-GPU C3 run)"})
        except Exception:
            pass

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
        native.set_stream(0)
        t0 = time.perf_counter()
        native.ba_solve(scene.n_cams, scene.pt_ptr, scene.cam_idx, uvn, scene.cams_init, scene.pts_init, LAMBDA, 3)
        out["host_buffer_path"] = {"seconds_for_3_iterations_incl_setup_and_pcie": time.perf_counter() - t0}

    if world > 1 or args.no_cpu_baseline:
        engine.close()
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
