"""`python bench.py --gpus N` with N > 1 and no launcher environment (what a driver's scaling run may issue) must start
its own ranks: the parent -- which has not imported torch and never touches a GPU -- spawns
`python -m torch.distributed.run --nproc-per-node N ... bench.py <same arguments>` as a CHILD, relays rank 0's single JSON
line and returns the worst exit code (contract: SURVEY.md section 8(e), the prompt's bench contract).
CPU only: the ranks run with SFM_BENCH_DRY=1 (rendezvous + one gloo all-reduce, no GPU work)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *argv, timeout=280):
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True, timeout=timeout,
                          env=env, cwd=REPO)


def test_parent_spawns_two_ranks_and_relays_the_line():
    out = _run({"SFM_BENCH_DRY": "1", "SFM_BENCH_TRACE_LAUNCH": "1"}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] is True and d["n_gpus"] == 2 and d["rank_sum"] == 3.0
    # the parent decided to spawn before importing torch, and it spawned a child (no exec)
    assert "torch imported in the parent: False" in out.stderr
    assert "-m torch.distributed.run" in out.stderr and "--nproc-per-node 2" in out.stderr and "--master-addr 127.0.0.1" in out.stderr


def test_parent_returns_the_ranks_failure():
    """Without a GPU the ranks refuse to run (no CPU fallback); the parent must hand that failure on, not print a line."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: the ranks would run")
    out = _run({}, "--gpus", "2", "--steps", "2", "--warmup", "1")
    assert out.returncode != 0
    assert not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "needs an MI355X" in out.stderr


def test_single_gpu_invocation_does_not_spawn():
    """--gpus 1 stays one process: on this GPU-less container it fails with the no-CPU-fallback message of the rank itself."""
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a GPU is visible: the bench would run")
    out = _run({"SFM_BENCH_TRACE_LAUNCH": "1"}, "--gpus", "1", "--steps", "2", "--warmup", "1")
    assert out.returncode != 0 and "launcher" not in out.stderr and "needs an MI355X" in out.stderr
