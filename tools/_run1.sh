mkdir -p gpurun_out/r3d
python -m pytest tests/test_gpu_parity.py tests/test_gpu_append.py -m gpu -q -x --timeout 300 --timeout-method=thread 2>&1 | tail -2 | tee gpurun_out/r3d/pytest.txt
python tools/time_solve_paths.py 2>&1 | grep -v amdgpu | tee gpurun_out/r3d/time_solve_paths.txt
python bench.py --no-cpu-baseline --steps 100 --warmup 10 2>&1 | grep '^{' > gpurun_out/r3d/bench_c3.json
python bench.py --no-cpu-baseline --steps 100 --warmup 10 --debug 1024 2>&1 | grep '^{' > gpurun_out/r3d/bench_c3_oldsplit.json
python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | grep '^{' > gpurun_out/r3d/bench_c4share.json
python - <<'PY'
import json
for f in ("bench_c3", "bench_c3_oldsplit", "bench_c4share"):
    d = json.load(open("gpurun_out/r3d/%s.json" % f))
    print(f, "value %.1f ms %.4f" % (d["value"], d["ms_per_step"]), {k: round(v * 1e3, 1) for k, v in d["kernel_ms"].items()}, "dom %.1f us" % (d["roofline"]["avg_launch_ms"] * 1e3))
PY
