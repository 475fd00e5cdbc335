bash tools/gpu_guard.sh r3guard3 tests/test_gpu_sharded_tri_pnp.py tests/test_gpu_two_view.py tests/test_gpu_linear_and_incremental.py
mkdir -p gpurun_out/r3h
python -m pytest tests/test_gpu_linear_and_incremental.py tests/test_gpu_parity.py -m gpu -q -x --timeout 300 --timeout-method=thread -k "incremental or processors or drop" 2>&1 | tail -3
python bench.py --config C5 --steps 18 --warmup 1 2>&1 | grep '^{' > gpurun_out/r3h/bench_c5.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r3h/bench_c5.json"))
print("C5 views/s %.1f ms/view %.2f" % (d["value"], d["ms_per_step"]))
for v in d["per_view"]:
    print({k: (round(x * 1e3, 3) if k.endswith("_s") else x) for k, x in v.items()})
PY
python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>&1 | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('C3', d['value'])"
