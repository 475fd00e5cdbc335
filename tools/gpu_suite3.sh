#!/bin/bash
# The GPU suite three times: plain, SFM_POOL_REDZONE=1, SFM_POOL_GUARD=1; then smoke and the driver-argument bench.   bash tools/gpu_suite3.sh <tag>
tag=${1:-suite3}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread > "$out/pytest_gpu.log" 2>&1; echo "plain rc=$?" | tee -a "$out/steps.log"; tail -2 "$out/pytest_gpu.log"
SFM_POOL_REDZONE=1 timeout -k 10 900 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread > "$out/pytest_gpu_redzone.log" 2>&1; echo "redzone rc=$?" | tee -a "$out/steps.log"; tail -1 "$out/pytest_gpu_redzone.log"
SFM_POOL_GUARD=1 timeout -k 10 1000 python -m pytest tests -m gpu -q --maxfail=3 --timeout 300 --timeout-method=thread > "$out/pytest_gpu_guard.log" 2>&1; echo "guard rc=$?" | tee -a "$out/steps.log"; tail -1 "$out/pytest_gpu_guard.log"
timeout -k 10 120 python __graft_entry__.py smoke > "$out/smoke.log" 2>&1; echo "smoke rc=$?" | tee -a "$out/steps.log"; tail -1 "$out/smoke.log"
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 3 > "$out/bench_driver_args.log" 2>&1; echo "bench rc=$?" | tee -a "$out/steps.log"; tail -c 300 "$out/bench_driver_args.log"
