#!/bin/bash
# the GPU suite (default pool), smoke, the default bench     bash tools/gpu_suite4.sh <tag>
tag=${1:-suite4}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SFM_TRACE_EXAMPLES="$out/examples.log"
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys > $out/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee $out/steps.log; tail -4 $out/pytest_gpu.log
[ "$(tail -1 $out/steps.log)" = "pytest rc=0" ] || exit 1
timeout -k 10 120 python __graft_entry__.py smoke > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee -a $out/steps.log
timeout -k 10 300 python bench.py > $out/bench.log 2>&1; echo "bench rc=$?" | tee -a $out/steps.log; tail -c 1500 $out/bench.log
