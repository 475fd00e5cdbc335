#!/bin/bash
# Exploration pass of the property tests: fresh (not derandomised) hypothesis examples, more of them, every example
# appended to a trace file before it runs.   bash tools/gpu_explore.sh <tag> [examples per test]
tag=${1:-explore}
n=${2:-400}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
export SFM_HYPOTHESIS_RANDOM=1 SFM_HYPOTHESIS_EXAMPLES=$n SFM_TRACE_EXAMPLES="$out/examples.log"
timeout -k 10 1000 python -m pytest tests/test_gpu_property.py -m gpu -q --timeout 900 --timeout-method=thread --capture=sys > "$out/pytest_explore.log" 2>&1
echo "rc=$?" | tee -a "$out/steps.log"
tail -5 "$out/pytest_explore.log"
wc -l "$out/examples.log"
exit 0
