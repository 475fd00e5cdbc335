#!/bin/bash
# A short gpurun call for the inner loop of a round: GPU tests (default pool), smoke, one bench line per config,
# small-scene latencies.   bash tools/gpu_quick.sh <tag> [pytest -k expression]
tag=${1:-q}
kexpr=${2:-}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
step() {   # step <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s)" | tee -a "$out/steps.log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a "$out/steps.log"
  if [ $rc -ge 124 ]; then echo "step killed; stopping" | tee -a "$out/steps.log"; exit $rc; fi
  return 0
}
export SFM_TRACE_EXAMPLES="$out/examples.log"
if [ -n "$kexpr" ]; then
  step 900 "$out/pytest_gpu.log" python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys -k "$kexpr"
else
  step 900 "$out/pytest_gpu.log" python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys
fi
tail -5 "$out/pytest_gpu.log"
step 120 "$out/smoke.log" python __graft_entry__.py smoke
tail -1 "$out/smoke.log"
step 300 "$out/bench.log" python bench.py
tail -c 600 "$out/bench.log"
step 240 "$out/bench_tri.log" python bench.py --config TRI --steps 20 --warmup 2
tail -c 400 "$out/bench_tri.log"
step 240 "$out/bench_pnp.log" python bench.py --config PNP --steps 20 --warmup 2
tail -c 400 "$out/bench_pnp.log"
step 300 "$out/bench_c5.log" python bench.py --config C5 --steps 18 --warmup 1
tail -c 400 "$out/bench_c5.log"
step 120 "$out/time_small.log" python tools/time_small.py
grep -v amdgpu.ids "$out/time_small.log" | tail -20
exit 0
