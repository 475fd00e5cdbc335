#!/bin/bash
# Round 4: the chain goldens + the whole GPU suite + the self-spawning bench rehearsed with 2 ranks on the one GPU + the C3 line.
tag=${1:-r4c}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s)" | tee -a "$out/steps.log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a "$out/steps.log"
  if [ $rc -ge 124 ]; then echo "step killed; stopping" | tee -a "$out/steps.log"; exit $rc; fi
  return 0
}
step 300 "$out/pytest_chain.log" python -m pytest tests/test_gpu_chain_golden.py -q -x -s --timeout 280 --timeout-method=thread
grep -E "chain seed|Q13:|passed|failed|Error|assert" "$out/pytest_chain.log" | tail -30
step 900 "$out/pytest_gpu.log" python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys
tail -3 "$out/pytest_gpu.log"
step 300 "$out/bench.log" python bench.py --steps 20 --warmup 3
tail -c 600 "$out/bench.log"
SFM_BENCH_REHEARSAL=1 step 600 "$out/rehearse2_selfspawn.log" python3 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 3
grep '^{' "$out/rehearse2_selfspawn.log" | cut -c 1-1500
SFM_BENCH_REHEARSAL=1 step 300 "$out/rehearse2_repair.log" python3 bench.py --gpus 2 --steps 5 --warmup 2 --repeats 2 --pts 4000 --collective reduce_broadcast --scaling strong
grep '^{' "$out/rehearse2_repair.log" | cut -c 1-600
exit 0
