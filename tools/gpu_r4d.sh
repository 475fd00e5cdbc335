#!/bin/bash
# Round 4: same-box A/B of ba_linearize builds (gpurun_ab/*.so), the split PnP class (tests, per-size timing, C5).
tag=${1:-r4d}
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
step 600 "$out/pytest_pnp.log" python -m pytest tests/test_gpu_sharded_tri_pnp.py tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -q -x --timeout 280 --timeout-method=thread -k "pnp or PnP"
tail -3 "$out/pytest_pnp.log"
step 300 "$out/time_pnp_stages.txt" python tools/time_pnp_stages.py
cat "$out/time_pnp_stages.txt" | grep "^n "
SFM_PNP_SPLIT_MIN=1025 step 300 "$out/time_pnp_stages_split1025.txt" python tools/time_pnp_stages.py
cat "$out/time_pnp_stages_split1025.txt" | grep "^n "
step 300 "$out/bench_c5.log" python bench.py --config C5 --steps 18 --warmup 1
grep '^{' "$out/bench_c5.log" | cut -c 1-400
# linearize variants, interleaved three times
for rep in 1 2 3; do for v in ${AB_VARIANTS:-base A B3 B}; do
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a "$out/ab_linearize.txt"
done; done
exit 0
