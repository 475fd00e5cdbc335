#!/bin/bash
# Same-box A/B of gpurun_ab/{base,new}.so after the BA parity tests with the tree's library.   bash tools/gpu_ab.sh <tag> [pytest -k expr]
tag=${1:-ab}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py tests/test_gpu_append.py tests/test_gpu_linear_and_incremental.py -m gpu -q -x --timeout 280 --timeout-method=thread > "$out/pytest_ba.log" 2>&1
echo "pytest rc=$?" | tee "$out/steps.log"; tail -3 "$out/pytest_ba.log"
[ "$(tail -1 $out/steps.log)" = "pytest rc=0" ] || exit 1
for rep in 1 2 3; do for v in ${AB_VARIANTS:-base new}; do
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a "$out/ab.txt"
done; done
for v in ${AB_VARIANTS:-base new}; do
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4share $v', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a "$out/ab.txt"
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 100 python tools/time_small.py 2>/dev/null | grep "mode 0 debug 0 graph 0" | sed "s/^/$v /" | tee -a "$out/ab.txt"
done
exit 0
