#!/bin/bash
# rows-path checks (parity tests of the sparse products) + the C4 share
out=gpurun_out/ab2; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -q -x -k "rows or schur or sparse or property or stats or cost" --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/steps.log; tail -3 $out/pytest.log
[ "$(tail -1 $out/steps.log)" = "pytest rc=0" ] || exit 1
for i in 1 2; do
timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4share', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $out/bench.txt
done
