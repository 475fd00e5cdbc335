#!/bin/bash
out=gpurun_out/flow2; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_append.py tests/test_gpu_linear_and_incremental.py -m gpu -q -x --timeout 200 --timeout-method=thread > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/steps.log; tail -3 $out/pytest.log
for rep in 1 2; do timeout -k 10 300 python bench.py --config C5 --steps 18 --warmup 1 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', round(d['value'],1), round(d['ms_per_step'],3))" | tee -a $out/bench.txt; done
