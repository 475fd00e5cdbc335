#!/bin/bash
# small scenes: reduce tests, then per-iteration latency with the reduce inside the solve's launch and as its own launch
out=gpurun_out/ab3; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -m gpu -q -x -k "reduce or flow or property or small or graph" --timeout 300 > $out/pytest.log 2>&1; echo "pytest rc=$?" | tee $out/steps.log; tail -3 $out/pytest.log
[ "$(tail -1 $out/steps.log)" = "pytest rc=0" ] || exit 1
timeout -k 10 300 python tools/time_small.py 2>/dev/null | grep "mode 0 debug 0 graph 0\|mode 0 debug 16384" | cut -c1-250 | tee $out/small.txt
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3', round(d['value'],1), round(d['ms_per_step']*1e3,2))" | tee -a $out/small.txt
