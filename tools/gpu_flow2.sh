#!/bin/bash
out=gpurun_out/flow2; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
for rep in 1 2 3; do for v in base reduce_u8; do
SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['rmse_px']['after_3_iterations'])" | tee -a $out/bench.txt
done; done
