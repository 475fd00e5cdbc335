#!/bin/bash
mkdir -p gpurun_out/r4v
for rep in 1 2 3; do for x in 0 1; do
  if [ $x = 1 ]; then export SFM_REDUCE_XCD=1; else unset SFM_REDUCE_XCD; fi
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/reduce_xcd.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reduce_xcd=$x', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['rmse_px']['after_3_iterations'])" | tee -a gpurun_out/r4v/ab_reduce_xcd.txt
done; done
