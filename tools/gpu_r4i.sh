#!/bin/bash
mkdir -p gpurun_out/r4i
for rep in 1 2; do for g in 4 2 8 16; do
  SFM_REDUCE_SLICES=$g SFM_HIP_LIBRARY=$PWD/gpurun_ab/slices.so timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('slices $g', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a gpurun_out/r4i/ab_reduce_slices.txt
done; done
