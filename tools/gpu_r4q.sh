#!/bin/bash
mkdir -p gpurun_out/r4q
for dbg in 0 2048 4096 16384 6144 22528; do
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/ablate_solve.so timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 10 --warmup 2 --no-cpu-baseline --repeats 2 --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug $dbg', round(d['ms_per_step']*1e3,1), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a gpurun_out/r4q/ablate_trailing.txt
done
