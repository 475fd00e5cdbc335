#!/bin/bash
mkdir -p gpurun_out/r4n
for p8 in 0 1; do
  SFM_ROWS_PITCH8=$p8 timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 280 -k "c4 or C4 or rows or schur" > gpurun_out/r4n/pytest_$p8.log 2>&1; echo "pitch8=$p8 pytest rc=$?"; tail -1 gpurun_out/r4n/pytest_$p8.log
done
for rep in 1 2 3; do for p8 in 0 1; do
  SFM_ROWS_PITCH8=$p8 timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4share pitch8=$p8', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a gpurun_out/r4n/ab_rows_pitch.txt
done; done
for p8 in 0 1; do SFM_ROWS_PITCH8=$p8 timeout -k 10 200 python bench.py --schur rows --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C3 rows pitch8=$p8', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a gpurun_out/r4n/ab_rows_pitch.txt; done
