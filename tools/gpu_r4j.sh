#!/bin/bash
mkdir -p gpurun_out/r4j
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sharded_tri_pnp.py tests/test_gpu_property.py -m gpu -q -x --timeout 280 -k "tri" > gpurun_out/r4j/pytest_tri.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/r4j/pytest_tri.log
for rep in 1 2 3; do for v in base new2; do
  SFM_HIP_LIBRARY=$PWD/gpurun_ab/$v.so timeout -k 10 200 python bench.py --config TRI --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TRI $v', '%.4g' % d['value'], round(d['ms_per_step'],4), round(d['roofline']['frac'],4))" | tee -a gpurun_out/r4j/ab_tri.txt
done; done
