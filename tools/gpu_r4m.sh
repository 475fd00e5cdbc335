#!/bin/bash
mkdir -p gpurun_out/r4m
timeout -k 10 500 python -m pytest tests/test_gpu_sharded_tri_pnp.py tests/test_gpu_linear_and_incremental.py tests/test_gpu_chain_golden.py -m gpu -q -x --timeout 280 > gpurun_out/r4m/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4m/pytest.log
timeout -k 10 200 python tools/time_pnp_stages.py 2>/dev/null | grep -E "^n +(3000|3336|5000|9601)" | tee gpurun_out/r4m/time_pnp_flags.txt
