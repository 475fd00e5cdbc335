#!/bin/bash
# the resident RANSAC -> PnP session: its tests, the chain goldens through it, C5
mkdir -p gpurun_out/r4k
timeout -k 10 500 python -m pytest tests/test_gpu_linear_and_incremental.py tests/test_gpu_chain_golden.py -m gpu -q -x --timeout 280 > gpurun_out/r4k/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r4k/pytest.log
for rep in 1 2 3; do timeout -k 10 280 python bench.py --config C5 --steps 36 --warmup 2 > gpurun_out/r4k/bench_c5_$rep.log 2>&1; grep '^{' gpurun_out/r4k/bench_c5_$rep.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['per_view'][-1]
print('C5', v.get('pnp_native_calls_ms'), round(d['value'],1), 'views/s', round(d['ms_per_step'],3), 'ms; view 10: pnp %.2f (native %.2f) tri %.2f ba %.2f (native %.2f)' % (v['pnp_s']*1e3, v['pnp_native_s']*1e3, v['triangulate_s']*1e3, v['ba_s']*1e3, v['ba_native_s']*1e3))"; done
