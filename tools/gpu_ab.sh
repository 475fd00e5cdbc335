#!/bin/bash
# correctness of the data-flow solve, the chain's stamps, and C3 / C4-share benches against the column steps (debug 1024)
out=gpurun_out/ab; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 300 python tools/flow_check.py > $out/check.txt 2>&1; echo "check rc=$?" | tee $out/steps.log; grep -v amdgpu $out/check.txt | tail -4
[ "$(tail -1 $out/steps.log)" = "check rc=0" ] || exit 1
timeout -k 10 120 python tools/flow_check.py stamps > $out/stamps.txt 2>&1; echo "stamps rc=$?" | tee -a $out/steps.log; grep "^step  [0-2]\|^chain\|^deferred" $out/stamps.txt | cut -c1-330
for dbg in 0 16384 0 16384; do
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug $dbg', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['rmse_px']['after_3_iterations'])" | tee -a $out/bench.txt
done
for dbg in 0; do
timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4share debug $dbg', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $out/bench.txt
done
for v in 37 50 90; do timeout -k 10 100 python tools/ab_defer.py $v 2>/dev/null | tee -a $out/defer.txt; done
