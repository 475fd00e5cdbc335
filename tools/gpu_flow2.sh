#!/bin/bash
out=gpurun_out/flow2; mkdir -p $out; export TMPDIR=/tmp; rm -f $out/*
timeout -k 10 300 python tools/flow_check.py > $out/check.txt 2>&1; echo "check rc=$?" | tee $out/steps.log; grep -v amdgpu $out/check.txt | tail -3
[ "$(tail -1 $out/steps.log)" = "check rc=0" ] || exit 1
for a in 0 16384; do
timeout -k 10 120 python tools/flow_check.py stamps $a > $out/stamps_$a.txt 2>&1; echo "stamps $a rc=$?" | tee -a $out/steps.log; grep "^step\|^chain" $out/stamps_$a.txt | cut -c1-210
done
for dbg in 0 16384 0 16384; do
timeout -k 10 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline --repeats 3 --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('debug $dbg', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()}, d['rmse_px']['after_3_iterations'])" | tee -a $out/bench.txt
done
for dbg in 0 16384; do
timeout -k 10 200 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline --repeats 3 --debug $dbg 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4share debug $dbg', round(d['value'],1), round(d['ms_per_step']*1e3,2), {k: round(v*1e3,1) for k,v in d['kernel_ms'].items()})" | tee -a $out/bench.txt
done
