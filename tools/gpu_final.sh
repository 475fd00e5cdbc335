#!/bin/bash
# what the driver runs at round end: smoke, then the bench with its arguments   bash tools/gpu_final.sh <tag>
tag=${1:-final}; out=gpurun_out/$tag; mkdir -p $out; export TMPDIR=/tmp
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $out/smoke.log 2>&1; echo "smoke rc=$?" | tee $out/steps.log; tail -1 $out/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench_driver_args.log 2>&1; echo "bench rc=$?" | tee -a $out/steps.log
python3 -c "import json,sys; d=json.loads(open('$out/bench_driver_args.log').read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['value_runs'], d['roofline']['frac'])"
