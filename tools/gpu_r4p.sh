#!/bin/bash
mkdir -p gpurun_out/r4p
timeout -k 10 300 python bench.py --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/r4p/bench_c4_share.log 2>&1; grep '^{' gpurun_out/r4p/bench_c4_share.log | cut -c 1-400
timeout -k 10 300 python bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline > gpurun_out/r4p/bench_c4_full.log 2>&1; grep '^{' gpurun_out/r4p/bench_c4_full.log | cut -c 1-400
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 280 -k "c4 or C4" > gpurun_out/r4p/pytest_c4.log 2>&1; echo "pytest rc=$?"; tail -1 gpurun_out/r4p/pytest_c4.log
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p/prof_c4share -- python3 bench.py --config C4 --pts 12500 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r4p/rocprof.log 2>&1
f=$(find gpurun_out/r4p/prof_c4share -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" gpurun_out/r4p/kernel_stats_c4share.csv && head -6 "$f" | cut -d, -f1-4
python3 tools/trace_iteration.py gpurun_out/r4p/prof_c4share > gpurun_out/r4p/trace_iteration_c4share.txt 2>&1; tail -2 gpurun_out/r4p/trace_iteration_c4share.txt
