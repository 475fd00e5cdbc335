#!/bin/bash
# rocprofv3 kernel trace of the C4 share (or "$@" bench arguments) and the timeline of its last iteration
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
args=${*:---config C4 --pts 12500 --steps 10 --warmup 2 --no-cpu-baseline}
rm -rf gpurun_out/prof_trace
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_trace -- python3 bench.py $args > gpurun_out/prof_trace.log 2>&1
f=$(find gpurun_out/prof_trace -name "*kernel_stats.csv" | head -1)
cut -d, -f1-4 "$f" | head -8
python3 tools/trace_iteration.py gpurun_out/prof_trace > gpurun_out/trace_iteration.txt
tail -64 gpurun_out/trace_iteration.txt
