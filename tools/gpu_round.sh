#!/bin/bash
# One gpurun call = GPU tests + smoke + benches + rocprofv3 kernel stats + PMC traffic passes (every call pays minutes
# of box acquisition, so everything rides in one).  Usage (from the repo root on the GPU box):
#   bash tools/gpu_round.sh [tag]
# Stops at the first step that is killed / timed out (rc >= 124); ordinary test failures do not stop it.
tag=${1:-r4}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp
step() {   # step <seconds> <logfile> <cmd...>
  local secs=$1 log=$2; shift 2
  echo "== $* (limit ${secs}s)" | tee -a "$out/steps.log"
  timeout -k 10 "$secs" "$@" > "$log" 2>&1
  local rc=$?
  echo "   rc=$rc" | tee -a "$out/steps.log"
  if [ $rc -ge 124 ]; then echo "step killed; stopping" | tee -a "$out/steps.log"; exit $rc; fi
  return 0
}
export SFM_TRACE_EXAMPLES="$out/examples.log"
step 900 "$out/pytest_gpu.log" python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys
tail -3 "$out/pytest_gpu.log"
# the same suite with guard zones around every device buffer (out-of-bounds writes abort with a message)
SFM_POOL_REDZONE=1 step 900 "$out/pytest_gpu_redzone.log" python -m pytest tests -m gpu -q -x --timeout 300 --timeout-method=thread --capture=sys
tail -1 "$out/pytest_gpu_redzone.log"
step 120 "$out/smoke.log" python __graft_entry__.py smoke
tail -1 "$out/smoke.log"
step 300 "$out/bench.log" python bench.py
tail -c 400 "$out/bench.log"
step 240 "$out/bench_tri.log" python bench.py --config TRI --steps 20 --warmup 2
step 240 "$out/bench_pnp.log" python bench.py --config PNP --steps 20 --warmup 2
step 300 "$out/bench_c5.log" python bench.py --config C5 --steps 18 --warmup 1
step 240 "$out/bench_c4_share.log" python bench.py --config C4 --pts 12500 --steps 10 --warmup 2 --no-cpu-baseline
step 240 "$out/bench_c4_full.log" python bench.py --config C4 --steps 5 --warmup 1 --no-cpu-baseline
step 240 "$out/bench_tri_pnp.log" python tools/bench_tri_pnp.py
step 120 "$out/probe_solve.log" python tools/probe_solve.py
step 120 "$out/flow_stamps.txt" python tools/flow_check.py stamps
step 300 "$out/time_solve_paths.txt" python tools/time_solve_paths.py
step 120 "$out/time_schur.log" python tools/time_schur.py
step 120 "$out/time_small.log" python tools/time_small.py
step 60 "$out/stamps_small.log" python tools/stamps_small.py
step 60 "$out/microbench_solve.txt" tools/bin/microbench_solve
step 60 "$out/microbench_elim.txt" tools/bin/microbench_elim
# kernel durations of the two small solvers (roofline of SURVEY.md section 8(d) item 4)
for cfg in tri pnp; do
  up=$(echo $cfg | tr a-z A-Z)
  step 300 "$out/rocprof_$cfg.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$cfg" -- python3 bench.py --config $up --steps 20 --warmup 2 --no-cpu-baseline
  f=$(find "$out/prof_$cfg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_$cfg.csv" && head -4 "$f"
done
for cfg in c3 c4share; do
  if [ $cfg = c3 ]; then args="--no-cpu-baseline"; pargs="--steps 5 --warmup 3 --no-cpu-baseline"; else args="--config C4 --pts 12500 --steps 10 --warmup 2 --no-cpu-baseline"; pargs=$args; fi
  step 300 "$out/rocprof_$cfg.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof_$cfg" -- python3 bench.py $args
  f=$(find "$out/prof_$cfg" -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && cp "$f" "$out/kernel_stats_$cfg.csv" && head -12 "$f"
  python3 tools/trace_iteration.py "$out/prof_$cfg" > "$out/trace_iteration_$cfg.txt" 2>&1
  # HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), no other trace domains
  step 300 "$out/pmc_fetch_$cfg.log" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch_$cfg" -- python3 bench.py $pargs
  step 300 "$out/pmc_write_$cfg.log" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write_$cfg" -- python3 bench.py $pargs
  if [ $cfg = c3 ]; then blog="$out/bench.log"; else blog="$out/bench_c4_share.log"; fi
  key=$(python3 -c "import json,sys; print(json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])['roofline']['traffic_key'])" "$blog")
  python3 tools/parse_pmc.py "$out/pmc_fetch_$cfg" "$out/pmc_write_$cfg" "$out/traffic.json" "$key"
done
# matrix-pipe / LDS counters of every kernel (own pass, SQ block), C3
step 300 "$out/pmc_sq.log" rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$out/pmc_sq" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
python3 - "$out/pmc_sq" "$out/pmc_sq_summary.csv" <<'PYEOF'
import csv, glob, os, sys, collections
files = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)
acc = collections.defaultdict(lambda: [0, 0.0])
for f in files:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"].split("(")[0][-60:], r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
w = csv.writer(open(sys.argv[2], "w")); w.writerow(["kernel", "counter", "dispatches", "avg_value"])
for (k, c), (n, v) in sorted(acc.items()):
    w.writerow([k, c, n, v / n])
    if "schur_mfma" in k: print(k, c, n, v / n)
PYEOF
exit 0
