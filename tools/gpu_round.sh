#!/bin/bash
# One gpurun call = GPU tests + smoke + bench + rocprofv3 kernel stats (each call pays ~10 min of
# box acquisition, so everything rides in one).  Usage (from the repo root on the GPU box):
#   bash tools/gpu_round.sh [tag]
# Stops at the first step that is killed/timed out (rc >= 124); ordinary test failures do not stop it.
tag=${1:-r01}
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
step 600 "$out/pytest_gpu.log" python -m pytest tests -m gpu -q -x --timeout 300
tail -5 "$out/pytest_gpu.log"
step 120 "$out/smoke.log" python __graft_entry__.py smoke
tail -2 "$out/smoke.log"
step 240 "$out/bench.log" python bench.py --steps 20 --warmup 3
tail -1 "$out/bench.log"
step 240 "$out/bench_tri_pnp.log" python tools/bench_tri_pnp.py
tail -1 "$out/bench_tri_pnp.log"
step 240 "$out/bench_pairs.log" python bench.py --steps 5 --warmup 1 --schur pairs --no-cpu-baseline
tail -1 "$out/bench_pairs.log"
step 120 "$out/probe_schur.log" python tools/probe_schur.py
tail -1 "$out/probe_schur.log"
step 300 "$out/rocprof.log" rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline
find "$out/prof" -name "*kernel_stats*" | head -3
f=$(find "$out/prof" -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && head -14 "$f"
# HBM traffic of every kernel: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), no other trace domains
step 300 "$out/pmc_fetch.log" rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
step 300 "$out/pmc_write.log" rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline
python3 tools/parse_pmc.py "$out/pmc_fetch" "$out/pmc_write" "$out/traffic.json"
# matrix-pipe / LDS counters of every kernel (own pass, SQ block)
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
