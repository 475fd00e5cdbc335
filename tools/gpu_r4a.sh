#!/bin/bash
# Round 4, first GPU call: (1) can the reduced solve's launch chain run under a chip-filling launch (tools/microbench_overlap),
# (2) the SQ counter series that names what bounds ba_linearize, one rocprofv3 --pmc pass per group of
# <= 8 SQ counters, kernel-trace only, the program directly behind "--".   bash tools/gpu_r4a.sh [tag]
tag=${1:-r4a}
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
step 120 "$out/microbench_overlap.txt" tools/bin/microbench_overlap
cat "$out/microbench_overlap.txt"
step 120 "$out/counters_available.txt" rocprofv3 -L
grep -c "SQ_" "$out/counters_available.txt"
run_pmc() {   # run_pmc <name> <counters...>
  local name=$1; shift
  step 300 "$out/pmc_$name.log" rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/pmc_$name" -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-preroll
}
run_pmc g1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY
run_pmc g2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU
run_pmc g3 SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA
run_pmc g4 SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES
python3 - "$out" <<'PYEOF'
import csv, glob, os, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: [0, 0.0])
for f in glob.glob(os.path.join(out, "pmc_g*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        if "ba_linearize" in name: name = "ba_linearize"
        elif "ba_schur_mfma" in name: name = "ba_schur_mfma"
        elif "ba_chol_step" in name: name = "ba_chol_step"
        elif "ba_schur_reduce" in name: name = "ba_schur_reduce"
        elif "ba_inv_apply" in name: name = "ba_inv_apply"
        else: continue
        k = (name, r["Counter_Name"])
        acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
w = csv.writer(open(os.path.join(out, "pmc_sq_linearize.csv"), "w"))
w.writerow(["kernel", "counter", "dispatches", "avg_value_per_dispatch"])
for (k, c), (n, v) in sorted(acc.items()):
    w.writerow([k, c, n, v / n])
    if k == "ba_linearize": print(k, c, n, v / n)
PYEOF
exit 0
