#!/bin/bash
# Copy the judged summaries of one tools/gpu_round.sh run (gpurun_out/<tag>/) into profiles/<dest>/ and merge its
# per-workload HBM traffic into profiles/traffic.json.   bash tools/collect_profiles.sh r2p r2
src=gpurun_out/${1:?tag}
dst=profiles/${2:-r3}
mkdir -p "$dst"
for f in trace_iteration_c3.txt trace_iteration_c4share.txt kernel_stats_c3.csv kernel_stats_c4share.csv kernel_stats_tri.csv kernel_stats_pnp.csv pmc_sq_summary.csv traffic.json microbench_solve.txt microbench_elim.txt steps.log; do
  [ -f "$src/$f" ] && cp "$src/$f" "$dst/$f"
done
for f in bench bench_tri bench_pnp bench_c5 bench_c4_share bench_c4_full bench_tri_pnp probe_solve time_schur; do
  [ -f "$src/$f.log" ] && grep -E '^\{|kernel us|^\[' "$src/$f.log" | tail -20 > "$dst/$f.json"
done
for f in time_small stamps_small flow_stamps time_solve_paths; do
  [ -f "$src/$f.log" ] && grep -v amdgpu.ids "$src/$f.log" > "$dst/$f.txt"
  [ -f "$src/$f.txt" ] && grep -v amdgpu.ids "$src/$f.txt" > "$dst/$f.txt"
done
tail -3 "$src/pytest_gpu.log" > "$dst/pytest_gpu_tail.txt"
[ -f "$src/pytest_gpu_redzone.log" ] && tail -1 "$src/pytest_gpu_redzone.log" > "$dst/pytest_gpu_redzone_tail.txt"
tail -2 "$src/smoke.log" > "$dst/smoke_tail.txt"
python3 - "$src/traffic.json" profiles/traffic.json <<'PY'
import json, sys
new = json.load(open(sys.argv[1]))
try:
    doc = json.load(open(sys.argv[2]))
except Exception:
    doc = {}
for k, v in new.items():
    if "/" in k:
        doc[k] = v
doc["_format"] = "workload key = <config>/<cams>cams_<points per rank>pts_per_rank/<schur kernel>; values = HBM bytes per launch per kernel (tools/parse_pmc.py: (2 x FETCH_SIZE + WRITE_SIZE) x 1024, separate --pmc passes)"
json.dump(doc, open(sys.argv[2], "w"), indent=1, sort_keys=True)
print("traffic keys:", [k for k in doc if "/" in k])
PY
