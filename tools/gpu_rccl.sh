#!/bin/bash
# bench.py under torch.distributed.run with ONE rank: RCCL (backend "nccl") in the loop of today's code on the one-GPU
# box -- process group, the all-reduce of the packed [S | rhs] tensor on the engine's stream, the MAX-over-ranks timing --
# plus the N = 2 / 4 rehearsals (gloo, all ranks on GPU 0).   bash tools/gpu_rccl.sh <tag>
tag=${1:-rccl}
out=gpurun_out/$tag
mkdir -p "$out"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
tr() { port=$1; shift; timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port "$port" bench.py --gpus 1 "$@"; }
tr 29541 --steps 100 --warmup 10 --no-cpu-baseline > "$out/torchrun_c3.log" 2>&1; echo "c3 rc=$?" | tee -a "$out/steps.log"
grep '^{' "$out/torchrun_c3.log" > "$out/bench_torchrun_1rank_rccl_c3.json"
# round 4: the library's own communicator inside sfm_ba_iterate (one C-ABI call for all K iterations), same box, interleaved
for rep in 1 2; do
  tr 2955$rep --steps 100 --warmup 10 --no-cpu-baseline --collective library > "$out/torchrun_c3_library$rep.log" 2>&1; echo "c3 library $rep rc=$?" | tee -a "$out/steps.log"
  grep '^{' "$out/torchrun_c3_library$rep.log" > "$out/bench_torchrun_1rank_library_c3_$rep.json"
  tr 2956$rep --steps 100 --warmup 10 --no-cpu-baseline > "$out/torchrun_c3_allreduce$rep.log" 2>&1; echo "c3 allreduce $rep rc=$?" | tee -a "$out/steps.log"
  grep '^{' "$out/torchrun_c3_allreduce$rep.log" > "$out/bench_torchrun_1rank_rccl_c3_$rep.json"
done
tr 29542 --config C4 --pts 12500 --steps 20 --warmup 3 --no-cpu-baseline > "$out/torchrun_c4share.log" 2>&1; echo "c4share rc=$?" | tee -a "$out/steps.log"
grep '^{' "$out/torchrun_c4share.log" > "$out/bench_torchrun_1rank_rccl_c4share.json"
tr 29543 --config TRI --steps 20 --warmup 2 --no-cpu-baseline > "$out/torchrun_tri.log" 2>&1; echo "tri rc=$?" | tee -a "$out/steps.log"
grep '^{' "$out/torchrun_tri.log" > "$out/bench_torchrun_1rank_rccl_tri.json"
tr 29544 --config PNP --steps 20 --warmup 2 --no-cpu-baseline > "$out/torchrun_pnp.log" 2>&1; echo "pnp rc=$?" | tee -a "$out/steps.log"
grep '^{' "$out/torchrun_pnp.log" > "$out/bench_torchrun_1rank_rccl_pnp.json"
for n in 2 4; do
  timeout -k 10 600 bash tools/rehearse_ranks.sh $n > "$out/rehearse$n.log" 2>&1; echo "rehearse $n rc=$?" | tee -a "$out/steps.log"
  grep '^{' "$out/rehearse$n.log" > "$out/rehearse$n.txt"
done
python - "$out" <<'PY'
import json, sys, glob
for f in sorted(glob.glob(sys.argv[1] + "/bench_torchrun*.json")) + sorted(glob.glob(sys.argv[1] + "/rehearse*.txt")):
    for line in open(f):
        d = json.loads(line)
        print(f.split("/")[-1], d["config"].get("collective_backend"), d["n_gpus"], "value %.4g" % d["value"], "ms/step %.4f" % d["ms_per_step"], d.get("max_camera_deviation_across_ranks"))
PY
exit 0
