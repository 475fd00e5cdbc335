#!/bin/bash
# The N > 1 control flow of bench.py on a one-GPU box: N ranks on GPU 0, gloo where RCCL sits (SFM_BENCH_REHEARSAL).
# Not a measurement: it checks rendezvous, sharding, the all-reduce / gather glue, the cross-rank camera check and the
# rank-0 JSON line of every multi-rank config.   bash tools/rehearse_ranks.sh [N]
n=${1:-2}
export SFM_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
run() { port=$1; shift; python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port "$port" bench.py --gpus "$n" "$@"; }
# round 4: a plain `python3 bench.py --gpus N` (no launcher: the parent spawns the ranks itself), the full-size C3 line with
# its strong-scaled companion, and the reduce + broadcast exchange forced
python3 bench.py --gpus "$n" --steps 5 --warmup 2 --repeats 3 &&
python3 bench.py --gpus "$n" --steps 5 --warmup 2 --repeats 2 --pts 4000 --collective reduce_broadcast --scaling strong &&
run 29533 --steps 5 --warmup 2 --pts 4000 &&
run 29534 --steps 3 --warmup 1 --config C4 &&
run 29535 --steps 5 --warmup 1 --config TRI --pts 200000 &&
run 29536 --steps 5 --warmup 1 --config PNP --pts 500
