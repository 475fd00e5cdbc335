#!/bin/bash
# The N > 1 control flow of bench.py on a one-GPU box: N ranks on GPU 0, gloo where RCCL sits (SFM_BENCH_REHEARSAL).
# Not a measurement: it checks rendezvous, sharding, the all-reduce glue, the cross-rank camera check and the
# rank-0 JSON line.   bash tools/rehearse_ranks.sh [N]
n=${1:-2}
export SFM_BENCH_REHEARSAL=1 HSA_ENABLE_IPC_MODE_LEGACY=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus "$n" --steps 5 --warmup 2 --pts 4000 &&
python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus "$n" --steps 3 --warmup 1 --config C4
