#!/bin/bash
# Round 6, GPU call 34: the driver's MULTI-GPU launch form of bench.py on the final tree, at the world size this box allows (1):
# torch.distributed.run -> RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment, RCCL process group, barrier + max over ranks.
# Inference (replicas) and the training leg (DDP over RCCL).
O=gpurun_out/r6c34; mkdir -p $O
export TMPDIR=/tmp
( time timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --cpu-seconds 0 --no-eager-leg --no-other-configs ) > $O/bench_torchrun_infer.json 2> $O/bench_torchrun_infer.err
tail -1 $O/bench_torchrun_infer.json | cut -c1-600; tail -3 $O/bench_torchrun_infer.err | cut -c1-200
( time timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --mode train --steps 5 --warmup 2 ) > $O/bench_torchrun_train.json 2> $O/bench_torchrun_train.err
tail -1 $O/bench_torchrun_train.json | cut -c1-900; tail -3 $O/bench_torchrun_train.err | cut -c1-200
