#!/bin/bash
# the three ways bench.py gets started: plain (N = 1), under torch.distributed.run (what the driver does for N > 1), and asking for more GPUs than the box has
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-260; echo "torchrun form rc=$?"
timeout -k 10 300 python bench.py --gpus 2 --steps 5 --warmup 1 --no-cpu-baseline > /tmp/two.out 2>&1; echo "two GPUs asked on a one-GPU box rc=$?"; tail -2 /tmp/two.out | cut -c1-200
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 tools/bench_train.py --gpus 1 --steps 10 --warmup 3 2>/dev/null | cut -c1-200; echo "train torchrun form rc=$?"
