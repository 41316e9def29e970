#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
SMH_HEADS_STAMPS=1 timeout -k 10 300 python tools/bench_train.py --steps 3 --warmup 1 2>&1 | grep "heads_train_kernel" | tail -3
