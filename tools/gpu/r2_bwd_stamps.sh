#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
SMH_BWD_STAMPS=1 timeout -k 10 300 python tools/bench_train.py --steps 3 --warmup 1 2>&1 | grep "tcn_backward_mfma_kernel" | tail -3
