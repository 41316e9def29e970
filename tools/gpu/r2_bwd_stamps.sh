#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for batch in 510 48; do
SMH_BWD_STAMPS=1 timeout -k 10 300 python tools/bench_train.py --batch $batch --steps 3 --warmup 1 2>&1 | grep -A1 "tcn_backward_mfma_kernel" | tail -2
done
