#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/train && mkdir -p gpurun_out/prof/train
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/train -- python3 tools/time_train.py > gpurun_out/prof/train.log 2>&1; echo "rc=$?"
grep -v amdgpu gpurun_out/prof/train.log | tail -3
find gpurun_out/prof/train -name "*kernel_stats.csv" | head -1 | xargs -r head -14 | cut -c1-170
