#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
timeout -k 10 300 python tools/time_cnn.py > gpurun_out/time_cnn.log 2>&1; rc=$?; grep -v amdgpu.ids gpurun_out/time_cnn.log
if [ $rc -ne 0 ]; then exit $rc; fi
rm -rf gpurun_out/prof/cnn
ONLY=Doukhan timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof/cnn -- python3 tools/time_cnn.py > gpurun_out/prof/cnn.log 2>&1
find gpurun_out/prof/cnn -name "*kernel_stats.csv" | head -1 | xargs -r head -12 | cut -c1-200
exit 0
