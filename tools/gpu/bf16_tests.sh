#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -q -s --timeout 300 -p no:cacheprovider > gpurun_out/pytest_bf16.log 2>&1; rc=$?
echo "pytest rc=$rc"; grep -E "bf16 vs|passed|failed|Error|assert" gpurun_out/pytest_bf16.log | tail -20
exit $rc
