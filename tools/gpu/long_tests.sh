#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -k "randomised_lengths" > gpurun_out/pytest_long.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -25 gpurun_out/pytest_long.log
exit $rc
