#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_cnn_gpu.py -m gpu -q --timeout 600 -p no:cacheprovider -x > gpurun_out/pytest_cnn.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -40 gpurun_out/pytest_cnn.log
exit $rc
