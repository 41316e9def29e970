#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 1000 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
