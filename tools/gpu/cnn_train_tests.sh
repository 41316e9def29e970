#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_cnn_train_gpu.py -x -q -s --timeout 500 -p no:cacheprovider > gpurun_out/cnn_train_tests.log 2>&1
echo "rc=$?"; tail -40 gpurun_out/cnn_train_tests.log
