#!/bin/bash
# round 2: training-surface tests (callbacks, compile, optimiser state, DP proof, sub-model Nadam) + the rest of the suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_cnn_train_gpu.py tests/test_inference_gpu.py -q -m gpu > gpurun_out/r2/train_tests.log 2>&1; echo "train tests rc=$?"
tail -30 gpurun_out/r2/train_tests.log
timeout -k 10 300 python tools/bench_train.py > gpurun_out/r2/bench_train.json 2> gpurun_out/r2/bench_train.err; echo "bench_train rc=$?"
cat gpurun_out/r2/bench_train.json; tail -3 gpurun_out/r2/bench_train.err
