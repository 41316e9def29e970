#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -x -q -k "b3mtl or layer0 or odd_large or gradients_and_losses_vs or two_conv" > gpurun_out/r3/dense_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r3/dense_pytest.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 200 python tools/trace_model_small.py 256 2>/dev/null
timeout -k 10 200 python tools/trace_model_small.py 1024 x0 2>/dev/null
