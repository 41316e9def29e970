#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_ragged_gpu.py tests/test_inference_gpu.py -q -m gpu -x > gpurun_out/r2/feat_tests.log 2>&1; rc=$?; echo "feature tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/feat_tests.log | tail -5 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/time_features.py 2>&1 | grep -v amdgpu.ids | grep "half-clip" | tee gpurun_out/r2/time_features.log
