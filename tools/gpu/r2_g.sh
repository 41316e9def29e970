#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py tests/test_bench_path_gpu.py tests/test_bf16_gpu.py -q -m gpu -x > gpurun_out/r2/g_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/g_tests.log | tail -5 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
for r in 1 2; do timeout -k 10 300 python tools/bench_train.py 2>/dev/null | cut -c1-260 || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | cut -c1-300
