#!/bin/bash
# schedule rule by tile count: parity tests (both schedules forced), model timings over batch sizes, default bench, training bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_parity_gpu.py tests/test_inference_gpu.py tests/test_bench_path_gpu.py -q -m gpu -x > gpurun_out/r2/sched_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/sched_tests.log | tail -4 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
TIME_N=48,256,510,768,1024 timeout -k 10 400 python tools/time_model_sizes.py 2>&1 | grep -v amdgpu.ids
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | cut -c1-400
timeout -k 10 300 python tools/bench_train.py 2>/dev/null | cut -c1-900
timeout -k 10 300 python tools/bench_train.py --batch 48 2>/dev/null | cut -c1-400
