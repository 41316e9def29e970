#!/bin/bash
# round 2, first GPU pass: the new bench-path parity tests, the whole -m gpu suite, a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 900 python -m pytest tests/test_bench_path_gpu.py tests/test_bench_gpu.py -x -q -m gpu > gpurun_out/r2/bench_path_tests.log 2>&1; echo "bench-path tests rc=$?"
tail -5 gpurun_out/r2/bench_path_tests.log
timeout -k 10 300 python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; echo "bench rc=$?"
cat gpurun_out/r2/bench.json
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r2/all_gpu_tests.log 2>&1; echo "all gpu tests rc=$?"
tail -5 gpurun_out/r2/all_gpu_tests.log
