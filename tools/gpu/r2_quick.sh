#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py -q -m gpu -x -k "schedules_agree or residency" > gpurun_out/r2/quick.log 2>&1; rc=$?; echo "rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/quick.log | tail -15 | cut -c1-220
