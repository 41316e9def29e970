#!/bin/bash
# round 2: the whole -m gpu suite, then the bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2/all_gpu_tests.log 2>&1; echo "all gpu tests rc=$?"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/all_gpu_tests.log | tail -40 | cut -c1-220
timeout -k 10 300 python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; echo "bench rc=$?"
cut -c1-1500 gpurun_out/r2/bench.json; tail -3 gpurun_out/r2/bench.err
