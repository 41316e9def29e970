#!/bin/bash
# round 2 closing pass: whole -m gpu suite, smoke, default bench (with the CPU baseline), rocprof kernel stats + PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 1000 python -m pytest tests -q -m gpu > gpurun_out/r2/all_gpu_tests.log 2>&1; echo "all gpu tests rc=$?"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/all_gpu_tests.log | tail -8 | cut -c1-220
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2
timeout -k 10 300 python bench.py > gpurun_out/r2/bench.json 2> gpurun_out/r2/bench.err; echo "bench rc=$?"
bash tools/gpu/collect_profiles.sh r02 > gpurun_out/r2/collect.log 2>&1; echo "profiles rc=$?"
tail -3 gpurun_out/r2/collect.log
timeout -k 10 300 python tools/bench_train.py > gpurun_out/r2/bench_train.json 2>/dev/null; echo "bench_train rc=$?"
timeout -k 10 200 python tools/time_features.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/time_features.log
TUNE_X0=1 timeout -k 10 200 python tools/tune_model.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r2/tune_model_x0.log
