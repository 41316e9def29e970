#!/bin/bash
# round 2: the skewed (barrier-free) block schedule of the B3_MTL forward -- correctness first, then timing
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_inference_gpu.py -q -m gpu -x -k "b3mtl or layer0 or single_feature or timed or golden or odd_large or head" > gpurun_out/r2/skew_tests.log 2>&1
rc=$?; echo "model tests rc=$rc"; tail -8 gpurun_out/r2/skew_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
for v in "1 0" "0 0"; do
set -- $v
echo "== skew=$1 tune=$2"
SMH_TCN_SKEW=$1 SMH_TCN_TUNE=$2 TRACE_OUT=gpurun_out/r2/tcn_trace_$2.npy timeout -k 10 200 python tools/trace_model.py 2>&1 | grep -v amdgpu.ids | grep -v "mod 8" | head -8 || exit 1
SMH_TCN_SKEW=$1 TUNE_X0=1 TUNE_SHORT=1 timeout -k 10 200 python tools/tune_model.py 2>&1 | grep -v amdgpu.ids | head -4 || exit 1
done
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_skew.json 2> gpurun_out/r2/bench_skew.err; echo "bench rc=$?"; cut -c1-300 gpurun_out/r2/bench_skew.json
