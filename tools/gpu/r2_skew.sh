#!/bin/bash
# the skewed (barrier-free) block schedule of the B3_MTL forward -- correctness first, then the whole step in steady state
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_inference_gpu.py -q -m gpu -x -k "b3mtl or layer0 or single_feature or timed or golden or odd_large or head or schedules" > gpurun_out/r2/skew_tests.log 2>&1
rc=$?; echo "model tests rc=$rc"; tail -3 gpurun_out/r2/skew_tests.log | cut -c1-200
[ $rc -eq 0 ] || exit $rc
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k != 'preprocess_signal'}, d['roofline']['frac'])"; }
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "skew   " || exit 1
SMH_TCN_SKEW=0 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "barrier" || exit 1
done
