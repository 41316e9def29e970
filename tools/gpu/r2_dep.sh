#!/bin/bash
# A/B: dependency window of the skew schedule (dilations >= T reach nothing) -- SMH_TCN_TUNE=4 restores the wide window
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_inference_gpu.py -q -m gpu -x -k "schedules or golden or b3mtl or timed" > gpurun_out/r2/dep_tests.log 2>&1; rc=$?; echo "tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/dep_tests.log | tail -3 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k != 'preprocess_signal'}, d['roofline']['frac'])"; }
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "narrow window" || exit 1
SMH_TCN_TUNE=4 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "wide window  " || exit 1
done
