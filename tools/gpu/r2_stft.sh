#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_ragged_gpu.py -q -m gpu -x -k "stft or timed or golden or ragged or frontend or featuregram" > gpurun_out/r2/stft_tests.log 2>&1; rc=$?; echo "stft tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/stft_tests.log | tail -4 | cut -c1-220
[ $rc -eq 0 ] || exit $rc
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k in ('stft','median')}, d['parity']['max_abs_logit_diff_vs_oracle_golden'])"; }
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "stft" || exit 1
done
