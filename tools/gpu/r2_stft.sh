#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_ragged_gpu.py -q -m gpu -x > gpurun_out/r2/stft_tests.log 2>&1; echo "tests rc=$?"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/stft_tests.log | tail -12 | cut -c1-220
run() { python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items() if k!='preprocess_signal'}, d['parity']['max_abs_logit_diff_vs_oracle_golden'])"; }
run "default"; run "default"
SMH_STFT_FRAMES=14,256 run "frames14"
SMH_STFT_FRAMES=20,256 run "frames20"
SMH_STFT_FRAMES=25,512 run "f25,512t"
