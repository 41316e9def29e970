#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_bench_gpu.py -q -m gpu -x -s > gpurun_out/r2/bf16_tests.log 2>&1; rc=$?; echo "bf16 tests rc=$rc"
grep -v "amdgpu.ids\|^\[W\|Gloo" gpurun_out/r2/bf16_tests.log | grep "vs f32\|passed\|failed\|Error" | tail -14 | cut -c1-200
[ $rc -eq 0 ] || exit $rc
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['value'], d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()}, d['roofline']['frac'])"; }
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --model-dtype bf16 2>/dev/null | show "bf16 (split) from x0" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --model-dtype bf16 --no-fuse-l0 2>/dev/null | show "bf16 (split) from patches" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "f32" || exit 1
