#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -15
timeout -k 10 300 python bench.py --no-cpu-baseline --model-dtype bf16 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bf16 split', d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"
