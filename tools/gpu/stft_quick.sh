#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "stft or frontend or fused or golden or e2e" --timeout 500 -p no:cacheprovider > gpurun_out/stft_tests.log 2>&1
echo "tests rc=$?"; tail -2 gpurun_out/stft_tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{\"metric\"'):
        d = json.loads(l); print(d['ms_per_step'], {k: v['ms'] for k, v in d['kernels'].items()})"; done
