#!/bin/bash
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py -x -q -k "featuregram or timed or full_batch or randomised or layer0 or single_feature or residency or fused" > gpurun_out/r3/feat_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r3/feat_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['parity']['max_abs_logit_diff_vs_oracle_golden'])
" || exit 1
done
