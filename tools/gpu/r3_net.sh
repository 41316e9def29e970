#!/bin/bash
# round 3: the 16-wave skew schedule (weights in an LDS ring) against the 8-wave one: parity tests, then A/B of the default bench
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -k "b3mtl or skew or odd_large or layer0 or single_feature" > gpurun_out/r3/net_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r3/net_pytest.log
[ $rc -eq 0 ] || exit $rc
for v in 1 0 1 0; do
  SMH_TCN_SKEW16=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('skew16=$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()}, d['roofline']['frac'], d['parity']['max_abs_logit_diff_vs_oracle_golden'])
" || exit 1
done
