#!/bin/bash
# round 3: the median kernel taking the clips last-first (the data the kernel in front wrote last is the data still cached),
# against the plain order (SMH_MEDIAN_REVERSE=0), same build, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_bench_path_gpu.py tests/test_parity_gpu.py -x -q -k "median or bench or golden or hpss" 2>&1 | tail -2 || exit 1
for i in 1 2 3; do
  for v in 0 1; do
    SMH_MEDIAN_REVERSE=$v timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('reverse=$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
  done
done
