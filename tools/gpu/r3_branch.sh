#!/bin/bash
# round 3: skew schedule with fewer branches in the task body -- parity, launch time (compare with the figure of the same tool before)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -x -q -k "b3mtl or schedule or give_up or odd_large or gradients" > gpurun_out/r3/branch_pytest.log 2>&1; rc=$?
tail -3 gpurun_out/r3/branch_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2 3; do timeout -k 10 120 python3 tools/model_only.py 1024 300 2>/dev/null || exit 1; done
timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
"
