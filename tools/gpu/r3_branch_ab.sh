#!/bin/bash
# round 3: the network kernel with fewer branches in the task body against the committed build (tools/ab/libsmh_base.so), same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -x -q -k "b3mtl or schedule or give_up or odd_large or gradients" 2>&1 | tail -1
for i in 1 2 3; do
  SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_base.so timeout -k 10 120 python3 tools/model_only.py 1024 300 2>/dev/null | sed 's/^/base /' || exit 1
  timeout -k 10 120 python3 tools/model_only.py 1024 300 2>/dev/null | sed 's/^/new  /' || exit 1
done
for v in base new; do
  if [ $v = base ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_base.so; else unset SMH_LIBSMH_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
done
