#!/bin/bash
# round 3: backward phase 3, a lone last-round tile as two accumulator chains on two waves -- training tests, residency, A/B of the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_training_gpu.py tests/test_driver_sequence_gpu.py -x -q > gpurun_out/r3/bwd_split_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r3/bwd_split_pytest.log
[ $rc -eq 0 ] || exit $rc
for sp in 1 0 1 0 1 0; do
SMH_BWD_SPLIT=$sp timeout -k 10 300 python tools/bench_train.py --steps 200 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('bwd split $sp', d['ms_per_step'], d['stages_ms_serial'])
" || exit 1
done
