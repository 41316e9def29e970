#!/bin/bash
# round 3: the lone last-round tile of the barrier schedule as two half tiles on two waves -- parity, timing over batch sizes, training bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py tests/test_training_gpu.py -x -q -k "b3mtl or schedule or gradients or give_up or odd_large" > gpurun_out/r3/split_pytest.log 2>&1; rc=$?
tail -5 gpurun_out/r3/split_pytest.log
[ $rc -eq 0 ] || exit $rc
TIME_N=1,48,128,256,510,1024 timeout -k 10 300 python tools/time_model_sizes.py 2>/dev/null || exit 1
TIME_W=99 TIME_N=48,510 timeout -k 10 300 python tools/time_model_sizes.py 2>/dev/null || exit 1
for sp in 1 0 1 0; do
SMH_TCN_SPLIT=$sp timeout -k 10 300 python tools/bench_train.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('split $sp', d['ms_per_step'], d['stages_ms_serial'])
" || exit 1
done
