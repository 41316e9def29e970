#!/bin/bash
# round 3: deterministic gradients -- tests, then what the mode costs per training step (config 4's shape)
mkdir -p gpurun_out/r3
timeout -k 10 600 python -m pytest tests/test_training_gpu.py tests/test_cnn_train_gpu.py -x -q -k "deterministic or growing or config4 or data_parallel" > gpurun_out/r3/det_pytest.log 2>&1; rc=$?
tail -4 gpurun_out/r3/det_pytest.log
[ $rc -eq 0 ] || exit $rc
for flag in "" "--deterministic" "" "--deterministic"; do
  timeout -k 10 300 python tools/bench_train.py $flag 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('det' if d['deterministic_gradients'] else 'atomics', d['ms_per_step'], d['value'], d['stages_ms_serial'])
" || exit 1
done
