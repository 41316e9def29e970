#!/bin/bash
# round 3: device RNG kernels -- their tests, the training suites that draw masks / noise through them, and the training bench
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_rng_gpu.py tests/test_training_gpu.py tests/test_driver_sequence_gpu.py tests/test_ragged_gpu.py -x -q > gpurun_out/r3/rng_pytest.log 2>&1; rc=$?
tail -15 gpurun_out/r3/rng_pytest.log
[ $rc -eq 0 ] || exit $rc
for i in 1 2; do
timeout -k 10 300 python tools/bench_train.py 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print(d['ms_per_step'], d['stages_ms_serial'], d['last_losses']['loss'])
" || exit 1
done
