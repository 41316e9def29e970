#!/bin/bash
# round 3: the training suite under the alternative implementations of the step
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r3
run() {
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests/test_training_gpu.py -q -p no:cacheprovider > gpurun_out/r3/tvariant_$name.log 2>&1
  echo "$name rc=$? $(tail -1 gpurun_out/r3/tvariant_$name.log)"; grep "^FAILED" gpurun_out/r3/tvariant_$name.log | cut -c1-160
}
run deterministic SMH_DETERMINISTIC=1
run nosplit SMH_TCN_SPLIT=0 SMH_BWD_SPLIT=0
run train_valu SMH_TRAIN_VALU=1
run dwh_valu SMH_DWH_VALU=1
run heads_global SMH_HEADS_GLOBAL=1
run skew2 SMH_TCN_SKEW=2
