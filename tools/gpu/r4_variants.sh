#!/bin/bash
# round 4: the parity suites under the alternative implementations the environment switches select: every variant must be
# green -- tests that assert WHICH implementation ran skip themselves under a forcing switch (tests/test_*: _skip_if_forced)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
: > gpurun_out/r4/variants.txt
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_ragged_gpu.py tests/test_inference_gpu.py -q -p no:cacheprovider > gpurun_out/r4/variant_$name.log 2>&1
  echo "$name $* rc=$? $(tail -1 gpurun_out/r4/variant_$name.log)" | tee -a gpurun_out/r4/variants.txt; grep "^FAILED" gpurun_out/r4/variant_$name.log | cut -c1-160 | tee -a gpurun_out/r4/variants.txt
}
run skew2 SMH_TCN_SKEW=2
run skew0_nosplit SMH_TCN_SKEW=0 SMH_TCN_SPLIT=0
run feat_nopair SMH_FEAT_NOPAIR=1
run stft_generic SMH_STFT_GENERIC=1
run feat_two_kernels SMH_FEAT_TWO_KERNELS=1
run median_nosplit SMH_MEDIAN_NOSPLIT=1
run median_persist SMH_MEDIAN_PERSIST=1
run stft_plain_grid SMH_STFT_XCD=0
run ragged_per_file SMH_RAGGED_PERFILE=1
run dense_patches SMH_DENSE_PATCHES=1
tr() {
  name=$1; shift
  env "$@" timeout -k 10 600 python -m pytest tests/test_training_gpu.py -q -p no:cacheprovider > gpurun_out/r4/variant_train_$name.log 2>&1
  echo "train_$name $* rc=$? $(tail -1 gpurun_out/r4/variant_train_$name.log)" | tee -a gpurun_out/r4/variants.txt; grep "^FAILED" gpurun_out/r4/variant_train_$name.log | cut -c1-160 | tee -a gpurun_out/r4/variants.txt
}
tr nosplit SMH_TCN_SPLIT=0 SMH_BWD_SPLIT=0
tr train_valu SMH_TRAIN_VALU=1
tr dwh_valu SMH_DWH_VALU=1
tr heads_global SMH_HEADS_GLOBAL=1
tr skew2 SMH_TCN_SKEW=2
tr deterministic SMH_DETERMINISTIC=1
tr bwd_f32 SMH_BWD_BF16=0
tr dtrunk_in_kernel SMH_DTRUNK=0
