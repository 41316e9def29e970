#!/bin/bash
# round 2: B3_MTL forward kernel variants -- correctness first, then timing by parts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py tests/test_bench_path_gpu.py tests/test_training_gpu.py tests/test_inference_gpu.py -q -m gpu -x -k "b3mtl or layer0 or single_feature or timed or golden or gradients or sgd_step or sub_model or odd_large or head" > gpurun_out/r2/model_tests.log 2>&1; echo "model tests rc=$?"
tail -15 gpurun_out/r2/model_tests.log
for v in "8 1" "9 0"; do
  set -- $v
  echo "== x0 path, waves=$1 prefetch=$2"
  TUNE_X0=1 SMH_TCN_WAVES=$1 SMH_TCN_PREFETCH=$2 timeout -k 10 200 python tools/tune_model.py 2>&1 | grep -v amdgpu.ids | head -9
done
echo "== patches path, default"
timeout -k 10 200 python tools/tune_model.py 2>&1 | grep -v amdgpu.ids | head -9
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/r2/bench_model.json 2> gpurun_out/r2/bench_model.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r2/bench_model.json
