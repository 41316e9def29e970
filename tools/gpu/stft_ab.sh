#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -m gpu -q --timeout 300 -p no:cacheprovider -x -k "stft or frontend or fused or pipeline or golden" > gpurun_out/pytest_stft.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 gpurun_out/pytest_stft.log
if [ $rc -ne 0 ]; then exit $rc; fi
show() { tail -1 $1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})"; }
for f in 25,256 25,512 33,512 49,512 20,256 20,512; do
SMH_STFT_FRAMES=$f timeout -k 10 300 python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/ab_stft_$f.log 2>&1 || exit 1; echo -n "F<=$f  "; show gpurun_out/ab_stft_$f.log
done
