#!/bin/bash
# A/B of the whole bench step in steady state (bench.py defaults) on one box: schedule of the network's blocks, layer-0 weights of the feature kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k != 'preprocess_signal'}, d['roofline']['frac'])"; }
for rep in 1 2; do
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "default              " || exit 1
SMH_TCN_SKEW=0 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "barrier schedule     " || exit 1
SMH_FEAT_W0LDS=1 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "layer-0 weights in LDS" || exit 1
SMH_STFT_FRAMES=25,256 timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "stft 25 frames       " || exit 1
done
