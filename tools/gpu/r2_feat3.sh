#!/bin/bash
# features_half_kernel: layer-0 weights in LDS (two workgroups per CU) vs from L2 (three per CU), A/B on one box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items()}, d['parity']['max_abs_logit_diff_vs_oracle_golden'])"; }
for rep in 1 2 3; do
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "w0 from L2 (default)" || exit 1
SMH_FEAT_W0LDS=1 timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>/dev/null | show "w0 in LDS (old)" || exit 1
done
