#!/bin/bash
# does the length of the untimed warm-up move the timed figure? (clock ramp of a cold box)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
show() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', d['ms_per_step'], {k:round(v['ms']*1000,1) for k,v in d['kernels'].items() if k != 'preprocess_signal'}, d['roofline']['frac'])"; }
timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | show "defaults" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 2>/dev/null | show "steps=20 warmup=3 (+preroll 50)" || exit 1
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 20 --warmup 3 --preroll 0 2>/dev/null | show "steps=20 warmup=3 preroll=0" || exit 1
timeout -k 10 300 python tools/bench_train.py 2>/dev/null | cut -c90-260
timeout -k 10 300 python tools/bench_train.py --steps 20 --warmup 3 2>/dev/null | cut -c90-260
timeout -k 10 600 python -m pytest tests/test_bench_gpu.py -q -m gpu 2>&1 | tail -2
