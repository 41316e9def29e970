#!/bin/bash
# BASELINE config 5's shape on one GPU: 5-class B3_MTL, bf16 (split-operand) network behind the f32 front end (21 x 11 medians)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r2
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --classes 5 --model-dtype bf16 --l-harm 21 --l-perc 11 > gpurun_out/r2/bench_config5.json 2>gpurun_out/r2/bench_config5.err || { tail -5 gpurun_out/r2/bench_config5.err; exit 1; }
cut -c1-700 gpurun_out/r2/bench_config5.json
timeout -k 10 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 --classes 5 --l-harm 21 --l-perc 11 2>/dev/null | cut -c1-300
