#!/bin/bash
# round 3: the small measurement artifacts DESIGN.md quotes, as text files for profiles/
mkdir -p gpurun_out/r3/art
A=gpurun_out/r3/art
timeout -k 10 300 python tools/time_stft.py 20,256 25,256 33,512 20,256 25,256 33,512 17,256 14,256 20,256 2>/dev/null > $A/r03_stft_frames_sweep.txt || exit 1
( timeout -k 10 200 python tools/trace_model_small.py 256 2>/dev/null; timeout -k 10 200 python tools/trace_model_small.py 256 x0 2>/dev/null; timeout -k 10 200 python tools/trace_model_small.py 1024 x0 2>/dev/null; SMH_ENABLE_PROBES=1 SMH_TCN_NOHEADS=1 timeout -k 10 200 python tools/trace_model_small.py 1024 x0 2>/dev/null | sed 's/^/[without the Dense product] /' ) > $A/r03_model_phases.txt || exit 1
timeout -k 10 300 python tools/graph_probe.py 2>/dev/null | grep "^B =" > $A/r03_graph_probe.txt || exit 1
timeout -k 10 300 python tools/split_stream_probe.py 2>/dev/null > $A/r03_split_stream_probe.txt || exit 1
for flag in "" "--deterministic"; do timeout -k 10 300 python tools/bench_train.py $flag 2>/dev/null | tail -1; done > $A/r03_bench_train.jsonl || exit 1
bash tools/gpu/r3_net.sh 2>/dev/null | grep "^skew16" > $A/r03_skew16_ab.txt || exit 1
cat $A/*.txt | head -60
