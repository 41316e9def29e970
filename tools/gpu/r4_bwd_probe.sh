#!/bin/bash
# round 4: timing probes of the split-bf16 backward kernel (results invalid under a probe): step time with phases switched off
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
export SMH_ENABLE_PROBES=1
for v in ${PROBES:-0 1 2 4 8 6 14 15}; do
  printf "probe %2d: " $v
  SMH_BWD_PROBE=$v timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])"
done | tee gpurun_out/r4/bwd_bf16_probe.txt
