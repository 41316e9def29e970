#!/bin/bash
# round 4: where the split-bf16 backward kernel's time goes (in-kernel stamps of workgroup 0, waves 0 and 5)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
SMH_BWD_STAMPS=1 timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 --steps 3 --warmup 2 2>&1 | grep -a "tcn_backward" | tail -4 | tee gpurun_out/r4/bwd_bf16_stamps.txt
for v in "$@"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 tools/bench_train.py --serial --dtype bf16 2>/dev/null | tail -1 | cut -c1-200
done
