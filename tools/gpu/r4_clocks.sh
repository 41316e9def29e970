#!/bin/bash
# round 4: what the engine clock does under the bench's load (rocm-smi, read-only), idle / one XCD busy / full chip
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r4
O=gpurun_out/r4/clocks.txt
: > $O
echo "== idle" >> $O
rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|fclk|power|socket" >> $O
python3 bench.py --no-cpu-baseline --steps 20000 --warmup 50 --steady-steps 0 > gpurun_out/r4/clocks_bench.json 2>/dev/null &
BP=$!
sleep 2.5
for i in 1 2 3; do
  echo "== full chip, sample $i" >> $O
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i -E "sclk|mclk|fclk|power|socket" >> $O
  sleep 0.7
done
wait $BP
tail -1 gpurun_out/r4/clocks_bench.json | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'])" >> $O
cat $O
