#!/bin/bash
# round 3: the percussive rows of the median kernel stored with plain stores (the product build) against nontemporal stores
# (tools/ab/libsmh_nt.so, -DSMH_NT_STORES): kernel time and WRITE_SIZE, layout 2, 1024 clips, alternating
# A/B library (not tracked): hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -fno-honor-nans -DSMH_NT_STORES -Iinclude -c sm_hpss_mtl_amd/csrc/smh_median_split.hip -o /tmp/m.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libsmh_nt.so /tmp/m.o $(ls sm_hpss_mtl_amd/csrc/build/*.o | grep -v smh_median_split.o)
# (when profiles/r03_store_policies.txt was taken the roles were the other way round: product nontemporal, A/B build plain)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/median_nt
for i in 1 2 3; do
  for v in nt plain; do
    if [ $v = nt ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_nt.so; else unset SMH_LIBSMH_PATH; fi
    LAYOUT=2 ITERS=200 timeout -k 10 120 python3 tools/median_only.py | head -1 | sed "s/^/$v /" || exit 1
  done
done
for v in nt plain; do
  if [ $v = nt ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_nt.so; else unset SMH_LIBSMH_PATH; fi
  for c in WRITE_SIZE FETCH_SIZE; do
    LAYOUT=2 ITERS=20 timeout -k 10 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/median_nt/pmc_${c}_$v -- python3 tools/median_only.py > /dev/null 2>&1 || exit 1
    python3 - $v $c <<'PY'
import csv, glob, sys
v, c = sys.argv[1], sys.argv[2]
f = glob.glob("gpurun_out/median_nt/pmc_%s_%s/*/*_counter_collection.csv" % (c, v))[0]
vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "hpss_median_split" in r["Kernel_Name"] and r["Counter_Name"] == c]
print("%s %s per launch: %.1f MB over %d launches" % (v, c, sum(vals) / len(vals) * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6, len(vals)))
PY
  done
done
for v in nt plain; do
  if [ $v = nt ]; then export SMH_LIBSMH_PATH=$PWD/tools/ab/libsmh_nt.so; else unset SMH_LIBSMH_PATH; fi
  timeout -k 10 300 python bench.py --no-cpu-baseline --steady-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v', d['ms_per_step'], {k:v['ms'] for k,v in d['kernels'].items()})
" || exit 1
done
