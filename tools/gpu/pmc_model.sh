#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 -L 2>/dev/null | grep -iE "MFMA|SQ_INSTS_VALU|SQ_BUSY_CU|SQ_INST_CYCLES|LDS_BANK|SQ_LDS" | cut -c1-160 | sort -u | head -40 > gpurun_out/prof/counters_list.txt
cat gpurun_out/prof/counters_list.txt
rm -rf gpurun_out/prof/pmc_m1 gpurun_out/prof/pmc_m2
timeout -k 10 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d gpurun_out/prof/pmc_m1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_m1.log 2>&1; echo "pmc1 rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/prof/pmc_m2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/prof/pmc_m2.log 2>&1; echo "pmc2 rc=$?"; tail -3 gpurun_out/prof/pmc_m2.log | cut -c1-300
python3 - <<'PY'
import csv, glob, collections
for d in ("pmc_m1","pmc_m2"):
    fs = glob.glob("gpurun_out/prof/%s/*/*_counter_collection.csv" % d)
    if not fs: print(d, "no output"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        agg[(r["Kernel_Name"][:36], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for k,v in sorted(agg.items()):
        if any(s in k[0] for s in ("median","hp_feat","std_patch","stft","b3mtl")):
            print(k, "%.4g" % (sum(v)/len(v)))
PY
exit 0
