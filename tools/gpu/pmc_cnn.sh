#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/kc && mkdir -p gpurun_out/prof/kc
export ONLY=Doukhan N=64
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/kc/$n -- python3 tools/time_cnn.py > gpurun_out/prof/kc/$n.log 2>&1; echo "$n rc=$?"; }
pass p1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS
pass p2 SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES
python3 - <<'PY'
import csv, glob, collections
for d in ("p1", "p2"):
    fs = glob.glob("gpurun_out/prof/kc/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "conv_gemm_kernel<128>" in r["Kernel_Name"] and int(r["Grid_Size"]) > 4000000:
            agg[(r["Grid_Size"], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for c, v in sorted(agg.items()):
        print("%-40s %.4g (n=%d)" % (c, sum(v) / len(v), len(v)))
PY
