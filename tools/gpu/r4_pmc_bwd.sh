#!/bin/bash
# round 4: SQ counters of the split-bf16 backward kernel and of the exact-f32 one (tools/bench_train.py --serial, 510 clips, few steps);
# counters only, one pass per --pmc set (never together with a trace)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r4; rm -rf gpurun_out/prof/bw && mkdir -p gpurun_out/prof/bw
pass() { n=$1; dt=$2; shift 2; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/bw/$n -- python3 tools/bench_train.py --serial --dtype $dt --steps 6 --warmup 2 > gpurun_out/prof/bw/$n.log 2>&1; echo "$n rc=$?"; }
for dt in bf16 f32; do
pass ${dt}_p1 $dt SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
pass ${dt}_p2 $dt SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH
pass ${dt}_p3 $dt SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_LEVEL_LDS
done
python3 - <<'PY' 
import csv, glob, collections
for dt, key in (("bf16", "tcn_backward_bf16_kernel"), ("f32", "tcn_backward_mfma_kernel")):
    print("==", key)
    for d in ("p1", "p2", "p3"):
        fs = glob.glob("gpurun_out/prof/bw/%s_%s/*/*_counter_collection.csv" % (dt, d))
        if not fs:
            print(d, "no csv"); continue
        agg = collections.defaultdict(list)
        for r in csv.DictReader(open(fs[0])):
            if key in r["Kernel_Name"]:
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in sorted(agg.items()):
            print("%-32s %.6g  (per launch, n=%d)" % (c, sum(v) / len(v), len(v)))
PY
