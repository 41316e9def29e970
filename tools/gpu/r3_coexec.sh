#!/bin/bash
# round 3: do exact-f32 MFMA and the VALU overlap in the network kernel?  SQ_VALU_MFMA_COEXEC_CYCLES (cycles in which vector and
# matrix instructions execute together) for the 8-wave and the 16-wave skew schedule
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
rm -rf gpurun_out/prof/cx && mkdir -p gpurun_out/prof/cx
rocprofv3 -L 2>/dev/null | grep -o "SQ_VALU_MFMA_[A-Z_]*\|SQ_INST_CYCLES_VALU\|SQ_ACTIVE_INST_VALU\|SQ_VALU_[A-Z_]*BUSY[A-Z_]*" | sort -u > gpurun_out/prof/cx/counters.txt
cat gpurun_out/prof/cx/counters.txt
pass() { n=$1; shift; timeout -k 10 300 rocprofv3 --pmc "$@" --output-format csv -d gpurun_out/prof/cx/$n -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --steady-steps 0 > gpurun_out/prof/cx/$n.log 2>&1; echo "$n rc=$?"; }
for v in 0 1; do
  export SMH_TCN_SKEW16=$v
  pass a$v SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES
  pass b$v SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS
done
python3 - <<'PY'
import csv, glob, collections, json
out = {}
for d in ("a0", "b0", "a1", "b1"):
    fs = glob.glob("gpurun_out/prof/cx/%s/*/*_counter_collection.csv" % d)
    if not fs:
        print(d, "no csv"); continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "b3mtl_forward" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    key = "skew8" if d.endswith("0") else "skew16"
    out.setdefault(key, {}).update({c: sum(v) / len(v) for c, v in agg.items()})
json.dump(out, open("gpurun_out/prof/cx/r03_model_coexec.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True))
PY
exit 0
