#!/bin/bash
# SQ counters of one GEMM launch shape: where do the waves spend their cycles?  usage: gpu_pmc_sq.sh tag M N K la lb
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT \
    --kernel-trace -d $O/sq -o pmc --output-format csv -- python $R/tools/gemm_probe.py "$@" > $O/sq.log 2>&1
cd $R
python - <<PY
import csv, glob, collections
f = glob.glob("$O/sq/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    if "gemm" in r["Kernel_Name"] or "x3" in r["Kernel_Name"]:
        agg[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    print(k)
    wc = sum(d["SQ_WAVE_CYCLES"]) / len(d["SQ_WAVE_CYCLES"])
    for c, v in sorted(d.items()):
        m = sum(v) / len(v)
        print(f"   {c:28s} {m:14.0f}  {m / wc:7.3f} of WAVE_CYCLES")
PY
tail -1 $O/sq.log
