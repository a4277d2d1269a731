#!/bin/bash
# LDS bank-conflict survey of the headline step: one rocprofv3 --pmc pass (SQ_LDS_BANK_CONFLICT, SQ_LDS_IDX_ACTIVE, SQ_BUSY_CYCLES)
# over two eager steps, per kernel symbol -> gpurun_out/<tag>/lds_conflicts.txt    usage: bash tools/gpu_lds_conflicts.sh <tag> [workload]
R=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=${1:-lds}; W=${2:-ex2_darcy141}
O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp; cd /tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d $O/ldspmc/p -o pmc --output-format csv -- \
    python $R/bench.py --workload $W --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-roofline --no-f32-leg --no-accuracy --strong-global-batch 0 > $O/lds.log 2>&1
cd $R
python - "$O/ldspmc" > $O/lds_conflicts.txt <<'PY'
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:110]
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_BUSY_CYCLES":
            calls[k] += 1
out = []
for k, c in rows.items():
    act, conf = c.get("SQ_LDS_IDX_ACTIVE", 0.0), c.get("SQ_LDS_BANK_CONFLICT", 0.0)
    out.append((conf, conf / act if act else 0.0, calls[k], k))
print("# bank-conflict cycles (sum over launches), conflict / active LDS cycles, launches, kernel")
for conf, ratio, n, k in sorted(out, reverse=True)[:40]:
    print(f"{conf:16.0f}  {ratio:6.3f}  {n:4d}  {k}")
PY
rm -rf $O/ldspmc
head -30 $O/lds_conflicts.txt
