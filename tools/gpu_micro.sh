#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r02micro}; mkdir -p $O
cd $R
for v in "" _x3nomfma _x3nosplit _x3nosplitb _x3nostore _x3onlyload; do
  GT_HIP_LIB=libgt_hip$v.so timeout 200 python tools/x3_micro.py 2>/dev/null | tail -1 >> $O/micro.jsonl
done
GT_X3_PACKED=0 timeout 200 python tools/x3_micro.py 2>/dev/null | tail -1 | sed 's/libgt_hip.so/libgt_hip.so(unpacked)/' >> $O/micro.jsonl
python - <<PY
import json
rows=[json.loads(l) for l in open("$O/micro.jsonl")]
keys=[k for k in rows[0] if k not in("lib","T")]
print("%-22s"%"shape"+"".join("%14s"%r["lib"].replace("libgt_hip","").replace(".so","")[:13] for r in rows))
for k in keys: print("%-22s"%k+"".join("%14.1f"%r[k]["us"] for r in rows))
PY
