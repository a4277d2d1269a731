#!/usr/bin/env python3
"""The `cpu_baseline` leg of bench.py with the REFERENCE implementation (kind: "reference"), run where /root/reference
exists -- the build container -- and committed as profiles/cpu_baseline_reference.json; bench.py quotes it next to the number
it measures on the GPU box's own host cores (there only the oracle port is available: /root/reference does not travel)."""
import json
import os
import platform
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "galerkin-transformer_amd"))
import torch
import bench

load1 = os.getloadavg()[0]
if load1 > 1.0 and os.environ.get("GT_CPU_BASELINE_FORCE") != "1":
    # round 4's record (0.792 samples/s) was taken while a compile was running: 6.8x below what the idle box gives
    sys.exit(f"refusing to time the CPU baseline: 1-minute load average {load1:.2f} > 1 (something else is running)")

model, cfg = bench.build_model("ex2_darcy141")
sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
rec = bench.cpu_baseline_leg(sd, cfg, budget_s=float(sys.argv[1]) if len(sys.argv) > 1 else 25.0)
rec["host"] = {"cpu": platform.processor() or platform.machine(), "logical_cores": os.cpu_count(), "torch_threads": torch.get_num_threads(),
               "where": "build container (no GPU)",
               "loadavg_1min_before": round(load1, 2), "loadavg_1min_after": round(os.getloadavg()[0], 2)}
print(json.dumps(rec, indent=1))
if rec["kind"] == "reference":
    with open(os.path.join(ROOT, "profiles", "cpu_baseline_reference.json"), "w") as f:
        json.dump(rec, f, indent=1)
