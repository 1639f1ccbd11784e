"""How an 'uncertainty' rollout phase ends as a function of the DKL-limit scale and the sample budget (bench.py's
budget-binding sub-results need a phase of >= 12 steps that ends because max_samples bound).
    python tools/probe_budget.py [B]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda:0")
for scale in (2.0, 4.0, 6.0, 8.0, 12.0):
    for frac in (0.5,):
        try:
            out = bench.run_config(f"probe_s{scale}_f{frac}", "AntSafe-v2", B, 35, "uncertainty", dev, reps=1,
                                   dkl_scale=scale, budget_frac=frac)
            print(json.dumps({k: out[k] for k in ("name", "steps_per_phase", "samples_per_phase", "max_samples",
                                                  "n_budget_terminated", "ended_by", "us_per_step", "value")}), flush=True)
        except Exception as e:  # noqa: BLE001
            print("probe failed", scale, frac, repr(e), flush=True)
