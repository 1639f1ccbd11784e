"""Runs a few Fisher-vector products of the default path at N samples (profiling target; dev tool).  usage: probe_fvp_only.py [N] [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd.cpo_update import PolicyOps
from worlds import make_update_batch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_400_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
D, A = 29, 8
rng = np.random.default_rng(0)
params, batch = make_update_batch(rng, N, D, A, 128, 0.3, 1.0, 35)
v = rng.standard_normal(params.shape).astype(np.float32)
ops = PolicyOps(D, A, 128, device="cuda:0")
ops.set_params(params)
ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"], batch["mu_old"], batch["log_std_old"])
ops.loss_grad(0)
for _ in range(reps):
    ops.fvp(v)
ops.evals()
torch.cuda.synchronize()
print("done")
