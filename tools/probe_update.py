"""Perf probe of CPOPolicy.update_policy at N samples (dev tool)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd.cpo_policy import CPOPolicy
from worlds import make_update_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
constrained = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
D, A, T = 29, 8, 35
rng = np.random.default_rng(0)
params, batch = make_update_batch(rng, N, D, A, 128, 0.3, 1.0, T)


class S:
    def __init__(self, d): self.shape = (d,)


pol = CPOPolicy(S(D), S(A), a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128), vf_ensemble_size=3,
                vf_elites=2, vf_activation="swish", vf_loss="MSE", device="cuda:0", constrain_cost=constrained,
                cost_lim=10.0, target_kl=0.01, max_path_length=T)
dev = pol.device
pol.ops.keep_activations = os.environ.get("CMBPO_KEEP_ACT", "1") != "0"   # A/B of the saved-activation products
z = torch.zeros(N, device=dev)
t = lambda a: torch.from_numpy(a).to(dev)
buf = [t(batch["obs"]), t(batch["act"]), t(batch["adv"]), t(batch["cadv"]), z, z, t(batch["logp_old"]), z, z,
       t(batch["cost"]), t(batch["log_std_old"]), t(batch["mu_old"])]
times = []
for i in range(reps + 1):
    pol.set_params(params)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    info = pol.update_policy(buf)
    torch.cuda.synchronize()
    times.append((time.perf_counter() - t0) * 1e3)
print(f"N={N} constrained={constrained}: case {info['OptimCase']} backtrack {info['BacktrackIters']} "
      f"update ms: first {times[0]:.2f} median {np.median(times[1:]):.2f} min {min(times[1:]):.2f}")
from cmbpo_amd import _lib
print("cg graph launches:", _lib.lib().cmbpo_pi_cg_graph_launches())
