"""Perf / agreement probe of the policy kernels (loss_grad, Fisher-vector product, eval) on the two matrix paths (dev tool).
usage: probe_pi.py [N] [D] [A] [hidden]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd import _lib
from cmbpo_amd.cpo_update import PolicyOps
from worlds import make_update_batch

N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_400_000
D = int(sys.argv[2]) if len(sys.argv) > 2 else 29
A = int(sys.argv[3]) if len(sys.argv) > 3 else 8
H = int(sys.argv[4]) if len(sys.argv) > 4 else 128      # 256: one kernel set (fp32 MFMAs) whatever the path
rng = np.random.default_rng(0)
params, batch = make_update_batch(rng, N, D, A, H, 0.3, 1.0, 35)
L = _lib.lib()
v = rng.standard_normal(params.shape).astype(np.float32)
res = {}
for path in (0, 1):
    L.cmbpo_set_pi_matrix_path(path)
    for keep in (False, True):
        ops = PolicyOps(D, A, H, device="cuda:0")
        ops.keep_activations = keep
        ops.set_params(params)
        ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"],
                 batch["mu_old"], batch["log_std_old"])
        g, _ = ops.loss_grad(0)
        hv = ops.fvp(v)
        s = ops.evals()
        res[(path, keep)] = (g, hv, np.array(s))

        def timeit(fn, reps=5):
            fn(); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps): fn()
            e1.record(); torch.cuda.synchronize()
            return e0.elapsed_time(e1) / reps
        print(f"path {path} keep {int(keep)} N={N} D={D} A={A}: loss_grad {timeit(lambda: ops.loss_grad(0)):.3f} ms  "
              f"fvp {timeit(lambda: ops.fvp(v)):.3f} ms (x5) {timeit(lambda: ops.fvp(v), 40):.3f} ms (x40)  "
              f"eval {timeit(ops.evals):.3f} ms", flush=True)
        del ops
for keep in (False, True):
    for i, name in enumerate(("grad", "fvp", "sums")):
        a, b = res[(0, keep)][i], res[(1, keep)][i]
        print(f"keep {int(keep)} {name}: max|f16 - fp32| / max|fp32| = {np.max(np.abs(a - b)) / np.max(np.abs(a)):.3e}")
