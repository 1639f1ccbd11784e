"""Diagnostic (GPU box; usage: probe_pi_diag.py [-DDIAG_SKIP]): FVP timing with the fp32 small products skipped (wrong results; upper bound of what converting them buys)."""
import glob, os, subprocess, sys, shutil
import re
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
srcs = [f for f in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) if not f.endswith("policy_update.hip")]
tmp = "/tmp/diag_src"; os.makedirs(tmp, exist_ok=True)
for f in glob.glob(os.path.join(PKG, "csrc", "*.h")): shutil.copy(f, tmp)
src = open(os.path.join(PKG, "csrc", "policy_update.hip")).read()
src = re.sub(r'\n(\s*)(mfma_layer<1, 1, true>\([^;]*;)', r'\n#ifndef DIAG_SKIP\n\1\2\n#endif', src)
src = re.sub(r'\n(\s*)(wgrad_tile\(gW2[^;]*;)', r'\n#ifndef DIAG_SKIP\n\1\2\n#endif', src)
src = re.sub(r'\n#pragma unroll\n(\s*)(for \(int t = 0; t < N_IT; \+\+t\) wgrad_tile\(gW0[^;]*;)', r'\n#ifndef DIAG_SKIP\n\1\2\n#endif', src)
assert src.count("#ifndef DIAG_SKIP") >= 10
open(os.path.join(tmp, "policy_update.hip"), "w").write(src)
so = "/tmp/libcmbpo_diag.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                       "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(PKG, "csrc")] + sys.argv[1:2] + srcs + [os.path.join(tmp, "policy_update.hip")] + ["-o", so])
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd import _lib
_lib.LIB_PATH = so
from cmbpo_amd.cpo_update import PolicyOps
from worlds import make_update_batch
L = _lib.lib()
N, D, A = 3_400_000, 29, 8
rng = np.random.default_rng(0)
params, batch = make_update_batch(rng, N, D, A, 128, 0.3, 1.0, 35)
v = rng.standard_normal(params.shape).astype(np.float32)
ops = PolicyOps(D, A, 128, device="cuda:0")
ops.set_params(params)
ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"], batch["mu_old"], batch["log_std_old"])
ops.loss_grad(0)
def timeit(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
print(sys.argv[1:2], f"loss_grad {timeit(lambda: ops.loss_grad(0)):.3f} fvp {timeit(lambda: ops.fvp(v)):.3f} eval {timeit(ops.evals):.3f} ms", flush=True)
