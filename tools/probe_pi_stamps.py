"""Diagnostic (GPU box): build libcmbpo_hip with -DCMBPO_STAMPS into /tmp, run one Fisher-vector product on each matrix
path and print where wave 0 of a workgroup spends its cycles (shares: the stamps forbid overlaps the real kernel has)."""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
so = "/tmp/libcmbpo_stamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-DCMBPO_STAMPS"]
                      + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) + ["-o", so])
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd import _lib
_lib.LIB_PATH = so
from cmbpo_amd.cpo_update import PolicyOps
from worlds import make_update_batch
L = _lib.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3_400_000
D, A = 29, 8
rng = np.random.default_rng(0)
params, batch = make_update_batch(rng, N, D, A, 128, 0.3, 1.0, 35)
v = rng.standard_normal(params.shape).astype(np.float32)
h = C.CDLL(so)
h.cmbpo_debug_set_pi_stamps.argtypes = [C.c_void_p]
names = ["stage / saved activations + barrier", "x dW0 -> dh1", "h1 dW1", "barrier", "dh1 W1 -> dh2", "h2 dW2", "barrier",
         "dh2 W2 + reduction image + barrier", "element phase + barrier", "cot W2^T -> delta2 + barrier", "delta2 W1^T -> delta1",
         "gW1", "gW2 + barrier", "gW0 + bias sums + ring + barrier"]
for path in (0, 1):
    L.cmbpo_set_pi_matrix_path(path)
    ops = PolicyOps(D, A, 128, device="cuda:0")
    ops.set_params(params)
    ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"], batch["mu_old"],
             batch["log_std_old"])
    ops.loss_grad(0)
    ops.fvp(v)
    stamps = torch.zeros(512 * 16, dtype=torch.int64, device="cuda")
    h.cmbpo_debug_set_pi_stamps(stamps.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); ops.fvp(v); e1.record(); torch.cuda.synchronize()
    h.cmbpo_debug_set_pi_stamps(None)
    st = stamps.cpu().numpy().reshape(512, 16).astype(np.float64)
    tiles = (N + 31) // 32 / 512.0
    tot = st[:, :14].sum(axis=1)
    print(f"path {path}: kernel ms (stamped build) {e0.elapsed_time(e1):.3f}; clock MHz {np.median(tot / np.maximum(st[:, 14], 1) * 100):.0f}; "
          f"cycles per tile {tot.mean() / tiles:.0f}")
    for k, n in enumerate(names):
        print(f"  {n:40s} {st[:, k].mean() / tiles:8.0f} cycles/tile {st[:, k].mean() / tot.mean() * 100:5.1f}%")
