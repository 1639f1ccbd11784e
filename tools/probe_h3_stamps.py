"""Diagnostic (GPU box): build libcmbpo_hip with -DCMBPO_STAMPS into /tmp, run the f16 ensemble forward and print where
wave 0 of a workgroup spends its cycles (shares, not absolute times: the stamps forbid overlaps the real kernel has)."""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
so = "/tmp/libcmbpo_stamps.so"
EXTRA = sys.argv[2].split() if len(sys.argv) > 2 else []      # extra -D flags of the ens_h3.hip variant under test
objs = []
for src in sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))):      # every file: MlpKernelArgs carries the stamp pointer
    o = "/tmp/st_" + os.path.basename(src)[:-4] + ".o"
    if not (os.path.exists(o) and os.path.getmtime(o) > os.path.getmtime(src)) or src.endswith("ens_h3.hip"):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
                               "-DCMBPO_STAMPS"] + (["-fno-slp-vectorize"] + EXTRA if src.endswith("ens_h3.hip") else []) +
                              ["-c", src, "-o", o])
    objs.append(o)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", so])
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd import _lib
_lib.LIB_PATH = so
from cmbpo_amd import synthetic
from cmbpo_amd.pens import PE
lib = _lib.lib()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
rng = np.random.default_rng(0)
D, A, E = 29, 8, 7
ws, bs = synthetic.ensemble_weights(rng, E, D + A, 512, 2 * (D + 1))
m = PE(D + A, D + 1, hidden_dims=(512, 512), num_networks=E, num_elites=5, loss="MSPE", use_scaler_in=True,
       use_scaler_out=True, device="cuda:0")
m.set_weights(ws, bs, synthetic.scaler(rng, D + A), synthetic.scaler(rng, D + 1))
obs = torch.randn(B, D, device="cuda") * 0.5
act = torch.rand(B, A, device="cuda") * 2 - 1
mean = torch.empty(E, B, D + 1, device="cuda"); var = torch.empty_like(mean)
lib.cmbpo_set_ens_matrix_path(2)
for _ in range(3):
    m.predict_ensemble(obs, act=act, out=(mean, var))
stamps = torch.zeros(256 * 16, dtype=torch.int64, device="cuda")
h = C.CDLL(so)
h.cmbpo_debug_set_stamps.argtypes = [C.c_void_p]
h.cmbpo_debug_set_stamps(stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
m.predict_ensemble(obs, act=act, out=(mean, var))
e1.record()
torch.cuda.synchronize()
print("kernel ms (stamped build)", e0.elapsed_time(e1))
st = stamps.cpu().numpy().reshape(256, 16).astype(np.float64)
items = (B + 127) // 128 * E / 256.0
names = ["stage", "stage barrier", "prologue L0 + barrier", "chunk: L0 part", "chunk: L1 slabs", "chunk: W0 write + barrier",
         "prefetch issue", "tail: h2 epilogue", "tail: L2 (+partials)", "tail: barrier + head", "tail: barrier + stores", "tail: barrier"]
tot = st[:, :12].sum(axis=1)
clk = tot / np.maximum(st[:, 12], 1) * 100.0
print("in-kernel clock MHz (median over workgroups): %.0f" % np.median(clk))
print("cycles per item (mean over workgroups): %.0f   items per workgroup %.1f" % (tot.mean() / items, items))
for k, n in enumerate(names):
    print(f"  {n:28s} {st[:, k].mean() / items:9.0f} cycles/item  {st[:, k].mean() / tot.mean() * 100:5.1f}%")
print("ideal MFMA cycles per item and SIMD (2 waves): L0 %d, L1 %d, L2 %d" % (2 * 72 * 32, 2 * 768 * 32, 2 * 96 * 32))
