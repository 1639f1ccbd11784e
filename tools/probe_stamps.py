"""Diagnostic (GPU box): build libcmbpo_hip with -DCMBPO_STAMPS into /tmp, run the ensemble forward once and
print the median cycles wave 0 of a workgroup spends per phase (shares, not absolute times)."""
import ctypes as C, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = os.path.join(ROOT, "constrained-model-based-policy-optimization_amd")
so = "/tmp/libcmbpo_stamps.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                       "-ffp-contract=off", "-DCMBPO_STAMPS"] + (["-DCMBPO_DIAG_NOLOAD"] if os.environ.get("NOLOAD") else []) + sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip"))) + ["-o", so])
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
tiles = (B + 31) // 32
stamps = torch.zeros(E * tiles * 12, dtype=torch.int64, device="cuda")
for _ in range(3):
    m.predict_ensemble(obs, act=act, out=(mean, var))
h = C.CDLL(so)
pad = int(sys.argv[2]) if len(sys.argv) > 2 else 0
h.cmbpo_debug_set_lds_pad(pad)
h.cmbpo_set_stagger(int(sys.argv[3]) if len(sys.argv) > 3 else 0)
h.cmbpo_debug_set_stamps.argtypes = [C.c_void_p]
h.cmbpo_debug_set_stamps(stamps.data_ptr())
print('stagger', sys.argv[3] if len(sys.argv) > 3 else 0)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
m.predict_ensemble(obs, act=act, out=(mean, var))
e1.record()
torch.cuda.synchronize()
print("lds pad", pad, "kernel ms", e0.elapsed_time(e1))
st10 = stamps.cpu().numpy().reshape(-1, 12).astype(np.int64)
st = st10[:, :8]
clk = (st10[:, 7] - st10[:, 0]) / np.maximum(st10[:, 9] - st10[:, 8], 1) * 100.0
print('in-kernel clock MHz: median %.0f  p10 %.0f  p90 %.0f' % (np.median(clk), np.percentile(clk, 10), np.percentile(clk, 90)))
print('wall span of the grid (realtime ticks @100MHz):', int(st10[:, 9].max() - st10[:, 8].min()), '-> ms', (st10[:, 9].max() - st10[:, 8].min()) / 1e5)
d = np.diff(st, axis=1)
names = ["x-stage", "bias+L0 mfma", "barrier+h1 store+barrier", "L1 mfma", "act+L2 mfma", "barrier", "red+epilogue"]
tot = st[:, 7] - st[:, 0]
print("workgroups", len(st), "median lifetime cycles", int(np.median(tot)))
for k, n in enumerate(names):
    print(f"  {n:28s} median {int(np.median(d[:, k])):8d}  mean {d[:, k].mean():10.0f}  share {d[:, k].mean() / tot.mean() * 100:5.1f}%")
print("ideal MFMA cycles per wave: L0 %d, L1 %d, L2 %d (x2 when the SIMD is shared)" % (80 * 64, 1024 * 64, 128 * 64))

# per-CU timelines from HW_ID (cu_id bits 8-11, sh_id 12, se_id 13-15 on gfx9) + XCC_ID
hw, xcc = st10[:, 10], st10[:, 11] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 50 + cu
uk = np.unique(key)
print("distinct CUs seen:", len(uk), "workgroups per CU: min %d max %d" % (np.bincount(np.searchsorted(uk, key)).min(), np.bincount(np.searchsorted(uk, key)).max()))
t0 = st10[:, 8].min()
occ, gaps = [], []
for k in uk[:64]:
    m = key == k
    s_, e_ = st10[m, 8] - t0, st10[m, 9] - t0
    order = np.argsort(s_)
    s_, e_ = s_[order], e_[order]
    busy = (e_ - s_).sum()
    span = e_.max() - s_.min()
    occ.append(busy / span)
    # time between a workgroup ending and the next one starting on this CU
    ends = np.sort(e_)
    for x in ends[:-2]:
        nxt = s_[s_ >= x]
        if len(nxt):
            gaps.append(nxt.min() - x)
print("mean concurrent workgroups per CU: %.2f" % np.mean(occ))
g = np.array(gaps)
print("end->next-start gap (us): median %.2f  p90 %.2f  mean %.2f" % (np.median(g) / 100, np.percentile(g, 90) / 100, g.mean() / 100))

occ = np.array(occ)
print("per-CU concurrency quantiles:", np.round(np.percentile(occ, [0, 10, 50, 90, 100]), 2))
print("gap quantiles (us):", np.round(np.percentile(g, [0, 10, 25, 50, 75, 90, 100]) / 100, 2))
# timeline of one CU
k = uk[5]
m = key == k
s_, e_ = (st10[m, 8] - t0) / 100.0, (st10[m, 9] - t0) / 100.0
order = np.argsort(s_)
print("CU timeline (us) start,end of first 12 workgroups:")
for a, b in list(zip(s_[order], e_[order]))[:12]:
    print("   %8.1f %8.1f  (%.1f)" % (a, b, b - a))
simd = (hw >> 4) & 3
print("wave-0 SIMD ids of the workgroups on that CU:", simd[m][order][:12])
