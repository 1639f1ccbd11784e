"""Dev probe of the three matrix paths of the 512-wide ensemble forward: error against the float64 oracle and time per
launch.    python tools/probe_h3.py [B] [iters] [task]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cmbpo_amd  # noqa: F401
from cmbpo_amd import _lib, synthetic
from cmbpo_amd.pens import PE
from oracle import refcpu

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
task = sys.argv[3] if len(sys.argv) > 3 else "AntSafe-v2"
paths = [int(p) for p in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0, 1, 2]
obs_dim, act_dim = synthetic.ENV_DIMS[task]
rng = np.random.default_rng(0)
E = 7
ws, bs = synthetic.ensemble_weights(rng, E, obs_dim + act_dim, 512, 2 * (obs_dim + 1), bias_scale=0.05)
sc_in, sc_out = synthetic.scaler(rng, obs_dim + act_dim), synthetic.scaler(rng, obs_dim + 1)
m = PE(obs_dim + act_dim, obs_dim + 1, hidden_dims=(512, 512), num_networks=E, num_elites=5, loss="MSPE",
       use_scaler_in=True, use_scaler_out=True, device="cuda:0")
m.set_weights(ws, bs, sc_in, sc_out)
lib = _lib.lib()
dev = m.device

# ---- accuracy on 777 rows against a float64 evaluation of the same network -------------------------------------------
n = 777
x = rng.standard_normal((n, obs_dim + act_dim)).astype(np.float32)
x[5] *= 1e3
x[6] *= 1e-4
ws64 = [w.astype(np.float64) for w in ws]
bs64 = [b.astype(np.float64) for b in bs]
r32 = refcpu.ens_forward(x, ws, bs, sc_in, sc_out)


def f64_forward(x):
    mu, var = sc_in
    sig = np.maximum(np.sqrt(var.astype(np.float32)), np.float32(1e-2)).astype(np.float64)
    h = (x.astype(np.float64) - mu.astype(np.float64)) / sig
    h = h[None]
    for l in range(3):
        h = h @ ws64[l] + bs64[l].reshape(E, 1, -1)
        if l < 2:
            h = h / (1.0 + np.exp(-h))
    omu, ovar = sc_out
    osig = np.maximum(np.sqrt(ovar.astype(np.float32)), np.float32(1e-2)).astype(np.float64)
    D = obs_dim + 1
    return osig * h[..., :D] + omu.astype(np.float64), np.exp(2 * np.log(osig) + h[..., D:])


mean64, var64 = f64_forward(x)
scale = np.abs(mean64).max(axis=(0, 1))
print("oracle(fp32 numpy) vs f64: max|d mean|/colscale = %.3e" % (np.abs(r32[0] - mean64) / scale).max())
for path in paths:
    _lib.check(lib.cmbpo_set_ens_matrix_path(path), "path")
    mean, var = m.predict_ensemble(x)
    em = (np.abs(mean - mean64) / scale).max()
    ev = (np.abs(var - var64) / np.abs(var64)).max()
    bad = int((~np.isfinite(mean)).sum())
    print(f"path {path}: max|d mean|/colscale = {em:.3e}  max rel d var = {ev:.3e}  non-finite = {bad}", flush=True)

# ---- time per launch ---------------------------------------------------------------------------------------------------
obs = torch.randn(B, obs_dim, device=dev) * 0.5
act = torch.rand(B, act_dim, device=dev) * 2 - 1
mean = torch.empty(E, B, obs_dim + 1, device=dev)
var = torch.empty_like(mean)
flop = 2.0 * E * ((obs_dim + act_dim) * 512 + 512 * 512 + 512 * 2 * (obs_dim + 1)) * B
res = {}
for rep in range(2):
    for path in paths:
        _lib.check(lib.cmbpo_set_ens_matrix_path(path), "path")
        for _ in range(3):
            m.predict_ensemble(obs, act=act, out=(mean, var))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            m.predict_ensemble(obs, act=act, out=(mean, var))
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        res.setdefault(path, []).append(ms)
        print(f"B={B} path={path}: {ms:.3f} ms/launch  {flop / ms / 1e9:.1f} TFLOP/s f32-equivalent", flush=True)
