"""Dev probe: time per launch of the bf16 (path 1) and f16 (path 2) ensemble forwards over small row counts, to place
cmbpo_set_ens_f16_min_rows.   python tools/probe_h3_small.py [task]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cmbpo_amd  # noqa: F401
from cmbpo_amd import _lib, synthetic
from cmbpo_amd.pens import PE

task = sys.argv[1] if len(sys.argv) > 1 else "AntSafe-v2"
obs_dim, act_dim = synthetic.ENV_DIMS[task]
rng = np.random.default_rng(0)
E = 7
ws, bs = synthetic.ensemble_weights(rng, E, obs_dim + act_dim, 512, 2 * (obs_dim + 1), bias_scale=0.05)
m = PE(obs_dim + act_dim, obs_dim + 1, hidden_dims=(512, 512), num_networks=E, num_elites=5, loss="MSPE",
       use_scaler_in=True, use_scaler_out=True, device="cuda:0")
m.set_weights(ws, bs, synthetic.scaler(rng, obs_dim + act_dim), synthetic.scaler(rng, obs_dim + 1))
lib = _lib.lib()
lib.cmbpo_set_ens_f16_min_rows(0)
for B in (128, 256, 512, 1000, 2000, 3000, 4000, 5000, 6000, 8000, 10000, 13000, 16000, 20000, 25000, 35000, 50000):
    obs = torch.randn(B, obs_dim, device="cuda") * 0.5
    act = torch.rand(B, act_dim, device="cuda") * 2 - 1
    mean = torch.empty(E, B, obs_dim + 1, device="cuda")
    var = torch.empty_like(mean)
    out = []
    for path, rt in ((1, 0), (2, 4), (2, 2), (2, 1), (2, 0)):
        lib.cmbpo_set_ens_matrix_path(path)
        lib.cmbpo_set_ens_f16_row_tiles(rt)
        for _ in range(5):
            m.predict_ensemble(obs, act=act, out=(mean, var))
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        iters = 50
        s.record()
        for _ in range(iters):
            m.predict_ensemble(obs, act=act, out=(mean, var))
        e.record()
        torch.cuda.synchronize()
        out.append(s.elapsed_time(e) / iters * 1e3)
    print(f"B={B:6d}  bf16x6 {out[0]:8.1f} us   f16x3 128-row {out[1]:8.1f}  64-row {out[2]:8.1f}  32-row {out[3]:8.1f}  cost model {out[4]:8.1f} us", flush=True)
lib.cmbpo_set_ens_f16_row_tiles(0)
