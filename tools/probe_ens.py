"""Quick perf probe of the fused ensemble forward (dev tool, not the bench contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cmbpo_amd
from cmbpo_amd import _lib, synthetic
from cmbpo_amd.pens import PE

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows_list = [int(r) for r in sys.argv[3].split(",")] if len(sys.argv) > 3 else [32, 64]
task = "AntSafe-v2"
obs_dim, act_dim = synthetic.ENV_DIMS[task]
rng = np.random.default_rng(0)
E = 7
ws, bs = synthetic.ensemble_weights(rng, E, obs_dim + act_dim, 512, 2 * (obs_dim + 1))
m = PE(obs_dim + act_dim, obs_dim + 1, hidden_dims=(512, 512), num_networks=E, num_elites=5, loss="MSPE",
       use_scaler_in=True, use_scaler_out=True, device="cuda:0")
m.set_weights(ws, bs, synthetic.scaler(rng, obs_dim + act_dim), synthetic.scaler(rng, obs_dim + 1))
dev = m.device
obs = torch.randn(B, obs_dim, device=dev) * 0.5
act = torch.rand(B, act_dim, device=dev) * 2 - 1
mean = torch.empty(E, B, obs_dim + 1, device=dev); var = torch.empty_like(mean)
flop = 2.0 * E * ((obs_dim + act_dim) * 512 + 512 * 512 + 512 * 2 * (obs_dim + 1)) * B
stag_list = [int(r) for r in sys.argv[4].split(",")] if len(sys.argv) > 4 else [10]
mode_list = [int(r) for r in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
for rows, stag, mode in [(r, st, md) for r in rows_list for md in mode_list for st in stag_list]:
    _lib.check(_lib.lib().cmbpo_set_dispatch_mode(mode), "mode")
    _lib.check(_lib.lib().cmbpo_set_block_rows(rows), "rows")
    _lib.check(_lib.lib().cmbpo_set_stagger(stag), "stagger")
    for _ in range(3):
        m.predict_ensemble(obs, act=act, out=(mean, var))
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        m.predict_ensemble(obs, act=act, out=(mean, var))
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    print(f"B={B} block_rows={rows} mode={mode} stagger={stag}: {ms:.3f} ms/step  {flop/ms/1e9:.1f} TFLOP/s  ({flop/ms/1e9/157.3*100:.1f}% of 157.3)  {B/ms*1e3/1e6:.2f} M branch-steps/s", flush=True)
