"""Time the ensemble training step (tools, not a test): python tools/probe_train.py [E I H D loss batch steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmbpo_amd.pens import PE  # noqa: E402


def run(E, I, H, D, loss, batch, steps, n=200000):
    rng = np.random.RandomState(0)
    pe = PE(I, D, hidden_dims=(H, H), num_networks=E, num_elites=max(1, E - 2), loss=loss, use_scaler_in=True,
            use_scaler_out=True, device="cuda:0", lr=1e-3, decay=1e-6)
    pe.init_weights(rng)
    tr = pe._ensure_trainer(batch)
    x = torch.randn(n, I, device="cuda")
    t = torch.randn(n, D, device="cuda")
    idx = torch.randint(0, n, (E, n), dtype=torch.int32, device="cuda")
    nb = n // batch
    for k in range(5):
        tr.step(x, t, idx.data_ptr() + 4 * k * batch, n, batch)
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(steps):
        tr.step(x, t, idx.data_ptr() + 4 * (k % nb) * batch, n, batch)
    torch.cuda.synchronize()
    dt = (time.time() - t0) / steps
    fl = 6.0 * E * batch * (I * H + H * H + H * (2 * D if loss == "MSPE" else D))
    print(f"E={E} I={I} H={H} D={D} {loss} batch={batch}: {dt * 1e6:.1f} us/step, {fl / dt / 1e12:.2f} TFLOP/s (3x fwd), "
          f"{E * batch / dt / 1e6:.2f} M member-rows/s")
    ne = (n // batch) * batch
    eidx = idx[:, :ne].contiguous()
    tr.epoch(x, t, eidx, batch)
    torch.cuda.synchronize()
    t0 = time.time()
    tr.epoch(x, t, eidx, batch)
    torch.cuda.synchronize()
    print(f"   epoch entry point: {(time.time() - t0) / (ne // batch) * 1e6:.1f} us/step over {ne // batch} steps")
    hold = torch.arange(5000, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.time()
    for k in range(20):
        tr.losses(x, t, hold, 0, 5000)
    torch.cuda.synchronize()
    print(f"   holdout losses (5000 rows): {(time.time() - t0) / 20 * 1e6:.1f} us")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = sys.argv[1:]
        run(int(a[0]), int(a[1]), int(a[2]), int(a[3]), a[4], int(a[5]), int(a[6]))
    else:
        run(7, 37, 512, 30, "MSPE", 2048, 200)
        run(3, 29, 128, 1, "MSE", 2048, 500)
