"""Dev probe: time of ModelBuffer.get()'s kernels (offsets / moments / flatten) after a full-size rollout, and the HBM
rate of the flatten (algorithmic bytes: every stored float read once and written once).
    python tools/probe_get.py [B] [task] [maxroll] [schedule|uncertainty] [dkl scale]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import bench
import cmbpo_amd  # noqa: F401
from cmbpo_amd import synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
task = sys.argv[2] if len(sys.argv) > 2 else "AntSafe-v2"
maxroll = int(sys.argv[3]) if len(sys.argv) > 3 else bench.MAXROLL
mode = sys.argv[4] if len(sys.argv) > 4 else "schedule"
w = bench.build_world(0, task)
dev = torch.device("cuda:0")
sampler, pool, env, policy = bench.build_hip(w, task, B, dev, maxroll=maxroll, mode=mode)
start = torch.from_numpy(synthetic.start_states(np.random.default_rng(1), B, task)).to(dev)
if mode == "uncertainty":
    sampler.set_rollout_dkl(float(sampler.compute_dynamics_dkl(start[:5000], depth=5)) * (float(sys.argv[5]) if len(sys.argv) > 5 else 0.6))
sampler.reset(start)
while sampler.any_alive():
    sampler.sample()
sampler.finish_all_paths()
t = pool.t
D, A = pool.obs_dim, pool.act_dim


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


us_prep = timed(lambda: pool._call("cmbpo_buffer_prepare", t["offsets"].data_ptr(), t["stats"].data_ptr()))
us_off = timed(lambda: pool._call("cmbpo_buffer_offsets", t["offsets"].data_ptr()))
us_m0 = timed(lambda: pool._call("cmbpo_buffer_moments", 0, t["stats"].data_ptr()))
pool._call("cmbpo_buffer_moments", 1, t["stats"].data_ptr())
us_m2 = timed(lambda: pool._call("cmbpo_buffer_moments", 2, t["stats"].data_ptr()))
pool._call("cmbpo_buffer_moments", 3, t["stats"].data_ptr())
n = int(t["offsets"][B].item())
f = dict(dtype=torch.float32, device=dev)
dims = [D, A, 0, 0, 0, 0, 0, 0, 0, 0, A, A]
outs = [torch.empty((n, d) if d else (n,), **f) for d in dims]
ptrs = (C.c_void_p * 12)(*[o.data_ptr() for o in outs])
us_fl = timed(lambda: pool._call("cmbpo_buffer_flatten", t["offsets"].data_ptr(), t["stats"].data_ptr(), ptrs))
bytes_fl = 2.0 * 4.0 * n * (D + 3 * A + 8)
print(f"B={B} {task} maxroll {maxroll} {mode}: steps taken {pool.ptr}")
print(f"samples {n}: offsets {us_off:.1f} us, moments pass0 {us_m0:.1f} us, pass2 {us_m2:.1f} us, flatten {us_fl:.1f} us "
      f"= {bytes_fl / us_fl / 1e6:.2f} TB/s ({bytes_fl / 1e6:.0f} MB algorithmic)")
print(f"one-GPU path: prepare (scan + both moment passes, two launches) {us_prep:.1f} us; with flatten {us_prep + us_fl:.1f} us")
tot = us_off + us_m0 + us_m2 + us_fl
bytes_all = bytes_fl + 4.0 * n * (5 + 1)       # + the moment passes' reads
print(f"get() kernels {tot:.1f} us, {bytes_all / tot / 1e6:.2f} TB/s over all of them")
