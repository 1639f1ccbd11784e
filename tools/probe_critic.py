"""Dev probe: both critics at B rows in one launch (critic_pair_kernel / critic_big_kernel) and the FakeEnv post kernel alone.
    python tools/probe_critic.py [B] [task]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
from cmbpo_amd import _lib, synthetic

B = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
task = sys.argv[2] if len(sys.argv) > 2 else "AntSafe-v2"
dev = torch.device("cuda:0")
w = bench.build_world(0, task)
sampler, pool, env, policy = bench.build_hip(w, task, B, dev, None, bench.MAXROLL, "schedule")
start = torch.from_numpy(synthetic.start_states(np.random.default_rng(1), B, task)).to(dev)
sampler.reset(start)
sampler.sample()


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


us_c = timed(lambda: sampler._critics("cur_obs", "v_t", "vc_t", B))
t = pool.t
v1 = t["v_t"].clone()
sampler._critics("cur_obs", "v_t", "vc_t", B)
print(f"B={B} {task}: both critics {us_c:.1f} us; repeatable {bool(torch.equal(v1, t['v_t']))}; finite {bool(torch.isfinite(t['v_t']).all())}")
E, O = env._model.num_nets, env.output_dim
mean = torch.randn((E, B, O), device=dev) * 0.1
var = torch.rand((E, B, O), device=dev) * 0.01 + 1e-4
inds = torch.randint(0, E, (B,), dtype=torch.int32, device=dev)
out = dict(next_obs=t["next_obs"], rew=t["rew_t"], term=t["term_t"], cost=t["cost_t"], dkl_path=t["dkl_t"], ep_var_mean=t["epv_t"])
lib = _lib.lib()


def post():
    _lib.check(lib.cmbpo_fakeenv_post(env._task_id, E, env.obs_dim, env.act_dim, mean.data_ptr(), var.data_ptr(), B,
                                      t["cur_obs"].data_ptr(), t["act_t"].data_ptr(), inds.data_ptr(), None, None, B,
                                      out["next_obs"].data_ptr(), out["rew"].data_ptr(), out["term"].data_ptr(),
                                      out["cost"].data_ptr(), out["dkl_path"].data_ptr(), out["ep_var_mean"].data_ptr(), None,
                                      _lib.current_stream()), "post")


us_p = timed(post)
print(f"FakeEnv post {us_p:.1f} us = {2.0 * E * O * 4 * B / us_p / 1e6:.2f} TB/s of (mean, var) reads")
