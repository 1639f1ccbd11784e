"""Time per rollout step at a small batch (default 1000 AntSafe branches): loop of sample() against sample_many()
(cmbpo_rollout_run), and the fixed parts of a phase (reset, finish_all_paths, get).  Dev tool."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
from cmbpo_amd import synthetic
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
task = sys.argv[2] if len(sys.argv) > 2 else "AntSafe-v2"
dev = torch.device("cuda:0")
w = bench.build_world(0, task)
sampler, pool, env, policy = bench.build_hip(w, task, B, dev, None, bench.MAXROLL, "schedule")
start = synthetic.start_states(np.random.default_rng(1), B, task)
def sync(): torch.cuda.synchronize()
res = {}
for name in ("loop", "many", "loop", "many"):
    t = dict(reset=0.0, steps=0.0, finish=0.0, get=0.0); nsteps = 0
    for rep in range(5):
        sync(); t0 = time.perf_counter()
        sampler.reset(start); sync(); t1 = time.perf_counter()
        if name == "loop":
            while sampler.any_alive() and pool.has_room:
                sampler.sample()
        else:
            while sampler.any_alive() and pool.has_room:
                sampler.sample_many()
        sync(); t2 = time.perf_counter()
        sampler.finish_all_paths(); sync(); t3 = time.perf_counter()
        pool.get(as_tensors=True); sync(); t4 = time.perf_counter()
        if rep > 0:
            t["reset"] += t1 - t0; t["steps"] += t2 - t1; t["finish"] += t3 - t2; t["get"] += t4 - t3; nsteps += sampler._n_episodes
    print(f"{name}: {t['steps'] / nsteps * 1e6:.1f} us/step over {nsteps // 4} steps; per phase: reset {t['reset'] / 4 * 1e6:.0f} us, "
          f"finish {t['finish'] / 4 * 1e6:.0f} us, get {t['get'] / 4 * 1e6:.0f} us", flush=True)
