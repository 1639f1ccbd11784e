"""Dev tool: an 'uncertainty'-mode rollout with a sample budget (the default rollout_mode of the shipped configs) at
bench size -- per-step wall time against the alive count, to see what the bookkeeping costs next to the ensemble."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from cmbpo_amd import synthetic  # noqa: E402


def main():
    task, B = "AntSafe-v2", int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    w = bench.build_world(0, task)
    sampler, pool, env, policy = bench.build_hip(w, task, B, torch.device("cuda:0"))
    sampler.rollout_mode = "uncertainty"
    rng = np.random.default_rng(5)
    start = torch.from_numpy(synthetic.start_states(rng, B, task)).cuda()
    # calibrate the DKL limit like the trainer does (algorithms/cmbpo.py:197-199), then tighten it so branches die
    sampler.reset(start)
    lim = sampler.compute_dynamics_dkl(start[:5000], depth=5)
    for scale in (1.0, 0.6):
        sampler.set_rollout_dkl(lim * scale)
        for rep in range(2):
            sampler.reset(start)
            torch.cuda.synchronize()
            rows, t0 = [], time.perf_counter()
            while sampler.any_alive() and pool.has_room:
                n = pool.n_alive
                ts = time.perf_counter()
                _, _, _, info = sampler.sample(max_samples=int(2.5e6))
                torch.cuda.synchronize()
                rows.append((n, (time.perf_counter() - ts) * 1e3))
            diag = sampler.finish_all_paths()
            res, bd = pool.get(as_tensors=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
        tot = bd["poolm_batch_size"]
        print(f"dkl_lim x{scale}: {len(rows)} steps, {tot} samples in {dt * 1e3:.1f} ms = {tot / dt / 1e6:.2f} M steps/s")
        for n, ms in rows[:4] + rows[-3:]:
            print(f"   alive {n:7d}: {ms:6.3f} ms/step = {ms * 1e3 / max(n, 1) * 1e3:.2f} ns/row")


if __name__ == "__main__":
    main()
