"""System-level run of the CMBPO loop at AntSafe shapes on a synthetic environment (dev tool, not a test):
python tools/run_loop_antsafe.py [rollout_batch] [batch_size_policy] [epochs]
Prints the per-phase wall time of each epoch (rollout / real sampling / model training / policy + critic updates)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmbpo_amd import synthetic  # noqa: E402
from cmbpo_amd.cmbpo import CMBPO  # noqa: E402
from cmbpo_amd.cpo_policy import CPOPolicy  # noqa: E402
from cmbpo_amd.cpo_sampler import CpoSampler  # noqa: E402
from cmbpo_amd.cpobuffer import CPOBuffer  # noqa: E402


class _Space:
    def __init__(self, d):
        self.shape = (d,)


class LinearEnv:
    """Stable random linear dynamics with AntSafe's dimensions; z (obs[0]) kept inside the alive range."""

    def __init__(self, seed=0):
        self.observation_space, self.action_space = _Space(29), _Space(8)
        r = np.random.RandomState(seed)
        self.A = 0.95 * np.eye(29) + 0.01 * r.standard_normal((29, 29))
        self.B = 0.05 * r.standard_normal((8, 29))
        self.rng = r

    def reset(self):
        self.s = synthetic.start_states(np.random.default_rng(int(self.rng.randint(1 << 30))), 1, "AntSafe-v2")[0].astype(np.float64)
        return self.s.astype(np.float32)

    def step(self, a):
        a = np.clip(np.asarray(a, np.float64).reshape(-1)[:8], -1, 1)
        self.s = self.s @ self.A + a @ self.B + 0.01 * self.rng.standard_normal(29)
        self.s[0] = np.clip(self.s[0], 0.3, 0.9)
        return self.s.astype(np.float32), float(self.s[1]), False, {"cost": float(abs(self.s[-1]) > 1.0)}

    def close(self):
        pass


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
    bsp = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
    epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    np.random.seed(0)
    env = LinearEnv()
    T = 200
    policy = CPOPolicy(env.observation_space, env.action_space, a_hidden_layer_sizes=(128, 128),
                       vf_hidden_layer_sizes=(128, 128), vf_ensemble_size=3, vf_elites=2, vf_activation="swish",
                       vf_loss="MSE", vf_lr=3e-4, vf_epochs=8, vf_batch_size=2048, device="cuda:0", max_path_length=T,
                       cost_lim=10.0, target_kl=0.01)
    policy.set_params(synthetic.policy_params(np.random.default_rng(2), 29, 8, 128))
    rng = np.random.RandomState(1)
    policy.v.init_weights(rng)
    policy.vc.init_weights(rng)
    buf = CPOBuffer(6000, 300000, env.observation_space, env.action_space)
    algo = CMBPO(env, policy, buf, sampler=CpoSampler(max_path_length=T), task="AntSafe-v2", n_env_interacts=10 ** 9,
                 eval_every_n_steps=1, m_train_freq=1000, m_networks=7, m_elites=5, m_hidden_dims=(512, 512),
                 rollout_batch_size=B, rollout_mode="schedule", rollout_schedule=[0, 1, 10, 10], maxroll=11,
                 initial_real_samples_per_epoch=1500, min_real_samples_per_epoch=1000, batch_size_policy=bsp,
                 n_initial_exploration_steps=5000, n_epochs=epochs,
                 initial_model_train_kwargs=dict(min_epochs=20, max_epochs=40), model_train_kwargs=dict(min_epochs=1, max_epochs=10))
    t0 = time.perf_counter()
    k = 0
    for d in algo.train():
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        times = {key[6:]: round(v, 3) for key, v in d.items() if key.startswith("times/")}
        print(f"epoch {k}: wall {t1 - t0:.2f} s | {times} | samples_added {d.get('model/samples_added', 0):.0f} "
              f"n_real {d.get('model/n_real_samples', 0):.0f} val_loss {d.get('model/DynEns/val_loss', float('nan')):.4f} "
              f"LossV {d.get('LossVEnsemble', float('nan')):.4f} KL {d.get('KL', float('nan')):.5f}", flush=True)
        t0, k = t1, k + 1
        if d.get("done") or k >= epochs:
            break


if __name__ == "__main__":
    main()
