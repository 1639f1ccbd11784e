"""Dev tool: does the whole algorithm learn?  CMBPO on the toy point-mass environment of tests/test_cmbpo_loop_gpu.py,
printing the average episode return / cost per epoch.  python tools/run_loop_point.py [use_model 0|1] [epochs] [goal_x]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from cmbpo_amd import synthetic  # noqa: E402
from cmbpo_amd.cmbpo import CMBPO  # noqa: E402
from cmbpo_amd.cpo_policy import CPOPolicy  # noqa: E402
from cmbpo_amd.cpo_sampler import CpoSampler  # noqa: E402
from cmbpo_amd.cpobuffer import CPOBuffer  # noqa: E402
from test_cmbpo_loop_gpu import PointEnv  # noqa: E402


def main():
    use_model = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
    epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    np.random.seed(0)
    goal = (float(sys.argv[3]), 0.0) if len(sys.argv) > 3 else (0.0, 0.0)     # goal x > 1 lies in the cost region
    env = PointEnv(seed=1, goal=goal)
    T = 50
    policy = CPOPolicy(env.observation_space, env.action_space, a_hidden_layer_sizes=(128, 128),
                       vf_hidden_layer_sizes=(128, 128), vf_ensemble_size=3, vf_elites=2, vf_activation="swish",
                       vf_loss="MSE", vf_lr=1e-3, vf_epochs=4, vf_batch_size=256, device="cuda:0", max_path_length=T,
                       cost_lim=5.0, target_kl=0.01, discount=0.97, lam=0.95)
    policy.set_params(synthetic.policy_params(np.random.default_rng(2), 6, 2, 128))
    rng = np.random.RandomState(1)
    policy.v.init_weights(rng)
    policy.vc.init_weights(rng)
    buf = CPOBuffer(2000, 100000, env.observation_space, env.action_space)
    algo = CMBPO(env, policy, buf, sampler=CpoSampler(max_path_length=T), task="default", n_env_interacts=10 ** 9,
                 eval_every_n_steps=1, use_model=use_model, m_train_freq=1000, m_networks=5, m_elites=3,
                 m_hidden_dims=(128, 128), rollout_batch_size=2000, rollout_mode="schedule", rollout_schedule=[0, 1, 5, 5],
                 maxroll=6, initial_real_samples_per_epoch=1000, min_real_samples_per_epoch=500, batch_size_policy=10000,
                 n_initial_exploration_steps=2000 if use_model else 0, n_epochs=epochs,
                 initial_model_train_kwargs=dict(min_epochs=10, max_epochs=30, batch_size=256),
                 model_train_kwargs=dict(min_epochs=1, max_epochs=5, batch_size=256))
    rets = []
    for k, d in enumerate(algo.train()):
        rets.append(d.get("RetEpAverage", float("nan")))
        print(f"epoch {k:3d}: RetEp {d.get('RetEpAverage', float('nan')):8.3f}  CostEp {d.get('CostEpAverage', float('nan')):6.3f} "
              f"KL {d.get('KL', float('nan')):.4f} LossV {d.get('LossVEnsemble', float('nan')):.3f} "
              f"n_real {d.get('model/n_real_samples', 0):.0f} samples {d.get('model/samples_added', 0):.0f}", flush=True)
        if d.get("done") or k + 1 >= epochs:
            break
    print("first 5 epochs mean return %.3f, last 5 %.3f" % (np.nanmean(rets[:5]), np.nanmean(rets[-5:])))


if __name__ == "__main__":
    main()
