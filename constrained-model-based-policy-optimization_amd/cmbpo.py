"""CMBPO trainer loop -- host-side mirror of ``algorithms/cmbpo.py:30-542``.

The epoch structure is the reference's: Boltzmann-weighted start states from the archive -> imagined rollouts until the
model batch is full -> real-environment sampling (as many steps as the model's current uncertainty asks for) ->
dynamics-model training every ``m_train_freq`` real samples -> ``update_real_c`` / ``update_policy`` /
``update_critic`` on real + imagined samples -> diagnostics.  Everything data-parallel in it runs in the HIP kernels
behind the other modules of this package; imagined samples stay on the device from the rollout to the updates
(``ModelBuffer.get(as_tensors=True)``), the real samples of an epoch are uploaded once.

Differences from the reference, all outside the numerics: gtimer stamps are a plain ``time.perf_counter`` dict; the
logger is the in-memory ``logger.EpochLogger``; ``m_learn_cost`` (a learned cost head) is not supported, as in
``fake_env.FakeEnv``.
"""
import time
import warnings
from collections import OrderedDict
from itertools import count

import numpy as np
import torch

from .cpo_sampler import CpoSampler
from .fake_env import FakeEnv
from .logger import EpochLogger
from .model_sampler import ModelSampler
from .modelbuffer import ModelBuffer
from .pens import build_PE

EPS = 1e-8


def update_dict(dict_a, dict_b, weight_a=.5, weight_b=.5):
    """models/pens/logger.py:7-17: keys of b overwrite a, keys in both are blended."""
    out = dict(dict_a)
    out.update(dict_b)
    for k in dict_b:
        if k in dict_a:
            out[k] = weight_b * dict_b[k] + weight_a * dict_a[k]
    return out


def format_samples_for_dyn(samples, append_r=True, append_c=False):
    """models/pens/pe_factory.py:74-107: inputs = [obs | act], outputs = [next_obs - obs | rew (| cost)]."""
    obs, act, next_obs = samples['observations'], samples['actions'], samples['next_observations']
    inputs = np.concatenate((obs, act), axis=-1)
    outputs = next_obs - obs
    if append_r:
        outputs = np.concatenate((outputs, np.squeeze(samples['rewards'])[..., None]), axis=-1)
    if append_c:
        outputs = np.concatenate((outputs, np.squeeze(samples['costs'])[..., None]), axis=-1)
    return inputs, outputs


class CMBPO:
    """Constrained Model-Based Policy Optimization (``algorithms/cmbpo.py:30``)."""

    def __init__(self, env, policy, buffer, sampler=None, task='default', static_fns=None, n_env_interacts=1e7,
                 eval_every_n_steps=5e3, use_model=True, m_learn_cost=False, m_train_freq=250, m_loss_type='MSPE',
                 m_use_scaler_in=True, m_use_scaler_out=True, m_lr=1e-3, m_networks=7, m_elites=5,
                 m_hidden_dims=(512, 512), max_model_t=None, rollout_batch_size=10e3, sampling_alpha=1,
                 rollout_mode='uncertainty', rollout_schedule=(20, 100, 1, 1), maxroll=80,
                 initial_real_samples_per_epoch=5000, min_real_samples_per_epoch=500, batch_size_policy=25000,
                 n_epochs=int(10e7), n_initial_exploration_steps=0, initial_exploration_policy=None, epoch_length=1000,
                 model_train_kwargs=None, initial_model_train_kwargs=None, shuffle_on_device=True, device=None,
                 session=None, **_unused):
        if m_learn_cost:
            raise NotImplementedError("m_learn_cost: the learned cost head is unused by every shipped config")
        # RLAlgorithm.__init__ (algorithms/rl_algorithm.py:22-74)
        self.sampler = sampler if sampler is not None else CpoSampler(max_path_length=getattr(policy, "max_path_length", 1000))
        self._n_epochs, self._epoch_length = n_epochs, epoch_length
        self._n_initial_exploration_steps = n_initial_exploration_steps
        self._epoch = self._timestep = self._num_train_steps = 0

        self.obs_space, self.act_space = env.observation_space, env.action_space
        self.obs_dim = int(np.prod(env.observation_space.shape))
        self.act_dim = int(np.prod(env.action_space.shape))
        self.n_env_interacts = n_env_interacts
        self._task = task
        self.eval_every_n_steps = eval_every_n_steps
        self._training_environment = env
        self._policy = policy
        self._initial_exploration_policy = initial_exploration_policy or policy
        self.sampling_alpha = sampling_alpha
        self.device = torch.device(device) if device is not None else policy.device

        self._buffer = buffer
        pi_info_shapes = policy.pi_info_shapes
        self._buffer.initialize(pi_info_shapes, gamma=policy.gamma, lam=policy.lam, cost_gamma=policy.cost_gamma,
                                cost_lam=policy.cost_lam)
        self._use_model = use_model
        self._m_train_freq = m_train_freq
        self._rollout_batch_size = int(rollout_batch_size)
        self._rollout_schedule = list(rollout_schedule)
        self._max_model_t = max_model_t
        # train_model arguments of algorithms/cmbpo.py:452,335 (first fit: 150..500 epochs; later: 1..10)
        self._initial_model_train_kwargs = dict(min_epochs=150, max_epochs=500)
        self._initial_model_train_kwargs.update(initial_model_train_kwargs or {})
        self._model_train_kwargs = dict(min_epochs=1, max_epochs=10)
        self._model_train_kwargs.update(model_train_kwargs or {})
        # per-epoch shuffle_rows of PE.train on the GPU (same distribution, other random numbers): the host argsort of
        # an [E, n] array costs more than the epoch's kernels from n ~ 1e5 on.  False restores the reference's draws.
        self._shuffle_on_device = bool(shuffle_on_device)

        if use_model:
            self._model = build_PE(in_dim=self.obs_dim + self.act_dim, out_dim=self.obs_dim + 1, name='DynEns',
                                   loss=m_loss_type, hidden_dims=m_hidden_dims, lr=m_lr, num_networks=m_networks,
                                   num_elites=m_elites, use_scaler_in=m_use_scaler_in, use_scaler_out=m_use_scaler_out,
                                   decay=1e-6, max_logvar=.5, min_logvar=-10, device=self.device)
            self.fake_env = FakeEnv(true_environment=env, task=self._task, model=self._model, predicts_delta=True,
                                    predicts_rew=True, predicts_cost=False)
            self.rollout_mode = rollout_mode
            self.model_buf = ModelBuffer(batch_size=self._rollout_batch_size, obs_dim=self.obs_dim, act_dim=self.act_dim,
                                         max_path_length=maxroll, device=self.device)
            self.model_buf.initialize(pi_info_shapes, gamma=policy.gamma, lam=policy.lam, cost_gamma=policy.cost_gamma,
                                      cost_lam=policy.cost_lam)
            self.model_sampler = ModelSampler(max_path_length=maxroll, batch_size=self._rollout_batch_size,
                                              logger=None, rollout_mode=self.rollout_mode)
        self.init_real_samples = initial_real_samples_per_epoch
        self.min_real_samples = min_real_samples_per_epoch
        self.batch_size_policy = batch_size_policy
        self.logger = EpochLogger()
        self._policy.set_logger(self.logger)
        self.sampler.set_logger(self.logger)
        self.times = {}

    # -- bookkeeping of RLAlgorithm ------------------------------------------------------------------------
    @property
    def _total_timestep(self):
        return self.sampler._total_samples

    @property
    def _training_started(self):
        return self._total_timestep > 0

    @property
    def ready_to_train(self):
        return self.sampler.batch_ready()

    def _do_sampling(self, timestep):
        return self.sampler.sample(timestep=timestep)

    def _stamp(self, name, t0):
        self.times[name] = self.times.get(name, 0.0) + time.perf_counter() - t0
        return time.perf_counter()

    # -- hooks ---------------------------------------------------------------------------------------------
    def _initial_exploration_hook(self, env, initial_exploration_policy, pool):
        """algorithms/cmbpo.py:432-453: gather n_initial_exploration_steps real samples, then fit the model."""
        if self._n_initial_exploration_steps < 1:
            return
        if not initial_exploration_policy:
            raise ValueError("Initial exploration policy must be provided when n_initial_exploration_steps > 0.")
        self.sampler.initialize(env, initial_exploration_policy, pool)
        while True:
            self.sampler.sample(timestep=0)
            if self.sampler._total_samples >= self._n_initial_exploration_steps:
                self.sampler.finish_all_paths(append_val=True, append_cval=True, reset_path=False)
                pool.get()       # moves the policy samples to the archive
                break
        if self._use_model:
            self.train_model(**self._initial_model_train_kwargs)

    def train_model(self, min_epochs=5, max_epochs=100, batch_size=2048):
        """algorithms/cmbpo.py:455-486"""
        model_samples = self._buffer.get_archive(['observations', 'actions', 'next_observations', 'rewards', 'costs',
                                                  'terminals', 'epochs'])
        dyn_ins, dyn_outs = format_samples_for_dyn(model_samples, append_r=True, append_c=False)
        return self._model.train(dyn_ins, dyn_outs, batch_size=batch_size, max_epochs=max_epochs,
                                 min_epoch_before_break=min_epochs, holdout_ratio=0.2, max_t=self._max_model_t,
                                 shuffle_on_device=self._shuffle_on_device)

    def _set_rollout_length(self):
        """algorithms/cmbpo.py:494-512"""
        min_epoch, max_epoch, min_length, max_length = self._rollout_schedule
        if self._epoch <= min_epoch:
            y = min_length
        else:
            dx = min((self._epoch - min_epoch) / (max_epoch - min_epoch), 1)
            y = dx * (max_length - min_length) + min_length
        self._rollout_length = int(y)
        self.model_sampler.set_max_path_length(self._rollout_length)

    def _to_device(self, arrays):
        return [a if isinstance(a, torch.Tensor) else torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32,
                                                                      device=self.device) for a in arrays]

    # -- the loop ------------------------------------------------------------------------------------------
    def _train(self):
        """Generator over diagnostics dicts; ``{'done': True, ...}`` ends it (algorithms/cmbpo.py:177-430)."""
        env_r, policy, pool = self._training_environment, self._policy, self._buffer
        if not self._training_started:
            self._initial_exploration_hook(env_r, self._initial_exploration_policy, pool)
        self.sampler.initialize(env_r, policy, pool)
        if self._use_model:
            self.model_sampler.initialize(self.fake_env, policy, self.model_buf)
            obs0 = self._buffer.rand_batch_from_archive(5000, fields=['observations'])['observations']
            rollout_dkl_lim = self.model_sampler.compute_dynamics_dkl(obs_batch=obs0, depth=self._rollout_schedule[2])
            self.model_sampler.set_rollout_dkl(rollout_dkl_lim)
            self.initial_model_dkl = self.model_sampler.dyn_dkl
            self.approx_model_batch = self.batch_size_policy - self.init_real_samples
        self.policy_epoch = 0
        self.new_real_samples = 0
        self.diag_counter = 0
        running_diag = {}

        for self._epoch in range(self._epoch, self._n_epochs):
            t0 = time.perf_counter()
            self.times = {}
            samples_added = 0
            model_samples = None
            keep_rolling = True
            metrics = {}
            if self._use_model:
                if self.rollout_mode == 'schedule':
                    self._set_rollout_length()
                while keep_rolling:
                    # starting states: Boltzmann distribution over the archived epochs, temperature = policy KL
                    ep_b = self._buffer.epoch_batch(batch_size=self._rollout_batch_size, epochs=self._buffer.epochs_list,
                                                    fields=['observations', 'pi_infos'])
                    kls = np.clip(policy.compute_DKL(ep_b['observations'], ep_b['mu'], ep_b['log_std']), a_min=0, a_max=None)
                    btz_dist = self._buffer.boltz_dist(kls, alpha=self.sampling_alpha)
                    btz_b = self._buffer.distributed_batch_from_archive(self._rollout_batch_size, btz_dist,
                                                                        fields=['observations', 'pi_infos'])
                    self.model_sampler.reset(btz_b['observations'])
                    # the reference's loop over sample() (:352-360: stop at 99 % of the batch or at alive_ratio <= 0.1),
                    # taken in native code
                    self.model_sampler.sample_many(max_samples=int(self.approx_model_batch - samples_added),
                                                   stop_total=.99 * self.approx_model_batch - samples_added,
                                                   min_alive_ratio=0.1)
                    if self.model_sampler.global_total_samples + samples_added >= .99 * self.approx_model_batch:
                        keep_rolling = False
                    rollout_diagnostics = self.model_sampler.finish_all_paths()
                    model_samples_new, buffer_diagnostics_new = self.model_buf.get(as_tensors=True)
                    model_samples = [torch.cat((o, n), dim=0) for o, n in zip(model_samples, model_samples_new)] \
                        if model_samples else model_samples_new
                    new_n_samples = len(model_samples_new[0]) + EPS
                    w_old = samples_added / (new_n_samples + samples_added)
                    w_new = new_n_samples / (new_n_samples + samples_added)
                    metrics = update_dict(metrics, rollout_diagnostics, weight_a=w_old, weight_b=w_new)
                    metrics = update_dict(metrics, buffer_diagnostics_new, weight_a=w_old, weight_b=w_new)
                    if buffer_diagnostics_new['poolm_batch_size'] > 0:
                        model_data_diag = {k + '_m': v for k, v in policy.run_diagnostics(model_samples_new).items()}
                        metrics = update_dict(metrics, model_data_diag, weight_a=w_old, weight_b=w_new)
                    samples_added += new_n_samples
                    metrics.update({'samples_added': samples_added})
                metrics.update({'cached_var': np.mean(self._model.scaler_out.cached_var)})
                metrics.update({'cached_mu': np.mean(self._model.scaler_out.cached_mu)})
                t0 = self._stamp('epoch_rollout_model', t0)

            # ---- real sampling: as many steps as the model's uncertainty (vs its calibration) asks for ----------
            if self._use_model:
                n_real_samples = self.model_sampler.dyn_dkl / self.initial_model_dkl * self.init_real_samples
                n_real_samples = max(n_real_samples, self.min_real_samples)
            else:
                n_real_samples = self.batch_size_policy
            metrics.update({'n_real_samples': n_real_samples})
            start_samples = self.sampler._total_samples
            for i in count():
                self._timestep = self.sampler._total_samples - start_samples
                self._do_sampling(timestep=self.policy_epoch)
                if self.ready_to_train or self._timestep > n_real_samples:
                    self.sampler.finish_all_paths(append_val=True, append_cval=True, reset_path=False)
                    self.new_real_samples += self._timestep
                    break
            t0 = self._stamp('sample', t0)

            if self.new_real_samples > self._m_train_freq and self._use_model:
                metrics.update(self.train_model(**self._model_train_kwargs))
                self.new_real_samples = 0
            t0 = self._stamp('train_model', t0)

            real_samples, buf_diag = self._buffer.get()
            metrics.update({k + '_r': v for k, v in policy.run_diagnostics(real_samples).items()})
            metrics.update(buf_diag)

            # ---- updates on real + imagined samples ------------------------------------------------------------
            if model_samples:
                train_samples = [torch.cat((r, m), dim=0) for r, m in zip(self._to_device(real_samples), model_samples)]
            else:
                train_samples = real_samples
            policy.update_real_c(real_samples)
            policy.update_policy(train_samples)
            policy.update_critic(train_samples, train_vc=bool((train_samples[-3] > 0).any()),
                                 shuffle_on_device=self._shuffle_on_device)
            if self._use_model:
                self.approx_model_batch = self.batch_size_policy - n_real_samples
            self.policy_epoch += 1
            policy.log()
            t0 = self._stamp('train', t0)

            self.sampler.log()
            new_diagnostics = dict(self.logger.dump_tabular(print_out=False))
            new_diagnostics.update(OrderedDict((
                *((f'times/{k}', self.times[k]) for k in sorted(self.times)),
                *((f'model/{k}', metrics[k]) for k in sorted(metrics)),
            )))
            old_ts = running_diag.get('timestep', 0)
            new_ts = self._total_timestep - self.diag_counter - old_ts
            w_old, w_new = old_ts / (new_ts + old_ts), new_ts / (new_ts + old_ts)
            running_diag = update_dict(running_diag, new_diagnostics, weight_a=w_old, weight_b=w_new)
            running_diag.update({'timestep': new_ts + old_ts})
            if new_ts + old_ts > self.eval_every_n_steps:
                running_diag.update({'epoch': self._epoch, 'timesteps_total': self._total_timestep,
                                     'train-steps': self._num_train_steps})
                self.diag_counter = self._total_timestep
                diag, running_diag = running_diag.copy(), {}
                yield diag
            if self._total_timestep >= self.n_env_interacts:
                self.sampler.terminate()
                yield {'done': True, **running_diag}
                break

    def train(self, *args, **kwargs):
        return self._train(*args, **kwargs)

    def get_diagnostics(self, iteration, obs_batch=None, training_paths=None, evaluation_paths=None):
        warnings.warn('diagnostics not implemented yet!')     # as the reference (algorithms/cmbpo.py:520-533)
        return {}

    def save(self, savedir):
        if self._use_model:
            self._model.save(savedir, self._epoch)
