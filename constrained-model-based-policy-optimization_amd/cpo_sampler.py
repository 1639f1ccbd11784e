"""CpoSampler -- host-side mirror of ``samplers/cpo_sampler.py:10-261``: one real-environment step per call, the
policy's device kernels for the action / values, samples into the CPOBuffer.  The environment itself is serial host
code (MuJoCo / safety-gym, out of scope); any object with gym's ``reset() / step(a) -> (obs, r, done,
info)`` (``info['cost']`` optional) and ``action_space.shape`` works.
"""
from collections import OrderedDict, defaultdict

import numpy as np

from .logger import EpochLogger


class CpoSampler:
    def __init__(self, max_path_length, render_mode=None, logger=None):
        self._max_path_length = max_path_length
        self._path_length = 0
        self._path_return = 0
        self._path_cost = 0
        self.cum_cost = 0
        self.logger = logger if logger else EpochLogger()
        self._render_mode = render_mode
        self._current_path = defaultdict(list)
        self._last_path_return = 0
        self._max_path_return = -np.inf
        self._n_episodes = 0
        self._current_observation = None
        self._total_samples = 0
        self._last_action = None
        self.env = self.policy = self.pool = None

    def initialize(self, env, policy, pool):
        self.env, self.policy, self.pool = env, policy, pool

    def set_policy(self, policy):
        self.policy = policy

    def set_logger(self, logger):
        self.logger = logger

    def terminate(self):
        if hasattr(self.env, "close"):
            self.env.close()

    def get_diagnostics(self):
        return OrderedDict({'pool-size': self.pool.size, 'max-path-return': self._max_path_return,
                            'last-path-return': self._last_path_return, 'episodes': self._n_episodes,
                            'total-samples': self._total_samples})

    @property
    def max_path_length(self):
        return self._max_path_length

    def batch_ready(self):
        return self.pool.size >= self.pool.max_size

    # -- one environment step --------------------------------------------------------------------------
    def _begin_episode(self):
        self._current_observation = np.squeeze(self.env.reset())
        self._last_action = np.zeros(shape=self.env.action_space.shape)

    def _account(self, obs, act, nxt, rew, cost, done, info):
        """Path statistics and the per-path record (cpo_sampler.py:151-176)."""
        self.cum_cost += cost
        self._path_cost += cost
        self._path_return += rew
        self._path_length += 1
        self._total_samples += 1
        rec = self._current_path
        rec['observations'].append(obs)
        rec['actions'].append(act)
        rec['rewards'].append([rew])
        rec['cost'].append([cost])
        rec['terminals'].append([done])
        rec['next_observations'].append(nxt)
        rec['infos'].append(info)

    def sample(self, timestep):
        """One step of the real environment with the current policy (cpo_sampler.py:125-190): action, value and cost
        value from the device kernels, transition into the pool, path bookkeeping, path end handling."""
        if self._current_observation is None:
            self._begin_episode()
        obs = self._current_observation
        pol = self.policy.get_action_outs(obs)
        act = pol['pi']
        nxt, rew, done, info = self.env.step(act)
        if self._render_mode:
            self.env.render(self._render_mode)
        nxt, rew, done = np.squeeze(nxt), np.squeeze(rew), np.squeeze(done)
        cost = info.get('cost', 0)
        self.pool.store(obs, act, nxt, rew, pol['v'], cost, pol['vc'], pol['logp_pi'], pol['pi_info'], done, timestep)
        self.logger.store(VVals=pol['v'], CostVVals=pol['vc'])
        self._account(obs, act, nxt, rew, cost, done, info)
        self._current_observation, self._last_action = nxt, act
        timed_out = self._path_length >= self._max_path_length
        if done or timed_out:
            # a time-out is not a terminal state: the value target is bootstrapped then; the cost value always is
            self.finish_all_paths(append_val=bool(timed_out or not done), append_cval=True)
        return nxt, rew, done, info

    def finish_all_paths(self, append_val=False, append_cval=False, reset_path=True):
        """Close the pool's open path with the right bootstrap values, optionally end the episode
        (cpo_sampler.py:192-236)."""
        obs = self._current_observation
        if obs is None:
            return
        zero = np.zeros((1,))
        self.pool.finish_path(self.policy.get_v(obs) if append_val else zero,
                              self.policy.get_vc(obs) if append_cval else zero)
        if not reset_path:
            return
        n = self._path_length
        self.logger.store(RetEp=self._path_return, EpLen=n, CostEp=self._path_cost,
                          CostFullEp=self._path_cost / n * self._max_path_length)
        self.last_path = {k: np.array(v) for k, v in self._current_path.items() if k != 'infos'}
        self._last_path_return = self._path_return
        self._max_path_return = max(self._max_path_return, self._path_return)
        self._n_episodes += 1
        self.policy.reset()
        self._current_observation = None
        self._last_action = np.zeros(shape=self.env.action_space.shape)
        self._current_path = defaultdict(list)
        self._path_length = self._path_return = self._path_cost = 0

    def log(self):
        """cpo_sampler.py:238-261 (single process: mpi_sum is the identity)"""
        logger = self.logger
        cost_rate = self.cum_cost / max(self._total_samples, 1)
        logger.log_tabular('RetEp', with_min_and_max=True)
        logger.log_tabular('CostEp', with_min_and_max=True)
        logger.log_tabular('CostFullEp', average_only=True)
        logger.log_tabular('EpLen', average_only=True)
        logger.log_tabular('CostCumulative', self.cum_cost)
        logger.log_tabular('CostRate', cost_rate)
        logger.log_tabular('VVals', with_min_and_max=True)
        logger.log_tabular('CostVVals', with_min_and_max=True)
