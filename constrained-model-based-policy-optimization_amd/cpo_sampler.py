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

    def sample(self, timestep):
        """cpo_sampler.py:125-190"""
        if self._current_observation is None:
            self._current_observation = np.squeeze(self.env.reset())
            self._last_action = np.zeros(shape=self.env.action_space.shape)
        outs = self.policy.get_action_outs(self._current_observation)
        a, v_t, vc_t = outs['pi'], outs['v'], outs['vc']
        logp_t, pi_info_t = outs['logp_pi'], outs['pi_info']

        next_observation, reward, terminal, info = self.env.step(a)
        if self._render_mode:
            self.env.render(self._render_mode)
        next_observation = np.squeeze(next_observation)
        reward = np.squeeze(reward)
        terminal = np.squeeze(terminal)
        c = info.get('cost', 0)

        self.pool.store(self._current_observation, a, next_observation, reward, v_t, c, vc_t, logp_t, pi_info_t,
                        terminal, timestep)
        self.logger.store(VVals=v_t, CostVVals=vc_t)
        self.cum_cost += c
        self._path_length += 1
        self._path_return += reward
        self._path_cost += c
        self._total_samples += 1
        for key, value in (('observations', self._current_observation), ('actions', a), ('rewards', [reward]),
                           ('cost', [c]), ('terminals', [terminal]), ('next_observations', next_observation),
                           ('infos', info)):
            self._current_path[key].append(value)
        self._current_observation = next_observation
        self._last_action = a

        if terminal or self._path_length >= self._max_path_length:
            # an env time-out is not a true terminal state: the value target is bootstrapped; costs always are
            if terminal and not (self._path_length >= self._max_path_length):
                self.finish_all_paths(append_val=False, append_cval=True)
            else:
                self.finish_all_paths(append_val=True, append_cval=True)
        return next_observation, reward, terminal, info

    def finish_all_paths(self, append_val=False, append_cval=False, reset_path=True):
        """cpo_sampler.py:192-236"""
        if self._current_observation is None:
            return
        last_val = self.policy.get_v(self._current_observation) if append_val else np.zeros((1,))
        last_cval = self.policy.get_vc(self._current_observation) if append_cval else np.zeros((1,))
        self.pool.finish_path(last_val, last_cval)
        if reset_path:
            self.logger.store(RetEp=self._path_return, EpLen=self._path_length, CostEp=self._path_cost,
                              CostFullEp=self._path_cost / self._path_length * self._max_path_length)
            self.last_path = {k: np.array(v) for k, v in self._current_path.items() if k != 'infos'}
            self._max_path_return = max(self._max_path_return, self._path_return)
            self._last_path_return = self._path_return
            self.policy.reset()
            self._current_observation = None
            self._last_action = np.zeros(shape=self.env.action_space.shape)
            self._path_length = 0
            self._path_return = 0
            self._path_cost = 0
            self._current_path = defaultdict(list)
            self._n_episodes += 1

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
