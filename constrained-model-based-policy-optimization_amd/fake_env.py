"""FakeEnv -- host-side mirror of ``models/fake_env.py:15-197``.

Same constructor, ``step(obs, act, deterministic=True)`` signature and return tuple
``(next_obs, r, terms, info)`` with the info keys of ``models/fake_env.py:164-170``.  The ensemble
forward and everything after it run in HIP (``cmbpo_ens_forward`` + ``cmbpo_fakeenv_post``).

Differences that are part of the contract:
  * the per-branch elite draw (``random_inds``, :174-178, global NumPy RNG) can be injected through
    ``model_inds=`` so a test / a sharded run can reproduce it; by default it is drawn with a
    seeded ``numpy.random.Generator`` owned by this object;
  * ``deterministic=False`` (mean + std, :105-106), the 3-D input path (:84-101, unused by the
    trainer and not an inverse) and learned costs (``predicts_cost``,
    ``m_learn_cost=False`` in every config) raise NotImplementedError.
"""
import numpy as np
import torch

from . import _lib


class FakeEnv:
    def __init__(self, true_environment, task, model, predicts_delta, predicts_rew, predicts_cost,
                 seed=0):
        self.env = true_environment
        self.obs_dim = int(np.prod(self.observation_space.shape))
        self.act_dim = int(np.prod(self.action_space.shape))
        self._task = task
        self._model = model
        self._uses_ensemble = model.is_ensemble
        self._is_probabilistic = model.is_probabilistic
        if not (self._uses_ensemble and self._is_probabilistic):
            raise NotImplementedError("the HIP path needs a probabilistic ensemble (all CMBPO configs)")
        if not (predicts_delta and predicts_rew) or predicts_cost:
            raise NotImplementedError("HIP path: predicts_delta=True, predicts_rew=True, predicts_cost=False "
                                      "(algorithms/cmbpo.py:137-142)")
        self._predicts_delta, self._predicts_rew, self._predicts_cost = True, True, False
        self.input_dim = model.in_dim
        self.output_dim = model.out_dim
        assert self.input_dim == self.obs_dim + self.act_dim and self.output_dim == self.obs_dim + 1
        self._task_id = _lib.TASK_IDS.get(task, _lib.TASK_DEFAULT)
        self._rng = np.random.default_rng(seed)
        self.device = model.device
        # bench.py sets this to a list to collect (start, end) HIP events around the dominant kernel
        self.kernel_events = None

    @property
    def observation_space(self):
        return self.env.observation_space

    @property
    def action_space(self):
        return self.env.action_space

    def random_inds(self, size):
        """One elite per branch (models/fake_env.py:174-178), from this object's seeded generator."""
        elites = np.asarray(self._model.elite_inds, dtype=np.int32)
        return elites[self._rng.integers(0, len(elites), size=size)]

    def step_device(self, obs, act, model_inds, out, row_idx=None, n_rows=None, scratch=None):
        """Device-resident step: all arguments are CUDA tensors indexed by branch slot.

        out: dict with next_obs[B,obs], rew[B], term[B] u8, cost[B], dkl_path[B], ep_var_mean[B]
        (and optionally ep_var[B,obs]).  scratch: (mean, var)[E,B,obs+1].
        """
        B = obs.shape[0]
        n = B if row_idx is None else (row_idx.shape[0] if n_rows is None else n_rows)
        E = self._model.num_nets
        if scratch is None:
            mean = torch.empty((E, B, self.output_dim), dtype=torch.float32, device=self.device)
            var = torch.empty_like(mean)
        else:
            mean, var = scratch
        lib = _lib.lib()
        stream = _lib.current_stream()
        ev = None
        if self.kernel_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()      # same stream the kernel is launched on (torch's current stream)
        _lib.check(lib.cmbpo_ens_forward(self._model.mlp.handle, _lib.ptr(obs), self.obs_dim,
                                         _lib.ptr(act), self.act_dim, _lib.ptr(row_idx), None, n, B,
                                         _lib.ptr(mean), _lib.ptr(var), stream), "cmbpo_ens_forward")
        if ev is not None:
            ev[1].record()
            self.kernel_events.append((ev[0], ev[1], n))
        _lib.check(lib.cmbpo_fakeenv_post(self._task_id, E, self.obs_dim, self.act_dim, _lib.ptr(mean),
                                          _lib.ptr(var), B, _lib.ptr(obs), _lib.ptr(act),
                                          _lib.ptr(model_inds), _lib.ptr(row_idx), None, n,
                                          _lib.ptr(out["next_obs"]), _lib.ptr(out["rew"]),
                                          _lib.ptr(out["term"]), _lib.ptr(out["cost"]),
                                          _lib.ptr(out["dkl_path"]), _lib.ptr(out["ep_var_mean"]),
                                          _lib.ptr(out.get("ep_var")), stream), "cmbpo_fakeenv_post")
        return out

    def step(self, obs, act, deterministic=True, model_inds=None):
        assert len(obs.shape) == len(act.shape)
        assert obs.shape[-1] == self.obs_dim and act.shape[-1] == self.act_dim
        if not deterministic:
            raise NotImplementedError("deterministic=False (mean + std) is never used by the trainer")
        if len(obs.shape) == 3:
            raise NotImplementedError("3-D inputs (forward_shuffle) are never used by the trainer")
        single = len(obs.shape) == 1
        if single:
            obs, act = obs[None], act[None]
        was_np = not isinstance(obs, torch.Tensor)
        with torch.cuda.device(self.device):
            o = torch.as_tensor(np.ascontiguousarray(obs, dtype=np.float32) if was_np else obs,
                                dtype=torch.float32, device=self.device).contiguous()
            a = torch.as_tensor(np.ascontiguousarray(act, dtype=np.float32) if was_np else act,
                                dtype=torch.float32, device=self.device).contiguous()
            n = o.shape[0]
            if model_inds is None:
                model_inds = self.random_inds(n)
            inds = torch.as_tensor(np.asarray(model_inds, dtype=np.int32), device=self.device) \
                if not isinstance(model_inds, torch.Tensor) else model_inds.to(self.device, torch.int32)
            f = dict(dtype=torch.float32, device=self.device)
            out = dict(next_obs=torch.empty((n, self.obs_dim), **f), rew=torch.empty(n, **f),
                       term=torch.empty(n, dtype=torch.uint8, device=self.device),
                       cost=torch.empty(n, **f), dkl_path=torch.empty(n, **f),
                       ep_var_mean=torch.empty(n, **f), ep_var=torch.empty((n, self.obs_dim), **f))
            self.step_device(o, a, inds, out)
        next_obs, r, terms = out["next_obs"], out["rew"][:, None], out["term"].bool()[:, None]
        c = out["cost"][:, None]
        if self._task_id == _lib.TASK_DEFAULT:
            c = c.bool()  # np.zeros_like(terms) is a bool array (models/fake_env.py:145-146)
        dkl_path, ep_var = out["dkl_path"], out["ep_var"]
        dkl_mean = dkl_path.mean()
        if was_np:
            next_obs, r, terms, c = (t.cpu().numpy() for t in (next_obs, r, terms, c))
            dkl_path, ep_var = dkl_path.cpu().numpy(), ep_var.cpu().numpy()
            dkl_mean = float(np.mean(dkl_path))
        if single:
            next_obs, r, c, terms = next_obs[0], r[0], c[0], terms[0]
        info = {"ensemble_dkl_mean": dkl_mean, "ensemble_dkl_path": dkl_path,
                "ensemble_ep_var": ep_var, "rew": r, "cost": c}
        return next_obs, r, terms, info

    def close(self):
        pass
