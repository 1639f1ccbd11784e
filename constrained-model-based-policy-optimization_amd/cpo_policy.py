"""CPOPolicy -- host-side mirror of ``policies/cpo_policy.py:317-931`` for the hot path.

Acting side (this file, rollout path):
  ``get_action_outs(obs) -> {pi, logp_pi, pi_info{mu, log_std}, v, vc}`` (:801-823), ``get_v`` /
  ``get_vc`` (:825-835), attributes ``pi_info_shapes``, ``gamma, lam, cost_gamma, cost_lam``
  (:358-362; note the policy reads 'discount', not 'gamma'), ``agent.reward_penalized``.
The actor is the tanh-MLP Gaussian policy of ``network/ac_network.py:99-123``; V and VC are
3-member 'MSE' PE ensembles with in/out scalers (``policies/cpo_policy.py:452-468``).

The update side (``update_policy`` / ``CPOAgent.update_pi``) lives in ``cpo_update.py``.
All arithmetic is HIP behind the C-ABI; NumPy in -> NumPy out, CUDA tensors in -> CUDA tensors out.
"""
import numpy as np
import torch

from . import _lib
from .cpo_update import CPOAgent, PolicyOps, _NullLogger
from .pens import PE, EnsembleMLP, _to_dev


class GaussianActor:
    """Parameters [W0,b0,W1,b1,W2,b2,log_std] (creation order of get_vars('pi'), ac_network.py:35-36)."""

    def __init__(self, obs_dim, act_dim, hidden_sizes=(128, 128), device=None):
        hidden_sizes = tuple(int(h) for h in hidden_sizes)
        if len(hidden_sizes) != 2 or hidden_sizes[0] != hidden_sizes[1]:
            raise ValueError("HIP actor supports two equal hidden layers; got %r" % (hidden_sizes,))
        self.obs_dim, self.act_dim, self.hidden = int(obs_dim), int(act_dim), hidden_sizes[0]
        self.mlp = EnsembleMLP(1, self.obs_dim, self.hidden, self.act_dim, "tanh", _lib.HEAD_GAUSS_PI, device)
        self.device = self.mlp.device
        self.shapes = [(obs_dim, self.hidden), (self.hidden,), (self.hidden, self.hidden), (self.hidden,),
                       (self.hidden, act_dim), (act_dim,), (act_dim,)]
        self.sizes = [int(np.prod(s)) for s in self.shapes]
        self.n_params = int(sum(self.sizes))
        self.params = None
        self.version = 0      # bumped by set_params so dependants can re-sync lazily

    def set_params(self, params, device_flat=None):
        """params: list of 7 arrays, or one flat vector in the same order (trust_region.py:21-25).  device_flat: the same
        flat vector as a float32 device tensor (the update's accepted parameters): the rollout handle is then packed from
        it on the device, `params` only refreshes the host copy."""
        if not isinstance(params, (list, tuple)):
            flat = np.asarray(params, dtype=np.float32).reshape(-1)
            assert flat.size == self.n_params
            params, off = [], 0
            for shp, sz in zip(self.shapes, self.sizes):
                params.append(flat[off:off + sz].reshape(shp))
                off += sz
        self.params = [np.ascontiguousarray(p, dtype=np.float32).reshape(s) for p, s in zip(params, self.shapes)]
        w0, b0, w1, b1, w2, b2, ls = self.params
        self.version += 1
        if device_flat is not None and getattr(self, "_loaded_once", False):
            assert device_flat.dtype == torch.float32 and device_flat.numel() == self.n_params and device_flat.is_contiguous()
            with torch.cuda.device(self.mlp.device):
                _lib.check(_lib.lib().cmbpo_mlp_load_policy_flat(self.mlp.handle, device_flat.data_ptr(), _lib.current_stream()),
                           "cmbpo_mlp_load_policy_flat")
            return
        self.mlp.load([w0[None], w1[None], w2[None]], [b0[None], b1[None], b2[None]], log_std=ls)
        self._loaded_once = True

    def get_flat_params(self):
        return np.concatenate([p.reshape(-1) for p in self.params]).astype(np.float32)

    def forward_device(self, obs, eps, out, row_idx=None, n_rows=None):
        n = obs.shape[0] if row_idx is None else (row_idx.shape[0] if n_rows is None else n_rows)
        _lib.check(_lib.lib().cmbpo_policy_forward(
            self.mlp.handle, _lib.ptr(obs), self.obs_dim, _lib.ptr(eps), _lib.ptr(row_idx), None, n,
            _lib.ptr(out["pi"]), _lib.ptr(out["logp_pi"]), _lib.ptr(out["mu"]), _lib.ptr(out["log_std"]),
            _lib.current_stream()), "cmbpo_policy_forward")
        return out


class CPOPolicy:
    def __init__(self, obs_space, act_space, session=None, logger=None, device=None, seed=0, **kwargs):
        self.obs_space, self.act_space = obs_space, act_space
        self.obs_dim = int(np.prod(obs_space.shape))
        self.act_dim = int(np.prod(act_space.shape))
        kw = kwargs
        self.hidden_sizes_a = kw.get("a_hidden_layer_sizes", (128, 128))
        self.hidden_sizes_c = kw.get("vf_hidden_layer_sizes", (128, 128))
        # critic training arguments, cpo_policy.py:332-350
        self.vf_lr = kw.get("vf_lr", 1e-4)
        self.vf_epochs = kw.get("vf_epochs", 10)
        self.vf_batch_size = kw.get("vf_batch_size", 64)
        self.vf_holdout = 0.1
        self.vf_train_kwargs = dict(batch_size=self.vf_batch_size, min_epoch_before_break=self.vf_epochs,
                                    max_epochs=self.vf_epochs, holdout_ratio=self.vf_holdout)
        self.vf_decay = kw.get("vf_decay", 1e-6)
        if kw.get("vf_clipping", False):
            raise NotImplementedError("vf_clipping is off in every shipped config")
        self.vf_ensemble = kw.get("vf_ensemble_size", 5)
        self.vf_elites = kw.get("vf_elites", 3)
        self.vf_activation = kw.get("vf_activation", "ReLU")
        self.vf_loss = kw.get("vf_loss", "MSE")
        if self.vf_activation != "swish" or self.vf_loss != "MSE":
            raise NotImplementedError("HIP critics: vf_activation='swish', vf_loss='MSE' (every shipped config)")
        self.ent_reg = kw.get("ent_reg", 0.0)
        self.cost_lim = kw.get("cost_lim", 25)
        self.constrain_cost = kw.get("constrain_cost", True)
        self.target_kl = kw.get("target_kl", 0.01)
        self.cost_lam = kw.get("cost_lam", 0.97)
        self.cost_gamma = kw.get("cost_gamma", 0.99)
        self.lam = kw.get("lam", 0.97)
        self.gamma = kw.get("discount", 0.99)          # cpo_policy.py:362 reads 'discount'
        self.max_path_length = kw.get("max_path_length", 1)
        self.real_c_buffer = [self.cost_lim] * 300     # cpo_policy.py:356
        self.logger = logger if logger is not None else _NullLogger()
        self.comm = kw.get("comm", None)
        self.actor = GaussianActor(self.obs_dim, self.act_dim, self.hidden_sizes_a, device)
        self.device = self.actor.device
        # cpo_policy.py:366-374: the agent's hyper-parameters
        self.agent = CPOAgent(constrained=self.constrain_cost, learn_margin=True, c_gamma=self.cost_gamma,
                              max_path_length=self.max_path_length, ent_reg=self.ent_reg)
        self.agent.set_logger(self.logger)
        self.ops = PolicyOps(self.obs_dim, self.act_dim, self.actor.hidden, self.device, comm=self.comm)
        common = dict(hidden_dims=self.hidden_sizes_c, num_networks=self.vf_ensemble,
                      num_elites=self.vf_elites, loss="MSE", activation="swish",
                      use_scaler_in=True, use_scaler_out=True, device=self.device, lr=self.vf_lr,
                      decay=self.vf_decay)
        self.v = PE(self.obs_dim, 1, name="VEnsemble", **common)
        self.vc = PE(self.obs_dim, 1, name="VCEnsemble", **common)
        # pi_info placeholders' shapes (network/ac_network.py:113,120; algorithms/cmbpo.py:94)
        self.pi_info_shapes = {"mu": [self.act_dim], "log_std": [self.act_dim]}
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(int(seed))

    def reset(self):
        pass

    def set_logger(self, logger):
        """cpo_policy.py:588-598: the logger is shared with the agent."""
        self.logger = logger
        self.agent.set_logger(logger)

    def log(self):
        """cpo_policy.py:896-931: the agent's optimisation measures and the policy's own stored keys -> logger row."""
        self.agent.log()
        for k in ('LossPi', 'SurrCost', 'SurrAdv', 'Entropy', 'KL', 'LossPiDelta', 'SurrCostDelta', 'SurrAdvDelta',
                  'Loss' + self.v.name, 'Loss' + self.vc.name, 'Loss' + self.v.name + 'Delta',
                  'Loss' + self.vc.name + 'Delta'):
            self.logger.log_tabular(k, average_only=True)
        self.logger.log_tabular('Penalty', 0)
        self.logger.log_tabular('PenaltyDelta', 0)

    def set_params(self, params):
        """Actor parameters ([W0,b0,W1,b1,W2,b2,log_std] or one flat vector) -> rollout + update handles."""
        self.actor.set_params(params)
        self._sync_ops()

    def _sync_ops(self):
        if getattr(self, "_ops_version", None) != self.actor.version:
            self.ops.set_params(self.actor.get_flat_params())
            self._ops_version = self.actor.version

    # -- update (policies/cpo_policy.py:600-656, 733-754, 837-845) ------------------------------------
    def _bind(self, buf_inputs):
        # buf_fields order (cpo_policy.py:472-477): obs, act, adv, cadv, ret, cret, logp_old, v, vc, cost, log_std, mu
        obs, act, adv, cadv, _ret, _cret, logp_old, _v, _vc, cost, log_std, mu = buf_inputs
        self.ops.bind(obs, act, adv, cadv, logp_old, cost, mu, log_std)

    def update_real_c(self, buf_inputs):
        """cpo_policy.py:733-737: running window of real epoch costs that drives the margin."""
        cost = buf_inputs[9]
        m = float(cost.mean()) if isinstance(cost, torch.Tensor) else float(np.mean(cost))
        self.real_c_buffer.append(m * self.max_path_length)
        self.real_c_buffer.pop(0)

    def update_policy(self, buf_inputs):
        """cpo_policy.py:600-656: pre-measures, CPOAgent.update_pi, post-measures and deltas."""
        self._sync_ops()
        self._bind(buf_inputs)
        info = self.agent.update_pi(self.ops, self.target_kl, self.cost_lim, self.real_c_buffer)
        # (the batch's mean cost comes out of the update's own loss kernel -- the same sum, over every rank's samples --
        # instead of a reduction and a blocking read of their own in front of it)
        if float(info["cur_cret_avg"]) - self.cost_lim > 0 and self.agent.cares_about_cost:
            self.logger.log('Warning! Safety constraint is already violated.', 'red')
        # pre / post measures come out of the update's own kernels (the reference spends two extra
        # sess.run calls on them, cpo_policy.py:613-618,647-656)
        pre, post = info["pre"], info["post"]
        self.logger.store(LossPi=pre["LossPi"], SurrCost=pre["SurrCost"], SurrAdv=pre["SurrAdv"],
                          Entropy=pre["Entropy"])
        # the accepted (or restored) parameters now live in ops; mirror them into the rollout actor
        self.actor.set_params(info["params"] if info.get("params") is not None else self.ops.get_params(),
                              device_flat=self.ops.params)
        self._ops_version = self.actor.version
        deltas = {k + "Delta": post[k] - pre[k] for k in ("LossPi", "SurrCost", "SurrAdv")}
        self.logger.store(KL=post["KL"], **deltas)
        return info

    def run_diagnostics(self, buf_inputs):
        """cpo_policy.py:739-754: actor measures and both critic validation losses."""
        self._sync_ops()
        self._bind(buf_inputs)
        m = self.agent.measures(self.ops)
        out = dict(LossPi=m["LossPi"], SurrCost=m["SurrCost"], Entropy=m["Entropy"])
        if self.v.finalized and self.vc.finalized:
            out.update(self.compute_v_losses(buf_inputs))
        return out

    def compute_DKL(self, obs_batch, mu_batch, logstd_batch):
        """cpo_policy.py:837-845: mean KL(current || stored) per archived epoch ([n_epochs, B, .] or 2-D)."""
        self._sync_ops()

        def one(o, m, l):
            n = o.shape[0]
            z = np.zeros(n, np.float32)
            self.ops.bind(o, np.zeros((n, self.act_dim), np.float32), z, z, z, z, m, l)
            s = self.ops.evals()
            return s[3] / s[0]
        if len(obs_batch.shape) == 3:
            return np.array([one(o, m, l) for o, m, l in zip(obs_batch, mu_batch, logstd_batch)])
        return one(obs_batch, mu_batch, logstd_batch)

    def update_critic(self, buf_inputs, train_vc=True, **kwargs):
        """cpo_policy.py:658-698: loss measures, train the return critic (and the cost critic) on the buffer, loss
        measures again, deltas to the logger.  ``kwargs`` override vf_train_kwargs (and may carry ``rng`` /
        ``shuffle_on_device`` for PE.train)."""
        obs, ret, cret = buf_inputs[0], buf_inputs[4], buf_inputs[5]
        pre = self.compute_v_losses(buf_inputs)
        self.logger.store(**pre)
        train_kwargs = self.vf_train_kwargs.copy()
        train_kwargs.update(kwargs)
        self.train_vf(obs, ret, **train_kwargs)
        if train_vc:
            self.train_vc(obs, cret, **train_kwargs)
        if self.comm is not None and self.comm.world > 1:    # replicas: rank 0's critics are everyone's
            self.v.sync_weights(self.comm)
            self.vc.sync_weights(self.comm)
        post = self.compute_v_losses(buf_inputs)
        deltas = {k + "Delta": post[k] - pre[k] for k in post if k in pre}
        self.logger.store(**deltas)
        return dict(pre=pre, post=post)

    def train_vf(self, obs, ret, **kwargs):
        """cpo_policy.py:700-715"""
        return self.v.train(obs, ret[:, None], **kwargs)

    def train_vc(self, obs, cret, **kwargs):
        """cpo_policy.py:717-731"""
        return self.vc.train(obs, cret[:, None], **kwargs)

    def compute_v_losses(self, buf_inputs, rng=None):
        """cpo_policy.py:756-774: validation loss of both critics on 5000 rows drawn with replacement."""
        obs, ret, cret = buf_inputs[0], buf_inputs[4], buf_inputs[5]
        rand_inds = (np.random if rng is None else rng).randint(0, obs.shape[0], 5000)
        if isinstance(obs, torch.Tensor):
            rand_inds = torch.from_numpy(rand_inds).to(obs.device)
        v_loss = self.v.validate(obs[rand_inds], ret[rand_inds][:, None])
        vc_loss = self.vc.validate(obs[rand_inds], cret[rand_inds][:, None])
        return {"Loss" + self.v.name: v_loss, "Loss" + self.vc.name: vc_loss}

    # -- checkpoints ---------------------------------------------------------------------
    def save(self, checkpoint_dir, timestep=0):
        """Actor parameters as ``policy_<timestep>.npz`` (keys W0 b0 W1 b1 W2 b2 log_std: the variables of the
        reference's ``pi`` scope in creation order, network/ac_network.py:26-36,104) and both critics in the ensemble
        format (``PE.save``).  The reference writes a TF SavedModel here (cpo_policy.py:890-894, utilities/logx.py:202-259),
        which only TensorFlow can read or write; a maintainer dumps ``sess.run(get_vars('pi'))`` into this npz instead."""
        import os
        names = ("W0", "b0", "W1", "b1", "W2", "b2", "log_std")
        path = os.path.join(checkpoint_dir, "policy_%s.npz" % timestep)
        np.savez(path, **dict(zip(names, self.actor.params)))
        if self.v.finalized:
            self.v.save(checkpoint_dir, timestep)
        if self.vc.finalized:
            self.vc.save(checkpoint_dir, timestep)
        return path

    def load(self, checkpoint_dir, timestep=0):
        import os
        z = np.load(os.path.join(checkpoint_dir, "policy_%s.npz" % timestep))
        self.set_params([z[k] for k in ("W0", "b0", "W1", "b1", "W2", "b2", "log_std")])
        for critic in (self.v, self.vc):
            if os.path.exists(os.path.join(checkpoint_dir, "%s_%s.nns" % (critic.name, timestep))):
                critic.load(checkpoint_dir, timestep)

    # -- acting ----------------------------------------------------------------------------
    def format_obs(self, obs):
        if len(obs.shape) == len(self.obs_space.shape):
            obs = obs[None]
        return obs

    def get_action_outs(self, obs, eps=None):
        """policies/cpo_policy.py:801-823.  eps (optional) injects the N(0,1) draw of ac_network.py:109."""
        obs = self.format_obs(obs)
        if len(obs.shape) > 2:
            raise NotImplementedError("bad observation shape")
        with torch.cuda.device(self.device):
            o, was_np = _to_dev(obs, self.device)
            n = o.shape[0]
            if eps is None:
                e = torch.randn((n, self.act_dim), generator=self._gen, dtype=torch.float32, device=self.device)
            else:
                e, _ = _to_dev(eps, self.device)
            # one flat device buffer behind every output: a NumPy caller (the real-environment sampler steps one
            # observation at a time) pays one device-to-host copy instead of six
            A = self.act_dim
            flat = torch.empty(n * (3 * A + 3), dtype=torch.float32, device=self.device)
            cut = lambda k, shape: flat[k[0]:k[0] + int(np.prod(shape))].view(shape)
            offs = np.cumsum([0, n * A, n * A, n * A, n, n, n])
            out = dict(pi=cut(offs[0:1], (n, A)), mu=cut(offs[1:2], (n, A)), log_std=cut(offs[2:3], (n, A)),
                       logp_pi=cut(offs[3:4], (n,)))
            v2, vc2 = cut(offs[4:5], (n, 1)), cut(offs[5:6], (n, 1))
            self.actor.forward_device(o, e, out)
            self.v.predict(o, out=v2)
            self.vc.predict(o, out=vc2)
        if was_np:
            h = flat.cpu().numpy()
            hcut = lambda i, shape: h[offs[i]:offs[i + 1]].reshape(shape)
            return {"pi": hcut(0, (n, A)), "logp_pi": hcut(3, (n,)),
                    "pi_info": {"mu": hcut(1, (n, A)), "log_std": hcut(2, (n, A))}, "v": hcut(4, (n,)), "vc": hcut(5, (n,))}
        return {"pi": out["pi"], "logp_pi": out["logp_pi"], "pi_info": {"mu": out["mu"], "log_std": out["log_std"]},
                "v": v2[:, 0], "vc": vc2[:, 0]}

    def get_v(self, obs):
        o = self.format_obs(obs)
        r = self.v.predict(o)
        return r[:, 0]

    def get_vc(self, obs):
        o = self.format_obs(obs)
        r = self.vc.predict(o)
        return r[:, 0]

    def actions(self, obs):
        return self.get_action_outs(obs)["pi"]

    def actions_np(self, obs):
        return np.array(self.actions(obs))
