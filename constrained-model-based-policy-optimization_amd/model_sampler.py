"""ModelSampler -- host-side mirror of ``samplers/model_sampler.py:13-444`` over device-resident state.

Same constructor / ``initialize(env, policy, pool)`` / ``reset(observations)`` / ``sample(max_samples)``
/ ``finish_all_paths()`` / ``compute_dynamics_dkl`` / ``set_rollout_dkl`` / ``set_max_path_length`` and the
``dyn_dkl`` / ``_total_samples`` attributes the trainer reads (``algorithms/cmbpo.py:197-201,251-266``).

One ``sample()`` = one imagined step of every alive branch, entirely on the GPU:

    policy forward (HIP)  ->  ensemble forward + FakeEnv post (HIP)  ->  decide  ->  finish(PRE)
    ->  store  ->  critics on next_obs (HIP)  ->  finish(POST)  ->  compact

Event order, bootstraps and the budget rule follow ``model_sampler.py:239-375`` (see the kernels in
``csrc/rollout_state.hip``).  Two restructurings that do not change results:
  * V / VC of the *next* observation are evaluated once after the store: they bootstrap the horizon /
    terminal finishes of this step (``:350-367``) and ARE the ``v_t`` / ``vc_t`` of the next step (the critics
    are deterministic), so the extra ``get_v`` / ``get_vc`` calls of ``_finish_paths`` (``:401-407``) vanish;
  * branches never move: an ordered alive list replaces the boolean-mask compaction (``:300-311``).

``sample()`` returns ``(next_obs, reward, terminal, info)`` like the reference, but as slot-indexed CUDA
tensors (the trainer only reads ``info['alive_ratio']``, ``algorithms/cmbpo.py:254-263``).
"""
import ctypes as C
from collections import OrderedDict

import numpy as np
import torch

from . import _lib
from .dist import budget_plan

EPS = 1e-8  # utilities/utils.py:19


class ModelSampler:
    def __init__(self, max_path_length, batch_size=1000, rollout_mode=False, logger=None, seed=0,
                 comm=None):
        self._max_path_length = int(max_path_length)
        self.batch_size = int(batch_size)
        self.rollout_mode = rollout_mode
        self.logger = logger
        self.comm = comm
        self.dkl_lim = float("inf")
        self.env = self.policy = self.pool = None
        self._n_episodes = 0
        self._seed = int(seed)
        self._gen = None
        self._host = dict(total_samples=0.0, total_dkl=0.0)
        self._calib_total_dkl = 0.0
        self._calib_total_samples = 0.0
        self._diag = None

    # -- wiring -----------------------------------------------------------------------------------
    def initialize(self, env, policy, pool):
        self.env, self.policy, self.pool = env, policy, pool
        self.device = pool.device
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(self._seed)
        self._elites = None
        self._handles = None

    def set_policy(self, policy):
        self.policy = policy
        self._handles = None

    def set_logger(self, logger):
        self.logger = logger

    def terminate(self):
        self.env.close()

    def set_rollout_dkl(self, dkl):
        self.dkl_lim = float(dkl)

    def set_max_path_length(self, path_length):
        self._max_path_length = int(path_length)

    def batch_ready(self):
        return self.pool.size >= self.pool.max_size

    # -- accumulators ---------------------------------------------------------------------------------
    def _read_scalars(self):
        isc, dsc = self.pool.read_scalars()
        self._dsc = dsc
        return isc, dsc

    @property
    def _total_samples(self):
        return self._host["total_samples"]

    @property
    def global_total_samples(self):
        """`_total_samples` summed over the shards (== `_total_samples` on one GPU): what a stop rule on the job's sample
        count has to read (algorithms/cmbpo.py:356-357)."""
        if self.comm is not None and self.comm.world > 1:
            return getattr(self, "_global_total_samples", 0.0)
        return self._host["total_samples"]

    @property
    def _total_dkl(self):
        return self._host["total_dkl"]

    @property
    def dyn_dkl(self):
        """model_sampler.py:175-177."""
        return self._host["total_dkl"] / (self._host["total_samples"] + EPS)

    def get_diagnostics(self):
        """model_sampler.py:89-133 (same keys; variance terms the reference never fills stay 0)."""
        _, d = self._read_scalars()
        L = _lib
        tot = d[L.D_TOTAL_SAMPLES]
        diagnostics = OrderedDict({"pool-size": self.pool.size})
        diagnostics.update({
            "msampler/samples_added": tot,
            "msampler/rollout_H_max": self._n_episodes,
            "msampler/rollout_H_mean": tot / (self.batch_size + EPS),
            "msampler/rew_var_perstep": 0.0,
            "msampler/cost_var_perstep": 0.0,
            "msampler/dyn_var_perstep": d[L.D_TOTAL_DYN_EP_VAR] / (tot + EPS),
            "msampler/cost_rate": d[L.D_SUM_PATH_COST] / (tot + EPS),
            "msampler/rew_rate": d[L.D_SUM_PATH_RET] / (tot + EPS),
            "msampler/v_mean": d[L.D_TOTAL_VS] / (tot + EPS),
            "msampler/cv_mean": d[L.D_TOTAL_CVS] / (tot + EPS),
            "msampler/ens_DKL": d[L.D_TOTAL_DKL] / (tot + EPS),
            "msampler/ens_mean_var": 0.0,
            "msampler/max_path_return": d[L.D_MAX_PATH_RETURN],
            "msampler/max_dkl": d[L.D_MAX_DKL],
        })
        return diagnostics

    # -- rollout -------------------------------------------------------------------------------------
    def _critics(self, obs_key, v_key, vc_key, n):
        t = self.pool.t
        idx = t["alive_idx"]
        lib, stream = _lib.lib(), _lib.current_stream()
        v, vc = self.policy.v.mlp.handle, self.policy.vc.mlp.handle
        if lib.cmbpo_critic_pair_supported(v, vc):       # both critics in one launch
            _lib.check(lib.cmbpo_critic_pair_predict(v, vc, t[obs_key].data_ptr(), self.pool.obs_dim, idx.data_ptr(), None, n,
                                                     t[v_key].data_ptr(), t[vc_key].data_ptr(), stream),
                       "cmbpo_critic_pair_predict")
            return
        for net, key in ((self.policy.v, v_key), (self.policy.vc, vc_key)):
            _lib.check(lib.cmbpo_ens_predict_mean(net.mlp.handle, t[obs_key].data_ptr(), self.pool.obs_dim,
                                                  idx.data_ptr(), None, n, t[key].data_ptr(), stream),
                       "cmbpo_ens_predict_mean")

    def reset(self, observations):
        """model_sampler.py:203-237: start all branches from `observations` [B, obs]."""
        self.batch_size = int(observations.shape[0])
        self.policy.reset()
        pool = self.pool
        with torch.cuda.device(self.device):
            pool.reset(self.batch_size)
            obs = observations if isinstance(observations, torch.Tensor) else \
                torch.from_numpy(np.ascontiguousarray(observations, dtype=np.float32))
            pool.t["cur_obs"].copy_(obs.to(self.device, torch.float32))
            pool.rs.max_path_length = self._max_path_length
            pool.rs.uncertainty_mode = 1 if self.rollout_mode == "uncertainty" else 0
            pool.rs.dkl_lim = float(self.dkl_lim)
            if self.comm is not None:
                pool.rs.rank, pool.rs.world = self.comm.rank, self.comm.world
            # critics at the start states: v_t / vc_t of the first step
            self._critics("cur_obs", "v_t", "vc_t", self.batch_size)
        self._n_episodes = 0
        self.global_alive = 1
        self._global_total_samples = 0.0
        self.n_budget_terminated = 0     # branches the `max_samples` rule ended early (model_sampler.py:282-287), this shard
        self._host = dict(total_samples=0.0, total_dkl=0.0)
        elites = np.asarray(self.env._model.elite_inds, dtype=np.int32)
        if getattr(self, "_elites_host", None) is None or not np.array_equal(self._elites_host, elites):
            # (uploaded again only when the elites changed: a host-to-device copy per reset otherwise)
            self._elites_host = elites.copy()
            self._elites = torch.as_tensor(elites, device=self.device)
        self._draws = None
        self._handles = None

    def _scatter(self, compact, idx, width=None, dtype=torch.float32):
        """Test hook: a draw given in the reference's compact (alive-only) order -> slot order."""
        B = self.batch_size
        shape = (B,) if width is None else (B, width)
        full = torch.zeros(shape, dtype=dtype, device=self.device)
        full[idx.long()] = torch.as_tensor(np.ascontiguousarray(compact), device=self.device).to(dtype)
        return full

    def sample(self, max_samples=None, eps=None, model_inds=None):
        """One imagined step of every alive branch (model_sampler.py:239-375).

        eps / model_inds (optional, compact alive-only order like the reference's arrays) inject the
        N(0,1) action noise (ac_network.py:109) and the per-branch elite draw (fake_env.py:174-178).
        """
        pool, env, pol = self.pool, self.env, self.policy
        assert pool.has_room                       # pool full! empty before sampling.
        sharded = self.comm is not None and self.comm.world > 1
        if sharded and pool.n_alive == 0:
            # this shard has nothing left but the others may: keep the collectives of the step matched
            return self._idle_step(max_samples)
        assert pool.n_alive > 0                    # reset before sampling !
        if not sharded and eps is None and model_inds is None and env.kernel_events is None:
            if torch.cuda.current_device() != self.device.index:
                with torch.cuda.device(self.device):
                    return self._sample_fast(max_samples)
            return self._sample_fast(max_samples)
        self._n_episodes += 1
        t, rs = pool.t, pool.rs
        n, B, A = pool.n_alive, self.batch_size, pool.act_dim
        with torch.cuda.device(self.device):
            idx = t["alive_idx"]
            if eps is None or model_inds is None:
                # action noise and elite picks are drawn for several steps at a time (three torch launches per chunk
                # instead of per step: at small rollout batches a step is a chain of launch latencies)
                ck = getattr(self, "_draws", None)
                if ck is None or ck[0] >= ck[1].shape[0] or ck[1].shape[1] != B:
                    ck = self._draw_chunk()
                k = ck[0]
                ck[0] = k + 1
            if eps is None:
                eps_t = ck[1][k]
            else:
                eps_t = self._scatter(eps, idx[:n], A)
            if model_inds is None:
                inds_t = ck[2][k]
            else:
                inds_t = self._scatter(model_inds, idx[:n], None, torch.int32)
            if getattr(self, "_scratch", None) is None or self._scratch[0].shape[1] != B:
                E, O = env._model.num_nets, env.output_dim
                self._scratch = (torch.empty((E, B, O), dtype=torch.float32, device=self.device),
                                 torch.empty((E, B, O), dtype=torch.float32, device=self.device))
            rs.max_samples = int(max_samples) if max_samples else 0
            rs.dkl_lim = float(self.dkl_lim)
            rs.max_path_length = self._max_path_length
            rs.use_host_budget = 0
            exchange = sharded and rs.max_samples != 0      # the reference's `if max_samples:` (negative budgets count)
            compacted = False
            if not exchange and env.kernel_events is None:
                # the whole step in one call: at small rollout batches a step is bound by host latency
                rc = _lib.lib().cmbpo_rollout_step(
                    C.byref(rs), n, pol.actor.mlp.handle, env._model.mlp.handle, pol.v.mlp.handle, pol.vc.mlp.handle,
                    env._task_id, env._model.num_nets, eps_t.data_ptr(), inds_t.data_ptr(),
                    self._scratch[0].data_ptr(), self._scratch[1].data_ptr(), _lib.current_stream())
                if rc == 1:         # small batch: finish(POST) and the compaction ran inside the call
                    compacted = True
                else:
                    _lib.check(rc, "cmbpo_rollout_step")
            else:
                # policy: pi, logp, mu, log_std at the current observations
                pol.actor.forward_device(t["cur_obs"], eps_t,
                                         dict(pi=t["act_t"], logp_pi=t["logp_t"], mu=t["mu_t"], log_std=t["ls_t"]),
                                         row_idx=idx, n_rows=n)
                # dynamics ensemble + FakeEnv post-processing
                env.step_device(t["cur_obs"], t["act_t"], inds_t,
                                dict(next_obs=t["next_obs"], rew=t["rew_t"], term=t["term_t"], cost=t["cost_t"],
                                     dkl_path=t["dkl_t"], ep_var_mean=t["epv_t"]),
                                row_idx=idx, n_rows=n, scratch=self._scratch)
                if exchange:
                    # budget rule across shards: gather {n_alive, n_unc, total}, rank the survivors globally
                    pool._call("cmbpo_rollout_count")
                    rows = self.comm.all_gather_i32(t["iscal"][8:12]).cpu().numpy()
                    excess, rank_off = budget_plan(rows, self.comm.rank, rs.max_samples)
                    rs.use_host_budget, rs.host_excess, rs.host_rank_off = 1, int(excess), int(rank_off)
                pool._call("cmbpo_rollout_decide")
                pool._call("cmbpo_rollout_finish", 0)
                pool._call("cmbpo_rollout_store")
                self._critics("next_obs", "v_n", "vc_n", n)
                pool._call("cmbpo_rollout_finish", 1)
            # the host sync of the step: counters of what finished / was stored + the accumulators
            isc, dsc = self._read_scalars()
            self.n_budget_terminated += int(isc[_lib.I_N_FIN_PRE]) - int(isc[_lib.I_N_UNC])
            if compacted:
                pool.swap("alive_idx", "alive_idx_out")
            elif int(isc[_lib.I_N_FIN_PRE]) + int(isc[_lib.I_N_FIN_POST]) > 0:
                pool._call("cmbpo_rollout_compact")      # the alive list only changes when a branch finished
                pool.swap("alive_idx", "alive_idx_out")
                # the survivors' count follows from the counters just read: no second host synchronisation
                pool._n_alive = n - int(isc[_lib.I_N_FIN_PRE]) - int(isc[_lib.I_N_FIN_POST])
            pool.swap("cur_obs", "next_obs")
            pool.swap("v_t", "v_n")
            pool.swap("vc_t", "vc_n")
            pool.ptr += 1
            rs.ptr = pool.ptr
        self._host["total_samples"] = float(dsc[_lib.D_TOTAL_SAMPLES])
        self._host["total_dkl"] = float(dsc[_lib.D_TOTAL_DKL])
        alive = pool.n_alive
        if self.comm is not None and self.comm.world > 1:
            g = self.comm.all_reduce_host([alive, self.batch_size, self._host["total_samples"]])
            alive_ratio, self._global_total_samples = g[0] / g[1], g[2]
            self.global_alive = int(g[0])
        else:
            alive_ratio = alive / self.batch_size
        info = {"alive_ratio": alive_ratio, "ensemble_dkl_path": t["dkl_t"], "cost": t["cost_t"]}
        # after the swap the step's next_obs is the new cur_obs
        return t["cur_obs"], t["rew_t"], t["term_t"], info

    def _draw_chunk(self):
        """Action noise and elite picks for several steps at a time (three torch launches per chunk instead of per step:
        at small rollout batches a step is a chain of launch latencies).  Returns the chunk list
        [next step, eps [K, B, A], elite indices [K, B]]."""
        B, A = self.batch_size, self.pool.act_dim
        step_bytes = max(B * A * 4, 1)
        K = max(1, min(40, max(2 ** 23 // step_bytes, min(8, 2 ** 28 // step_bytes))))    # a whole rollout at small batches
        e_ck = torch.randn((K, B, A), generator=self._gen, dtype=torch.float32, device=self.device)
        d_ck = torch.randint(0, len(self._elites), (K, B), generator=self._gen, device=self.device)
        self._draws = [0, e_ck, self._elites[d_ck]]
        return self._draws

    def _sample_fast(self, max_samples):
        """sample() on one rank with the sampler's own draws: the whole step is ONE C call (cmbpo_rollout_step) plus the
        read of the step's counters.  Same results as the general path below -- at the shipped configurations' 1e3 - 1e4
        branches a step is ~100 us of kernels, so the host side is kept to pointer arithmetic."""
        pool, env, pol = self.pool, self.env, self.policy
        self._n_episodes += 1
        t, rs = pool.t, pool.rs
        n, B, A = pool.n_alive, self.batch_size, pool.act_dim
        ck = self._draws
        if ck is None or ck[0] >= ck[1].shape[0] or ck[1].shape[1] != B:
            ck = self._draw_chunk()
        k = ck[0]
        ck[0] = k + 1
        eps_ptr = ck[1].data_ptr() + k * B * A * 4
        inds_ptr = ck[2].data_ptr() + k * B * ck[2].element_size()
        if getattr(self, "_scratch", None) is None or self._scratch[0].shape[1] != B:
            E, O = env._model.num_nets, env.output_dim
            self._scratch = (torch.empty((E, B, O), dtype=torch.float32, device=self.device),
                             torch.empty((E, B, O), dtype=torch.float32, device=self.device))
            self._handles = None
        if getattr(self, "_handles", None) is None:
            self._handles = (pol.actor.mlp.handle, env._model.mlp.handle, pol.v.mlp.handle, pol.vc.mlp.handle,
                             self._scratch[0].data_ptr(), self._scratch[1].data_ptr())
        h = self._handles
        rs.max_samples = int(max_samples) if max_samples else 0
        rs.dkl_lim = float(self.dkl_lim)
        rs.max_path_length = self._max_path_length
        rs.use_host_budget = 0
        stream = _lib.current_stream()
        rc = _lib.lib().cmbpo_rollout_step(C.byref(rs), n, h[0], h[1], h[2], h[3], env._task_id, env._model.num_nets,
                                           eps_ptr, inds_ptr, h[4], h[5], stream)
        if rc != 1:
            _lib.check(rc, "cmbpo_rollout_step")
        isc, dsc = self._read_scalars()         # the host sync of the step
        self.n_budget_terminated += int(isc[_lib.I_N_FIN_PRE]) - int(isc[_lib.I_N_UNC])
        if rc == 1:                             # small batch: finish(POST) and the compaction ran inside the call
            pool.swap("alive_idx", "alive_idx_out")
        elif int(isc[_lib.I_N_FIN_PRE]) + int(isc[_lib.I_N_FIN_POST]) > 0:
            pool._call("cmbpo_rollout_compact")
            pool.swap("alive_idx", "alive_idx_out")
            # the survivors' count follows from the counters just read: no second host synchronisation
            pool._n_alive = n - int(isc[_lib.I_N_FIN_PRE]) - int(isc[_lib.I_N_FIN_POST])
        pool.swap("cur_obs", "next_obs")
        pool.swap("v_t", "v_n")
        pool.swap("vc_t", "vc_n")
        pool.ptr += 1
        rs.ptr = pool.ptr
        self._host["total_samples"] = float(dsc[_lib.D_TOTAL_SAMPLES])
        self._host["total_dkl"] = float(dsc[_lib.D_TOTAL_DKL])
        info = {"alive_ratio": pool.n_alive / self.batch_size, "ensemble_dkl_path": t["dkl_t"], "cost": t["cost_t"]}
        return t["cur_obs"], t["rew_t"], t["term_t"], info

    def sample_many(self, max_steps=None, max_samples=None, stop_total=None, min_alive_ratio=None):
        """Consecutive sample() calls without the interpreter between them (cmbpo_rollout_run): the rollout loop of
        algorithms/cmbpo.py:352-360.  Steps until `max_steps` are taken, no branch is alive, the buffer is full,
        alive / batch_size <= min_alive_ratio (:358-359) or the sampler's total_samples >= stop_total (:356-357) -- each
        test made after a step, as the reference's loop makes them.  `max_samples` is the budget every step is given.
        Returns (steps taken, info of the last step).  On a sharded sampler, with injected draws or with kernel events
        on, the same loop runs over sample()."""
        pool, env, pol = self.pool, self.env, self.policy
        sharded = self.comm is not None and self.comm.world > 1
        min_alive = int(np.floor(min_alive_ratio * self.batch_size + 1e-9)) if min_alive_ratio is not None else 0
        left = int(max_steps) if max_steps is not None else (1 << 30)
        steps, info = 0, None
        # shards without a budget or stop rule between them roll independently: the native loop runs each shard to its
        # end and the counts are exchanged ONCE (one collective per call instead of one per step)
        local_only = sharded and not max_samples and stop_total is None and min_alive_ratio is None and max_steps is None
        if (sharded and not local_only) or env.kernel_events is not None:
            while left > 0 and self.any_alive() and pool.has_room:
                _, _, _, info = self.sample(max_samples=max_samples)
                steps += 1
                left -= 1
                # every quantity of the stop tests is GLOBAL on a sharded sampler (the budget and alive_ratio already are):
                # all ranks leave the loop after the same step, so the step's collectives stay matched
                if stop_total is not None and self.global_total_samples >= stop_total:
                    break
                if min_alive_ratio is not None and info["alive_ratio"] <= min_alive_ratio:
                    break
            return steps, info
        t, rs = pool.t, pool.rs
        B, A = self.batch_size, pool.act_dim
        lib = _lib.lib()
        with torch.cuda.device(self.device):
            if getattr(self, "_scratch", None) is None or self._scratch[0].shape[1] != B:
                E, O = env._model.num_nets, env.output_dim
                self._scratch = (torch.empty((E, B, O), dtype=torch.float32, device=self.device),
                                 torch.empty((E, B, O), dtype=torch.float32, device=self.device))
                self._handles = None
            if getattr(self, "_handles", None) is None:
                self._handles = (pol.actor.mlp.handle, env._model.mlp.handle, pol.v.mlp.handle, pol.vc.mlp.handle,
                                 self._scratch[0].data_ptr(), self._scratch[1].data_ptr())
            h = self._handles
            if getattr(self, "_run_host", None) is None:
                self._run_host = torch.empty((64, 384), dtype=torch.uint8, pin_memory=True)
                self._run_out = (C.c_int(0), C.c_int(0), C.c_int(0))
            rs.max_samples = int(max_samples) if max_samples else 0
            rs.dkl_lim = float(self.dkl_lim)
            rs.max_path_length = self._max_path_length
            rs.use_host_budget = 0
            stream = _lib.current_stream()
            stop = float("nan") if stop_total is None else float(stop_total)     # NaN: no stop rule on the total
            done_o, alive_o, swaps_o = self._run_out
            while left > 0 and pool.n_alive > 0 and pool.has_room:
                ck = self._draws
                if ck is None or ck[0] >= ck[1].shape[0] or ck[1].shape[1] != B:
                    ck = self._draw_chunk()
                k = ck[0]
                take = min(left, ck[1].shape[0] - k, 64)
                _lib.check(lib.cmbpo_rollout_run(
                    C.byref(rs), pool.n_alive, h[0], h[1], h[2], h[3], env._task_id, env._model.num_nets,
                    ck[1].data_ptr() + k * B * A * 4, ck[2].data_ptr() + k * B * ck[2].element_size(), B * A, B,
                    h[4], h[5], take, stop, min_alive, self._run_host.data_ptr(), C.byref(done_o), C.byref(alive_o),
                    C.byref(swaps_o), stream), "cmbpo_rollout_run")
                done = done_o.value
                if done == 0:
                    break
                # the struct was advanced in native code: bring the tensor table to the same state
                if done & 1:
                    for a, b in (("cur_obs", "next_obs"), ("v_t", "v_n"), ("vc_t", "vc_n")):
                        t[a], t[b] = t[b], t[a]
                if swaps_o.value & 1:
                    t["alive_idx"], t["alive_idx_out"] = t["alive_idx_out"], t["alive_idx"]
                ck[0] = k + done
                pool.ptr += done
                self._n_episodes += done
                steps += done
                left -= done
                blk = self._run_host[:done, :128].view(torch.int32).numpy()
                self.n_budget_terminated += int(blk[:, _lib.I_N_FIN_PRE].sum() - blk[:, _lib.I_N_UNC].sum())
                last = self._run_host[done - 1]
                isc, dsc = last[:128].view(torch.int32).numpy().copy(), last[128:].view(torch.float64).numpy().copy()
                pool._n_alive, pool._size = alive_o.value, int(isc[_lib.I_SIZE])
                self._dsc = dsc
                self._host["total_samples"] = float(dsc[_lib.D_TOTAL_SAMPLES])
                self._host["total_dkl"] = float(dsc[_lib.D_TOTAL_DKL])
                info = {"alive_ratio": pool.n_alive / B, "ensemble_dkl_path": t["dkl_t"], "cost": t["cost_t"]}
                if done < take:       # a stop test fired inside the call
                    break
        if local_only:
            # (a shard whose buffer is full cannot continue: its rows do not keep the other shards in the caller's loop --
            # every shard leaves `while any_alive() and has_room` after the same call, the collectives stay matched)
            g = self.comm.all_reduce_host([pool.n_alive, self.batch_size, self._host["total_samples"],
                                           pool.n_alive if pool.has_room else 0])
            self.global_alive, self._global_total_samples = int(g[3]), g[2]
            info = {"alive_ratio": g[0] / g[1], "ensemble_dkl_path": t["dkl_t"], "cost": t["cost_t"]}
        return steps, info

    def _idle_step(self, max_samples):
        pool, t = self.pool, self.pool.t
        with torch.cuda.device(self.device):
            if max_samples:
                pool.rs.max_samples = int(max_samples)
                pool._call("cmbpo_rollout_count")
                self.comm.all_gather_i32(t["iscal"][8:12])
        g = self.comm.all_reduce_host([0, self.batch_size, self._host["total_samples"]])
        self.global_alive, self._global_total_samples = int(g[0]), g[2]
        info = {"alive_ratio": g[0] / g[1], "ensemble_dkl_path": t["dkl_t"], "cost": t["cost_t"]}
        return t["cur_obs"], t["rew_t"], t["term_t"], info

    def any_alive(self):
        """True while any shard still has alive branches (== pool.n_alive > 0 on one GPU)."""
        if self.comm is not None and self.comm.world > 1:
            return getattr(self, "global_alive", 1) > 0
        return self.pool.n_alive > 0

    def finish_all_paths(self):
        """model_sampler.py:418-444: bootstrap-finish whatever is still alive, return diagnostics."""
        if self.pool.n_alive > 0:
            with torch.cuda.device(self.device):
                self.pool._call("cmbpo_rollout_finish", 2)
                self.pool._call("cmbpo_rollout_compact")
                self.pool.swap("alive_idx", "alive_idx_out")
            self.pool.sync_counters()
            assert self.pool.n_alive == 0   # something went wrong with finishing all paths
        return self.get_diagnostics()

    def compute_dynamics_dkl(self, obs_batch, depth=1):
        """model_sampler.py:151-167: mean per-step ensemble DKL along `depth` policy steps (calibration)."""
        obs = obs_batch if isinstance(obs_batch, torch.Tensor) else \
            torch.from_numpy(np.ascontiguousarray(obs_batch, dtype=np.float32))
        obs = obs.to(self.device, torch.float32)
        for _ in range(depth):
            n = obs.shape[0]
            if n == 0:
                break
            outs = self.policy.get_action_outs(obs)
            next_obs, _, terminal, info = self.env.step(obs, outs["pi"])
            self._host["total_dkl"] += float(info["ensemble_dkl_mean"]) * n
            self._host["total_samples"] += n
            obs = next_obs[~terminal[:, 0]]
        return self.dyn_dkl * depth
