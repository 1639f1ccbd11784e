"""ModelBuffer -- host-side mirror of ``buffers/modelbuffer.py:18-226`` over device-resident state.

Same constructor / ``initialize`` / ``reset`` / ``store_multiple`` / ``finish_path_multiple`` / ``get``
and the ``size`` / ``has_room`` / ``alive_paths`` properties.  The arrays live on the GPU as torch
tensors and every operation is a HIP kernel behind the C-ABI (``csrc/rollout_state.hip``).

Layout differences (performance only, invisible through the API):
  * buffers are time-major ``[T, B, ...]`` and allocated once; ``reset()`` zeroes pointers instead of
    re-allocating 17 arrays (``modelbuffer.py:53-98``);
  * ``next_obs`` / ``dyn_error`` / ``term`` / ``roll_lengths`` buffers are never read by ``get()`` in the
    reference (``modelbuffer.py:212-218``) and are not kept;
  * ``populated_mask[b, t]`` is represented as ``t < len[b]`` (alive branches share ``ptr``).

``get()`` returns the reference's 12-array list ``[obs, act, adv, cadv, ret, cret, logp, val, cval, cost,
log_std, mu]`` (``pi_info`` sorted by key) in branch-major, time-minor order.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import RolloutStruct

EPS = 1e-8  # utilities/utils.py:19


def _np_or_t(x, device, dtype=torch.float32):
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=dtype)
    return torch.as_tensor(np.ascontiguousarray(x), device=device).to(dtype)


class ModelBuffer:
    def __init__(self, batch_size, obs_dim, act_dim, max_path_length, device=None, comm=None,
                 *args, **kwargs):
        self.max_path_length = int(max_path_length)
        self.batch_size = int(batch_size)
        self.capacity = int(batch_size)
        self.obs_shape, self.act_shape = obs_dim, act_dim
        self.obs_dim = int(np.prod(obs_dim))
        self.act_dim = int(np.prod(act_dim))
        self.device = torch.device(device if device is not None else "cuda")
        self.comm = comm                      # optional dist.Comm for sharded statistics
        self.pi_info_shapes = None
        self.gamma, self.lam, self.cost_gamma, self.cost_lam = 0.99, 0.95, 0.99, 0.95
        self._alloc(self.capacity)
        self.reset()

    # ------------------------------------------------------------------------------------------
    def _alloc(self, B):
        T, D, A, dev = self.max_path_length, self.obs_dim, self.act_dim, self.device
        f = dict(dtype=torch.float32, device=dev)
        d = dict(dtype=torch.float64, device=dev)
        t = {}
        for name, dim in (("obs_buf", D), ("act_buf", A), ("mu_buf", A), ("ls_buf", A)):
            t[name] = torch.empty((T, B, dim), **f)
        for name in ("rew_buf", "val_buf", "cost_buf", "cval_buf", "logp_buf",
                     "adv_buf", "ret_buf", "cadv_buf", "cret_buf"):
            t[name] = torch.empty((T, B), **f)
        t["alive_idx"] = torch.empty(B, dtype=torch.int32, device=dev)
        t["alive_idx_out"] = torch.empty(B, dtype=torch.int32, device=dev)
        # device scalar block: int32[32] counters | float64[32] accumulators, one D2H copy reads both
        t["scal_bytes"] = torch.zeros(32 * 4 + 32 * 8, dtype=torch.uint8, device=dev)
        t["iscal"] = t["scal_bytes"][:128].view(torch.int32)
        t["dscal"] = t["scal_bytes"][128:].view(torch.float64)
        t["alive"] = torch.empty(B, dtype=torch.uint8, device=dev)
        t["fin_code"] = torch.empty(B, dtype=torch.uint8, device=dev)
        t["len"] = torch.empty(B, dtype=torch.int32, device=dev)
        for name in ("dkl_acc", "path_ret", "path_cost", "path_dyn_var"):
            t[name] = torch.empty(B, **d)
        t["store_part"] = torch.empty(((B + 63) // 64) * 8, **d)
        # per-step values (slot indexed); cur/next and t/n pairs are swapped every step
        for name, dim in (("cur_obs", D), ("next_obs", D), ("act_t", A), ("mu_t", A), ("ls_t", A)):
            t[name] = torch.zeros((B, dim), **f)
        for name in ("logp_t", "v_t", "vc_t", "v_n", "vc_n", "rew_t", "cost_t", "dkl_t", "epv_t"):
            t[name] = torch.zeros(B, **f)
        t["term_t"] = torch.zeros(B, dtype=torch.uint8, device=dev)
        t["offsets"] = torch.empty(B + 1, dtype=torch.int32, device=dev)
        t["stats"] = torch.zeros(16, **d)
        self.t = t
        self.capacity = B
        self.rs = RolloutStruct()
        self.rs.T, self.rs.obs_dim, self.rs.act_dim = T, D, A
        self.rs.rank, self.rs.world = 0, 1
        self.rs.max_samples = 0
        self.rs.dkl_lim = float("inf")
        self.rs.max_path_length = T
        self._bind_all()

    def _bind_all(self):
        for name, _ in RolloutStruct._fields_:
            if name in self.t:
                setattr(self.rs, name, self.t[name].data_ptr())
        self.rs.use_host_budget = 0

    def swap(self, a, b):
        """Exchange two same-shaped state arrays (cur_obs <-> next_obs, v_t <-> v_n, ...)."""
        self.t[a], self.t[b] = self.t[b], self.t[a]
        setattr(self.rs, a, self.t[a].data_ptr())
        setattr(self.rs, b, self.t[b].data_ptr())

    def _call(self, fn, *args):
        with torch.cuda.device(self.device):
            _lib.check(getattr(_lib.lib(), fn)(C.byref(self.rs), *args, _lib.current_stream()), fn)

    # ------------------------------------------------------------------------------------------
    def initialize(self, pi_info_shapes, gamma=0.99, lam=0.95, cost_gamma=0.99, cost_lam=0.95):
        """modelbuffer.py:41-51."""
        self.pi_info_shapes = pi_info_shapes
        keys = sorted(pi_info_shapes.keys())
        if keys != ["log_std", "mu"]:
            raise NotImplementedError("HIP buffer stores the Gaussian pi_info {mu, log_std} (ac_network.py:120)")
        self.sorted_pi_info_keys = keys
        self.gamma, self.lam, self.cost_gamma, self.cost_lam = gamma, lam, cost_gamma, cost_lam
        self.rs.gamma, self.rs.lam = float(gamma), float(lam)
        self.rs.cost_gamma, self.rs.cost_lam = float(cost_gamma), float(cost_lam)

    def reset(self, batch_size=None):
        """modelbuffer.py:53-98 (pointers only; the buffers are reused unless the batch size changes)."""
        if batch_size is not None and int(batch_size) != self.capacity:
            self.batch_size = int(batch_size)
            self._alloc(self.batch_size)
        self.rs.B = self.batch_size
        self.rs.gamma, self.rs.lam = float(self.gamma), float(self.lam)
        self.rs.cost_gamma, self.rs.cost_lam = float(self.cost_gamma), float(self.cost_lam)
        self.ptr = 0
        self.rs.ptr = 0
        self.path_start_idx = 0
        self.max_size = self.max_path_length
        self._call("cmbpo_rollout_reset")
        self._n_alive = self.batch_size
        self._size = 0

    # ------------------------------------------------------------------------------------------
    def read_scalars(self):
        """One small D2H copy (the host sync of a step): (int32 counters, float64 accumulators)."""
        if getattr(self, "_scal_host", None) is None:
            # pinned once: the copy is a true async D2H and the NumPy views below never change
            self._scal_host = torch.empty(384, dtype=torch.uint8, pin_memory=True)
            self._scal_views = (self._scal_host[:128].view(torch.int32).numpy(), self._scal_host[128:].view(torch.float64).numpy())
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_rollout_read_scalars(C.byref(self.rs), self._scal_host.data_ptr(), _lib.current_stream()),
                       "cmbpo_rollout_read_scalars")
        isc, dsc = self._scal_views[0].copy(), self._scal_views[1].copy()     # (the pinned block is rewritten by the next read)
        self._n_alive = int(isc[_lib.I_N_ALIVE])
        self._size = int(isc[_lib.I_SIZE])
        return isc, dsc

    def sync_counters(self):
        return self.read_scalars()[0]

    @property
    def size(self):
        return self._size

    @property
    def has_room(self):
        return self.ptr < self.max_size

    @property
    def alive_paths(self):
        """bool[B], modelbuffer.py:110-112."""
        return self.t["alive"].bool().cpu().numpy()

    @property
    def n_alive(self):
        return self._n_alive

    def alive_indices(self):
        return self.t["alive_idx"][: self._n_alive]

    # -- API-parity entry points taking host arrays in compact (alive-only) order --------------
    def store_multiple(self, obs, act, next_obs, rew, val, cost, cval, dyn_error, logp, pi_info, term):
        """modelbuffer.py:114-135.  Arguments cover the currently alive paths, in index order."""
        assert self.ptr < self.max_size
        idx = self.alive_indices().long()
        t, dev = self.t, self.device
        t["cur_obs"][idx] = _np_or_t(obs, dev)
        t["act_t"][idx] = _np_or_t(act, dev)
        t["rew_t"][idx] = _np_or_t(rew, dev)
        t["v_t"][idx] = _np_or_t(val, dev)
        t["cost_t"][idx] = _np_or_t(cost, dev)
        t["vc_t"][idx] = _np_or_t(cval, dev)
        t["epv_t"][idx] = _np_or_t(dyn_error, dev)
        t["logp_t"][idx] = _np_or_t(logp, dev)
        t["mu_t"][idx] = _np_or_t(pi_info["mu"], dev)
        t["ls_t"][idx] = _np_or_t(pi_info["log_std"], dev)
        t["fin_code"].zero_()
        self._call("cmbpo_rollout_store")
        self.ptr += 1
        self.rs.ptr = self.ptr
        self.sync_counters()

    def finish_path_multiple(self, term_mask, last_val=0, last_cval=0):
        """modelbuffer.py:138-182.  term_mask covers the alive paths; last_* the paths to finish."""
        term_mask = np.asarray(term_mask, dtype=bool)
        if not term_mask.any():
            return
        assert self._n_alive == len(term_mask)
        idx = self.alive_indices().long()
        dev = self.device
        sel = idx[torch.as_tensor(term_mask, device=dev)]
        lv = np.asarray(last_val)
        # float64 zeros (append_vals=False, model_sampler.py:404) promote the reward deltas to float64
        zero_boot = lv.dtype == np.float64 and not lv.any()
        code = torch.zeros_like(self.t["fin_code"])
        code[sel] = 2 if zero_boot else 1
        self.t["fin_code"].copy_(code)
        self.t["v_t"][sel] = _np_or_t(np.broadcast_to(lv, (int(term_mask.sum()),)), dev)
        self.t["vc_t"][sel] = _np_or_t(np.broadcast_to(np.asarray(last_cval), (int(term_mask.sum()),)), dev)
        self._call("cmbpo_rollout_finish", 0)
        self._call("cmbpo_rollout_compact")
        self.swap("alive_idx", "alive_idx_out")
        self.sync_counters()

    # ------------------------------------------------------------------------------------------
    def get(self, as_tensors=False):
        """modelbuffer.py:184-226: normalise adv, centre cadv, flatten populated entries, reset."""
        self.sync_counters()
        assert self._n_alive == 0, "all paths have to be finished"          # modelbuffer.py:194
        t = self.t
        ev = None
        if getattr(self, "get_events", None) is not None:    # bench.py: HIP events around get()'s kernels
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True),
                  torch.cuda.Event(enable_timing=True))
        # the number of stored samples is the pool's size counter, read with the step counters: no device round trip here --
        # and the output block is sized and split BEFORE the first launch, so that the scan / moment launches and the flatten
        # follow each other on the stream without the interpreter between them
        n = int(self._size)
        D, A, dev = self.obs_dim, self.act_dim, self.device
        dims = [D, A, 0, 0, 0, 0, 0, 0, 0, 0, A, A]
        # one allocation for the twelve arrays (each starts on a 256-byte boundary) and one split: the list holds views of it
        sizes = []
        for d in dims:
            m = n * max(d, 1)
            sizes += [m, -m % 64]
        flat = torch.empty(max(sum(sizes), 1), dtype=torch.float32, device=dev)
        parts = flat.split_with_sizes(sizes)[0::2] if n > 0 else [flat[:0]] * 12
        outs = [p_.view(n, d) if d else p_ for p_, d in zip(parts, dims)]
        ptrs = (C.c_void_p * 12)(*[o.data_ptr() for o in outs])
        if ev is not None:
            ev[0].record()
        sharded = self.comm is not None and self.comm.world > 1
        if sharded:
            self._call("cmbpo_buffer_offsets", t["offsets"].data_ptr())
            self._call("cmbpo_buffer_moments", 0, t["stats"].data_ptr())
            self.comm.all_reduce_sum(t["stats"][8:13])
            self._call("cmbpo_buffer_moments", 1, t["stats"].data_ptr())
            self._call("cmbpo_buffer_moments", 2, t["stats"].data_ptr())
            self.comm.all_reduce_sum(t["stats"][13:14])
            self._call("cmbpo_buffer_moments", 3, t["stats"].data_ptr())
        else:
            # one GPU: the scan and both moment passes as two launches
            self._call("cmbpo_buffer_prepare", t["offsets"].data_ptr(), t["stats"].data_ptr())
        if n > 0:
            if ev is not None:
                ev[1].record()
            self._call("cmbpo_buffer_flatten", t["offsets"].data_ptr(), t["stats"].data_ptr(), ptrs)
            if ev is not None:
                ev[2].record()
                self.get_events.append((ev[0], ev[1], ev[2], n))
            st = t["stats"].cpu().numpy()
            ret_mean, cret_mean = float(st[4]), float(st[5])
        else:
            ret_mean, cret_mean = 0, 0
        diagnostics = dict(poolm_batch_size=n, poolm_ret_mean=ret_mean, poolm_cret_mean=cret_mean)
        res = outs if as_tensors else [o.cpu().numpy() for o in outs]
        self.reset()
        return res, diagnostics
