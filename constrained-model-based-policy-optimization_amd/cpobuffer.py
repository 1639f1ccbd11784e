"""CPOBuffer -- host-side mirror of ``buffers/cpobuffer.py:15-534`` (the real-env on-policy buffer).

Real-env samples arrive one at a time from a serial MuJoCo step on the host, so storage stays in host NumPy
arrays (as in the reference); the arithmetic of the path runs in HIP behind the C-ABI:

  * ``finish_path`` (``cpobuffer.py:179-207``): reward + cost GAE of the finished path -> ``cmbpo_gae_segments``
    (float64 ``lfilter`` recurrence, float32 stores, float64 deltas when the bootstrap is the float64 zeros
    of ``samplers/cpo_sampler.py:205-212`` -- NumPy's promotion in ``np.append``);
  * ``get`` (``cpobuffer.py:249-290``): advantage normalisation / cost-advantage centring -> ``cmbpo_adv_normalize``,
    then the reference's 12-array list ``[obs, act, adv, cadv, ret, cret, logp, val, cval, cost, log_std, mu]``
    truncated to ``ptr`` and a dump into the off-policy archive.

The archive accessors the trainer uses for start-state sampling (``epoch_batch``, ``boltz_dist``,
``distributed_batch_from_archive``, ``algorithms/cmbpo.py:241-245``) are kept as thin NumPy index plumbing; moving
them to the device is SURVEY §8(f) row N3.
"""
import warnings

import numpy as np
import torch

from . import _lib

EPS = 1e-8  # utilities/utils.py:19

_SCALARS = ("advantages", "rewards", "returns", "values", "cadvantages", "costs", "creturns", "cvalues", "log_probs")


class CPOBuffer:
    def __init__(self, size, archive_size, observation_space, action_space, device=None, *args, **kwargs):
        self.obs_shape = tuple(observation_space.shape)
        self.act_shape = tuple(action_space.shape)
        self.archive_size = int(archive_size)
        self.max_size = int(size)
        self.device = torch.device(device if device is not None else "cuda")
        self.pi_info_shapes = None
        self.gamma, self.lam, self.cost_gamma, self.cost_lam = 0.99, 0.95, 0.99, 0.95
        self.reset_buffers()
        self.reset_arch()

    # -- storage --------------------------------------------------------------------------------------------
    def _new(self, n, with_pi=True):
        d = {"observations": np.zeros((n,) + self.obs_shape, np.float32),
             "actions": np.zeros((n,) + self.act_shape, np.float32),
             "next_observations": np.zeros((n,) + self.obs_shape, np.float32),
             "terminals": np.zeros(n, np.bool_)}
        for k in _SCALARS:
            d[k] = np.zeros(n, np.float32)
        return d

    def reset_buffers(self):
        """cpobuffer.py:99-117."""
        self.buf_dict = self._new(self.max_size)
        self.buf_dict["epochs"] = np.ones(self.max_size, np.float32) * -1
        if self.pi_info_shapes:
            self.pi_info_bufs = {k: np.zeros([self.max_size] + list(v), np.float32)
                                 for k, v in self.pi_info_shapes.items()}
            self.buf_dict["pi_infos"] = self.pi_info_bufs
        self.ptr, self.path_start_idx, self.path_finished = 0, 0, False

    def reset_arch(self):
        """cpobuffer.py:119-138."""
        self.archive_full = False
        self.arch_dict = self._new(self.archive_size)
        self.arch_dict["epochs"] = np.ones(self.archive_size, np.int64) * -1
        if self.pi_info_shapes:
            self.pi_info_archive = {k: np.zeros([self.archive_size] + list(v), np.float32)
                                    for k, v in self.pi_info_shapes.items()}
            self.arch_dict["pi_infos"] = self.pi_info_archive
        self.archive_ptr = 0
        self.max_pointer = 0

    def initialize(self, pi_info_shapes, gamma=0.99, lam=0.95, cost_gamma=0.99, cost_lam=0.95):
        """cpobuffer.py:79-97."""
        self.pi_info_shapes = pi_info_shapes
        self.pi_info_bufs = {k: np.zeros([self.max_size] + list(v), np.float32) for k, v in pi_info_shapes.items()}
        self.pi_info_archive = {k: np.zeros([self.archive_size] + list(v), np.float32)
                                for k, v in pi_info_shapes.items()}
        self.buf_dict["pi_infos"] = self.pi_info_bufs
        self.arch_dict["pi_infos"] = self.pi_info_archive
        self.sorted_pi_info_keys = sorted(pi_info_shapes.keys())
        self.gamma, self.lam, self.cost_gamma, self.cost_lam = gamma, lam, cost_gamma, cost_lam

    # convenient views with the reference's attribute names
    @property
    def epoch_archive(self):
        return self.arch_dict["epochs"]

    @property
    def size(self):
        return self.ptr

    @property
    def arch_size(self):
        return self.max_pointer

    @property
    def max_ep(self):
        return int(np.max(self.epoch_archive))

    @property
    def min_ep(self):
        return int(np.min(self.epoch_archive[self.epoch_archive > -1]))

    @property
    def epochs_list(self):
        """cpobuffer.py:148-153: the epochs present in the archive."""
        return np.flatnonzero(np.bincount(self.epoch_archive[self.epoch_archive >= 0]))

    def store(self, obs, act, next_obs, rew, val, cost, cval, logp, pi_info, term, epoch):
        """cpobuffer.py:160-176."""
        assert self.ptr < self.max_size     # buffer has to have room so you can store
        b, p = self.buf_dict, self.ptr
        b["observations"][p], b["actions"][p], b["next_observations"][p] = obs, act, next_obs
        one = lambda x: np.asarray(x).reshape(-1)[0]     # the policy hands over batch-of-one arrays
        b["rewards"][p], b["values"][p], b["costs"][p], b["cvalues"][p] = one(rew), one(val), one(cost), one(cval)
        b["log_probs"][p], b["terminals"][p], b["epochs"][p] = one(logp), one(term), epoch
        for k in self.sorted_pi_info_keys:
            self.pi_info_bufs[k][p] = pi_info[k]
        self.ptr += 1
        self.path_finished = False

    # -- the path arithmetic (HIP) -----------------------------------------------------------------------------
    def finish_path(self, last_val=0, last_cval=0):
        """cpobuffer.py:179-207."""
        lo, hi = self.path_start_idx, self.ptr
        n = hi - lo
        b = self.buf_dict
        lv, lcv = np.asarray(last_val), np.asarray(last_cval)
        # np.append(float32 buffer, x): anything but a float32 bootstrap promotes the deltas to float64
        mask = (1 if lv.dtype != np.float32 else 0) | (2 if lcv.dtype != np.float32 else 0)
        if n > 0:
            dev = self.device
            with torch.cuda.device(dev):
                up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
                rew, val, cost, cval = (up(b[k][lo:hi]) for k in ("rewards", "values", "costs", "cvalues"))
                offs = torch.tensor([0, n], dtype=torch.int32, device=dev)
                lvd, lcvd = up(lv.reshape(-1)[:1]), up(lcv.reshape(-1)[:1])
                fm = torch.tensor([mask], dtype=torch.uint8, device=dev)
                out = [torch.empty(n, dtype=torch.float32, device=dev) for _ in range(4)]
                _lib.check(_lib.lib().cmbpo_gae_segments(
                    1, offs.data_ptr(), rew.data_ptr(), val.data_ptr(), cost.data_ptr(), cval.data_ptr(),
                    lvd.data_ptr(), lcvd.data_ptr(), fm.data_ptr(), float(self.gamma), float(self.lam),
                    float(self.cost_gamma), float(self.cost_lam), out[0].data_ptr(), out[1].data_ptr(),
                    out[2].data_ptr(), out[3].data_ptr(), _lib.current_stream()), "cmbpo_gae_segments")
                for k, t in zip(("advantages", "returns", "cadvantages", "creturns"), out):
                    b[k][lo:hi] = t.cpu().numpy()
        self.path_start_idx = self.ptr
        self.path_finished = True

    def dump_to_archive(self):
        """cpobuffer.py:210-248: copy the finished on-policy samples behind the archive pointer."""
        assert self.path_finished
        if self.archive_ptr >= self.archive_size - self.ptr:
            self.archive_full = True
            self.archive_ptr = 0
            warnings.warn('Archive is full, deleting old samples.')
        dst = slice(self.archive_ptr, self.archive_ptr + self.ptr)
        for k, a in self.arch_dict.items():
            if k == "pi_infos":
                for kk in self.sorted_pi_info_keys:
                    a[kk][dst] = self.pi_info_bufs[kk][:self.ptr]
            else:
                a[dst] = self.buf_dict[k][:self.ptr]
        self.archive_ptr += self.ptr
        self.max_pointer = max(self.archive_ptr, self.max_pointer)

    def get(self):
        """cpobuffer.py:249-290."""
        b, n = self.buf_dict, self.ptr
        if n > 0:
            dev = self.device
            with torch.cuda.device(dev):
                adv = torch.from_numpy(b["advantages"][:n].copy()).to(dev)
                cadv = torch.from_numpy(b["cadvantages"][:n].copy()).to(dev)
                stats = torch.zeros(16, dtype=torch.float64, device=dev)
                _lib.check(_lib.lib().cmbpo_adv_normalize(n, adv.data_ptr(), cadv.data_ptr(), stats.data_ptr(),
                                                         _lib.current_stream()), "cmbpo_adv_normalize")
                b["advantages"][:n] = adv.cpu().numpy()
                b["cadvantages"][:n] = cadv.cpu().numpy()
        self.dump_to_archive()
        keys = ["observations", "actions", "advantages", "cadvantages", "returns", "creturns", "log_probs", "values",
                "cvalues", "costs"]
        res = [b[k][:n].copy() for k in keys] + [self.pi_info_bufs[k][:n].copy() for k in self.sorted_pi_info_keys]
        diagnostics = dict(poolr_ret_mean=b["returns"][:n].mean(), poolr_cret_mean=b["creturns"][:n].mean())
        self.reset_buffers()
        return res, diagnostics

    # -- archive access (NumPy index plumbing; callers of the path) ----------------------------
    def _fields(self, fields):
        if fields is None:
            return ['observations', 'actions', 'next_observations', 'rewards', 'terminals'], False
        fields = list(fields)
        want_pi = 'pi_infos' in fields
        if 'all' in fields:
            return [k for k in self.arch_dict if k != 'pi_infos'], True
        return [f for f in fields if f != 'pi_infos'], want_pi

    def _take(self, idx, fields):
        names, want_pi = self._fields(fields)
        out = {k: self.arch_dict[k][idx] for k in names}
        if want_pi:
            out.update({k: self.pi_info_archive[k][idx] for k in self.sorted_pi_info_keys})
        return out

    def get_archive(self, fields=None):
        return self._take(slice(0, self.arch_size), fields)

    def rand_batch_from_archive(self, batch_size, fields=None):
        idx = np.random.randint(0, self.arch_size, batch_size) if self.arch_size else np.arange(0, 0)
        return self._take(idx, fields)

    def boltz_dist(self, kls, alpha=1):
        """cpobuffer.py:385-396: per-sample probabilities, Boltzmann over the epochs' mean policy KL."""
        ep_probs = np.exp(alpha * np.negative(kls))
        ep_probs /= np.sum(ep_probs)
        ea = self.epoch_archive
        sample_p = np.bincount(ea[ea >= 0]).astype(np.float32)
        sample_p[sample_p > 0] = ep_probs / sample_p[sample_p > 0]
        return np.where(ea >= 0, sample_p[ea], 0)

    def distributed_batch_from_archive(self, batch_size, dist, fields=None):
        idx = np.random.choice(np.arange(self.archive_size), size=batch_size, p=dist)
        return self._take(idx, fields)

    def epoch_batch(self, batch_size, epochs, fields=None):
        assert len(np.shape(epochs)) == 1
        ep = np.array(epochs)
        if np.any(ep > self.max_ep) or np.any(ep < self.min_ep):
            print('Warning: epoch not contained in buffer.')
            return None
        idx = np.array([np.random.choice(np.flatnonzero(self.epoch_archive == e), size=batch_size) for e in epochs])
        return self._take(idx, fields)
