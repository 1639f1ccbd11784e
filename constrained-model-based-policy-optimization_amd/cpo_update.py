"""CPO trust-region update -- host-side mirror of ``CPOAgent.update_pi`` (``policies/cpo_policy.py:139-300``).

The reference evaluates TF graph fetches through ``sess.run`` (a full feed of the batch per call: 1 gradient
eval, 11 or 22 Hessian-vector products, up to 11 line-search evals).  Here the batch is bound once on the
device and each fetch is one fused HIP kernel behind the C-ABI (``csrc/policy_update.hip``):

    flat_g, flat_b, pi_loss, surr_cost, cur_cret_avg  -> cmbpo_pi_loss_grad (x2)
    Hx = hvp + damping * v                             -> cmbpo_pi_fvp
    set_and_eval(step) -> [d_kl, pi_loss, surr_cost]   -> cmbpo_pi_set_params + cmbpo_pi_eval

The decision logic (c, margin, cases 0-4, dual (lam, nu), step, backtracking) is the reference's, line for
line in meaning, on host scalars; CG (``utilities/trust_region.py:32-45``) runs on float32 host vectors.
Reductions are sums weighted by sample counts (all-reduced over ranks when a ``dist.Comm`` is given) instead of
``mpi_avg`` of per-rank means (``utilities/mpi_tools.py:67-69``), and counts are never cast to float32.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

EPS = 1e-8  # utilities/utils.py:19
F32 = np.float32


class _NullLogger:
    def __init__(self):
        self.stored = {}

    def log(self, *a, **k):
        pass

    def store(self, **kw):
        self.stored.update(kw)


def cg(Ax, b, cg_iters=10):
    """utilities/trust_region.py:32-45: fixed iteration count, no early exit, EPS in the step length."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    r_dot_old = np.dot(r, r)
    for _ in range(cg_iters):
        z = Ax(p)
        alpha = r_dot_old / (np.dot(p, z) + EPS)
        x += alpha * p
        r -= alpha * z
        r_dot_new = np.dot(r, r)
        p = r + (r_dot_new / r_dot_old) * p
        r_dot_old = r_dot_new
    return x


class PolicyOps:
    """Device side of the update: owns the ``cmbpo_pi_t`` handle, the bound batch and scratch vectors."""

    def __init__(self, obs_dim, act_dim, hidden=128, device=None, comm=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.D, self.A = int(obs_dim), int(act_dim)
        self.comm = comm
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_pi_create(C.byref(self._h), self.D, int(hidden), self.A), "cmbpo_pi_create")
        self.P = _lib.lib().cmbpo_pi_num_params(self._h)
        f = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.P, **f)
        self.vec = torch.zeros(self.P, **f)
        self.dirv = torch.zeros(self.P, **f)
        self.sums = torch.zeros(8, dtype=torch.float64, device=self.device)
        self.batch = _lib.PiBatchStruct()
        self._keep = None
        self.n_global = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().cmbpo_pi_destroy(h)
            except Exception:
                pass

    def _stream(self):
        return _lib.current_stream()

    def set_params(self, flat):
        with torch.cuda.device(self.device):
            t = flat if isinstance(flat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(flat, dtype=F32))
            self.params.copy_(t.to(self.device, torch.float32).reshape(-1))
            _lib.check(_lib.lib().cmbpo_pi_set_params(self._h, self.params.data_ptr(), self._stream()),
                       "cmbpo_pi_set_params")

    def get_params(self):
        return self.params.cpu().numpy()

    def bind(self, obs, act, adv, cadv, logp_old, cost, mu_old, logstd_old):
        """Bind the actor feed (actor_phs, cpo_policy.py:479-487); arrays stay on the device for the whole update."""
        def dev(x):
            t = x if isinstance(x, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(x, dtype=F32))
            return t.to(self.device, torch.float32).contiguous()
        ts = [dev(x) for x in (obs, act, adv, cadv, logp_old, cost, mu_old, logstd_old)]
        n = ts[0].shape[0]
        assert ts[0].shape == (n, self.D) and ts[1].shape == (n, self.A) and ts[6].shape == (n, self.A)
        self._keep = ts
        b = self.batch
        b.n, b.obs_dim, b.act_dim = n, self.D, self.A
        for name, t in zip(("obs", "act", "adv", "cadv", "logp_old", "cost", "mu_old", "logstd_old"), ts):
            setattr(b, name, t.data_ptr())
        self.n_local = n
        self.n_global = n if self.comm is None else int(self.comm.all_reduce_host([n])[0])

    def _reduce(self, *tensors):
        if self.comm is not None and self.comm.world > 1:
            for t in tensors:
                self.comm.all_reduce_sum(t)

    def loss_grad(self, which):
        """(sum-gradient / N as float32 numpy, sums as float64 numpy)."""
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_pi_loss_grad(self._h, C.byref(self.batch), which, self.vec.data_ptr(),
                                                    self.sums.data_ptr(), self._stream()), "cmbpo_pi_loss_grad")
            self._reduce(self.vec, self.sums)
            return (self.vec / float(self.n_global)).cpu().numpy(), self.sums.cpu().numpy()

    def fvp(self, v):
        with torch.cuda.device(self.device):
            self.dirv.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=F32)))
            _lib.check(_lib.lib().cmbpo_pi_fvp(self._h, C.byref(self.batch), self.dirv.data_ptr(), self.vec.data_ptr(),
                                              self._stream()), "cmbpo_pi_fvp")
            self._reduce(self.vec)
            return (self.vec / float(self.n_global)).cpu().numpy()

    def evals(self):
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_pi_eval(self._h, C.byref(self.batch), self.sums.data_ptr(), self._stream()),
                       "cmbpo_pi_eval")
            self._reduce(self.sums)
            return self.sums.cpu().numpy()


class CPOAgent:
    """policies/cpo_policy.py:124-151: TrustRegionAgent + CPOAgent hyper-parameters and the learned margin."""

    def __init__(self, constrained=True, learn_margin=True, damping_coeff=0.1, backtrack_coeff=0.8,
                 backtrack_iters=10, c_gamma=0.99, max_path_length=1, ent_reg=0.0, **_unused):
        self.constrained = constrained
        self.learn_margin = learn_margin
        self.damping_coeff, self.backtrack_coeff, self.backtrack_iters = damping_coeff, backtrack_coeff, backtrack_iters
        self.margin = 0
        self.margin_lr = 0.0001
        self.margin_discount = .9999
        self.c_gamma = c_gamma
        self.max_path_length = max_path_length
        self.ent_reg = ent_reg
        self.reward_penalized = False
        self.trust_region = True
        self.cares_about_cost = constrained
        self.logger = _NullLogger()

    def set_logger(self, logger):
        self.logger = logger

    # -- fetches ------------------------------------------------------------------------------------------
    def _ent(self, log_std):
        return float(np.sum(log_std.astype(np.float64) + 0.5 * np.log(2 * np.pi * np.e)))   # ac_network.py:57-61

    def measures(self, ops):
        """LossPi, SurrCost, SurrAdv, Entropy, KL at the current parameters (cpo_policy.py:613-656)."""
        s = ops.evals()
        n = s[0]
        log_std = ops.get_params()[-ops.A:]
        ent = self._ent(log_std)
        surr_adv, surr_cost = s[1] / n, s[2] / n
        return dict(LossPi=F32(-(surr_adv + self.ent_reg * ent)), SurrCost=F32(surr_cost), SurrAdv=F32(surr_adv),
                    Entropy=F32(ent), KL=F32(s[3] / n))

    def update_pi(self, ops, target_kl, cost_lim, real_cost_buf):
        """policies/cpo_policy.py:153-300.  `ops` is a bound PolicyOps; returns the logged scalars."""
        A = ops.A
        g, sg = ops.loss_grad(0)
        b, _ = ops.loss_grad(1)
        n = sg[0]
        g[-A:] -= F32(self.ent_reg)                               # d(-ent_reg * ent) / d log_std
        ent = self._ent(ops.get_params()[-A:])
        pi_l_old = F32(-(sg[1] / n + self.ent_reg * ent))
        surr_cost_old = F32(sg[2] / n)
        cur_cret_avg = F32(sg[4] / n * self.max_path_length)      # cpo_policy.py:533
        damping = F32(self.damping_coeff)

        def Hx(x):
            x = np.asarray(x, dtype=F32)
            return ops.fvp(x) + damping * x                        # cpo_policy.py:550-552

        old_params = ops.get_params()
        rescale = 1 / self.max_path_length
        c = (cur_cret_avg - cost_lim) * rescale                    # :185
        if self.learn_margin:                                      # :188-196
            real_c = np.mean(real_cost_buf)
            self.margin *= self.margin_discount
            self.margin += self.margin_lr * (real_c - cost_lim) * rescale
            self.margin = max(0, self.margin)
        self.margin = F32(self.margin)                             # mpi_avg returns float32 (:201)
        c += self.margin

        v = cg(Hx, g)                                              # :210-212
        approx_g = Hx(v)
        q = np.dot(v, approx_g)
        if np.dot(b, b) <= 1e-8 and c < 0 or not self.constrained:  # :216-219
            w, r, s, A_, B_ = 0, 0, 0, 0, 0
            optim_case = 4
        else:
            w = cg(Hx, b)
            r = np.dot(w, approx_g)
            s = np.dot(w, Hx(w))
            A_ = q - r ** 2 / s
            B_ = 2 * target_kl - c ** 2 / s
            if c < 0 and B_ < 0:
                optim_case = 3
            elif c < 0 and B_ >= 0:
                optim_case = 2
            elif c >= 0 and B_ >= 0:
                optim_case = 1
                self.logger.log('Alert! Attempting feasible recovery!', 'yellow')
            else:
                optim_case = 0
                self.logger.log('Alert! Attempting infeasible recovery!', 'red')

        if optim_case in [3, 4]:                                   # :247-262
            lam = np.sqrt(q / (2 * target_kl))
            nu = 0
        elif optim_case in [1, 2]:
            LA, LB = [0, r / c], [r / c, np.inf]
            LA, LB = (LA, LB) if c < 0 else (LB, LA)
            proj = lambda x, L: max(L[0], min(L[1], x))
            lam_a = proj(np.sqrt(A_ / B_), LA)
            lam_b = proj(np.sqrt(q / (2 * target_kl)), LB)
            f_a = lambda lam: -0.5 * (A_ / (lam + EPS) + B_ * lam) - r * c / (s + EPS)
            f_b = lambda lam: -0.5 * (q / (lam + EPS) + 2 * target_kl * lam)
            lam = lam_a if f_a(lam_a) >= f_b(lam_b) else lam_b
            nu = max(0, lam * c - r) / (s + EPS)
        else:
            lam = 0
            nu = np.sqrt(2 * target_kl / (s + EPS))

        x = (1. / (lam + EPS)) * (v + nu * w) if optim_case > 0 else nu * w      # :266
        info = dict(Optim_A=A_, Optim_B=B_, Optim_c=c, Optim_q=q, Optim_r=r, Optim_s=s, Optim_Lam=lam,
                    Optim_Nu=nu, Penalty=nu, PenaltyDelta=0, Margin=self.margin, OptimCase=optim_case)
        self.logger.store(**info)

        def set_and_eval(step):                                    # :278-280
            ops.set_params(np.asarray(old_params - step * x, dtype=F32))
            sm = ops.evals()
            nn = sm[0]
            ent_new = self._ent(ops.get_params()[-A:])
            return F32(sm[3] / nn), F32(-(sm[1] / nn + self.ent_reg * ent_new)), F32(sm[2] / nn)

        accepted = False
        for j in range(self.backtrack_iters):                      # :285-300
            kl, pi_l_new, surr_cost_new = set_and_eval(step=self.backtrack_coeff ** j)
            if (kl <= target_kl and (pi_l_new <= pi_l_old if optim_case > 1 else True) and
                    surr_cost_new - surr_cost_old <= max(-c, 0)):
                self.logger.log('Accepting new params at step %d of line search.' % j)
                self.logger.store(BacktrackIters=j)
                info["BacktrackIters"] = j
                accepted = True
                break
            if j == self.backtrack_iters - 1:
                self.logger.log('Line search failed! Keeping old params.')
                self.logger.store(BacktrackIters=j)
                info["BacktrackIters"] = j
                kl, pi_l_new, surr_cost_new = set_and_eval(step=0.)
        info["accepted"] = accepted
        info["step"] = x
        return info
