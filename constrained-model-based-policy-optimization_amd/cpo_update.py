"""CPO trust-region update -- host-side mirror of ``CPOAgent.update_pi`` (``policies/cpo_policy.py:139-300``).

The reference evaluates TF graph fetches through ``sess.run`` (a full feed of the batch per call: 1 gradient
eval, 11 or 22 Hessian-vector products, up to 11 line-search evals) and does the vector algebra in NumPy in
between.  Here the batch is bound once on the device, every fetch is one fused HIP kernel behind the C-ABI
(``csrc/policy_update.hip``) and the parameter-sized vectors never leave the GPU (``csrc/vec_ops.hip``):

    flat_g, flat_b, pi_loss, surr_cost, cur_cret_avg  -> cmbpo_pi_loss_grad (x2)
    v = cg(Hx, g), w = cg(Hx, b)                       -> 10 x [cmbpo_pi_fvp, (all-reduce), cmbpo_cg_step] each
    q, r, s, b.b                                       -> cmbpo_vec_dots
    set_and_eval(step) -> [d_kl, pi_loss, surr_cost]   -> cmbpo_vec_lincomb + cmbpo_pi_set_params + cmbpo_pi_eval

Host synchronisations per update: one after the CG phase (8 scalars) and one per line-search trial (the
accept / reject decision is the reference's, on host scalars).  The decision logic (c, margin, cases 0-4, dual
(lam, nu), step, backtracking) follows the reference line for line in meaning.  Reductions are sums weighted
by sample counts (all-reduced over ranks when a ``dist.Comm`` is given) instead of ``mpi_avg`` of per-rank means
(``utilities/mpi_tools.py:67-69``), and counts are never cast to float32.
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

    def log_tabular(self, *a, **k):
        pass


def cg(Ax, b, cg_iters=10):
    """Host form of utilities/trust_region.py:32-45 (kept for callers that hold NumPy vectors)."""
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
    """Device side of the update: owns the ``cmbpo_pi_t`` handle, the bound batch and the work vectors."""

    def __init__(self, obs_dim, act_dim, hidden=128, device=None, comm=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.D, self.A = int(obs_dim), int(act_dim)
        self.comm = comm
        self.use_graph = True       # CG iterations as a cached hipGraph (single-GPU jobs)
        self.keep_activations = True    # Fisher-vector products reuse the hidden activations loss_grad computed
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_pi_create(C.byref(self._h), self.D, int(hidden), self.A), "cmbpo_pi_create")
        self.P = _lib.lib().cmbpo_pi_num_params(self._h)
        f = dict(dtype=torch.float32, device=self.device)
        self.params = torch.zeros(self.P, **f)
        self.vec = torch.zeros(self.P, **f)       # raw kernel output (sums over samples)
        self.dirv = torch.zeros(self.P, **f)      # staging of host directions
        self.work = {k: torch.zeros(self.P, **f) for k in
                     ("g", "b", "v", "w", "hv", "hw", "cg_r", "cg_p", "x", "old", "trial")}
        self.sums = torch.zeros(8, dtype=torch.float64, device=self.device)
        self.sums_g = torch.zeros(8, dtype=torch.float64, device=self.device)
        self.scal = torch.zeros(16, dtype=torch.float64, device=self.device)   # [0] cg r.r ; [8..] dot products
        self.batch = _lib.PiBatchStruct()
        self._keep = None
        self.n_global = 0

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().cmbpo_pi_cg_release(h)
                _lib.lib().cmbpo_pi_destroy(h)
            except Exception:
                pass

    def _s(self):
        return _lib.current_stream()

    def early_read(self, *tensors, tag=""):
        """Start copying small device results to the host behind everything enqueued on the current stream SO FAR -- on a
        side stream, so that the copy does not queue behind what the caller enqueues next (a CG solve of a millisecond) --
        and return a function that waits for the copies and yields them as NumPy arrays.  The host then decides the next
        step while the GPU is still busy, instead of the GPU idling through three blocking copies and the decision.
        Reads with the same `tag` share their pinned host buffers: take one's result before starting the next."""
        with torch.cuda.device(self.device):
            if getattr(self, "_rd_stream", None) is None:
                self._rd_stream = torch.cuda.Stream(device=self.device)
                self._rd_pinned = {}
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self._rd_stream.wait_event(ev)
            hosts = []
            with torch.cuda.stream(self._rd_stream):
                for k, t in enumerate(tensors):
                    key = (tag, k, t.dtype, tuple(t.shape))
                    h = self._rd_pinned.get(key)
                    if h is None:
                        h = self._rd_pinned[key] = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
                    h.copy_(t, non_blocking=True)
                    hosts.append(h)
                done = torch.cuda.Event()
                done.record(self._rd_stream)

        def wait():
            done.synchronize()
            return [h.numpy().copy() for h in hosts]
        return wait

    # -- parameters ----------------------------------------------------------------------------------------
    def set_params(self, flat):
        """flat: NumPy / torch vector [P] (host or device)."""
        with torch.cuda.device(self.device):
            t = flat if isinstance(flat, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(flat, dtype=F32))
            if t.data_ptr() != self.params.data_ptr():
                self.params.copy_(t.to(self.device, torch.float32).reshape(-1))
            _lib.check(_lib.lib().cmbpo_pi_set_params(self._h, self.params.data_ptr(), self._s()),
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
        # a new feed: whatever activations the handle saved belong to the previous one
        _lib.check(_lib.lib().cmbpo_pi_keep_activations(self._h, 1 if self.keep_activations else 0),
                   "cmbpo_pi_keep_activations")
        self.n_global = n if self.comm is None else int(self.comm.all_reduce_host([n])[0])

    def _reduce(self, *tensors):
        if self.comm is not None and self.comm.world > 1:
            for t in tensors:
                self.comm.all_reduce_sum(t)

    # -- device-resident building blocks ---------------------------------------------------------------------
    def lincomb(self, out, a, x, b=0.0, y=None):
        """out = a * x + b * y on the device."""
        _lib.check(_lib.lib().cmbpo_vec_lincomb(self.P, float(a), x.data_ptr(), float(b),
                                                None if y is None else y.data_ptr(), out.data_ptr(), self._s()),
                   "cmbpo_vec_lincomb")
        return out

    def dots(self, pairs, offset=8):
        """scal[offset + k] = x_k . y_k (device); read back with the next synchronisation."""
        n = len(pairs)
        xs = (C.c_void_p * n)(*[p[0].data_ptr() for p in pairs])
        ys = (C.c_void_p * n)(*[p[1].data_ptr() for p in pairs])
        _lib.check(_lib.lib().cmbpo_vec_dots(self.P, n, xs, ys, self.scal[offset:].data_ptr(), self._s()),
                   "cmbpo_vec_dots")

    def grad_dev(self, which, out, sums):
        """out = grad / N on the device; `sums` (device float64[8]) gets the loss sums."""
        _lib.check(_lib.lib().cmbpo_pi_loss_grad(self._h, C.byref(self.batch), which, self.vec.data_ptr(),
                                                sums.data_ptr(), self._s()), "cmbpo_pi_loss_grad")
        self._reduce(self.vec, sums)
        return self.lincomb(out, 1.0 / float(self.n_global), self.vec)

    def fvp_raw(self, v):
        """self.vec = sum_n J^T M J v (+ log_std diagonal) for the device vector v."""
        self.n_fvp = getattr(self, "n_fvp", 0) + 1
        _lib.check(_lib.lib().cmbpo_pi_fvp(self._h, C.byref(self.batch), v.data_ptr(), self.vec.data_ptr(), self._s()),
                   "cmbpo_pi_fvp")
        self._reduce(self.vec)
        return self.vec

    def hx_dev(self, v, out, damping):
        """out = Hx(v) = hvp / N + damping * v (cpo_policy.py:168, 550-552)."""
        self.fvp_raw(v)
        return self.lincomb(out, 1.0 / float(self.n_global), self.vec, damping, v)

    def cg_dev(self, b, x, damping, iters=10):
        """x = cg(Hx, b) (utilities/trust_region.py:32-45), entirely enqueued on the stream."""
        lib, r, p = _lib.lib(), self.work["cg_r"], self.work["cg_p"]
        if self.comm is None or self.comm.world == 1:
            # one C call; the iterations after the first replay as a cached hipGraph (no all-reduce to interleave)
            _lib.check(lib.cmbpo_pi_cg_solve(self._h, C.byref(self.batch), b.data_ptr(), 1.0 / float(self.n_global),
                                             float(damping), int(iters), x.data_ptr(), r.data_ptr(), p.data_ptr(),
                                             self.vec.data_ptr(), self.scal.data_ptr(), 1 if self.use_graph else 0,
                                             self._s()), "cmbpo_pi_cg_solve")
            self.n_fvp = getattr(self, "n_fvp", 0) + int(iters)
            return x
        _lib.check(lib.cmbpo_cg_init(self.P, b.data_ptr(), x.data_ptr(), r.data_ptr(), p.data_ptr(),
                                     self.scal.data_ptr(), self._s()), "cmbpo_cg_init")
        inv_n = 1.0 / float(self.n_global)
        for _ in range(iters):
            self.fvp_raw(p)
            _lib.check(lib.cmbpo_cg_step(self.P, self.vec.data_ptr(), inv_n, float(damping), x.data_ptr(), r.data_ptr(),
                                         p.data_ptr(), self.scal.data_ptr(), self._s()), "cmbpo_cg_step")
        return x

    def eval_dev(self, sums):
        _lib.check(_lib.lib().cmbpo_pi_eval(self._h, C.byref(self.batch), sums.data_ptr(), self._s()), "cmbpo_pi_eval")
        self._reduce(sums)
        return sums

    # -- NumPy-facing forms (tests, diagnostics) -----------------------------------------------------------------
    def loss_grad(self, which):
        with torch.cuda.device(self.device):
            g = self.grad_dev(which, self.work["g" if which == 0 else "b"], self.sums)
            return g.cpu().numpy(), self.sums.cpu().numpy()

    def fvp(self, v):
        with torch.cuda.device(self.device):
            self.dirv.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=F32)))
            self.fvp_raw(self.dirv)
            return (self.vec / float(self.n_global)).cpu().numpy()

    def evals(self):
        with torch.cuda.device(self.device):
            return self.eval_dev(self.sums).cpu().numpy()


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

    def log(self):
        """policies/cpo_policy.py:303-315"""
        for k in ('Optim_A', 'Optim_B', 'Optim_c', 'Optim_q', 'Optim_r', 'Optim_s', 'Optim_Lam', 'Optim_Nu',
                  'OptimCase', 'Margin', 'BacktrackIters'):
            self.logger.log_tabular(k, average_only=True)

    def _ent(self, log_std):
        return float(np.sum(np.asarray(log_std, np.float64) + 0.5 * np.log(2 * np.pi * np.e)))   # ac_network.py:57-61

    def measures(self, ops):
        """LossPi, SurrCost, SurrAdv, Entropy, KL at the current parameters (cpo_policy.py:613-656)."""
        s = ops.evals()
        n = s[0]
        ent = self._ent(ops.get_params()[-ops.A:])
        surr_adv, surr_cost = s[1] / n, s[2] / n
        return dict(LossPi=F32(-(surr_adv + self.ent_reg * ent)), SurrCost=F32(surr_cost), SurrAdv=F32(surr_adv),
                    Entropy=F32(ent), KL=F32(s[3] / n))

    def update_pi(self, ops, target_kl, cost_lim, real_cost_buf):
        """policies/cpo_policy.py:153-300.  `ops` is a bound PolicyOps.

        Returns the logged scalars plus `pre` / `post` measures (LossPi, SurrCost, SurrAdv, Entropy, KL) that the
        reference obtains with extra sess.run calls around the update (cpo_policy.py:613-656)."""
        A, wk = ops.A, ops.work
        damping = float(F32(self.damping_coeff))
        with torch.cuda.device(ops.device):
            wk["old"].copy_(ops.params)
            g = ops.grad_dev(0, wk["g"], ops.sums_g)
            b = ops.grad_dev(1, wk["b"], ops.sums)
            if self.ent_reg:
                g[-A:] -= float(self.ent_reg)                       # d(-ent_reg * ent) / d log_std
            ops.dots([(b, b)], offset=9)
            rd0 = ops.early_read(ops.sums_g, ops.scal[9:10], ops.params[-A:], tag="pre")   # (nothing enqueued below writes these)
            v = ops.cg_dev(g, wk["v"], damping)                      # :210 (enqueued before the first host wait)
            # host wait #0 -- losses, b.b, log_std: they arrive while the first CG solve runs, and the second solve is
            # enqueued behind it before it ends
            sg, bb_, log_std = rd0()
            bb = F32(bb_[0])
        n = sg[0]
        ent = self._ent(log_std)
        surr_adv_old = sg[1] / n
        pi_l_old = F32(-(surr_adv_old + self.ent_reg * ent))
        surr_cost_old = F32(sg[2] / n)
        cur_cret_avg = F32(sg[4] / n * self.max_path_length)       # cpo_policy.py:533
        pre = dict(LossPi=pi_l_old, SurrCost=surr_cost_old, SurrAdv=F32(surr_adv_old), Entropy=F32(ent))

        rescale = 1 / self.max_path_length
        c = (cur_cret_avg - cost_lim) * rescale                     # :185
        if self.learn_margin:                                       # :188-196
            real_c = np.mean(real_cost_buf)
            self.margin *= self.margin_discount
            self.margin += self.margin_lr * (real_c - cost_lim) * rescale
            self.margin = max(0, self.margin)
        self.margin = F32(self.margin)                              # mpi_avg returns float32 (:201)
        c += self.margin

        trpo_shortcut = bool(bb <= 1e-8 and c < 0 or not self.constrained)   # :216
        with torch.cuda.device(ops.device):
            approx_g = ops.hx_dev(v, wk["hv"], damping)              # :211
            pairs = [(v, approx_g)]
            if not trpo_shortcut:
                w = ops.cg_dev(b, wk["w"], damping)                  # :222
                hw = ops.hx_dev(w, wk["hw"], damping)
                pairs += [(w, approx_g), (w, hw)]                    # :223-224
            ops.dots(pairs, offset=10)
            sc = ops.scal.cpu().numpy()                               # <- host synchronisation #1: q, r, s
        q = F32(sc[10])

        if trpo_shortcut:                                           # :216-219
            r, s, A_, B_ = 0, 0, 0, 0
            optim_case = 4
        else:
            r, s = F32(sc[11]), F32(sc[12])
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

        if optim_case in [3, 4]:                                    # :247-262
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

        # x = (1 / (lam + EPS)) * (v + nu * w) if optim_case > 0 else nu * w      (:266)
        with torch.cuda.device(ops.device):
            x = wk["x"]
            if optim_case == 4 or (optim_case > 0 and nu == 0):
                ops.lincomb(x, 1.0 / (float(lam) + EPS), wk["v"])
            elif optim_case > 0:
                inv = 1.0 / (float(lam) + EPS)
                ops.lincomb(x, inv, wk["v"], inv * float(nu), wk["w"])
            else:
                ops.lincomb(x, float(nu), wk["w"])
        info = dict(Optim_A=A_, Optim_B=B_, Optim_c=c, Optim_q=q, Optim_r=r, Optim_s=s, Optim_Lam=lam,
                    Optim_Nu=nu, Penalty=nu, PenaltyDelta=0, Margin=self.margin, OptimCase=optim_case)
        self.logger.store(**info)

        trial = {}

        def set_and_eval(step):                                     # :278-280
            with torch.cuda.device(ops.device):
                ops.lincomb(ops.params, 1.0, wk["old"], -float(step), x)
                ops.set_params(ops.params)
                ops.eval_dev(ops.sums)
                # (the trial parameters -- and once the step, for the diagnostics -- travel to the host while the evaluation
                # runs: starting a read costs the host ~40 us, which the GPU must not spend idle; the caller mirrors the
                # accepted parameters into the rollout actor without a blocking copy of its own)
                rd = ops.early_read(ops.params, tag="trial") if "step" in trial else ops.early_read(ops.params, x, tag="trial")
                sm = ops.sums.cpu().numpy()                         # <- host synchronisation per trial
                got = rd()
                trial["params"] = got[0]
                if len(got) > 1:
                    trial["step"] = got[1]
                ls = got[0][-A:] if self.ent_reg else log_std
            nn = sm[0]
            ent_new = self._ent(ls)
            return F32(sm[3] / nn), F32(-(sm[1] / nn + self.ent_reg * ent_new)), F32(sm[2] / nn), ent_new

        accepted = False
        for j in range(self.backtrack_iters):                       # :285-300
            kl, pi_l_new, surr_cost_new, ent_new = set_and_eval(step=self.backtrack_coeff ** j)
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
                kl, pi_l_new, surr_cost_new, ent_new = set_and_eval(step=0.)
        info["accepted"] = accepted
        info["step"] = trial["step"] if "step" in trial else x.cpu().numpy()
        info["cur_cret_avg"] = cur_cret_avg
        info["params"] = trial.get("params")                        # the parameters ops holds now (accepted or restored)
        info["pre"] = pre
        info["post"] = dict(LossPi=pi_l_new, SurrCost=surr_cost_new, KL=kl, Entropy=F32(ent_new),
                            SurrAdv=F32(-float(pi_l_new) - self.ent_reg * ent_new))
        return info
