"""CPU oracle for the CPO trust-region update -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

TF half (parity unpinned by reference outputs: TensorFlow 1.14 cannot run here and the reference has
no fixtures): the policy graph of policies/cpo_policy.py:522-559 restated with torch-CPU autograd --
the same definitions (tf.gradients == reverse-mode autograd, hessian_vector_product == double
back-prop, utilities/trust_region.py:9-19) -- cross-checked in tests against central finite
differences (float64) and the explicit Gauss-Newton/Fisher form.

NumPy half (pinned by golden vectors recorded from the reference's own CPOAgent.update_pi driven with
a fake session, tests/golden/g7_update_pi*.npz): the case selection, dual solve, step and backtracking
of policies/cpo_policy.py:153-300, restated in `update_pi`.
"""
import numpy as np
import torch

EPS = 1e-8  # utilities/utils.py:19
LOG_2PI = float(np.log(2 * np.pi))


def split_params(flat, obs_dim, act_dim, hidden=128):
    """flat [P] -> [W0,b0,W1,b1,W2,b2,log_std] views (creation order, ac_network.py:35-36)."""
    shapes = [(obs_dim, hidden), (hidden,), (hidden, hidden), (hidden,), (hidden, act_dim), (act_dim,), (act_dim,)]
    out, off = [], 0
    for s in shapes:
        n = int(np.prod(s))
        out.append(flat[off:off + n].reshape(s))
        off += n
    assert off == flat.shape[0]
    return out


def n_params(obs_dim, act_dim, hidden=128):
    return obs_dim * hidden + hidden + hidden * hidden + hidden + hidden * act_dim + 2 * act_dim


class PolicyGraph:
    """The symbols of policies/cpo_policy.py:522-559 as functions of a flat parameter vector."""

    def __init__(self, obs_dim, act_dim, batch, ent_reg=0.0, max_path_length=1, dtype=torch.float32, hidden=128):
        self.D, self.A, self.H, self.dtype = obs_dim, act_dim, hidden, dtype
        self.ent_reg, self.T = ent_reg, max_path_length
        t = lambda a: torch.as_tensor(np.asarray(a), dtype=dtype)
        # batch: dict(obs, act, adv, cadv, logp_old, cost, mu_old, log_std_old)
        self.b = {k: t(v) for k, v in batch.items()}

    def _mu(self, p):
        w0, b0, w1, b1, w2, b2, ls = split_params(p, self.D, self.A, self.H)
        h = torch.tanh(self.b["obs"] @ w0 + b0)           # ac_network.py:26-33
        h = torch.tanh(h @ w1 + b1)
        return h @ w2 + b2, ls

    def logp(self, p):
        mu, ls = self._mu(p)
        pre = -0.5 * (((self.b["act"] - mu) / (torch.exp(ls) + EPS)) ** 2 + 2 * ls + LOG_2PI)   # :46-48
        return pre.sum(dim=1)

    def d_kl(self, p):
        """gaussian_kl(mu, log_std, old_mu, old_log_std) -- ac_network.py:50-55,114."""
        mu, ls = self._mu(p)
        var0, var1 = torch.exp(2 * ls), torch.exp(2 * self.b["log_std_old"])
        pre = 0.5 * (((self.b["mu_old"] - mu) ** 2 + var0) / (var1 + EPS) - 1) + self.b["log_std_old"] - ls
        return pre.sum(dim=1).mean()

    def ent(self, p):
        _, ls = self._mu(p)
        return (ls + 0.5 * float(np.log(2 * np.pi * np.e))).sum()        # gaussian_entropy, :57-61

    def surr(self, p):
        ratio = torch.exp(self.logp(p) - self.b["logp_old"])              # cpo_policy.py:522
        return (ratio * self.b["adv"]).mean(), (ratio * self.b["cadv"]).mean()

    def pi_loss(self, p):
        return -(self.surr(p)[0] + self.ent_reg * self.ent(p))             # :540-543

    def surr_cost(self, p):
        return self.surr(p)[1]

    def cur_cret_avg(self):
        return self.b["cost"].mean() * self.T                             # :533

    # -- fetches used by update_pi -------------------------------------------------------------------
    def _param(self, flat):
        return torch.tensor(np.asarray(flat), dtype=self.dtype, requires_grad=True)

    def grads(self, flat):
        """flat_g, flat_b, pi_loss, surr_cost (trust_region.py:9-13; cpo_policy.py:549,555)."""
        p = self._param(flat)
        lo = self.pi_loss(p)
        g, = torch.autograd.grad(lo, p)
        p2 = self._param(flat)
        sc = self.surr_cost(p2)
        b, = torch.autograd.grad(sc, p2)
        return g.numpy(), b.numpy(), float(lo.detach()), float(sc.detach())

    def hvp(self, flat, v, damping=0.1):
        """hessian_vector_product(d_kl, pi_params) + damping * v (trust_region.py:15-19; cpo_policy.py:550-552)."""
        p = self._param(flat)
        kl = self.d_kl(p)
        g, = torch.autograd.grad(kl, p, create_graph=True)
        vv = torch.as_tensor(np.asarray(v), dtype=self.dtype)
        hv, = torch.autograd.grad((g * vv).sum(), p)
        return (hv + damping * vv).numpy()

    def evals(self, flat):
        """[d_kl, pi_loss, surr_cost] at trial parameters (cpo_policy.py:278-280)."""
        with torch.no_grad():
            p = torch.as_tensor(np.asarray(flat), dtype=self.dtype)
            return float(self.d_kl(p)), float(self.pi_loss(p)), float(self.surr_cost(p))

    def fisher_vp(self, flat, v, damping=0.1):
        """Explicit Gauss-Newton form: mean_n J^T diag(1/(var_old+eps)) J v on the MLP block,
        diag(mean_n 2 var/(var_old+eps)) on log_std.  Equals hvp() when mu_old == mu(theta)."""
        p = self._param(flat)
        vv = torch.as_tensor(np.asarray(v), dtype=self.dtype)
        mu, ls = self._mu(p)
        var1 = torch.exp(2 * self.b["log_std_old"]) + EPS
        _, jv = torch.autograd.functional.jvp(lambda q: self._mu(q)[0], (p.detach(),), (vv,))
        w = jv / var1 / mu.shape[0]
        out, = torch.autograd.grad(mu, p, grad_outputs=w)
        coef = (2 * torch.exp(2 * ls.detach()) / var1).mean(dim=0)
        out = out.clone()
        out[-self.A:] += coef * vv[-self.A:]
        return (out + damping * vv).numpy()


def cg(Ax, b, cg_iters=10):
    """utilities/trust_region.py:32-45."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rr = np.dot(r, r)
    for _ in range(cg_iters):
        z = Ax(p)
        alpha = rr / (np.dot(p, z) + EPS)
        x += alpha * p
        r -= alpha * z
        rr_new = np.dot(r, r)
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x


class AgentState:
    """The mutable scalars of CPOAgent (policies/cpo_policy.py:139-151)."""

    def __init__(self, max_path_length, constrained=True, learn_margin=True, damping_coeff=0.1,
                 backtrack_coeff=0.8, backtrack_iters=10):
        self.margin = 0
        self.margin_lr = 0.0001
        self.margin_discount = .9999
        self.max_path_length = max_path_length
        self.constrained, self.learn_margin = constrained, learn_margin
        self.damping_coeff, self.backtrack_coeff, self.backtrack_iters = damping_coeff, backtrack_coeff, backtrack_iters


def update_pi(agent, ops, old_params, target_kl, cost_lim, real_cost_buf):
    """CPOAgent.update_pi (policies/cpo_policy.py:153-300) at world size 1.

    ops: grads() -> (g, b, pi_l_old, surr_cost_old, cur_cret_avg); Hx(v) -> Hv (incl. damping);
         set_and_eval(params) -> (kl, pi_l, surr_cost).
    Returns (new_params, info dict with the logged Optim_* scalars, OptimCase, BacktrackIters, accepted).
    """
    f32 = np.float32
    g, b, pi_l_old, surr_cost_old, cur_cret_avg = ops["grads"]()
    g, b = np.asarray(g, f32), np.asarray(b, f32)
    pi_l_old, surr_cost_old, cur_cret_avg = f32(pi_l_old), f32(surr_cost_old), f32(cur_cret_avg)   # mpi_op casts
    Hx = lambda x: np.asarray(ops["Hx"](x), f32)
    rescale = 1 / agent.max_path_length
    c = (cur_cret_avg - cost_lim) * rescale                                  # :185
    if agent.learn_margin:                                                   # :188-196
        real_c = np.mean(real_cost_buf)
        agent.margin *= agent.margin_discount
        agent.margin += agent.margin_lr * (real_c - cost_lim) * rescale
        agent.margin = max(0, agent.margin)
    agent.margin = f32(agent.margin)                                          # mpi_avg -> float32 (:201)
    c += agent.margin
    v = cg(Hx, g)                                                            # :210-212
    approx_g = Hx(v)
    q = np.dot(v, approx_g)
    if np.dot(b, b) <= 1e-8 and c < 0 or not agent.constrained:              # :216-219
        w, r, s, A, B = 0, 0, 0, 0, 0
        optim_case = 4
    else:
        w = cg(Hx, b)                                                        # :222-226
        r = np.dot(w, approx_g)
        s = np.dot(w, Hx(w))
        A = q - r ** 2 / s
        B = 2 * target_kl - c ** 2 / s
        if c < 0 and B < 0:
            optim_case = 3
        elif c < 0 and B >= 0:
            optim_case = 2
        elif c >= 0 and B >= 0:
            optim_case = 1
        else:
            optim_case = 0
    if optim_case in [3, 4]:                                                 # :247-262
        lam = np.sqrt(q / (2 * target_kl))
        nu = 0
    elif optim_case in [1, 2]:
        LA, LB = [0, r / c], [r / c, np.inf]
        LA, LB = (LA, LB) if c < 0 else (LB, LA)
        proj = lambda x, L: max(L[0], min(L[1], x))
        lam_a = proj(np.sqrt(A / B), LA)
        lam_b = proj(np.sqrt(q / (2 * target_kl)), LB)
        f_a = lambda lam: -0.5 * (A / (lam + EPS) + B * lam) - r * c / (s + EPS)
        f_b = lambda lam: -0.5 * (q / (lam + EPS) + 2 * target_kl * lam)
        lam = lam_a if f_a(lam_a) >= f_b(lam_b) else lam_b
        nu = max(0, lam * c - r) / (s + EPS)
    else:
        lam = 0
        nu = np.sqrt(2 * target_kl / (s + EPS))
    x = (1. / (lam + EPS)) * (v + nu * w) if optim_case > 0 else nu * w        # :266
    info = dict(Optim_A=A, Optim_B=B, Optim_c=c, Optim_q=q, Optim_r=r, Optim_s=s, Optim_Lam=lam, Optim_Nu=nu,
                Margin=agent.margin, OptimCase=optim_case)
    old_params = np.asarray(old_params, f32)
    accepted, params = False, old_params
    for j in range(agent.backtrack_iters):                                   # :285-300
        trial = old_params - agent.backtrack_coeff ** j * x
        kl, pi_l_new, surr_cost_new = [f32(t) for t in ops["set_and_eval"](trial)]
        if (kl <= target_kl and (pi_l_new <= pi_l_old if optim_case > 1 else True) and
                surr_cost_new - surr_cost_old <= max(-c, 0)):
            info["BacktrackIters"] = j
            accepted, params = True, trial
            break
        if j == agent.backtrack_iters - 1:
            info["BacktrackIters"] = j
            ops["set_and_eval"](old_params - 0. * x)
            params = old_params - 0. * x
    info["accepted"] = accepted
    info["step"] = x
    return params, info
