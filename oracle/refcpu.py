"""CPU oracle for the CMBPO hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A NumPy restatement of the reference's arithmetic for the imagined rollout and the
CPO update.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this module; the product package
(``constrained-model-based-policy-optimization_amd/``) never does.

Pinning (see DESIGN.md "Oracle"):
  * the NumPy half (statics, average_dkl, FakeEnv.step, discount_cumsum, ModelBuffer,
    ModelSampler, cg, mpi_statistics_scalar, CPOAgent.update_pi) is checked against
    golden vectors produced by running the reference's own code in the build container
    (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``).
  * the TF half (ensemble / policy forward, losses, gradients, HVP) cannot be executed
    (TensorFlow 1.14 is not installable offline) and the reference ships no tests or
    fixtures for it: that half is "parity unpinned" by reference outputs and rests on
    this restatement plus analytic cross-checks (finite differences, explicit Fisher).

Every function cites the reference file:line (relative to the reference tree) it follows.
"""
import numpy as np

F32 = np.float32


# ----------------------------------------------------------------------------------------
# Ensemble forward (TF half)
# ----------------------------------------------------------------------------------------
def scaler_sigma(var):
    """max(sqrt(var), 1e-2) -- models/pens/utils.py:156,167,187."""
    return np.maximum(np.sqrt(np.asarray(var, F32)), F32(1e-2)).astype(F32)


def swish(x):
    """x * sigmoid(x) -- models/pens/fc.py:19."""
    return (x / (F32(1.0) + np.exp(-x))).astype(F32)


def ens_layers(x, weights, biases, act):
    """FC.compute_output_tensor chain -- models/pens/fc.py:74-95.

    x: [B, in] shared by all members (einsum 'ij,ajk->aik') for the first layer, then
    batched matmul.  weights[l]: [E, in_l, out_l]; biases[l]: [E, 1, out_l] or [E, out_l].
    The last layer has no activation (pe.py:183-186).
    """
    h = None
    n_layers = len(weights)
    for l, (w, b) in enumerate(zip(weights, biases)):
        w = np.asarray(w, F32)
        b = np.asarray(b, F32).reshape(w.shape[0], 1, w.shape[2])
        if l == 0:
            h = np.einsum("ij,ajk->aik", np.asarray(x, F32), w).astype(F32) + b
        else:
            h = np.matmul(h, w).astype(F32) + b
        if l < n_layers - 1:
            h = act(h)
    return h.astype(F32)


def ens_forward(x, weights, biases, scaler_in=None, scaler_out=None, act=swish):
    """PE.predict_ensemble (2-D input) -- models/pens/pe.py:688-697, 789-838.

    Returns (mean, var), each [E, B, out].  No max/min-logvar clamp on this path.
    """
    x = np.asarray(x, F32)
    if scaler_in is not None:
        mu, var = scaler_in
        x = ((x - np.asarray(mu, F32).reshape(1, -1)) / scaler_sigma(var).reshape(1, -1)).astype(F32)
    o = ens_layers(x, weights, biases, act)
    half = o.shape[-1] // 2
    mean, logvar = o[..., :half], o[..., half:]
    if scaler_out is not None:
        mu, var = scaler_out
        sig = scaler_sigma(var).reshape(1, 1, -1)
        mean = (sig * mean + np.asarray(mu, F32).reshape(1, 1, -1)).astype(F32)
        logvar = (F32(2.0) * np.log(sig) + logvar).astype(F32)
    return mean.astype(F32), np.exp(logvar).astype(F32)


def ens_predict_mean(x, weights, biases, scaler_in=None, scaler_out=None, act=swish):
    """PE.predict for a deterministic ('MSE') ensemble: mean over ALL members.

    models/pens/pe.py:338-343 (reduce_mean over axis 0), :648-669, :815-823.  Returns [B, out].
    """
    x = np.asarray(x, F32)
    if scaler_in is not None:
        mu, var = scaler_in
        x = ((x - np.asarray(mu, F32).reshape(1, -1)) / scaler_sigma(var).reshape(1, -1)).astype(F32)
    o = ens_layers(x, weights, biases, act)
    if scaler_out is not None:
        mu, var = scaler_out
        o = (scaler_sigma(var).reshape(1, 1, -1) * o + np.asarray(mu, F32).reshape(1, 1, -1)).astype(F32)
    return (o.sum(axis=0, dtype=F32) / F32(o.shape[0])).astype(F32)


# ----------------------------------------------------------------------------------------
# Gaussian policy (TF half)
# ----------------------------------------------------------------------------------------
LOG_2PI = np.log(2 * np.pi)


def policy_mu(obs, params):
    """tanh MLP -- network/ac_network.py:26-33 (tf.layers.dense: y = x W + b, W[in,out]).

    params = [W0, b0, W1, b1, W2, b2, log_std] in creation order (get_vars('pi'), :35-36).
    """
    h = np.asarray(obs, F32)
    w0, b0, w1, b1, w2, b2 = [np.asarray(p, F32) for p in params[:6]]
    h = np.tanh(h @ w0 + b0).astype(F32)
    h = np.tanh(h @ w1 + b1).astype(F32)
    return (h @ w2 + b2).astype(F32)


def gaussian_likelihood(x, mu, log_std):
    """network/ac_network.py:46-48 (EPS = 1e-8, utilities/utils.py:19)."""
    pre = F32(-0.5) * (((x - mu) / (np.exp(log_std) + F32(1e-8))) ** 2 + F32(2) * log_std + F32(LOG_2PI))
    return pre.sum(axis=1, dtype=F32).astype(F32)


def policy_forward(obs, params, eps):
    """mlp_gaussian_policy forward -- network/ac_network.py:99-123.

    eps replaces tf.random_normal (:109).  Returns dict(pi, logp_pi, mu, log_std[B,A]).
    """
    mu = policy_mu(obs, params)
    log_std = np.asarray(params[6], F32)
    std = np.exp(log_std).astype(F32)
    pi = (mu + np.asarray(eps, F32) * std).astype(F32)
    logp_pi = gaussian_likelihood(pi, mu, log_std)
    ls_b = np.tensordot(np.ones(mu.shape[0], F32), log_std, axes=0).astype(F32)   # :119
    return dict(pi=pi, logp_pi=logp_pi, mu=mu, log_std=ls_b)


# ----------------------------------------------------------------------------------------
# Ensemble disagreement (NumPy half)
# ----------------------------------------------------------------------------------------
DKL_EPS = 1e-10  # models/pens/utils.py:7


def gaussian_kl_np(mu0, log_std0, mu1, log_std1):
    """Element-wise KL(N0 || N1), clipped to [0, 1/EPS] -- models/pens/utils.py:15-26."""
    v0 = np.exp(2 * log_std0)
    v1 = np.exp(2 * log_std1)
    kl = 0.5 * (((mu1 - mu0) ** 2 + v0) / (v1 + DKL_EPS) - 1) + log_std1 - log_std0
    return np.clip(kl, 0, 1 / DKL_EPS)


def average_dkl(mu, std):
    """Mean KL over all ordered member pairs -- models/pens/utils.py:30-57.

    mu, std: [E, ...]; output drops axis 0.  Accumulation order (i outer, j inner) and the
    i == j terms are kept, as is the divisor E(E-1) + EPS.
    """
    log_std = np.clip(np.log(std), -100, 1e8)
    E = len(mu)
    total = None
    for i in range(E):
        for j in range(E):
            term = gaussian_kl_np(mu[i], log_std[i], mu[j], log_std[j])
            total = term if total is None else total + term
    return total / (E * (E - 1) + DKL_EPS)


# ----------------------------------------------------------------------------------------
# Static termination / cost rules (NumPy half)
# ----------------------------------------------------------------------------------------
def no_done(obs, act, next_obs):
    """models/statics.py:3-8."""
    return np.zeros(obs.shape[:-1] + (1,), dtype=bool)


def hcs_cost_f(obs, act, next_obs):
    """models/statics.py:10-15: cost = |next_obs[-1] * 10| < 2."""
    return (np.abs(next_obs[..., -1] * 10) < 2.0).astype(F32)[..., None]


def _ant_notdone(next_obs):
    # models/statics.py:20-27 / :36-46.  `a * b * c * z_rot >= -0.7` parses as
    # ((a*b*c) * z_rot) >= -0.7 : the product of the three bools gates z_rot.
    z = next_obs[..., 0]
    q = next_obs[..., 1:5]
    z_rot = 1 - 2 * (q[..., 1] ** 2 + q[..., 2] ** 2)
    gate = np.isfinite(next_obs).all(axis=-1) * (z >= 0.2) * (z <= 1.0)
    return gate * z_rot >= -0.7


def antsafe_term_fn(obs, act, next_obs):
    """models/statics.py:17-31."""
    return (~_ant_notdone(next_obs))[..., None]


def antsafe_c_fn(obs, act, next_obs):
    """models/statics.py:33-53."""
    obj = np.any(np.abs(next_obs[..., -1:]) > 3.2, axis=-1)[..., None] * 1.0
    done = (~_ant_notdone(next_obs))[..., None] * 1.0
    return np.clip(done + obj, 0, 1)


TERMS_BY_TASK = {"default": no_done, "HalfCheetah-v2": no_done, "HalfCheetahSafe-v2": no_done,
                 "AntSafe-v2": antsafe_term_fn}                                     # statics.py:56-61
COST_BY_TASK = {"HalfCheetahSafe-v2": hcs_cost_f, "AntSafe-v2": antsafe_c_fn}       # statics.py:67-69


# ----------------------------------------------------------------------------------------
# FakeEnv.step (NumPy half), given the ensemble's (mean, var)
# ----------------------------------------------------------------------------------------
def fake_env_step(obs, act, pred_mean, pred_var, model_inds, task):
    """models/fake_env.py:104-172, 2-D obs, deterministic=True, predicts_delta/rew=True, no learned cost.

    model_inds replaces np.random.choice(elite_inds, n) (:174-178).
    Returns next_obs[n,obs], r[n,1], terms[n,1] bool, info dict.
    """
    obs_dim = obs.shape[-1]
    pred_std = np.sqrt(pred_var)
    next_all = pred_mean[..., :obs_dim]
    ep_var = np.var(next_all, axis=0)
    dkl_path = np.mean(average_dkl(next_all, pred_std[..., :obs_dim]), axis=-1)
    dkl_mean = np.mean(dkl_path)
    rows = np.arange(obs.shape[0])
    next_obs = next_all[model_inds, rows]
    next_obs = next_obs + obs
    terms = TERMS_BY_TASK.get(task, no_done)(obs, act, next_obs)
    if task in COST_BY_TASK:
        c = COST_BY_TASK[task](obs, act, next_obs)
    else:
        c = np.zeros_like(terms)
    r = pred_mean[..., -1:][model_inds, rows]
    info = dict(ensemble_dkl_mean=dkl_mean, ensemble_dkl_path=dkl_path, ensemble_ep_var=ep_var,
                rew=r, cost=c)
    return next_obs, r, terms, info


# ----------------------------------------------------------------------------------------
# GAE / statistics / CG (NumPy half)
# ----------------------------------------------------------------------------------------
def discount_cumsum(x, discount, lam):
    """utilities/utils.py:184-188, un-weighted branch, last axis.

    scipy.signal.lfilter([1], [1, -discount*lam]) on the reversed rows: a float64 first-order
    recurrence y_t = x_t + (discount*lam) * y_{t+1}, product and sum rounded separately.
    Input keeps its dtype for the deltas; output is float64 (lfilter promotes float32 input).
    """
    x = np.asarray(x)
    if x.size == 0:
        return np.array(x)
    c = float(discount * lam)
    y = np.zeros(x.shape, dtype=np.float64)
    acc = np.zeros(x.shape[:-1], dtype=np.float64)
    for t in range(x.shape[-1] - 1, -1, -1):
        acc = x[..., t].astype(np.float64) + c * acc
        y[..., t] = acc
    return y


def mpi_statistics_scalar(x):
    """utilities/mpi_tools.py:71-87 at world size 1: float32 two-pass mean / std."""
    x = np.array(x, dtype=F32)
    s = np.asarray([np.sum(x), len(x)], dtype=F32)
    mean = s[0] / s[1]
    sq = np.asarray(np.sum((x - mean) ** 2), dtype=F32)
    return mean, np.sqrt(sq / s[1])


def cg(Ax, b, cg_iters=10):
    """utilities/trust_region.py:32-45: plain CG from x = 0, fixed iteration count, EPS in alpha."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rr = np.dot(r, r)
    for _ in range(cg_iters):
        z = Ax(p)
        alpha = rr / (np.dot(p, z) + 1e-8)
        x += alpha * p
        r -= alpha * z
        rr_new = np.dot(r, r)
        p = r + (rr_new / rr) * p
        rr = rr_new
    return x


def gae_rows(rew, val, last_val, gamma, lam):
    """buffers/modelbuffer.py:163-170 for rows [n, L]: returns (adv float32, ret float32).

    dtype follows NumPy promotion in the reference: float32 rows with a float32 bootstrap stay
    float32 for the deltas; a float64 bootstrap (np.zeros, model_sampler.py:404) promotes them.
    """
    last_val = np.asarray(last_val)
    rews = np.append(rew, last_val[..., None], axis=-1)
    vals = np.append(val, last_val[..., None], axis=-1)
    deltas = rews[..., :-1] + gamma * vals[..., 1:] - vals[..., :-1]
    adv = discount_cumsum(deltas, gamma, lam).astype(F32)
    return adv, (adv + val).astype(F32)


# ----------------------------------------------------------------------------------------
# ModelSampler + ModelBuffer, restated on fixed branch slots (NumPy half)
# ----------------------------------------------------------------------------------------
class RolloutOracle:
    """samplers/model_sampler.py:203-444 + buffers/modelbuffer.py:53-226, without array compaction.

    model(x)->(mean,var)[E,n,out]; policy(obs, eps)->dict(pi, logp_pi, mu, log_std); v(obs), vc(obs)->[n].
    Random draws are inputs: eps[step][n_alive, act], inds[step][n_alive] in alive (index) order.
    """

    def __init__(self, model, policy, v, vc, task, obs_dim, act_dim, max_path_length, mode, dkl_lim,
                 gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5):
        self.model, self.policy, self.v, self.vc, self.task = model, policy, v, vc, task
        self.D, self.A, self.T = obs_dim, act_dim, max_path_length
        self.mode, self.dkl_lim = mode, dkl_lim
        self.g, self.l, self.cg_, self.cl = gamma, lam, cost_gamma, cost_lam

    def reset(self, start):
        B, T, D, A = start.shape[0], self.T, self.D, self.A
        self.B = B
        self.cur = np.array(start, dtype=F32)
        self.alive = np.ones(B, dtype=bool)
        self.len = np.zeros(B, dtype=np.int64)
        z = lambda *s: np.zeros(s, dtype=F32)
        self.buf = dict(obs=z(B, T, D), act=z(B, T, A), mu=z(B, T, A), log_std=z(B, T, A), rew=z(B, T),
                        val=z(B, T), cost=z(B, T), cval=z(B, T), logp=z(B, T), adv=z(B, T), ret=z(B, T),
                        cadv=z(B, T), cret=z(B, T))
        self.ptr = 0
        self.dkl_acc = np.zeros(B)
        self.path_ret = np.zeros(B)
        self.path_cost = np.zeros(B)
        self.tot = dict(samples=0, cost=0.0, rew=0.0, Vs=0.0, CVs=0.0, dkl=0.0, dyn_ep_var=0.0, max_dkl=0.0,
                        max_path_return=0.0)
        self.n_episodes = 0

    def _finish(self, slots, last_val, last_cval):
        """finish_path_multiple (modelbuffer.py:138-182) for the given slots."""
        if len(slots) == 0:
            return
        L = self.ptr
        if L > 0:
            b = self.buf
            adv, ret = gae_rows(b["rew"][slots, :L], b["val"][slots, :L], last_val, self.g, self.l)
            cadv, cret = gae_rows(b["cost"][slots, :L], b["cval"][slots, :L], last_cval, self.cg_, self.cl)
            b["adv"][slots, :L], b["ret"][slots, :L] = adv, ret
            b["cadv"][slots, :L], b["cret"][slots, :L] = cadv, cret
        self.alive[slots] = False

    def sample(self, eps, inds, max_samples=None):
        """One step (model_sampler.py:239-375).  Returns alive_ratio."""
        self.n_episodes += 1
        idx = np.flatnonzero(self.alive)
        obs = self.cur[idx]
        out = self.policy(obs, eps)
        a, v_t, vc_t = out["pi"], self.v(obs), self.vc(obs)
        mean, var = self.model(np.concatenate([obs, a], axis=-1))
        nobs, r, terms, info = fake_env_step(obs, a, mean, var, inds, self.task)
        r, terms = r[:, 0], terms[:, 0]
        c = np.squeeze(info["cost"])
        dkl_path, ep_var = info["ensemble_dkl_path"], info["ensemble_ep_var"]
        dkl_mean = info["ensemble_dkl_mean"]
        # uncertainty test on accumulated + new DKL, before storing (:275-279)
        if self.mode == "uncertainty":
            unc = self.dkl_acc[idx] + dkl_path >= self.dkl_lim
        else:
            unc = np.zeros(len(idx), dtype=bool)
        # budget: first n surviving rows by index (:282-287)
        if max_samples:
            n = self.tot["samples"] + len(idx) - unc.sum()
            n = max(n - max_samples, 0)
            surv = np.flatnonzero(~unc)
            unc = unc.copy()
            unc[surv[:n]] = True
        # finish with V/VC bootstrap of the PRE-step obs (:290, :401-407).  The reference calls the
        # critics again on the subset (BLAS results can differ in the last bit with the batch shape).
        if unc.any():
            self._finish(idx[unc], self.v(obs[unc]), self.vc(obs[unc]))
        keep = ~unc
        if not keep.any():
            return 0.0
        k = idx[keep]
        self.tot["samples"] += len(k)
        self.tot["cost"] += c[keep].sum()
        self.tot["rew"] += r[keep].sum()
        self.path_ret[k] += r[keep]
        self.path_cost[k] += c[keep]
        self.tot["dyn_ep_var"] += ep_var[keep].sum()
        self.tot["Vs"] += v_t[keep].sum()
        self.tot["CVs"] += vc_t[keep].sum()
        self.tot["dkl"] += dkl_mean * len(k)
        self.tot["max_dkl"] = max(self.tot["max_dkl"], np.max(dkl_path[keep]))
        self.dkl_acc[k] += dkl_path[keep]
        self.tot["max_path_return"] = max(self.tot["max_path_return"], np.max(self.path_ret))
        # store at column ptr (modelbuffer.py:114-135)
        b, p = self.buf, self.ptr
        b["obs"][k, p], b["act"][k, p] = obs[keep], a[keep]
        b["rew"][k, p], b["val"][k, p], b["cost"][k, p], b["cval"][k, p] = r[keep], v_t[keep], c[keep], vc_t[keep]
        b["logp"][k, p], b["mu"][k, p], b["log_std"][k, p] = out["logp_pi"][keep], out["mu"][keep], out["log_std"][keep]
        self.len[k] = p + 1
        self.ptr += 1
        self.cur[k] = nobs[keep]
        # horizon (:350-356): everything left finishes with V/VC(next_obs)
        if self.ptr >= self.T - 1:
            self._finish(k, self.v(self.cur[k]), self.vc(self.cur[k]))
            return 0.0
        # env terminals (:357-367): last_val = float64 zeros, last_cval = VC(next_obs)
        tk = k[terms[keep]]
        if len(tk):
            self._finish(tk, np.zeros(len(tk)), self.vc(self.cur[tk]))
        if not self.alive.any():
            return 0.0
        return self.alive.sum() / self.B

    def finish_all(self):
        idx = np.flatnonzero(self.alive)
        if len(idx):
            self._finish(idx, self.v(self.cur[idx]), self.vc(self.cur[idx]))

    def get(self):
        """modelbuffer.py:184-226."""
        assert not self.alive.any()
        mask = np.arange(self.T)[None, :] < self.len[:, None]
        b = self.buf
        adv, cadv = b["adv"].copy(), b["cadv"].copy()
        if mask.sum() > 0:
            m, s = mpi_statistics_scalar(adv[mask].flatten())
            adv[mask] = (adv[mask] - m) / (s + 1e-8)
            cm, _ = mpi_statistics_scalar(cadv[mask].flatten())
            cadv[mask] -= cm
            ret_mean, cret_mean = b["ret"][mask].mean(), b["cret"][mask].mean()
        else:
            ret_mean = cret_mean = 0
        res = [b["obs"], b["act"], adv, cadv, b["ret"], b["cret"], b["logp"], b["val"], b["cval"], b["cost"],
               b["log_std"], b["mu"]]
        res = [x[mask] for x in res]
        return res, dict(poolm_batch_size=int(mask.sum()), poolm_ret_mean=ret_mean, poolm_cret_mean=cret_mean)

    def diagnostics(self):
        """model_sampler.py:89-133 (the keys that carry information)."""
        t, e = self.tot, 1e-8
        return {"msampler/samples_added": t["samples"], "msampler/rollout_H_max": self.n_episodes,
                "msampler/rollout_H_mean": t["samples"] / (self.B + e),
                "msampler/dyn_var_perstep": t["dyn_ep_var"] / (t["samples"] + e),
                "msampler/cost_rate": self.path_cost.sum() / (t["samples"] + e),
                "msampler/rew_rate": self.path_ret.sum() / (t["samples"] + e),
                "msampler/v_mean": t["Vs"] / (t["samples"] + e), "msampler/cv_mean": t["CVs"] / (t["samples"] + e),
                "msampler/ens_DKL": t["dkl"] / (t["samples"] + e),
                "msampler/max_path_return": t["max_path_return"], "msampler/max_dkl": t["max_dkl"]}
