"""CPU oracle for ensemble training -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

TF half ("parity unpinned": TensorFlow 1.14 cannot run here, the reference has no fixtures): the training graph of
``models/pens/pe.py`` restated with torch-CPU autograd --
  * 3-D input forward without output scaling (``_compile_outputs(inputs, ret_log_var=True)``, :789-838; batched
    ``tf.matmul`` per member, ``models/pens/fc.py:89-95``);
  * MSPE train loss (:921-973): per-member mean squared error + variance-fit term with the stop-gradient ratio
    0.05 * mean(mse) / mean((var - mse)^2) taken over ALL members, + 0.05 * mean(log_var^2);
  * NLL train loss (:840-919 with inc_var_loss=True; the class default of PE): per-member mean 0.5 exp(-lv)(m - t)^2
    + mean 0.5 lv; its max_logvar / min_logvar variables never enter the network (:789-838) and only drift under the
    constant gradient of their regulariser (:263) -- :func:`logvar_bounds`;
  * MSE train loss for deterministic ensembles (critics; ``_nll_loss(inc_var_loss=False)``, :840-919);
  * weight decay ``decay_l * tf.nn.l2_loss(W_l)`` per layer (``models/pens/fc.py:167-168``, ``pe_factory.py:50-56``);
  * ``self.loss`` used for holdout / elite ranking = per-member 0.5 * mean((mean - target)^2) (:264, :840-919);
  * tf.train.AdamOptimizer (beta1 .9, beta2 .999, eps 1e-8): lr_t = lr sqrt(1-b2^t)/(1-b1^t), m += (g-m)(1-b1),
    v += (g^2-v)(1-b2), var -= lr_t m / (sqrt(v) + eps).
NumPy half (pinned by golden G9 recorded from the reference's own PE.train control flow driven with a fake session):
  the epoch / bootstrap-index / holdout / early-stopping / elite-selection logic of ``PE.train`` (:457-646) and the
  running scaler fit (``models/pens/utils.py:119-138,220-231``), restated in :func:`train_loop`.
"""
import numpy as np
import torch

F32 = np.float32


def sigma_of(var):
    return torch.clamp(torch.sqrt(var), min=1e-2)


def forward_raw(x, ws, bs, scaler_in=None):
    """x [E,B,in] -> raw network output [E,B,O] (no output scaler, no exp): pe.py:803-812 with 3-D inputs."""
    h = x
    if scaler_in is not None:
        mu, var = scaler_in
        h = (h - mu.reshape(1, 1, -1)) / sigma_of(var).reshape(1, 1, -1)
    for l, (w, b) in enumerate(zip(ws, bs)):
        h = torch.matmul(h, w) + b.reshape(w.shape[0], 1, w.shape[2])
        if l < len(ws) - 1:
            h = h * torch.sigmoid(h)
    return h


def scale_targets(t, scaler_out):
    if scaler_out is None:
        return t
    mu, var = scaler_out
    return (t - mu.reshape(1, 1, -1)) / sigma_of(var).reshape(1, 1, -1)


def mspe_losses(o, t_scaled):
    """pe.py:921-973 -> per-member total_losses [E]."""
    out = o.shape[-1] // 2
    mean, log_var = o[..., :out], o[..., out:]
    var_pred = torch.exp(log_var)
    mse_logit = (mean - t_scaled) ** 2
    var_logit = (var_pred - mse_logit.detach()) ** 2
    ratio = 0.05 * mse_logit.detach().mean() / var_logit.detach().mean()
    mse_losses = mse_logit.mean(dim=-1).mean(dim=-1)
    var_losses = (var_logit * ratio).mean(dim=-1).mean(dim=-1)
    var_reg = 0.05 * (log_var ** 2).mean()
    return mse_losses + var_losses + var_reg


def mse_losses(o, t_scaled):
    """_nll_loss(inc_var_loss=False) on a deterministic head, and `self.loss` on the mean head: pe.py:911-919."""
    return (0.5 * (o - t_scaled) ** 2).mean(dim=-1).mean(dim=-1)


def nll_losses(o, t_scaled):
    """_nll_loss(inc_var_loss=True), pe.py:840-919 (no clipping, no weights) -> per-member total_losses [E]."""
    out = o.shape[-1] // 2
    mean, log_var = o[..., :out], o[..., out:]
    inv_var = torch.exp(-log_var)
    var_losses = (0.5 * log_var).mean(dim=-1).mean(dim=-1)
    mse_losses_ = (0.5 * inv_var * (mean - t_scaled) ** 2).mean(dim=-1).mean(dim=-1)
    return mse_losses_ + var_losses


def logvar_bounds(steps, lr, max0, min0, b1=0.9, b2=0.999, eps=1e-8):
    """max_logvar / min_logvar of an 'NLL' model after `steps` Adam steps (pe.py:198-209,263): they never enter the
    network (_compile_outputs, :789-838), their gradient is the constant +-0.01 of the regulariser, so their
    trajectory is that of tf.train.AdamOptimizer under a constant gradient."""
    hi, lo = np.array(max0, np.float64).copy(), np.array(min0, np.float64).copy()
    for c, var in ((0.01, hi), (-0.01, lo)):
        m = v = 0.0
        for t in range(1, int(steps) + 1):
            m += (c - m) * (1 - b1)
            v += (c * c - v) * (1 - b2)
            var -= lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t) * m / (np.sqrt(v) + eps)
    return hi.astype(F32), lo.astype(F32)


def holdout_losses(o, t_scaled, probabilistic):
    mean = o[..., : o.shape[-1] // 2] if probabilistic else o
    return mse_losses(mean, t_scaled)


def train_loss(ws, bs, x, t, loss_type, decays, scaler_in=None, scaler_out=None):
    """The scalar the optimizer minimises: sum of per-member losses + decays (pe.py:252-274)."""
    o = forward_raw(x, ws, bs, scaler_in)
    ts = scale_targets(t, scaler_out)
    per_member = {"MSPE": mspe_losses, "NLL": nll_losses, "MSE": mse_losses}[loss_type](o, ts)
    loss = per_member.sum()
    for w, d in zip(ws, decays):
        loss = loss + d * 0.5 * (w ** 2).sum()
    return loss


class AdamTF:
    def __init__(self, params, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
        self.lr, self.b1, self.b2, self.eps, self.t = lr, b1, b2, eps, 0
        self.m = [torch.zeros_like(p) for p in params]
        self.v = [torch.zeros_like(p) for p in params]

    def step(self, params, grads):
        self.t += 1
        lr_t = self.lr * np.sqrt(1 - self.b2 ** self.t) / (1 - self.b1 ** self.t)
        out = []
        for p, g, m, v in zip(params, grads, self.m, self.v):
            m += (g - m) * (1 - self.b1)
            v += (g * g - v) * (1 - self.b2)
            out.append(p - lr_t * m / (torch.sqrt(v) + self.eps))
        return out


class EnsembleTrainer:
    """One optimisation state: weights [W0,W1,W2], biases, Adam moments; step(x, t) == sess.run(train_op)."""

    def __init__(self, ws, bs, loss_type="MSPE", decays=(2.5e-7, 5e-7, 1e-6), lr=1e-3, dtype=torch.float32):
        self.dtype = dtype
        self.ws = [torch.tensor(np.asarray(w), dtype=dtype) for w in ws]
        self.bs = [torch.tensor(np.asarray(b), dtype=dtype).reshape(w.shape[0], 1, w.shape[2]) for b, w in zip(bs, ws)]
        self.loss_type, self.decays = loss_type, decays
        self.opt = AdamTF(self.ws + self.bs, lr=lr)
        self.scaler_in = self.scaler_out = None

    def set_scalers(self, scaler_in, scaler_out):
        t = lambda a: torch.tensor(np.asarray(a), dtype=self.dtype).reshape(-1)
        self.scaler_in = None if scaler_in is None else (t(scaler_in[0]), t(scaler_in[1]))
        self.scaler_out = None if scaler_out is None else (t(scaler_out[0]), t(scaler_out[1]))

    def grads(self, x, t):
        ps = [p.clone().requires_grad_(True) for p in self.ws + self.bs]
        n = len(self.ws)
        loss = train_loss(ps[:n], ps[n:], torch.as_tensor(x, dtype=self.dtype), torch.as_tensor(t, dtype=self.dtype),
                          self.loss_type, self.decays, self.scaler_in, self.scaler_out)
        gs = torch.autograd.grad(loss, ps)
        return float(loss.detach()), gs

    def step(self, x, t):
        loss, gs = self.grads(x, t)
        new = self.opt.step(self.ws + self.bs, list(gs))
        n = len(self.ws)
        self.ws, self.bs = new[:n], new[n:]
        return loss

    def losses(self, x, t):
        """self.loss (pe.py:264): per-member 0.5 * mean (mean - target_scaled)^2, used for holdout / elites."""
        with torch.no_grad():
            o = forward_raw(torch.as_tensor(x, dtype=self.dtype), self.ws, self.bs, self.scaler_in)
            ts = scale_targets(torch.as_tensor(t, dtype=self.dtype), self.scaler_out)
            return holdout_losses(o, ts, self.loss_type in ("MSPE", "NLL")).numpy()


# --------------------------------------------------------------------------------------------------------
# NumPy half: scaler fit and the control flow of PE.train
# --------------------------------------------------------------------------------------------------------
class RunningScaler:
    """TensorStandardScaler.fit / running_mean_var_from_batch -- models/pens/utils.py:119-138,220-231."""

    def __init__(self, dim):
        self.count, self.mu, self.var = 0, np.zeros([1, dim]), np.ones([1, dim])

    def fit(self, data):
        n = data.shape[0]
        b_mu = np.mean(data, axis=0, keepdims=True)
        b_var = np.var(data, axis=0, keepdims=True)
        delta = b_mu - self.mu
        tot = self.count + n
        new_mu = self.mu + delta * n / tot
        m2 = self.var * self.count + b_var * n + np.square(delta) * self.count * n / tot
        self.mu, self.var, self.count = new_mu, m2 / tot, tot


def train_loop(ops, n_rows, num_nets, num_elites, rng, batch_size=32, max_epochs=None, max_epochs_since_update=5,
               min_epoch_before_break=0, holdout_ratio=0.0, max_logging=5000, max_grad_updates=None):
    """Control flow of PE.train (pe.py:480-646) with the numerics behind `ops`:
         ops.fit_scalers(train_rows) ; ops.train_step(batch_idxs[E,b]) ; ops.holdout_losses(holdout_rows) -> [E]
       `rng` provides permutation / randint / uniform in the order the reference draws them from np.random.
       Returns (elite indices, final holdout losses, epochs run, gradient updates)."""
    num_holdout = min(int(n_rows * holdout_ratio), max_logging)
    permutation = rng.permutation(n_rows)
    train_rows, holdout_rows = permutation[num_holdout:], permutation[:num_holdout]
    ops.fit_scalers(train_rows)
    n = len(train_rows)
    idxs = rng.randint(n, size=[num_nets, n])
    snapshots = {i: (None, 1e10) for i in range(num_nets)}
    epochs_since_update, grad_updates, epoch = 0, 0, 0
    import itertools
    for epoch in (range(max_epochs) if max_epochs else itertools.count()):
        for bn in range(int(np.ceil(idxs.shape[-1] / batch_size))):
            ops.train_step(train_rows[idxs[:, bn * batch_size:(bn + 1) * batch_size]])
            grad_updates += 1
        order = np.argsort(rng.uniform(size=idxs.shape), axis=-1)             # shuffle_rows
        idxs = idxs[np.arange(idxs.shape[0])[:, None], order]
        break_train = False
        if holdout_ratio >= 1e-12:
            hl = ops.holdout_losses(holdout_rows)
            updated = False
            for i in range(len(hl)):
                _, best = snapshots[i]
                if (best - hl[i]) / best > 0.01:
                    snapshots[i] = (epoch, hl[i])
                    updated = True
            epochs_since_update = 0 if updated else epochs_since_update + 1
            break_train = epochs_since_update > max_epochs_since_update
        if (break_train and epoch > min_epoch_before_break) or (max_grad_updates and grad_updates > max_grad_updates):
            break
    final = ops.holdout_losses(holdout_rows)
    elites = np.argsort(final)[:num_elites].tolist()
    return elites, final, epoch + 1, grad_updates
