"""CPU: the training oracle (oracle/reftrain.py) and the host control flow of PE.train against golden G9.

G9 (tests/golden/g9_pe_train.npz) was recorded from the reference's own ``PE.train`` / ``_save_best`` /
``_end_train`` / ``TensorStandardScaler.fit`` driven by a stand-in session (tests/golden/make_golden.py
``gen_pe_train``): the rows every train_op was fed, the scripted holdout losses it was answered with, the
resulting elites, validation loss and scaler moments.  The TF graph itself is "parity unpinned"; its torch
restatement is cross-checked here against closed forms (the output deltas the HIP loss kernel implements, Adam).
"""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import reftrain  # noqa: E402

G9 = os.path.join(HERE, "golden", "g9_pe_train.npz")
CASES = ["early_stop", "max_epochs", "grad_updates", "max_logging"]


def _case(name):
    z = np.load(G9)
    pre = name + "/"
    c = {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}
    kw = {k[3:]: (v.item()) for k, v in c.items() if k.startswith("kw_")}
    return c, kw


class _ScriptedOps:
    def __init__(self, c):
        self.c, self.steps, self.k = c, [], 0
        self.fit_rows = None

    def fit_scalers(self, rows):
        self.fit_rows = rows

    def train_step(self, rows):
        self.steps.append(np.asarray(rows, np.int32))

    def holdout_losses(self, rows):
        np.testing.assert_array_equal(rows, self.c["holdout_rows"])
        out = self.c["script"][self.k]
        self.k += 1
        return out


@pytest.mark.parametrize("name", CASES)
def test_oracle_train_loop_matches_reference_control_flow(name):
    c, kw = _case(name)
    ops = _ScriptedOps(c)
    elites, final, epochs, updates = reftrain.train_loop(
        ops, int(c["n"]), int(c["E"]), int(c["num_elites"]), np.random.RandomState(int(c["seed"])),
        batch_size=int(c["batch_size"]), **kw)
    widths = np.array([s.shape[1] for s in ops.steps], np.int32)
    np.testing.assert_array_equal(widths, c["step_widths"])
    np.testing.assert_array_equal(np.concatenate([s.reshape(-1) for s in ops.steps]), c["step_rows"])
    assert ops.k == c["script"].shape[0]
    np.testing.assert_array_equal(np.asarray(elites, np.int32), c["elites"])
    np.testing.assert_allclose(np.sort(final)[:int(c["num_elites"])].mean(), float(c["val_loss"]), rtol=1e-12)
    assert updates == len(c["step_widths"])


@pytest.mark.parametrize("name", CASES)
def test_product_train_control_matches_reference_control_flow(name):
    """The host half of cmbpo_amd.pens.PE.train (TrainControl) with recording hooks instead of the HIP trainer."""
    from cmbpo_amd.pens import TrainControl
    c, kw = _case(name)

    class Rec(TrainControl):
        num_nets, num_elites, name = int(c["E"]), int(c["num_elites"]), "G9"

        def __init__(self):
            self.steps, self.k = [], 0

        def _begin_train(self, inputs, targets, train_rows, holdout_rows, batch_size):
            np.testing.assert_array_equal(holdout_rows, c["holdout_rows"])
            self.train_rows, self.bs = train_rows, batch_size

        def _begin_epoch(self, idxs):
            self.gidx = self.train_rows[idxs]

        def _train_batch(self, bn, rows):
            blk = self.gidx[:, bn * self.bs: bn * self.bs + rows]
            assert blk.shape[1] == rows
            self.steps.append(blk.astype(np.int32))

        def _holdout_losses(self):
            out = c["script"][self.k]
            self.k += 1
            return out

        def _finish_train(self):
            pass

    r = Rec()
    out = r.train(c["inputs"], c["targets"], batch_size=int(c["batch_size"]), rng=np.random.RandomState(int(c["seed"])), **kw)
    np.testing.assert_array_equal(np.array([s.shape[1] for s in r.steps], np.int32), c["step_widths"])
    np.testing.assert_array_equal(np.concatenate([s.reshape(-1) for s in r.steps]), c["step_rows"])
    assert r.k == c["script"].shape[0]
    np.testing.assert_array_equal(np.asarray(r._model_inds, np.int32), c["elites"])
    np.testing.assert_allclose(out["G9/val_loss"], float(c["val_loss"]), rtol=1e-12)
    assert r.train_grad_updates == len(c["step_widths"])


@pytest.mark.parametrize("name", CASES)
def test_scaler_fit_matches_reference(name):
    from cmbpo_amd.pens import _CachedScaler
    c, kw = _case(name)
    n = int(c["n"])
    perm = np.random.RandomState(int(c["seed"])).permutation(n)
    train_rows = perm[len(c["holdout_rows"]):]
    x, t = c["inputs"][train_rows], c["targets"][train_rows]
    for cls in (_CachedScaler, None):
        if cls is None:                      # the oracle's float64 running scaler, rounded like the TF variables
            s_in, s_out = reftrain.RunningScaler(x.shape[1]), reftrain.RunningScaler(t.shape[1])
            s_in.fit(x); s_out.fit(t)
            np.testing.assert_allclose(s_in.mu.astype(np.float32), c["in_mu"], rtol=1e-6, atol=1e-7)
            np.testing.assert_allclose(s_out.var.astype(np.float32), c["out_var"], rtol=1e-6, atol=1e-7)
            continue
        s_in, s_out = cls(x.shape[1]), cls(t.shape[1])
        s_in.fit(x); s_out.fit(t)
        np.testing.assert_array_equal(s_in.cached_mu, c["in_mu"])
        np.testing.assert_array_equal(s_in.cached_var, c["in_var"])
        np.testing.assert_array_equal(s_out.cached_mu, c["out_mu"])
        np.testing.assert_array_equal(s_out.cached_var, c["out_var"])
        s_in.fit(c["more"])                  # running branch (count > 0), float32 state in between
        np.testing.assert_array_equal(s_in.cached_mu, c["in_mu2"])
        np.testing.assert_array_equal(s_in.cached_var, c["in_var2"])
        assert s_in.cached_count == float(c["in_count2"])


def test_output_deltas_closed_form():
    """d(train_loss)/d(raw output) as the HIP loss kernel computes it == autograd of the restated losses."""
    rng = np.random.default_rng(0)
    E, B, D = 3, 17, 4
    o = torch.tensor(rng.standard_normal((E, B, 2 * D)), dtype=torch.float64, requires_grad=True)
    t = torch.tensor(rng.standard_normal((E, B, D)), dtype=torch.float64)
    reftrain.mspe_losses(o, t).sum().backward()
    od = o.detach()
    mean, lv = od[..., :D], od[..., D:]
    mse, var = (mean - t) ** 2, torch.exp(lv)
    ratio = 0.05 * mse.sum() / ((var - mse) ** 2).sum()
    d_mean = 2 * (mean - t) / (B * D)
    d_lv = (2 * ratio * (var - mse) * var + 0.1 * lv) / (B * D)
    np.testing.assert_allclose(o.grad[..., :D].numpy(), d_mean.numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(o.grad[..., D:].numpy(), d_lv.numpy(), rtol=1e-12, atol=1e-15)
    o2 = torch.tensor(rng.standard_normal((E, B, 1)), dtype=torch.float64, requires_grad=True)
    t2 = torch.tensor(rng.standard_normal((E, B, 1)), dtype=torch.float64)
    reftrain.mse_losses(o2, t2).sum().backward()
    np.testing.assert_allclose(o2.grad.numpy(), ((o2.detach() - t2) / B).numpy(), rtol=1e-12, atol=1e-15)


def test_adam_tf_rule_against_torch_adam():
    """With eps = 0 the TensorFlow and the torch formulations of Adam coincide."""
    rng = np.random.default_rng(1)
    p0 = torch.tensor(rng.standard_normal(50), dtype=torch.float64)
    gs = [torch.tensor(rng.standard_normal(50) + 0.1, dtype=torch.float64) for _ in range(5)]
    opt = reftrain.AdamTF([p0.clone()], lr=1e-2, eps=0.0)
    p = [p0.clone()]
    q = p0.clone().requires_grad_(True)
    topt = torch.optim.Adam([q], lr=1e-2, eps=0.0)
    for g in gs:
        p = opt.step(p, [g])
        q.grad = g.clone()
        topt.step()
    np.testing.assert_allclose(p[0].numpy(), q.detach().numpy(), rtol=1e-10)


def test_weight_decay_and_member_sum_in_train_loss():
    """train_loss = sum over members + decay_l * 0.5 |W_l|^2: the gradient of a weight the data does not reach is
    decay * w, and the regulariser 0.05 mean(lv^2) counts once per member."""
    rng = np.random.default_rng(2)
    E, I, H, D, B = 2, 3, 5, 2, 6
    ws = [rng.standard_normal((E, I, H)) * .3, rng.standard_normal((E, H, H)) * .3, rng.standard_normal((E, H, 2 * D)) * .3]
    bs = [np.zeros((E, 1, H)), np.zeros((E, 1, H)), np.zeros((E, 1, 2 * D))]
    tr = reftrain.EnsembleTrainer(ws, bs, "MSPE", decays=(0.1, 0.2, 0.3), dtype=torch.float64)
    x = np.zeros((E, B, I)); x[..., 0] = rng.standard_normal((E, B))      # inputs 1, 2 are dead
    t = rng.standard_normal((E, B, D))
    _, gs = tr.grads(x, t)
    np.testing.assert_allclose(gs[0][:, 1:, :].numpy(), 0.1 * ws[0][:, 1:, :], rtol=1e-12)
    loss_e = reftrain.mspe_losses(reftrain.forward_raw(torch.tensor(x), tr.ws, tr.bs), torch.tensor(t))
    lv = reftrain.forward_raw(torch.tensor(x), tr.ws, tr.bs)[..., D:]
    assert loss_e.shape == (E,)
    base = loss_e - 0.05 * (lv ** 2).mean()
    assert float((loss_e.sum() - base.sum())) == pytest.approx(E * 0.05 * float((lv ** 2).mean()), rel=1e-12)


def test_nll_deltas_closed_form_and_bounds_drift():
    """'NLL' (the class default of PE): d(train_loss)/d(raw output) as the HIP kernel computes it == autograd of the
    restated loss; max_logvar / min_logvar follow Adam under their constant regulariser gradient, and the product's
    closed form of that drift == the step-by-step rule."""
    rng = np.random.default_rng(3)
    E, B, D = 3, 13, 5
    o = torch.tensor(rng.standard_normal((E, B, 2 * D)), dtype=torch.float64, requires_grad=True)
    t = torch.tensor(rng.standard_normal((E, B, D)), dtype=torch.float64)
    reftrain.nll_losses(o, t).sum().backward()
    od = o.detach()
    mean, lv = od[..., :D], od[..., D:]
    iv = torch.exp(-lv)
    np.testing.assert_allclose(o.grad[..., :D].numpy(), (iv * (mean - t) / (B * D)).numpy(), rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(o.grad[..., D:].numpy(), ((0.5 - 0.5 * iv * (mean - t) ** 2) / (B * D)).numpy(),
                               rtol=1e-12, atol=1e-15)
    # the value is the Gaussian negative log-likelihood up to the constant 0.5 log(2 pi)
    nll = -torch.distributions.Normal(mean, torch.exp(0.5 * lv)).log_prob(t).mean(dim=-1).mean(dim=-1)
    np.testing.assert_allclose((reftrain.nll_losses(od, t) + 0.5 * np.log(2 * np.pi)).numpy(), nll.numpy(), rtol=1e-12)
    # bounds: AdamTF on the two variables with gradient +-0.01 == logvar_bounds == the product's closed form
    hi0, lo0 = np.full((1, D), 0.5), np.full((1, D), -6.0)
    opt = reftrain.AdamTF([torch.tensor(hi0), torch.tensor(lo0)], lr=1e-3)
    ps = [torch.tensor(hi0), torch.tensor(lo0)]
    for _ in range(37):
        ps = opt.step(ps, [torch.full((1, D), 0.01, dtype=torch.float64), torch.full((1, D), -0.01, dtype=torch.float64)])
    hi, lo = reftrain.logvar_bounds(37, 1e-3, hi0, lo0)
    np.testing.assert_allclose(hi, ps[0].numpy(), rtol=1e-6)
    np.testing.assert_allclose(lo, ps[1].numpy(), rtol=1e-6)
    sys.path.insert(0, os.path.dirname(HERE)) if os.path.dirname(HERE) not in sys.path else None
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd.pens import PE
    drift = PE._const_grad_adam_drift
    np.testing.assert_allclose(0.5 - drift(0.01, 37, 1e-3), float(ps[0][0, 0]), rtol=1e-9)
    np.testing.assert_allclose(-6.0 - drift(-0.01, 37, 1e-3), float(ps[1][0, 0]), rtol=1e-9)
    assert drift(0.01, 0, 1e-3) == 0.0
