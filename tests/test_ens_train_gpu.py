"""GPU parity of ensemble training against oracle/reftrain.py.

The TensorFlow half of the oracle is "parity unpinned" (TF 1.14 cannot run here): the torch-autograd restatement
of the training graph is the checker, the analytic cross-checks of tests/test_oracle_train.py pin it.

Tolerances (fp32; sums over the batch in a different order than the oracle):
  * gradients (read back as 10 x the first Adam moment after step 1): |d| <= 2e-3 |ref| + 5e-5 max|ref| per tensor;
  * per-member losses: rtol 2e-4;
  * k Adam steps: || w_hip - w_ref || <= 3e-2 || w_ref - w_init || (the first steps are ~ lr * sign(g), so a
    near-zero gradient component may legitimately differ by a whole step).
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)

from oracle import reftrain  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _make(E, I, H, D, loss, n, seed, use_scalers=True, lr=1e-3, decay=1e-3):
    from cmbpo_amd.pens import PE
    rng = np.random.RandomState(seed)
    pe = PE(I, D, name="T", hidden_dims=(H, H), num_networks=E, num_elites=max(1, E - 2), loss=loss,
            use_scaler_in=use_scalers, use_scaler_out=use_scalers, device="cuda:0", lr=lr, decay=decay)
    ws, bs = pe.init_weights(rng)
    # non-zero biases and a livelier output layer so every gradient path carries signal
    bs = [(0.1 * rng.standard_normal(b.shape)).astype(np.float32) for b in bs]
    ws[2] = (ws[2] * 3).astype(np.float32)
    x = (rng.standard_normal((n, I)) * (1 + rng.rand(I)) + rng.standard_normal(I)).astype(np.float32)
    wtrue = rng.standard_normal((I, D)) / np.sqrt(I)
    t = (np.tanh(x @ wtrue) + 0.1 * rng.standard_normal((n, D))).astype(np.float32)
    sc_in = sc_out = None
    if use_scalers:
        sc_in = (x.mean(0, keepdims=True), x.var(0, keepdims=True))
        sc_out = (t.mean(0, keepdims=True), t.var(0, keepdims=True))
    pe.set_weights(ws, bs, sc_in, sc_out)
    ref = reftrain.EnsembleTrainer(ws, bs, loss_type=loss, decays=pe.decays, lr=lr)
    ref.set_scalers(sc_in, sc_out)
    return rng, pe, ref, x, t, ws, bs


def _close(got, ref, rtol, arel, msg):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=arel * float(np.max(np.abs(ref))) + 1e-12, err_msg=msg)


CASES = [  # E, I, H, D, loss, batch
    (3, 11, 128, 4, "MSPE", 100),
    (7, 37, 512, 30, "MSPE", 256),
    (3, 29, 128, 1, "MSE", 77),
    (2, 20, 512, 1, "MSE", 64),
    (4, 8, 128, 3, "MSPE", 32),
    (3, 45, 128, 2, "MSE", 100),     # fused critic step, two row tiles of W0, two outputs
    (2, 29, 128, 1, "MSE", 2048),    # fused critic step at the shipped batch size
    (7, 37, 512, 30, "MSPE", 2048),  # the dynamics ensemble of the AntSafe config at the shipped batch size
    (1, 64, 128, 8, "MSE", 33),      # fused step at its limits: in_dim 64, 8 outputs (the deterministic head's maximum), one member
    (2, 5, 128, 1, "MSE", 1),        # a single row
    (2, 100, 128, 64, "MSPE", 40),   # 128 raw outputs (4 output tiles), wide input
    (3, 23, 512, 11, "MSPE", 70),    # HalfCheetah-like dynamics (odd output width 22)
    (2, 100, 128, 2, "MSE", 40),     # input too wide for the fused kernel: the general path on a deterministic head
    (7, 37, 512, 30, "NLL", 256),    # the class-default loss of PE on the dynamics shapes
    (3, 11, 128, 4, "NLL", 100),
    (3, 29, 256, 1, "MSE", 100),     # the base config's default critic width (configs/baseconfig/base.py:16,19): the general path
    (4, 20, 256, 6, "MSPE", 150),    # ... and a probabilistic ensemble of that width
]


@pytest.mark.parametrize("E,I,H,D,loss,batch", CASES)
def test_first_step_gradients_and_losses(hip_lib, E, I, H, D, loss, batch):
    _need_gpu()
    rng, pe, ref, x, t, ws, bs = _make(E, I, H, D, loss, max(600, 2 * batch), seed=E * 1000 + I)
    tr = pe._ensure_trainer(batch)
    idx = rng.randint(0, x.shape[0], size=(E, batch)).astype(np.int32)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    idx_d = torch.from_numpy(idx).cuda()
    # `self.loss` before the step
    hold = rng.permutation(x.shape[0])[:157].astype(np.int32)
    got_l = tr.losses(xd, td, torch.from_numpy(hold).cuda(), 0, hold.shape[0]).cpu().numpy()
    ref_l = ref.losses(np.tile(x[hold][None], (E, 1, 1)), np.tile(t[hold][None], (E, 1, 1)))
    np.testing.assert_allclose(got_l, ref_l, rtol=2e-4)
    # per-member rows (3-D feed of the reference): losses on the bootstrap batch
    got_lb = tr.losses(xd, td, idx_d, batch, batch).cpu().numpy()
    np.testing.assert_allclose(got_lb, ref.losses(x[idx], t[idx]), rtol=2e-4)

    _, gs = ref.grads(x[idx], t[idx])
    tr.step(xd, td, idx_d.data_ptr(), batch, batch)
    mw, mb = tr.get_moments(0)
    n = len(ws)
    for l in range(n):
        _close(10.0 * mw[l], gs[l].numpy(), 2e-3, 5e-5, f"dW{l}")
        _close(10.0 * mb[l], gs[n + l].numpy().reshape(mb[l].shape), 2e-3, 5e-5, f"db{l}")
    vw, _ = tr.get_moments(1)
    _close(1000.0 * vw[1], gs[1].numpy() ** 2, 5e-3, 1e-6, "v1")
    assert tr.steps_done == 1


@pytest.mark.parametrize("E,I,H,D,loss,batch", CASES[:3] + CASES[-1:])
def test_several_adam_steps_track_oracle(hip_lib, E, I, H, D, loss, batch):
    _need_gpu()
    rng, pe, ref, x, t, ws, bs = _make(E, I, H, D, loss, 500, seed=7 + E)
    tr = pe._ensure_trainer(batch)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    w_init = [w.copy() for w in ws]
    for k in range(6):
        b = batch if k != 3 else batch - 9     # a ragged batch in the middle (last batch of an epoch)
        idx = rng.randint(0, x.shape[0], size=(E, b)).astype(np.int32)
        idx_d = torch.from_numpy(idx).cuda()
        tr.step(xd, td, idx_d.data_ptr(), b, b)
        ref.step(x[idx], t[idx])
    gw, gb = tr.get_weights()
    for l in range(3):
        rw = ref.ws[l].numpy()
        moved = float(np.linalg.norm(rw - w_init[l]))
        assert float(np.linalg.norm(gw[l] - rw)) <= 3e-2 * moved, (l, moved)
        rb = ref.bs[l].numpy().reshape(gb[l].shape)
        assert float(np.linalg.norm(gb[l] - rb)) <= 3e-2 * float(np.linalg.norm(rb - bs[l].reshape(rb.shape))) + 1e-7
    # the packed images the prediction kernels read follow the masters
    pe._weights_on_device = True
    xs = x[:50]
    if loss in ("MSPE", "NLL"):
        from oracle import refcpu
        mean, var = pe.predict_ensemble(xs)
        sc_in = (pe.scaler_in.cached_mu, pe.scaler_in.cached_var)
        sc_out = (pe.scaler_out.cached_mu, pe.scaler_out.cached_var)
        rm, rv = refcpu.ens_forward(xs, gw, gb, sc_in, sc_out)
        np.testing.assert_allclose(mean, rm, rtol=2e-4, atol=2e-4)
        np.testing.assert_allclose(var, rv, rtol=2e-3, atol=1e-6)


def test_step_is_bitwise_reproducible(hip_lib):
    """No atomics on the gradient path: two trainers fed the same batches hold identical weights."""
    _need_gpu()
    outs = []
    for rep in range(2):
        rng, pe, ref, x, t, ws, bs = _make(3, 11, 128, 4, "MSPE", 300, seed=99)
        tr = pe._ensure_trainer(64)
        xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
        for k in range(4):
            idx = torch.from_numpy(rng.randint(0, 300, size=(3, 64)).astype(np.int32)).cuda()
            tr.step(xd, td, idx.data_ptr(), 64, 64)
        outs.append(tr.get_weights())
    for a, b in zip(outs[0][0] + outs[0][1], outs[1][0] + outs[1][1]):
        np.testing.assert_array_equal(a, b)


class _OracleOps:
    """The numerics behind reftrain.train_loop, on the same data PE.train sees."""

    def __init__(self, ref, x, t, use_scalers):
        self.ref, self.x, self.t, self.use_scalers = ref, x, t, use_scalers
        self.sc_in, self.sc_out = reftrain.RunningScaler(x.shape[1]), reftrain.RunningScaler(t.shape[1])

    def fit_scalers(self, rows):
        if self.use_scalers:
            self.sc_in.fit(self.x[rows])
            self.sc_out.fit(self.t[rows])
            f = lambda s: (s.mu.astype(np.float32), s.var.astype(np.float32))
            self.ref.set_scalers(f(self.sc_in), f(self.sc_out))

    def train_step(self, rows):
        self.ref.step(self.x[rows], self.t[rows])

    def holdout_losses(self, rows):
        E = self.ref.ws[0].shape[0]
        return self.ref.losses(np.tile(self.x[rows][None], (E, 1, 1)), np.tile(self.t[rows][None], (E, 1, 1)))


@pytest.mark.parametrize("loss,D", [("MSPE", 3), ("MSE", 1), ("NLL", 3)])
def test_pe_train_follows_oracle_loop(hip_lib, loss, D):
    """PE.train end to end (same numpy RandomState on both sides): epochs, gradient updates, scaler moments,
    holdout losses and the elite ranking."""
    _need_gpu()
    E, I, H = 4, 9, 128
    # (the scaler moments _make loads are overwritten by the first fit: the running count starts at 0)
    rng, pe, ref, x, t, ws, bs = _make(E, I, H, D, loss, 700, seed=5, use_scalers=True, lr=1e-3)
    kw = dict(batch_size=64, max_epochs=6, holdout_ratio=0.2, max_epochs_since_update=5, min_epoch_before_break=2)
    out = pe.train(x, t, rng=np.random.RandomState(123), **kw)
    ops = _OracleOps(ref, x, t, True)
    elites, final, epochs, updates = reftrain.train_loop(ops, x.shape[0], E, pe.num_elites, np.random.RandomState(123), **kw)
    assert (pe.train_epochs, pe.train_grad_updates) == (epochs, updates)
    np.testing.assert_allclose(pe.scaler_in.cached_mu, ops.sc_in.mu, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(pe.scaler_out.cached_var, ops.sc_out.var, rtol=1e-5, atol=1e-6)
    got_final = pe._trainer.losses(torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda(),
                                   torch.arange(x.shape[0], dtype=torch.int32, device="cuda"), 0, x.shape[0]).cpu().numpy()
    full = ops.holdout_losses(np.arange(x.shape[0]))
    np.testing.assert_allclose(got_final, full, rtol=3e-2)
    srt = np.sort(final)
    if len(srt) > pe.num_elites and (srt[pe.num_elites] - srt[pe.num_elites - 1]) > 0.05 * srt[pe.num_elites - 1]:
        assert sorted(pe.elite_inds) == sorted(elites)
    np.testing.assert_allclose(out[f"{pe.name}/val_loss"], np.sort(final)[:pe.num_elites].mean(), rtol=3e-2)


def test_update_critic_trains_both_critics(hip_lib):
    """CPOPolicy.update_critic (policies/cpo_policy.py:658-698): both critics fit their returns, the logger gets the
    loss measures and their deltas, the rollout-side predict() reads the trained weights."""
    _need_gpu()
    from cmbpo_amd.cpo_policy import CPOPolicy

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    D, A, n = 12, 3, 6000
    rng = np.random.RandomState(3)
    pol = CPOPolicy(_Space(D), _Space(A), a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128),
                    vf_ensemble_size=3, vf_elites=2, vf_activation="swish", vf_loss="MSE", device="cuda:0",
                    vf_lr=3e-3, vf_epochs=6, vf_batch_size=256, max_path_length=10)
    pol.v.init_weights(rng)
    pol.vc.init_weights(rng)
    obs = rng.standard_normal((n, D)).astype(np.float32)
    ret = (np.sin(obs[:, 0]) + 0.5 * obs[:, 1] + 3.0).astype(np.float32)
    cret = (np.abs(obs[:, 2]) * 2.0).astype(np.float32)
    z = np.zeros(n, np.float32)
    buf = [obs, np.zeros((n, A), np.float32), z, z, ret, cret, z, z, z, z, np.zeros((n, A), np.float32),
           np.zeros((n, A), np.float32)]
    out = pol.update_critic(buf, rng=rng)
    assert out["post"]["LossVEnsemble"] < 0.5 * out["pre"]["LossVEnsemble"]
    assert out["post"]["LossVCEnsemble"] < 0.5 * out["pre"]["LossVCEnsemble"]
    st = pol.logger.stored
    for k in ("LossVEnsemble", "LossVCEnsemble", "LossVEnsembleDelta", "LossVCEnsembleDelta"):
        assert k in st
    assert pol.v.train_epochs == 6 and pol.v.train_grad_updates == 6 * int(np.ceil(5400 / 256))
    assert len(pol.v.elite_inds) == 2
    # get_v reads the packed weights the trainer maintains: mean over members, de-normalised by the fitted scaler
    from oracle import refcpu
    ws, bs = pol.v.get_weights()
    sc_in = (pol.v.scaler_in.cached_mu, pol.v.scaler_in.cached_var)
    sc_out = (pol.v.scaler_out.cached_mu, pol.v.scaler_out.cached_var)
    ref = refcpu.ens_predict_mean(obs[:200], ws, bs, sc_in, sc_out)[:, 0]
    np.testing.assert_allclose(pol.get_v(obs[:200]), ref, rtol=2e-4, atol=2e-4)
    assert float(np.mean((pol.get_v(obs) - ret) ** 2)) < 0.2 * float(np.var(ret))


def test_checkpoint_round_trip_through_device(hip_lib, tmp_path):
    """PE.save after training steps -> PE.load into a fresh ensemble: same masters, same predictions."""
    _need_gpu()
    rng, pe, ref, x, t, ws, bs = _make(3, 11, 128, 4, "MSPE", 300, seed=21)
    tr = pe._ensure_trainer(64)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    for k in range(3):
        idx = torch.from_numpy(rng.randint(0, 300, size=(3, 64)).astype(np.int32)).cuda()
        tr.step(xd, td, idx.data_ptr(), 64, 64)
    pe._weights_on_device = True
    pe.save(str(tmp_path), 7)
    from cmbpo_amd.pens import PE
    pe2 = PE(11, 4, name="T", hidden_dims=(128, 128), num_networks=3, num_elites=1, loss="MSPE", use_scaler_in=True,
             use_scaler_out=True, device="cuda:0")
    pe2.load(str(tmp_path), 7)
    for a, b in zip(pe.get_weights()[0] + pe.get_weights()[1], pe2.get_weights()[0] + pe2.get_weights()[1]):
        np.testing.assert_array_equal(a, b)
    m1, v1 = pe.predict_ensemble(x[:40])
    m2, v2 = pe2.predict_ensemble(x[:40])
    np.testing.assert_array_equal(m1, m2)
    np.testing.assert_array_equal(v1, v2)


def test_pe_train_edge_cases(hip_lib):
    """No holdout set (the reference ranks NaN losses: member order kept), a batch larger than the data, a second
    train() call that continues from the first (Adam state and running scaler moments carried over), tensors in."""
    _need_gpu()
    rng, pe, ref, x, t, ws, bs = _make(3, 11, 128, 4, "MSPE", 90, seed=31)
    out = pe.train(x, t, batch_size=256, max_epochs=3, holdout_ratio=0.0, rng=np.random.RandomState(1))
    assert pe.train_epochs == 3 and pe.train_grad_updates == 3          # one ragged batch per epoch
    assert list(pe.elite_inds) == [0]                                   # argsort of NaNs, num_elites = 1
    assert np.isnan(out["T/val_loss"])
    steps_before = pe._trainer.steps_done
    count_before = pe.scaler_in.cached_count
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    out2 = pe.train(xd, td, batch_size=32, max_epochs=2, holdout_ratio=0.25, rng=np.random.RandomState(2),
                    shuffle_on_device=True)
    n_train = 90 - int(90 * 0.25)
    assert pe._trainer.steps_done == steps_before + 2 * int(np.ceil(n_train / 32))
    assert pe.scaler_in.cached_count == count_before + n_train
    assert np.isfinite(out2["T/val_loss"])
    m, v = pe.predict_ensemble(x[:10])
    assert np.isfinite(m).all() and (v > 0).all()
    with pytest.raises(NotImplementedError):
        pe.train(x, t, weights=np.ones(90))


def test_policy_checkpoint_round_trip(hip_lib, tmp_path):
    _need_gpu()
    from cmbpo_amd import synthetic
    from cmbpo_amd.cpo_policy import CPOPolicy

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    kw = dict(a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128), vf_ensemble_size=3, vf_elites=2,
              vf_activation="swish", vf_loss="MSE", device="cuda:0", max_path_length=10)
    pol = CPOPolicy(_Space(9), _Space(3), **kw)
    pol.set_params(synthetic.policy_params(np.random.default_rng(0), 9, 3, 128))
    rng = np.random.RandomState(0)
    pol.v.init_weights(rng)
    pol.vc.init_weights(rng)
    pol.save(str(tmp_path), 12)
    pol2 = CPOPolicy(_Space(9), _Space(3), **kw)
    pol2.load(str(tmp_path), 12)
    obs = rng.standard_normal((33, 9)).astype(np.float32)
    eps = rng.standard_normal((33, 3)).astype(np.float32)
    a, b = pol.get_action_outs(obs, eps=eps), pol2.get_action_outs(obs, eps=eps)
    for k in ("pi", "logp_pi", "v", "vc"):
        np.testing.assert_array_equal(a[k], b[k])


def test_nll_checkpoint_carries_logvar_bounds(hip_lib, tmp_path):
    """An 'NLL' model saves max_logvar / min_logvar as its last two variables (pe.py:208-209,760-764); they drift with
    the optimiser steps as tf.train.AdamOptimizer moves them under the regulariser's constant gradient."""
    _need_gpu()
    from scipy.io import loadmat
    from cmbpo_amd.pens import PE
    E, I, H, D, batch = 3, 11, 128, 4, 64
    rng, pe, ref, x, t, ws, bs = _make(E, I, H, D, "NLL", 300, seed=5)
    np.testing.assert_array_equal(pe.max_logvar, np.full((1, D), 0.5, np.float32))
    np.testing.assert_array_equal(pe.min_logvar, np.full((1, D), -6.0, np.float32))
    tr = pe._ensure_trainer(batch)
    xd, td = torch.from_numpy(x).cuda(), torch.from_numpy(t).cuda()
    for _ in range(5):
        idx = torch.from_numpy(rng.randint(0, x.shape[0], size=(E, batch)).astype(np.int32)).cuda()
        tr.step(xd, td, idx.data_ptr(), batch, batch)
    pe._weights_on_device = True
    hi, lo = reftrain.logvar_bounds(5, pe.lr, np.full((1, D), 0.5), np.full((1, D), -6.0))
    np.testing.assert_allclose(pe.max_logvar, hi, rtol=1e-6)
    np.testing.assert_allclose(pe.min_logvar, lo, rtol=1e-6)
    nns, mat = pe.save(str(tmp_path), 7)
    d = loadmat(mat)
    n_vars = len([k for k in d if k.isdigit()])
    assert n_vars == 4 + 6 + 2                      # two scalers, three layers, the two bounds
    np.testing.assert_allclose(d[str(n_vars - 2)], hi, rtol=1e-6)
    np.testing.assert_allclose(d[str(n_vars - 1)], lo, rtol=1e-6)
    pe2 = PE(I, D, name="T", hidden_dims=(H, H), num_networks=E, num_elites=1, loss="NLL", use_scaler_in=True,
             use_scaler_out=True, device="cuda:0")
    pe2.load(str(tmp_path), 7)
    np.testing.assert_allclose(pe2.max_logvar, hi, rtol=1e-6)
    gw, gb = pe.get_weights()
    gw2, gb2 = pe2.get_weights()
    for a, b in zip(gw + gb, gw2 + gb2):
        np.testing.assert_array_equal(a, b)
    # an MSPE model refuses the NLL file (its variable list is two entries shorter) and vice versa
    pe3 = PE(I, D, name="T", hidden_dims=(H, H), num_networks=E, num_elites=1, loss="MSPE", use_scaler_in=True,
             use_scaler_out=True, device="cuda:0")
    with pytest.raises(ValueError):
        pe3.load(str(tmp_path), 7)
