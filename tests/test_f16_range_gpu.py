"""GPU: the three matrix paths of the 512-wide ensemble forward against a FLOAT64 evaluation of the same network
(models/pens/fc.py:74-95, models/pens/pe.py:789-838) where the two-piece f16 split could hurt:

  (i)   weights whose magnitudes are log-uniform over six decades inside every matrix, one member 1e3 x the others;
        and heavy-tailed (log-normal) weights, where single entries stand decades above the rest of their matrix;
  (ii)  input rows scaled by 1e3 and 1e-4, and a row with one huge feature;
  (iii) weights this repo's own PE.train produced (>= 200 Adam steps, the shipped weight decays of
        models/pens/pe_factory.py:50-60), and FakeEnv.step masks on those weights.

The float32 parameters and inputs are taken as exact; a path's error is measured per (member, row) relative to that
row's output scale.  The bound: the f16 path errs at most TWICE as much as the fp32-MFMA path (the arithmetic the
reference's float32 graph is closest to), with a floor of a few float32 roundings of the output scale.  This pins the
numerics of the default path against a ground truth, not against another float32 evaluation order.
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

F64 = np.float64
FLOOR = 6e-7          # ~5 float32 roundings of the row's output scale
PATHS = ((0, "fp32mfma"), (1, "splitbf16"), (2, "splitf16"))


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def f64_forward(x, ws, bs, sc_in, sc_out):
    """(mean, logvar)[E, B, out] in float64; the scaler sigma is the float32 one of models/pens/utils.py:156-187."""
    sig_in = np.maximum(np.sqrt(np.asarray(sc_in[1], np.float32)), np.float32(1e-2)).astype(F64).reshape(1, -1)
    sig_out = np.maximum(np.sqrt(np.asarray(sc_out[1], np.float32)), np.float32(1e-2)).astype(F64).reshape(1, 1, -1)
    h = (x.astype(F64) - np.asarray(sc_in[0], F64).reshape(1, -1)) / sig_in
    h = np.einsum("ij,ajk->aik", h, ws[0].astype(F64)) + bs[0].astype(F64).reshape(ws[0].shape[0], 1, -1)
    h = h / (1.0 + np.exp(-h))
    h = np.matmul(h, ws[1].astype(F64)) + bs[1].astype(F64).reshape(ws[1].shape[0], 1, -1)
    h = h / (1.0 + np.exp(-h))
    o = np.matmul(h, ws[2].astype(F64)) + bs[2].astype(F64).reshape(ws[2].shape[0], 1, -1)
    half = o.shape[-1] // 2
    mean = sig_out * o[..., :half] + np.asarray(sc_out[0], F64).reshape(1, 1, -1)
    logvar = 2.0 * np.log(sig_out) + o[..., half:]
    return mean, logvar


def path_errors(hip_lib, pe, x, ref_mean, ref_logvar):
    """{path: (err_mean[E, B], err_logvar[E, B])}: max over the outputs of |out - ref| / the (member, row)'s output scale."""
    out = {}
    before = hip_lib.cmbpo_get_ens_matrix_path()
    s_mean = np.maximum(np.abs(ref_mean).max(axis=2), 1e-30)
    s_lv = np.maximum(np.abs(ref_logvar).max(axis=2), 1.0)
    try:
        for path, name in PATHS:
            assert hip_lib.cmbpo_set_ens_matrix_path(path) == 0
            hip_lib.cmbpo_set_ens_f16_min_rows(0)
            mean, var = pe.predict_ensemble(x)
            assert np.isfinite(mean).all() and not np.isnan(var).any() and (var >= 0).all(), name
            e_mean = np.abs(mean.astype(F64) - ref_mean).max(axis=2) / s_mean
            # the variance head is exp(logvar) in float32: compared as log(var) where float32 can hold it; beyond that range
            # (a 1e3-scaled row drives the head to +-1e2) it must overflow to inf / underflow towards 0 like the reference
            ok = np.abs(ref_logvar) < 80.0
            with np.errstate(divide="ignore"):
                lv = np.log(var.astype(F64))
            e_lv = (np.where(ok, np.abs(lv - ref_logvar), 0.0)).max(axis=2) / s_lv
            assert np.isinf(var[ref_logvar > 89.0]).all() and (var[ref_logvar < -104.0] == 0).all(), name
            out[name] = (e_mean, e_lv)
    finally:
        hip_lib.cmbpo_set_ens_matrix_path(before)
        hip_lib.cmbpo_set_ens_f16_min_rows(0)
    return out


def assert_f16_within_twice_fp32(errs, what):
    """Every (member, row): err_f16 <= 2 * max(that member's worst fp32-MFMA row, the members' mean worst row, FLOOR) -- errors relative to the
    (member, row)'s own output scale, so a small row is not hidden behind a large one -- and the same over everything.
    (Row by row the two paths' errors are a few roundings in different orders: their ratio scatters by more than 2 both
    ways; tools/probe_f16_range.py prints the quantiles.)"""
    report = {k: (float(v[0].max()), float(v[1].max())) for k, v in errs.items()}
    for part, label in ((0, "mean"), (1, "logvar")):
        f16, f32 = errs["splitf16"][part], errs["fp32mfma"][part]
        # (a member's worst fp32 row is itself a maximum over a few thousand rounding patterns: it scatters by 2x from
        # member to member of one ensemble -- 1.2e-6 .. 2.2e-6 on the trained AntSafe model -- so a member is not held
        # below the ensemble's typical worst case)
        worst = f32.max(axis=1, keepdims=True)
        bound = 2.0 * np.maximum(np.maximum(worst, worst.mean()), FLOOR)
        assert (f16 <= bound).all(), \
            f"{what}/{label}: f16 path per member {f16.max(axis=1)} vs fp32-MFMA path {f32.max(axis=1)}; all: {report}"
        assert f16.max() <= 2.0 * max(f32.max(), FLOOR), f"{what}/{label}: {report}"
        # and in distribution: the f16 path's median and 99th percentile row are no worse than twice the fp32 path's
        for q in (50, 99):
            a, b = np.percentile(f16, q), np.percentile(f32, q)
            assert a <= 2.0 * max(b, FLOOR), f"{what}/{label}: {q}th percentile {a:.3e} vs {b:.3e}"
    return report


def _pe(E, I, O, ws, bs, sc_in, sc_out):
    from cmbpo_amd.pens import PE
    pe = PE(I, O, hidden_dims=(512, 512), num_networks=E, num_elites=min(5, E), loss="MSPE", use_scaler_in=True,
            use_scaler_out=True, device="cuda:0")
    pe.set_weights(ws, bs, sc_in, sc_out)
    return pe


def _wide_weights(rng, kind, E, I, H, O2):
    """Weights of the reference's layout W[E, in, out] whose magnitudes span decades inside every matrix."""
    ws, bs = [], []
    for li, (i, o) in enumerate(((I, H), (H, H), (H, O2))):
        sign = rng.choice([-1.0, 1.0], size=(E, i, o))
        if kind == "loguniform6":
            # |w| log-uniform over six decades, the top of the range ~10 x the initialisation's sigma (pre-activations
            # keep the scale of their inputs: E[w^2] = top^2 / (12 ln 10))
            mag = 10.0 ** rng.uniform(-6.0, 0.0, size=(E, i, o)) * (5.0 / np.sqrt(i))
        else:
            # heavy tail: log-normal, sigma 2 -- single entries four decades above the typical one, output units with
            # no large weight at all next to units dominated by one
            mag = np.exp(2.0 * rng.standard_normal((E, i, o))) * (0.01 / np.sqrt(i))
        w = (sign * mag).astype(np.float32)
        if li == 2:
            w *= np.float32(0.2)
        # member 1: first layer 1e3 x the others (hidden activations ~1e3), last layer 1e-3 x (outputs stay O(1): the
        # variance head is an exp); member 2: a second layer two decades below the others
        if li == 0:
            w[1] *= np.float32(1e3)
        if li == 2:
            w[1] *= np.float32(1e-3)
        if li == 1:
            w[2] *= np.float32(1e-2)
        ws.append(w)
        bs.append((rng.standard_normal((E, 1, o)) * 0.05).astype(np.float32))
    return ws, bs


@pytest.mark.parametrize("kind", ["loguniform6", "lognormal_tail"])
@pytest.mark.parametrize("task", ["AntSafe-v2", "HumanoidSafe-v2"])
def test_wide_range_weights_against_float64(hip_lib, kind, task):
    _cuda()
    from cmbpo_amd import synthetic
    rng = np.random.default_rng(zlib.crc32(f"{kind}/{task}".encode()))
    D, A = synthetic.ENV_DIMS[task]
    E, I, O = 7, D + A, D + 1
    ws, bs = _wide_weights(rng, kind, E, I, 512, 2 * O)
    sc_in, sc_out = synthetic.scaler(rng, I), synthetic.scaler(rng, O)
    pe = _pe(E, I, O, ws, bs, sc_in, sc_out)
    x = rng.standard_normal((777, I)).astype(np.float32)
    ref_mean, ref_lv = f64_forward(x, ws, bs, sc_in, sc_out)
    errs = path_errors(hip_lib, pe, x, ref_mean, ref_lv)
    assert_f16_within_twice_fp32(errs, f"{kind}/{task}")


@pytest.mark.parametrize("task", ["AntSafe-v2", "HalfCheetahSafe-v2"])
def test_wide_range_rows_against_float64(hip_lib, task):
    """(ii): rows scaled by 1e3 and 1e-4 next to ordinary ones in the same 128-row item, a row with one huge feature, a
    zero row -- ordinary weights.  Every row keeps its own lift (rows are MFMA columns)."""
    _cuda()
    from cmbpo_amd import synthetic
    rng = np.random.default_rng(zlib.crc32(f"rows/{task}".encode()))
    D, A = synthetic.ENV_DIMS[task]
    E, I, O = 7, D + A, D + 1
    ws, bs = synthetic.ensemble_weights(rng, E, I, 512, 2 * O, bias_scale=0.05)
    sc_in, sc_out = synthetic.scaler(rng, I, hit_clamp=False), synthetic.scaler(rng, O)
    pe = _pe(E, I, O, ws, bs, sc_in, sc_out)
    x = rng.standard_normal((900, I)).astype(np.float32)
    x[3::7] *= np.float32(1e3)
    x[5::7] *= np.float32(1e-4)
    x[6::49, 4] = np.float32(3e4)           # one huge feature
    x[13::49, :] = 0.0
    x[20::49, 0::2] *= np.float32(1e2)      # half of the features two decades above the others
    ref_mean, ref_lv = f64_forward(x, ws, bs, sc_in, sc_out)
    errs = path_errors(hip_lib, pe, x, ref_mean, ref_lv)
    assert_f16_within_twice_fp32(errs, f"rows/{task}")


def _trained_model(rng_seed, task, steps, H=512):
    """A dynamics ensemble fitted by this repo's PE.train (TrainControl + the HIP training kernels) on a smooth synthetic
    system with the shipped optimiser settings (lr 1e-3, decays decay/4, decay/2, decay with decay = 1e-4)."""
    from cmbpo_amd import synthetic
    from cmbpo_amd.pens import PE
    D, A = synthetic.ENV_DIMS[task]
    I, O = D + A, D + 1
    rng = np.random.RandomState(rng_seed)
    n = 6000
    obs = (rng.standard_normal((n, D)) * rng.uniform(0.05, 8.0, size=(1, D))).astype(np.float32)     # features of very different scale
    act = np.tanh(rng.standard_normal((n, A))).astype(np.float32)
    M = (rng.standard_normal((I, O)) / np.sqrt(I)).astype(np.float32)
    xin = np.concatenate([obs, act], 1)
    tgt = (0.05 * np.sin(xin @ M) + 0.02 * (xin @ M) + 0.01 * rng.standard_normal((n, O))).astype(np.float32)
    tgt[:, -1] = (np.abs(obs[:, 0]) * 0.3 + act[:, 0] ** 2).astype(np.float32)                       # a "reward" column
    pe = PE(I, O, hidden_dims=(H, H), num_networks=7, num_elites=5, loss="MSPE", use_scaler_in=True, use_scaler_out=True,
            device="cuda:0", lr=1e-3, decay=1e-4)
    pe.init_weights(rng)
    batch = 256
    epochs = int(np.ceil(steps / np.ceil(n * 0.9 / batch)))
    pe.train(xin, tgt, batch_size=batch, max_epochs=epochs, holdout_ratio=0.1, max_epochs_since_update=epochs + 1,
             min_epoch_before_break=epochs + 1, rng=rng)
    assert pe.train_grad_updates >= steps
    return pe, xin, (D, A)


def test_trained_weights_against_float64(hip_lib):
    """(iii): the f16 path on weights an optimiser produced (>= 200 Adam steps with the shipped weight decays), not on
    the initialisation's truncated normals."""
    _cuda()
    pe, xin, _ = _trained_model(7, "AntSafe-v2", 220)
    ws, bs = pe.get_weights()
    sc_in = (pe.scaler_in.cached_mu, pe.scaler_in.cached_var)
    sc_out = (pe.scaler_out.cached_mu, pe.scaler_out.cached_var)
    x = xin[:1500].copy()
    x[::11] *= np.float32(30.0)             # states far outside the training distribution, as a diverging rollout visits
    ref_mean, ref_lv = f64_forward(x, ws, bs, sc_in, sc_out)
    errs = path_errors(hip_lib, pe, x, ref_mean, ref_lv)
    rep = assert_f16_within_twice_fp32(errs, "trained")
    # spread of the trained matrices (what the lifts have to cover), for the record of the failure message above
    assert all(np.isfinite(w).all() for w in ws), rep


def test_fakeenv_masks_on_trained_weights_agree_across_paths(hip_lib):
    """FakeEnv.step on the trained ensemble: termination and cost masks bit-exact between the f16 path and the fp32-MFMA
    path (identical elite picks), next observations within the forward tolerance."""
    _cuda()
    from cmbpo_amd.fake_env import FakeEnv
    pe, xin, (D, A) = _trained_model(9, "AntSafe-v2", 200)

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    class _Env:
        observation_space, action_space = _Space(D), _Space(A)

    env = FakeEnv(_Env(), "AntSafe-v2", pe, predicts_delta=True, predicts_rew=True, predicts_cost=False)
    rng = np.random.default_rng(3)
    n = 4000
    obs = xin[:n, :D].copy()
    obs[:, 0] = rng.uniform(0.1, 1.1, n).astype(np.float32)         # torso height around the termination thresholds
    act = xin[:n, D:].copy()
    inds = rng.integers(0, 7, n).astype(np.int32)
    outs = {}
    before = hip_lib.cmbpo_get_ens_matrix_path()
    try:
        for path, name in PATHS:
            assert hip_lib.cmbpo_set_ens_matrix_path(path) == 0
            hip_lib.cmbpo_set_ens_f16_min_rows(0)
            next_obs, rew, term, info = env.step(obs, act, model_inds=inds)
            outs[name] = dict(next_obs=next_obs, rew=rew, term=term, cost=info["cost"])
    finally:
        hip_lib.cmbpo_set_ens_matrix_path(before)
        hip_lib.cmbpo_set_ens_f16_min_rows(0)
    a, b = outs["fp32mfma"], outs["splitf16"]
    scale = np.abs(a["next_obs"]).max()
    np.testing.assert_allclose(b["next_obs"], a["next_obs"], rtol=0, atol=4e-6 * scale)
    # a mask may differ only where the deciding quantity sits within the forward error of its threshold: none here
    np.testing.assert_array_equal(b["term"], a["term"])
    np.testing.assert_array_equal(b["cost"], a["cost"])
    np.testing.assert_allclose(b["rew"], a["rew"], rtol=0, atol=4e-6 * max(1.0, float(np.abs(a["rew"]).max())))
    assert 0 < int(a["term"].sum()) < n                              # the thresholds were actually exercised
