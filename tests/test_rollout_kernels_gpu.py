"""GPU parity: HIP rollout kernels (through the C-ABI) vs the CPU oracle on the same seeded inputs.

Tolerances (fp32 path, stated per check):
  * ensemble mean: |d| <= 2e-4 + 2e-4*|ref|  (K = 512 fp32 accumulations in a different order than
    NumPy's BLAS; hidden activations O(1)),  var: rtol 2e-3 (exp of a logvar with the same abs error);
  * policy / critic heads: 1e-4 abs+rel;
  * FakeEnv post-processing on IDENTICAL (mean, var): next_obs / reward bit-exact (one fp32 add / a
    copy), termination and cost masks bit-exact, dkl / variance rtol 1e-4 (device logf/expf vs libm).
"""
import zlib

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import refcpu  # noqa: E402


def _cuda():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _dyn_model(rng, task, hidden=512, E=7):
    from cmbpo_amd import synthetic
    from cmbpo_amd.pens import PE
    obs_dim, act_dim = synthetic.ENV_DIMS[task]
    ws, bs = synthetic.ensemble_weights(rng, E, obs_dim + act_dim, hidden, 2 * (obs_dim + 1), bias_scale=0.05)
    sc_in = synthetic.scaler(rng, obs_dim + act_dim)
    sc_out = synthetic.scaler(rng, obs_dim + 1)
    m = PE(obs_dim + act_dim, obs_dim + 1, hidden_dims=(hidden, hidden), num_networks=E, num_elites=5,
           loss="MSPE", use_scaler_in=True, use_scaler_out=True, device="cuda:0")
    m.set_weights(ws, bs, sc_in, sc_out)
    return m, ws, bs, sc_in, sc_out, obs_dim, act_dim


@pytest.mark.parametrize("ens_path", [0, 1, 2], indirect=True, ids=["fp32mfma", "splitbf16", "splitf16"])
@pytest.mark.parametrize("task", ["AntSafe-v2", "HalfCheetahSafe-v2", "HopperSafe-v2", "HumanoidSafe-v2"])
@pytest.mark.parametrize("n", [1, 31, 32, 33, 63, 64, 65, 257, 1200, 2431])   # more than 36 tiles of 32 rows x 7 members: 64-row items
def test_ens_forward_matches_oracle(hip_lib, ens_path, task, n):
    _cuda()
    rng = np.random.default_rng(zlib.crc32(f"{task}/{n}".encode()))     # stable across processes (str hash is salted)
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, task)
    x = rng.standard_normal((n, obs_dim + act_dim)).astype(np.float32)
    mean, var = m.predict_ensemble(x)
    rmean, rvar = refcpu.ens_forward(x, ws, bs, sc_in, sc_out)
    assert mean.shape == rmean.shape == (7, n, obs_dim + 1)
    np.testing.assert_allclose(mean, rmean, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(var, rvar, rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("rt", [1, 2, 4])
@pytest.mark.parametrize("task", ["AntSafe-v2", "HalfCheetahSafe-v2", "HumanoidSafe-v2"])
@pytest.mark.parametrize("n", [1, 33, 130, 1000])
def test_ens_forward_f16_item_shapes(hip_lib, rt, task, n):
    """The f16 matrix path at every item shape (32 / 64 / 128 rows per item), whatever the row count would choose."""
    _cuda()
    rng = np.random.default_rng(zlib.crc32(f"{task}/{n}/{rt}".encode()))
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, task)
    x = rng.standard_normal((n, obs_dim + act_dim)).astype(np.float32)
    before = hip_lib.cmbpo_get_ens_matrix_path()
    try:
        assert hip_lib.cmbpo_set_ens_matrix_path(2) == 0 and hip_lib.cmbpo_set_ens_f16_min_rows(0) == 0
        assert hip_lib.cmbpo_set_ens_f16_row_tiles(rt) == 0
        mean, var = m.predict_ensemble(x)
    finally:
        hip_lib.cmbpo_set_ens_f16_row_tiles(0)
        hip_lib.cmbpo_set_ens_f16_min_rows(0)
        hip_lib.cmbpo_set_ens_matrix_path(before)
    rmean, rvar = refcpu.ens_forward(x, ws, bs, sc_in, sc_out)
    np.testing.assert_allclose(mean, rmean, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(var, rvar, rtol=2e-3, atol=1e-7)
    assert hip_lib.cmbpo_set_ens_f16_row_tiles(3) < 0


@pytest.mark.parametrize("task,n,forced", [("AntSafe-v2", 35000, 4), ("AntSafe-v2", 10000, 4), ("HalfCheetahSafe-v2", 10000, 2),
                                           ("AntSafe-v2", 5000, 1), ("AntSafe-v2", 5761, 2), ("HumanoidSafe-v2", 10000, 4)])
def test_ens_forward_tail_round_as_shorter_items(hip_lib, task, n, forced):
    """The full rounds of one item size followed by the leftovers at a smaller one, in a launch of their own (ens_h3.hip; on
    256 CUs with 7 members: 35 000 and 20 000 rows = 128-row items + a round of 64-row items, 10 000 and 5 000 rows = 128-row
    items + 32-row items, 5 761 rows and Humanoid shapes at 10 000 rows = 64-row items + 32-row items).  Every row against the oracle, and bitwise against the
    same forward with one item size forced (a single launch, no split)."""
    _cuda()
    rng = np.random.default_rng(4242 + n)
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, task)
    x = rng.standard_normal((n, obs_dim + act_dim)).astype(np.float32)
    mean, var = m.predict_ensemble(x)
    rmean, rvar = refcpu.ens_forward(x, ws, bs, sc_in, sc_out)
    np.testing.assert_allclose(mean, rmean, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(var, rvar, rtol=2e-3, atol=1e-6)
    assert hip_lib.cmbpo_set_ens_f16_row_tiles(forced) == 0
    try:
        mean4, var4 = m.predict_ensemble(x)
    finally:
        hip_lib.cmbpo_set_ens_f16_row_tiles(0)
    np.testing.assert_array_equal(mean, mean4)
    np.testing.assert_array_equal(var, var4)


def test_ens_forward_no_scalers_hidden128(hip_lib):
    _cuda()
    from cmbpo_amd import synthetic
    from cmbpo_amd.pens import PE
    rng = np.random.default_rng(5)
    E, I, O = 3, 26, 21
    ws, bs = synthetic.ensemble_weights(rng, E, I, 128, 2 * O, bias_scale=0.1)
    m = PE(I, O, hidden_dims=(128, 128), num_networks=E, num_elites=2, loss="MSPE", device="cuda:0")
    m.set_weights(ws, bs)
    x = rng.standard_normal((100, I)).astype(np.float32)
    mean, var = m.predict_ensemble(x)
    rmean, rvar = refcpu.ens_forward(x, ws, bs)
    np.testing.assert_allclose(mean, rmean, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(var, rvar, rtol=1e-3, atol=1e-7)


@pytest.mark.parametrize("n", [1, 100, 4097])
def test_width_256_forward_heads(hip_lib, n):
    """Hidden width 256 (the base config's default for the policy and both critics, configs/baseconfig/base.py:7,16,19; every
    shipped experiment overrides it to 128): the three heads of the forward on the general fp32-MFMA kernel."""
    _cuda()
    from cmbpo_amd import synthetic
    from cmbpo_amd.pens import PE
    from cmbpo_amd.cpo_policy import GaussianActor
    rng = np.random.default_rng(256 + n)
    H, obs_dim, act_dim, E = 256, 29, 8, 3
    x = rng.standard_normal((n, obs_dim)).astype(np.float32)
    # a critic (mean over all members)
    ws, bs = synthetic.ensemble_weights(rng, E, obs_dim, H, 1, bias_scale=0.1)
    sc_in, sc_out = synthetic.scaler(rng, obs_dim), synthetic.scaler(rng, 1, hit_clamp=False)
    v = PE(obs_dim, 1, hidden_dims=(H, H), num_networks=E, num_elites=2, loss="MSE", use_scaler_in=True, use_scaler_out=True,
           device="cuda:0")
    v.set_weights(ws, bs, sc_in, sc_out)
    np.testing.assert_allclose(v.predict(x), refcpu.ens_predict_mean(x, ws, bs, sc_in, sc_out), rtol=1e-4, atol=1e-4)
    # a probabilistic ensemble
    O = 11
    ws, bs = synthetic.ensemble_weights(rng, E, obs_dim, H, 2 * O, bias_scale=0.1)
    m = PE(obs_dim, O, hidden_dims=(H, H), num_networks=E, num_elites=2, loss="MSPE", device="cuda:0")
    m.set_weights(ws, bs)
    mean, var = m.predict_ensemble(x)
    rmean, rvar = refcpu.ens_forward(x, ws, bs)
    np.testing.assert_allclose(mean, rmean, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(var, rvar, rtol=1e-3, atol=1e-7)
    # the actor
    shapes = [(obs_dim, H), (H,), (H, H), (H,), (H, act_dim), (act_dim,), (act_dim,)]
    params = [(rng.standard_normal(sh) * (1.0 / np.sqrt(sh[0]) if len(sh) == 2 else 0.1)).astype(np.float32) for sh in shapes]
    params[6] = rng.uniform(-1.0, 0.0, act_dim).astype(np.float32)
    actor = GaussianActor(obs_dim, act_dim, (H, H), device="cuda:0")
    actor.set_params(params)
    eps = rng.standard_normal((n, act_dim)).astype(np.float32)
    dev = actor.device
    f = dict(dtype=torch.float32, device=dev)
    out = dict(pi=torch.empty((n, act_dim), **f), logp_pi=torch.empty(n, **f), mu=torch.empty((n, act_dim), **f),
               log_std=torch.empty((n, act_dim), **f))
    actor.forward_device(torch.from_numpy(x).to(dev), torch.from_numpy(eps).to(dev), out)
    ref = refcpu.policy_forward(x, params, eps)
    for k in ("pi", "mu", "logp_pi", "log_std"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4, err_msg=k)


def test_ens_forward_row_gather_and_split_inputs(hip_lib):
    """obs/act passed separately, rows addressed through an index list into branch slots."""
    dev = _cuda()
    rng = np.random.default_rng(11)
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, "AntSafe-v2")
    B = 300
    obs = rng.standard_normal((B, obs_dim)).astype(np.float32)
    act = rng.standard_normal((B, act_dim)).astype(np.float32)
    idx = np.sort(rng.choice(B, size=77, replace=False)).astype(np.int32)
    mean = torch.full((7, B, obs_dim + 1), float("nan"), device=dev)
    var = torch.full((7, B, obs_dim + 1), float("nan"), device=dev)
    m.predict_ensemble(torch.from_numpy(obs).to(dev), act=torch.from_numpy(act).to(dev),
                       row_idx=torch.from_numpy(idx).to(dev), out=(mean, var))
    rmean, rvar = refcpu.ens_forward(np.concatenate([obs, act], -1)[idx], ws, bs, sc_in, sc_out)
    got = mean.cpu().numpy()
    np.testing.assert_allclose(got[:, idx], rmean, rtol=2e-4, atol=2e-4)
    untouched = np.setdiff1d(np.arange(B), idx)
    assert np.isnan(got[:, untouched]).all()     # rows outside the list are never written


@pytest.mark.parametrize("ens_path", [0, 1, 2], indirect=True, ids=["fp32mfma", "splitbf16", "splitf16"])
@pytest.mark.parametrize("obs_dim", [29, 45, 11])
@pytest.mark.parametrize("n", [1, 32, 63, 65, 100, 1000, 32768 + 37])    # from 32768 rows: the split path's own kernel
def test_critic_predict_mean(hip_lib, ens_path, n, obs_dim):
    _cuda()
    from cmbpo_amd import synthetic
    from cmbpo_amd.pens import PE
    rng = np.random.default_rng(n + obs_dim)
    E = 3
    ws, bs = synthetic.ensemble_weights(rng, E, obs_dim, 128, 1, bias_scale=0.1)
    sc_in, sc_out = synthetic.scaler(rng, obs_dim), synthetic.scaler(rng, 1, hit_clamp=False)
    v = PE(obs_dim, 1, hidden_dims=(128, 128), num_networks=E, num_elites=2, loss="MSE",
           use_scaler_in=True, use_scaler_out=True, device="cuda:0")
    v.set_weights(ws, bs, sc_in, sc_out)
    x = rng.standard_normal((n, obs_dim)).astype(np.float32)
    got = v.predict(x)
    ref = refcpu.ens_predict_mean(x, ws, bs, sc_in, sc_out)
    assert got.shape == ref.shape == (n, 1)
    np.testing.assert_allclose(got, ref, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("ens_path", [0, 2], indirect=True, ids=["fp32mfma", "splitf16"])
@pytest.mark.parametrize("n", [333, 1, 129, 4100])
@pytest.mark.parametrize("task", ["AntSafe-v2", "HumanoidSafe-v2", "HopperSafe-v2"])
def test_policy_forward(hip_lib, ens_path, task, n):
    """cmbpo_policy_forward on both matrix paths (the general fp32-MFMA kernel; policy_f16.hip: one wave per 32-row tile,
    three f16 MFMAs per product), with and without a row list."""
    _cuda()
    from cmbpo_amd import synthetic
    from cmbpo_amd.cpo_policy import GaussianActor
    rng = np.random.default_rng(3)
    obs_dim, act_dim = synthetic.ENV_DIMS[task]
    params = synthetic.policy_params(rng, obs_dim, act_dim)
    params[1] = (rng.standard_normal(128) * 0.1).astype(np.float32)
    params[5] = (rng.standard_normal(act_dim) * 0.1).astype(np.float32)
    params[6] = rng.uniform(-1.0, 0.0, act_dim).astype(np.float32)
    actor = GaussianActor(obs_dim, act_dim, (128, 128), device="cuda:0")
    actor.set_params(params)
    obs = rng.standard_normal((n, obs_dim)).astype(np.float32)
    obs[n // 2] *= 1e3                                     # one row far outside the others' range (its own lift)
    eps = rng.standard_normal((n, act_dim)).astype(np.float32)
    dev = actor.device
    f = dict(dtype=torch.float32, device=dev)
    out = dict(pi=torch.empty((n, act_dim), **f), logp_pi=torch.empty(n, **f),
               mu=torch.empty((n, act_dim), **f), log_std=torch.empty((n, act_dim), **f))
    actor.forward_device(torch.from_numpy(obs).to(dev), torch.from_numpy(eps).to(dev), out)
    ref = refcpu.policy_forward(obs, params, eps)
    for k in ("pi", "mu", "logp_pi", "log_std"):
        np.testing.assert_allclose(out[k].cpu().numpy(), ref[k], rtol=1e-4, atol=1e-4, err_msg=k)
    np.testing.assert_array_equal(out["log_std"].cpu().numpy(), ref["log_std"])
    if n >= 129:
        # a row list (the sampler's alive list): only the listed slots are evaluated and written
        idx = np.sort(rng.choice(n, size=n // 3, replace=False)).astype(np.int32)
        out2 = {k: torch.full_like(v, -7.0) for k, v in out.items()}
        actor.forward_device(torch.from_numpy(obs).to(dev), torch.from_numpy(eps).to(dev), out2,
                             row_idx=torch.from_numpy(idx).to(dev), n_rows=len(idx))
        rest = np.setdiff1d(np.arange(n), idx)
        for k in ("pi", "mu", "logp_pi"):
            got = out2[k].cpu().numpy()
            np.testing.assert_array_equal(got[idx], out[k].cpu().numpy()[idx], err_msg=k)
            assert np.all(got[rest] == -7.0), k


def _post_inputs(rng, task, n, E=7):
    from cmbpo_amd import synthetic
    obs_dim, act_dim = synthetic.ENV_DIMS[task]
    obs = synthetic.start_states(rng, n, task)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    mean = (rng.standard_normal((E, n, obs_dim + 1)) * 0.3).astype(np.float32)
    var = np.exp(rng.uniform(-12, 1, (E, n, obs_dim + 1))).astype(np.float32)
    var[:, ::7, 3] = 0.0                                   # std = 0 -> log clip at -100
    var[:, 1::11, 5] = np.float32(1e-30)
    inds = rng.choice(np.array([0, 2, 3, 5, 6], np.int32), size=n)
    return obs_dim, act_dim, obs, act, mean, var, inds.astype(np.int32)


def _run_post(task, obs, act, mean, var, inds, obs_dim, act_dim):
    from cmbpo_amd import _lib
    dev = torch.device("cuda:0")
    n, E = obs.shape[0], mean.shape[0]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    f = dict(dtype=torch.float32, device=dev)
    out = dict(next_obs=torch.empty((n, obs_dim), **f), rew=torch.empty(n, **f),
               term=torch.empty(n, dtype=torch.uint8, device=dev), cost=torch.empty(n, **f),
               dkl_path=torch.empty(n, **f), ep_var_mean=torch.empty(n, **f),
               ep_var=torch.empty((n, obs_dim), **f))
    d = [t(obs), t(act), t(mean), t(var), t(inds)]
    _lib.check(_lib.lib().cmbpo_fakeenv_post(
        _lib.TASK_IDS.get(task, 0), E, obs_dim, act_dim, _lib.ptr(d[2]), _lib.ptr(d[3]), n,
        _lib.ptr(d[0]), _lib.ptr(d[1]), _lib.ptr(d[4]), None, None, n,
        _lib.ptr(out["next_obs"]), _lib.ptr(out["rew"]), _lib.ptr(out["term"]), _lib.ptr(out["cost"]),
        _lib.ptr(out["dkl_path"]), _lib.ptr(out["ep_var_mean"]), _lib.ptr(out["ep_var"]),
        _lib.current_stream()), "cmbpo_fakeenv_post")
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


@pytest.mark.parametrize("E", [7, 5, 3])     # the kernel is compiled for 7 and 5 members; any other size at run time
@pytest.mark.parametrize("task", ["AntSafe-v2", "HalfCheetahSafe-v2", "HopperSafe-v2", "HumanoidSafe-v2"])
@pytest.mark.parametrize("n", [1, 8, 1001])
def test_fakeenv_post_matches_oracle(hip_lib, task, n, E):
    _cuda()
    rng = np.random.default_rng(zlib.crc32(f"{task}/{n}/post".encode()))
    obs_dim, act_dim, obs, act, mean, var, inds = _post_inputs(rng, task, n, E=E)
    if E < 7:
        inds = (inds % E).astype(np.int32)
    if task == "AntSafe-v2" and n > 8:
        # force every branch of the termination rule, incl. the precedence quirk and non-finite rows
        mean[:, 0, 0] = 5.0       # z > 1           -> gate 0 -> not done
        mean[:, 1, 0] = -5.0      # z < 0.2         -> gate 0 -> not done
        mean[:, 2, 2] = 2.0       # z_rot << -0.7, gate 1 (z ok) -> done
        obs[2, 0] = 0.5; mean[:, 2, 0] = 0.0
        mean[:, 3, 4] = np.nan    # non-finite, z_rot finite -> gate 0 -> not done
        mean[:, 4, 2] = np.inf    # z_rot = -inf, gate 0 -> 0 * -inf = nan -> done
        mean[:, 5, -2] = 10.0     # |y| > 3.2 -> cost 1 (obs slot is out_dim-2 = obs_dim-1)
    if n > 8:
        # members with an unbounded / undefined variance or a NaN mean: np.clip keeps the NaN of their own (i == i) KL
        # term, so the reference's KL of the branch is NaN (and `NaN >= dkl_lim` keeps the branch alive): same here
        var[1, 6, 3] = np.inf
        var[0, 7, 1] = np.nan
        mean[2, 8, 5] = np.nan
    got = _run_post(task, obs, act, mean, var, inds, obs_dim, act_dim)
    with np.errstate(all="ignore"):
        rn, rr, rt, info = refcpu.fake_env_step(obs, act, mean, var, inds, task)
    np.testing.assert_array_equal(got["next_obs"], rn)                       # bit-exact
    np.testing.assert_array_equal(got["rew"], rr[:, 0])                      # bit-exact
    np.testing.assert_array_equal(got["term"].astype(bool), rt[:, 0])        # bit-exact masks
    np.testing.assert_array_equal(got["cost"], np.asarray(info["cost"], np.float32)[:, 0])
    ok = np.isfinite(info["ensemble_dkl_path"])
    np.testing.assert_array_equal(np.isnan(got["dkl_path"]), np.isnan(info["ensemble_dkl_path"]))   # NaN where the reference's is
    np.testing.assert_array_equal(np.isinf(got["dkl_path"]), np.isinf(info["ensemble_dkl_path"]))
    if n > 8:
        assert np.isnan(got["dkl_path"][[6, 7, 8]]).all()
    np.testing.assert_allclose(got["dkl_path"][ok], info["ensemble_dkl_path"][ok], rtol=1e-4, atol=1e-7)
    evar = info["ensemble_ep_var"]
    okv = np.isfinite(evar)
    np.testing.assert_allclose(got["ep_var"][okv], evar[okv], rtol=1e-5, atol=1e-9)
    okm = np.isfinite(evar).all(-1)
    np.testing.assert_allclose(got["ep_var_mean"][okm], evar.mean(-1)[okm], rtol=1e-5, atol=1e-9)
    if task == "AntSafe-v2" and n > 8:
        assert list(got["term"][:5]) == [0, 0, 1, 0, 1]


def test_fake_env_step_end_to_end(hip_lib):
    """FakeEnv.step (HIP forward + post) vs oracle forward + oracle step; masks must be consistent
    with the kernel's own next_obs (bit-exact rule evaluation)."""
    _cuda()
    from cmbpo_amd.fake_env import FakeEnv
    rng = np.random.default_rng(21)
    task = "AntSafe-v2"
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, task)

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    class _Env:
        observation_space, action_space = _Space(obs_dim), _Space(act_dim)

    env = FakeEnv(_Env(), task, m, predicts_delta=True, predicts_rew=True, predicts_cost=False)
    from cmbpo_amd import synthetic
    n = 500
    obs = synthetic.start_states(rng, n, task)
    act = rng.uniform(-1, 1, (n, act_dim)).astype(np.float32)
    inds = rng.choice(np.asarray(m.elite_inds, np.int32), size=n).astype(np.int32)
    nobs, r, terms, info = env.step(obs, act, model_inds=inds)
    rmean, rvar = refcpu.ens_forward(np.concatenate([obs, act], -1), ws, bs, sc_in, sc_out)
    rn, rr, rt, rinfo = refcpu.fake_env_step(obs, act, rmean, rvar, inds, task)
    assert nobs.shape == (n, obs_dim) and r.shape == (n, 1) and terms.shape == (n, 1) and terms.dtype == bool
    np.testing.assert_allclose(nobs, rn, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(r, rr, rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(info["ensemble_dkl_path"], rinfo["ensemble_dkl_path"], rtol=5e-3, atol=1e-6)
    np.testing.assert_allclose(info["ensemble_ep_var"], rinfo["ensemble_ep_var"], rtol=5e-3, atol=1e-7)
    # rules re-evaluated by the oracle on the kernel's own next_obs: bit-exact
    np.testing.assert_array_equal(terms, refcpu.antsafe_term_fn(obs, act, nobs))
    np.testing.assert_array_equal(info["cost"], refcpu.antsafe_c_fn(obs, act, nobs).astype(np.float32))


@pytest.mark.parametrize("obs_dim", [20, 29, 47])
@pytest.mark.parametrize("n", [1, 31, 33, 257, 5000, 24576 + 45, 70001])   # from 24576 rows: members in turn, weights in LDS
def test_critic_pair_matches_oracle(hip_lib, n, obs_dim):
    """Both critics in one launch (three f16 MFMAs per float32 product, one wave per member; at large batches one member
    after the other with its weights in LDS) == PE.predict of each: the oracle's mean over all members, and the general
    kernel's value, through a row index list.  The two kernels carry the same arithmetic: bitwise equal values."""
    dev = _cuda()
    from cmbpo_amd import _lib, synthetic
    from cmbpo_amd.pens import PE
    rng = np.random.default_rng(zlib.crc32(f"pair/{n}/{obs_dim}".encode()))
    nets, refs = [], []
    B = n + 7
    obs = (rng.standard_normal((B, obs_dim)) * np.exp(rng.standard_normal((B, 1)))).astype(np.float32)   # rows of different scale
    idx = np.sort(rng.choice(B, size=n, replace=False)).astype(np.int32)
    for k in range(2):
        ws, bs = synthetic.ensemble_weights(rng, 3, obs_dim, 128, 1, bias_scale=0.1)
        sc_in, sc_out = synthetic.scaler(rng, obs_dim), synthetic.scaler(rng, 1)
        m = PE(obs_dim, 1, hidden_dims=(128, 128), num_networks=3, num_elites=2, loss="MSE", use_scaler_in=True,
               use_scaler_out=(k == 0), device="cuda:0")
        m.set_weights(ws, bs, sc_in, sc_out if k == 0 else None)
        nets.append(m)
        refs.append(refcpu.ens_predict_mean(obs[idx], ws, bs, sc_in, sc_out if k == 0 else None)[:, 0])
    lib = hip_lib
    assert lib.cmbpo_critic_pair_supported(nets[0].mlp.handle, nets[1].mlp.handle) == 1
    o = torch.from_numpy(obs).to(dev)
    ix = torch.from_numpy(idx).to(dev)
    out = [torch.full((B,), float("nan"), device=dev) for _ in range(2)]
    _lib.check(lib.cmbpo_critic_pair_predict(nets[0].mlp.handle, nets[1].mlp.handle, o.data_ptr(), obs_dim, ix.data_ptr(), None, n,
                                             out[0].data_ptr(), out[1].data_ptr(), _lib.current_stream()), "pair")
    for k in range(2):
        got = out[k].cpu().numpy()
        np.testing.assert_allclose(got[idx], refs[k], rtol=1e-4, atol=1e-4)
        assert np.isnan(got[np.setdiff1d(np.arange(B), idx)]).all()          # rows outside the list are never written
        single = nets[k].predict(obs[idx])[:, 0]                                # the general kernel (fp32 MFMAs)
        tol = 2e-5 if n < 20000 else 5e-5                                       # (the largest of 70 k differences sits further out)
        np.testing.assert_allclose(got[idx], single, rtol=tol, atol=tol)
    if n >= 24576:
        # the first 3000 listed rows through the small-batch kernel: the same bits
        sub = [torch.full((B,), float("nan"), device=dev) for _ in range(2)]
        _lib.check(lib.cmbpo_critic_pair_predict(nets[0].mlp.handle, nets[1].mlp.handle, o.data_ptr(), obs_dim, ix.data_ptr(), None,
                                                 3000, sub[0].data_ptr(), sub[1].data_ptr(), _lib.current_stream()), "pair")
        for k in range(2):
            np.testing.assert_array_equal(sub[k].cpu().numpy()[idx[:3000]], out[k].cpu().numpy()[idx[:3000]])
    # new weights are followed
    ws2, bs2 = synthetic.ensemble_weights(rng, 3, obs_dim, 128, 1, bias_scale=0.1)
    nets[1].set_weights(ws2, bs2, synthetic.scaler(rng, obs_dim), None)
    _lib.check(lib.cmbpo_critic_pair_predict(nets[0].mlp.handle, nets[1].mlp.handle, o.data_ptr(), obs_dim, ix.data_ptr(), None, n,
                                             out[0].data_ptr(), out[1].data_ptr(), _lib.current_stream()), "pair")
    np.testing.assert_allclose(out[1].cpu().numpy()[idx], nets[1].predict(obs[idx])[:, 0], rtol=2e-5, atol=2e-5)


def test_ens_matrix_paths_agree_and_follow_weight_updates(hip_lib):
    """The three matrix paths of the 512-wide forward differ by float32 rounding only (far inside the parity tolerance),
    and the split paths' bf16 / f16 weight images follow every change of the weights."""
    _cuda()
    rng = np.random.default_rng(77)
    m, ws, bs, sc_in, sc_out, obs_dim, act_dim = _dyn_model(rng, "AntSafe-v2")
    x = rng.standard_normal((1000, obs_dim + act_dim)).astype(np.float32)
    out = {}
    before = hip_lib.cmbpo_get_ens_matrix_path()
    hip_lib.cmbpo_set_ens_f16_min_rows(0)
    try:
        for path in (0, 1, 2):
            assert hip_lib.cmbpo_set_ens_matrix_path(path) == 0
            out[path] = m.predict_ensemble(x)
        scale = float(np.abs(out[0][0]).max())
        for path in (1, 2):
            assert float(np.abs(out[0][0] - out[path][0]).max()) <= 2e-5 * scale
            np.testing.assert_allclose(out[path][1], out[0][1], rtol=2e-4)
            assert float(np.abs(out[0][0] - out[path][0]).max()) > 0.0          # they ARE different instruction streams
        # new weights: both paths must see them
        ws2 = [(w * 1.1).astype(np.float32) for w in ws]
        m.set_weights(ws2, bs, sc_in, sc_out)
        ref = refcpu.ens_forward(x[:64], ws2, bs, sc_in, sc_out)
        for path in (0, 1, 2):
            hip_lib.cmbpo_set_ens_matrix_path(path)
            mean, var = m.predict_ensemble(x[:64])
            np.testing.assert_allclose(mean, ref[0], rtol=2e-4, atol=2e-4)
        assert hip_lib.cmbpo_set_ens_matrix_path(7) < 0
    finally:
        hip_lib.cmbpo_set_ens_matrix_path(before)
        hip_lib.cmbpo_set_ens_f16_min_rows(0)
