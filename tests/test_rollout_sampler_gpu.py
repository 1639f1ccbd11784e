"""GPU parity of the device-resident ModelSampler / ModelBuffer (HIP) against
  (a) golden traces recorded from the REFERENCE's own ModelSampler + ModelBuffer + FakeEnv code, and
  (b) the CPU oracle on fresh seeds / larger batches.

Masks, alive lists, sample counts and the branch-major order of get() must be bit-exact.  Continuous
values follow the fp32 tolerance of the forward kernels (the networks are evaluated by HIP here and by
NumPy in the reference run), amplified over a few steps of the recurrence: rtol/atol 2e-3 on values,
5e-3 on the normalised advantages.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

from oracle import refcpu  # noqa: E402

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
NAMES = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]
TOL = dict(obs=2e-3, act=2e-3, adv=5e-3, cadv=2e-3, ret=2e-3, cret=2e-3, logp=2e-3, val=2e-3, cval=2e-3,
           cost=0.0, log_std=0.0, mu=2e-3)


class _Space:
    def __init__(self, d):
        self.shape = (d,)


def hip_world(w, task, T, mode, dkl_lim, B, hidden, comm=None):
    from cmbpo_amd.cpo_policy import CPOPolicy
    from cmbpo_amd.fake_env import FakeEnv
    from cmbpo_amd.model_sampler import ModelSampler
    from cmbpo_amd.modelbuffer import ModelBuffer
    from cmbpo_amd.pens import PE
    D, A = w["obs_dim"], w["act_dim"]
    E = w["ws"][0].shape[0]
    model = PE(D + A, D + 1, hidden_dims=(hidden, hidden), num_networks=E, num_elites=len(w["elites"]),
               loss="MSPE", use_scaler_in=True, use_scaler_out=True, device="cuda:0")
    model.set_weights(w["ws"], w["bs"], w["sc_in"], w["sc_out"])
    model.set_elites(w["elites"])
    policy = CPOPolicy(_Space(D), _Space(A), a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128),
                       vf_ensemble_size=3, vf_elites=2, vf_activation="swish", vf_loss="MSE", device="cuda:0",
                       cost_gamma=0.97, cost_lam=0.5, lam=0.95, comm=comm)
    policy.actor.set_params(w["pol"])
    policy.v.set_weights(*w["v"])
    policy.vc.set_weights(*w["vc"])

    class _Env:
        observation_space, action_space = _Space(D), _Space(A)

    env = FakeEnv(_Env(), task, model, predicts_delta=True, predicts_rew=True, predicts_cost=False)
    pool = ModelBuffer(B, D, A, T, device="cuda:0", comm=comm)
    pool.initialize(policy.pi_info_shapes, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    sampler = ModelSampler(max_path_length=T, batch_size=B, rollout_mode=mode, comm=comm)
    sampler.initialize(env, policy, pool)
    sampler.set_rollout_dkl(dkl_lim)
    return sampler, pool


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.mark.parametrize("name", ["g5_trace_ant_unc", "g5_trace_ant_term", "g5_trace_hcs_sched",
                                  "g5_trace_hopper_budget", "g5_trace_humanoid_512"])
@pytest.mark.parametrize("ens_path", [0, 1, 2], indirect=True, ids=["fp32mfma", "splitbf16", "splitf16"])
def test_hip_sampler_reproduces_reference_trace(hip_lib, ens_path, name):
    _need_gpu()
    from worlds import build_world
    g = np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)
    task, B, T, hidden = str(g["task"]), int(g["B"]), int(g["T"]), int(g["hidden"])
    w = build_world(int(g["seed"]), task, hidden, out_scale=float(g["out_scale"]), q_boost=float(g["q_boost"]))
    sampler, pool = hip_world(w, task, T, str(g["mode"]), float(g["dkl_lim"]), B, hidden)
    sampler.reset(g["start"])
    budget = int(g["budget"]) or None
    for s in range(len(g["n_rows"])):
        n = int(g["n_rows"][s])
        assert pool.n_alive == n
        _, _, _, info = sampler.sample(max_samples=budget, eps=g["eps"][s, :n], model_inds=g["inds"][s, :n])
        np.testing.assert_array_equal(pool.alive_paths, g["alive"][s], err_msg=f"alive mask after step {s}")
        assert sampler._total_samples == g["total_samples"][s]
        assert info["alive_ratio"] == g["alive_ratio"][s]
    np.testing.assert_allclose(pool.t["dkl_acc"].cpu().numpy(), g["dkl_acc"], rtol=5e-3, atol=1e-9)
    diag = sampler.finish_all_paths()
    res, bdiag = pool.get()
    assert bdiag["poolm_batch_size"] == int(g["poolm_batch_size"])
    for k, arr in zip(NAMES, res):
        ref = g["get_" + k]
        assert arr.shape == ref.shape and arr.dtype == ref.dtype, k
        if TOL[k] == 0.0:
            np.testing.assert_array_equal(arr, ref, err_msg=k)       # cost masks / log_std copies: bit-exact
        else:
            np.testing.assert_allclose(arr, ref, rtol=TOL[k], atol=TOL[k], err_msg=k)
    np.testing.assert_allclose(bdiag["poolm_ret_mean"], float(g["poolm_ret_mean"]), rtol=2e-3, atol=2e-4)
    np.testing.assert_allclose(bdiag["poolm_cret_mean"], float(g["poolm_cret_mean"]), rtol=2e-3, atol=2e-4)
    for k in ("msampler/samples_added", "msampler/rollout_H_max"):
        assert diag[k] == float(g["diag_" + k.replace("/", "__")])
    for k in ("msampler/rollout_H_mean", "msampler/dyn_var_perstep", "msampler/cost_rate", "msampler/rew_rate",
              "msampler/v_mean", "msampler/cv_mean", "msampler/ens_DKL", "msampler/max_path_return",
              "msampler/max_dkl"):
        np.testing.assert_allclose(diag[k], float(g["diag_" + k.replace("/", "__")]), rtol=5e-3, atol=1e-6,
                                   err_msg=k)


def _oracle(w, task, T, mode, lim):
    model = lambda x: refcpu.ens_forward(x, w["ws"], w["bs"], w["sc_in"], w["sc_out"])
    policy = lambda obs, eps: refcpu.policy_forward(obs, w["pol"], eps)
    v = lambda obs: refcpu.ens_predict_mean(obs, *w["v"])[:, 0]
    vc = lambda obs: refcpu.ens_predict_mean(obs, *w["vc"])[:, 0]
    return refcpu.RolloutOracle(model, policy, v, vc, task, w["obs_dim"], w["act_dim"], T, mode, lim)


@pytest.mark.parametrize("task,B,T,hidden,budget", [
    ("AntSafe-v2", 1500, 10, 512, None),        # the production dynamics width, terminations + horizon
    ("HumanoidSafe-v2", 333, 6, 128, 1500),     # wide obs/act (3 output tiles), budget, no statics entry
    ("HalfCheetahSafe-v2", 2048, 5, 128, None),
    ("HalfCheetahSafe-v2", 300, 5, 128, -100),  # `if max_samples:` is true for a negative budget: every survivor is finished
])
def test_hip_sampler_matches_oracle_on_fresh_seeds(hip_lib, task, B, T, hidden, budget):
    _need_gpu()
    from worlds import build_world
    from cmbpo_amd import synthetic
    seed = 1234
    w = build_world(seed, task, hidden, q_boost=1.2 if task == "AntSafe-v2" else 0.0)
    rng = np.random.default_rng(seed + 1)
    start = synthetic.start_states(rng, B, task)
    sampler, pool = hip_world(w, task, T, "uncertainty", float("inf"), B, hidden)
    orc = _oracle(w, task, T, "uncertainty", float("inf"))
    sampler.reset(start)
    orc.reset(start)
    elites = np.asarray(w["elites"], np.int32)
    with np.errstate(all="ignore"):
        for s in range(T):
            n = pool.n_alive
            if n == 0:
                break
            assert n == int(orc.alive.sum())
            eps = rng.standard_normal((n, w["act_dim"])).astype(np.float32)
            inds = elites[rng.integers(0, len(elites), n)]
            ratio = orc.sample(eps, inds, max_samples=budget)
            _, _, _, info = sampler.sample(max_samples=budget, eps=eps, model_inds=inds)
            np.testing.assert_array_equal(pool.alive_paths, orc.alive, err_msg=f"step {s}")
            assert sampler._total_samples == orc.tot["samples"]
            assert info["alive_ratio"] == ratio
            if budget and orc.tot["samples"] >= .99 * budget:
                break
        orc.finish_all()
        sampler.finish_all_paths()
        ref, rdiag = orc.get()
    res, bdiag = pool.get()
    assert bdiag["poolm_batch_size"] == rdiag["poolm_batch_size"]
    for k, arr, r in zip(NAMES, res, ref):
        assert arr.shape == r.shape, k
        if TOL[k] == 0.0:
            np.testing.assert_array_equal(arr, r.astype(np.float32), err_msg=k)
        else:
            np.testing.assert_allclose(arr, r, rtol=TOL[k], atol=TOL[k], err_msg=k)


@pytest.mark.parametrize("task,B,T,mode,lim_scale,budget,stop_frac,min_ratio", [
    ("AntSafe-v2", 1000, 12, "uncertainty", 2.5, 9000, None, None),     # small-batch path (one-workgroup bookkeeping), budget
    ("AntSafe-v2", 1000, 12, "schedule", None, None, 0.5, 0.1),         # stop on total_samples / alive ratio (cmbpo.py:356-359)
    ("AntSafe-v2", 1000, 12, "schedule", None, None, -0.01, None),      # a threshold <= 0 is reached by the first step (as in the reference)
    ("HalfCheetahSafe-v2", 6000, 7, "uncertainty", 2.5, None, None, 0.1),  # > 4096 rows: separate bookkeeping calls + compaction
    ("HalfCheetahSafe-v2", 26000, 4, "schedule", None, None, None, None),  # >= 24576 rows: the critics' member-after-member kernel, the actor its last member
])
def test_sample_many_equals_a_loop_of_sample(hip_lib, task, B, T, mode, lim_scale, budget, stop_frac, min_ratio):
    """cmbpo_rollout_run (ModelSampler.sample_many) takes exactly the steps a Python loop of sample() takes: same
    stopping step, bit-identical buffers."""
    _need_gpu()
    from worlds import build_world
    from cmbpo_amd import synthetic
    w = build_world(77, task, 128, q_boost=1.2 if task == "AntSafe-v2" else 0.0)
    start = synthetic.start_states(np.random.default_rng(78), B, task)
    stop_total = None if stop_frac is None else stop_frac * B * T
    out = []
    for many in (False, True):
        sampler, pool = hip_world(w, task, T, mode, float("inf"), B, 128)
        if lim_scale is not None:
            # a limit some branches exceed: calibrate on one step of a throw-away sampler with the same seeds
            cal, _ = hip_world(w, task, T, mode, float("inf"), B, 128)
            cal._gen.manual_seed(5)
            cal.reset(start)
            _, _, _, info = cal.sample()
            sampler.set_rollout_dkl(lim_scale * float(np.median(info["ensemble_dkl_path"].cpu().numpy()[:B])))
        sampler._gen.manual_seed(5)
        sampler.reset(start)
        steps = 0
        if many:
            steps, info = sampler.sample_many(max_samples=budget, stop_total=stop_total, min_alive_ratio=min_ratio)
        else:
            while sampler.any_alive() and pool.has_room:
                _, _, _, info = sampler.sample(max_samples=budget)
                steps += 1
                if stop_total is not None and sampler._total_samples >= stop_total:
                    break
                if min_ratio is not None and info["alive_ratio"] <= min_ratio:
                    break
        state = (steps, pool.n_alive, pool.ptr, sampler._total_samples, info["alive_ratio"])
        diag = sampler.finish_all_paths()
        res, _ = pool.get()
        out.append((state, diag["msampler/samples_added"], res))
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    assert out[0][0][0] == 1 if (stop_frac is not None and stop_frac <= 0) else out[0][0][0] >= 2
    assert out[0][1] == out[1][1]
    for k, a, b in zip(NAMES, out[0][2], out[1][2]):
        np.testing.assert_array_equal(a, b, err_msg=k)


def test_modelbuffer_api_parity_with_host_arrays(hip_lib):
    """store_multiple / finish_path_multiple / get driven with host arrays (the reference's call
    pattern) against the oracle's GAE on ragged finish patterns: finish at ptr = 0, mid-way, and at
    the end, with float32 and float64-zero bootstraps."""
    _need_gpu()
    from cmbpo_amd.modelbuffer import ModelBuffer
    rng = np.random.default_rng(5)
    B, T, D, A = 37, 6, 5, 2
    buf = ModelBuffer(B, D, A, T, device="cuda:0")
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    alive = np.ones(B, bool)
    hist = {k: np.zeros((B, T), np.float32) for k in ("rew", "val", "cost", "cval", "logp")}
    obs_h = np.zeros((B, T, D), np.float32)
    length = np.zeros(B, int)
    boots = {}
    # finish three branches before anything is stored
    tm = np.zeros(B, bool); tm[[1, 5, 36]] = True
    buf.finish_path_multiple(tm, rng.standard_normal(3).astype(np.float32), rng.standard_normal(3).astype(np.float32))
    alive[[1, 5, 36]] = False
    for t in range(T):
        idx = np.flatnonzero(alive)
        n = len(idx)
        vals = {k: rng.standard_normal(n).astype(np.float32) for k in hist}
        obs = rng.standard_normal((n, D)).astype(np.float32)
        buf.store_multiple(obs, rng.standard_normal((n, A)).astype(np.float32), obs, vals["rew"], vals["val"],
                           vals["cost"], vals["cval"], np.zeros(n, np.float32), vals["logp"],
                           {"mu": np.zeros((n, A), np.float32), "log_std": np.zeros((n, A), np.float32)},
                           np.zeros(n, bool))
        for k in hist:
            hist[k][idx, t] = vals[k]
        obs_h[idx, t] = obs
        length[idx] = t + 1
        if t < T - 1:
            tm = rng.random(n) < 0.25
            if tm.any():
                zero = (t % 2 == 1)
                lv = np.zeros(int(tm.sum())) if zero else rng.standard_normal(int(tm.sum())).astype(np.float32)
                lcv = rng.standard_normal(int(tm.sum())).astype(np.float32)
                buf.finish_path_multiple(tm, lv, lcv)
                for b, a_, c_ in zip(idx[tm], lv, lcv):
                    boots[b] = (np.asarray(a_), np.float32(c_))
                alive[idx[tm]] = False
        np.testing.assert_array_equal(buf.alive_paths, alive)
    idx = np.flatnonzero(alive)
    lv, lcv = rng.standard_normal(len(idx)).astype(np.float32), rng.standard_normal(len(idx)).astype(np.float32)
    buf.finish_path_multiple(np.ones(len(idx), bool), lv, lcv)
    for b, a_, c_ in zip(idx, lv, lcv):
        boots[b] = (np.asarray(a_), np.float32(c_))
    assert buf.size == int(length.sum())
    res, diag = buf.get()
    # oracle GAE per branch, then the get() normalisation
    adv = np.zeros((B, T), np.float32); ret = adv.copy(); cadv = adv.copy(); cret = adv.copy()
    for b in range(B):
        L = length[b]
        if L == 0:
            continue
        a_, r_ = refcpu.gae_rows(hist["rew"][b:b + 1, :L], hist["val"][b:b + 1, :L], boots[b][0][None], 0.99, 0.95)
        ca_, cr_ = refcpu.gae_rows(hist["cost"][b:b + 1, :L], hist["cval"][b:b + 1, :L], boots[b][1][None], 0.97, 0.5)
        adv[b, :L], ret[b, :L], cadv[b, :L], cret[b, :L] = a_[0], r_[0], ca_[0], cr_[0]
    mask = np.arange(T)[None] < length[:, None]
    np.testing.assert_array_equal(res[4], ret[mask])        # ret: float64 recurrence + fp32 casts, bit-exact
    np.testing.assert_array_equal(res[5], cret[mask])
    np.testing.assert_array_equal(res[0], obs_h[mask])      # branch-major, time-minor order
    np.testing.assert_array_equal(res[6], hist["logp"][mask])
    m, s = refcpu.mpi_statistics_scalar(adv[mask])
    np.testing.assert_allclose(res[2], (adv[mask] - m) / (s + 1e-8), rtol=1e-5, atol=1e-6)
    cm, _ = refcpu.mpi_statistics_scalar(cadv[mask])
    np.testing.assert_allclose(res[3], cadv[mask] - cm, rtol=1e-5, atol=1e-6)
    assert diag["poolm_batch_size"] == int(mask.sum())
    # get() resets: everything alive again, nothing stored
    assert buf.alive_paths.all() and buf.size == 0 and buf.ptr == 0
