"""GPU: (1) the remaining reference call sites of the path (get_action_outs, compute_dynamics_dkl, compute_DKL,
run_diagnostics, update_real_c) against the oracle, and (2) size-independent properties of the rollout at the
bench size (B = 100 000 AntSafe branches), where the oracle is too slow to run.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)

from oracle import refcpu, refupdate  # noqa: E402


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _world(task, hidden=128, B=64, T=8, mode="uncertainty", seed=31):
    from worlds import build_world
    from test_rollout_sampler_gpu import hip_world
    w = build_world(seed, task, hidden)
    sampler, pool = hip_world(w, task, T, mode, float("inf"), B, hidden)
    return w, sampler, pool


def test_get_action_outs_numpy_in_numpy_out(hip_lib):
    _need_gpu()
    w, sampler, pool = _world("AntSafe-v2")
    pol = sampler.policy
    rng = np.random.default_rng(0)
    obs = rng.standard_normal((77, w["obs_dim"])).astype(np.float32)
    eps = rng.standard_normal((77, w["act_dim"])).astype(np.float32)
    out = pol.get_action_outs(obs, eps=eps)
    ref = refcpu.policy_forward(obs, w["pol"], eps)
    assert isinstance(out["pi"], np.ndarray) and out["pi"].shape == (77, w["act_dim"])
    np.testing.assert_allclose(out["pi"], ref["pi"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out["logp_pi"], ref["logp_pi"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out["pi_info"]["mu"], ref["mu"], rtol=1e-4, atol=1e-4)
    np.testing.assert_array_equal(out["pi_info"]["log_std"], ref["log_std"])
    np.testing.assert_allclose(out["v"], refcpu.ens_predict_mean(obs, *w["v"])[:, 0], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(out["vc"], refcpu.ens_predict_mean(obs, *w["vc"])[:, 0], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(pol.get_v(obs), out["v"], rtol=0, atol=0)
    # a single observation keeps the reference's batch-of-one convention (cpo_policy.py:786-794)
    one = pol.get_action_outs(obs[0], eps=eps[:1])
    assert one["pi"].shape == (1, w["act_dim"])


def test_compute_dynamics_dkl_matches_oracle(hip_lib):
    """samplers/model_sampler.py:151-167: calibration of the rollout DKL limit."""
    _need_gpu()
    w, sampler, pool = _world("HalfCheetahSafe-v2", B=200)
    from cmbpo_amd import synthetic
    rng = np.random.default_rng(4)
    obs0 = synthetic.start_states(rng, 200, "HalfCheetahSafe-v2")
    depth = 3
    # inject the same draws on both sides by seeding the policy / env generators and replaying them
    pol, env = sampler.policy, sampler.env
    draws = []
    orig_gao, orig_step = pol.get_action_outs, env.step

    def gao(o):
        n = o.shape[0]
        eps = rng.standard_normal((n, w["act_dim"])).astype(np.float32)
        inds = np.asarray(w["elites"], np.int32)[rng.integers(0, len(w["elites"]), n)]
        draws.append((eps, inds))
        return orig_gao(o, eps=eps)

    def step(o, a):
        return orig_step(o, a, model_inds=draws[-1][1])

    pol.get_action_outs, env.step = gao, step
    got = sampler.compute_dynamics_dkl(obs0, depth=depth)
    pol.get_action_outs, env.step = orig_gao, orig_step
    # oracle
    obs, tot_dkl, tot_n = obs0, 0.0, 0
    for k in range(depth):
        eps, inds = draws[k]
        out = refcpu.policy_forward(obs, w["pol"], eps)
        mean, var = refcpu.ens_forward(np.concatenate([obs, out["pi"]], -1), w["ws"], w["bs"], w["sc_in"], w["sc_out"])
        nobs, _, term, info = refcpu.fake_env_step(obs, out["pi"], mean, var, inds, "HalfCheetahSafe-v2")
        tot_dkl += info["ensemble_dkl_mean"] * obs.shape[0]
        tot_n += obs.shape[0]
        obs = nobs[~term[:, 0]]
    np.testing.assert_allclose(got, tot_dkl / (tot_n + 1e-8) * depth, rtol=5e-3)
    np.testing.assert_allclose(sampler.dyn_dkl, tot_dkl / (tot_n + 1e-8), rtol=5e-3)


def test_compute_dkl_run_diagnostics_update_real_c(hip_lib):
    _need_gpu()
    from worlds import make_update_batch
    from cmbpo_amd.cpo_policy import CPOPolicy
    D, A, n, T = 20, 6, 500, 35
    rng = np.random.default_rng(8)
    params, batch = make_update_batch(rng, n, D, A, 128, 0.2, 1.0, T)

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    pol = CPOPolicy(_Space(D), _Space(A), a_hidden_layer_sizes=(128, 128), vf_hidden_layer_sizes=(128, 128),
                    vf_ensemble_size=3, vf_elites=2, vf_activation="swish", vf_loss="MSE", device="cuda:0",
                    max_path_length=T, cost_lim=10)
    moved = (params + 0.05 * rng.standard_normal(params.shape)).astype(np.float32)
    pol.set_params(moved)
    graph = refupdate.PolicyGraph(D, A, batch, max_path_length=T, hidden=128)
    kl_ref, lo_ref, sc_ref = graph.evals(moved)
    # compute_DKL: 2-D and the [n_epochs, B, .] form of algorithms/cmbpo.py:241-242
    kl = pol.compute_DKL(batch["obs"], batch["mu_old"], batch["log_std_old"])
    np.testing.assert_allclose(kl, kl_ref, rtol=2e-4, atol=1e-7)
    kl3 = pol.compute_DKL(np.stack([batch["obs"]] * 2), np.stack([batch["mu_old"]] * 2), np.stack([batch["log_std_old"]] * 2))
    assert kl3.shape == (2,)
    np.testing.assert_allclose(kl3, [kl_ref, kl_ref], rtol=2e-4, atol=1e-7)
    z = np.zeros(n, np.float32)
    buf = [batch["obs"], batch["act"], batch["adv"], batch["cadv"], z, z, batch["logp_old"], z, z, batch["cost"],
           batch["log_std_old"], batch["mu_old"]]
    diag = pol.run_diagnostics(buf)
    np.testing.assert_allclose(diag["LossPi"], lo_ref, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(diag["SurrCost"], sc_ref, rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(diag["Entropy"], float(graph.ent(torch.as_tensor(moved))), rtol=1e-6)
    before = list(pol.real_c_buffer)
    pol.update_real_c(buf)
    assert len(pol.real_c_buffer) == 300 and pol.real_c_buffer[:-1] == before[1:]
    np.testing.assert_allclose(pol.real_c_buffer[-1], float(batch["cost"].mean()) * T, rtol=1e-6)


def test_rollout_properties_at_bench_size(hip_lib):
    """B = 100 000 AntSafe branches, 512-wide ensemble, 6-step horizon: properties that hold at any size."""
    _need_gpu()
    import bench
    from cmbpo_amd import synthetic
    task, B, T = "AntSafe-v2", 100000, 7
    w = bench.build_world(0, task)
    sampler, pool, env, policy = bench.build_hip(w, task, B, torch.device("cuda:0"), maxroll=T)
    rng = np.random.default_rng(5)
    start = synthetic.start_states(rng, B, task)
    sampler.reset(torch.from_numpy(start).cuda())
    alive_hist, first_obs = [], None
    while pool.n_alive > 0:
        idx = pool.t["alive_idx"][: pool.n_alive].clone()
        assert bool((idx[1:] > idx[:-1]).all())                       # ordered alive list, no duplicates
        alive_hist.append(pool.n_alive)
        _, _, _, info = sampler.sample()
        assert abs(info["alive_ratio"] - pool.n_alive / B) < 1e-12
    lens = pool.t["len"].cpu().numpy()
    diag = sampler.finish_all_paths()
    res, bdiag = pool.get(as_tensors=True)
    n = int(lens.sum())
    # no uncertainty / budget finishes in this mode: every branch alive at a step stores exactly one sample
    assert bdiag["poolm_batch_size"] == n == int(diag["msampler/samples_added"]) == sum(alive_hist)
    assert lens.max() == T - 1 and alive_hist[0] == B and all(a >= b for a, b in zip(alive_hist, alive_hist[1:]))
    obs, act, adv, cadv, ret, cret, logp, val, cval, cost, ls, mu = res
    assert obs.shape == (n, 29) and act.shape == (n, 8) and adv.shape == (n,)
    # branch-major order: the first sample of every branch is its start state
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    has = lens > 0
    np.testing.assert_array_equal(obs[torch.from_numpy(offs[has]).cuda()].cpu().numpy(), start[has])
    # normalisation: mean 0 / std 1 for adv, mean 0 for cadv (float32 statistics over 6e5 samples)
    assert abs(float(adv.double().mean())) < 1e-4 and abs(float(adv.double().std(unbiased=False)) - 1.0) < 1e-4
    assert abs(float(cadv.double().mean())) < 1e-5
    # ret = adv_raw + val, cret = cadv_raw + cval: finite, and cost in {0, 1}
    for t_ in (ret, cret, logp, val, cval, mu):
        assert bool(torch.isfinite(t_).all())
    assert set(np.unique(cost.cpu().numpy())) <= {0.0, 1.0}
    assert bool((ls == -0.5).all())
    # the buffer is reset and reusable: a second rollout from the same states with the same draws is identical
    assert pool.n_alive == B and pool.size == 0


def test_rollout_properties_at_config5_shard(hip_lib):
    """BASELINE config 5's per-rank shape (1 M branches x horizon 25 over 8 GPUs = 125 000 branches, maxroll 26) on one
    GPU, 'uncertainty' mode with a sample budget that binds: size-independent properties of the whole phase."""
    _need_gpu()
    import bench
    from cmbpo_amd import synthetic
    task, B, T = "AntSafe-v2", 125000, 26
    w = bench.build_world(0, task)
    sampler, pool, env, policy = bench.build_hip(w, task, B, torch.device("cuda:0"), maxroll=T, mode="uncertainty")
    start = synthetic.start_states(np.random.default_rng(6), B, task)
    start_t = torch.from_numpy(start).cuda()
    sampler.reset(start_t)
    lim = float(sampler.compute_dynamics_dkl(start_t[:5000], depth=5)) * bench.BIND_DKL_SCALE
    sampler.set_rollout_dkl(lim)
    budget = int(0.5 * B * (T - 1))
    sampler.reset(start_t)
    steps = 0
    while sampler.any_alive() and pool.has_room:
        k, info = sampler.sample_many(max_samples=budget)
        steps += k
    lens = pool.t["len"].cpu().numpy()
    diag = sampler.finish_all_paths()
    res, bdiag = pool.get(as_tensors=True)
    n = int(lens.sum())
    # the budget rule is exact: the step that would pass max_samples stores only what is left of it, the next one ends
    # every surviving branch (samplers/model_sampler.py:282-287)
    assert n == budget == int(diag["msampler/samples_added"]) == bdiag["poolm_batch_size"]
    assert sampler.n_budget_terminated > 0 and steps >= 12 and lens.max() <= T - 1
    obs, act, adv, cadv, ret, cret, logp, val, cval, cost, ls, mu = res
    assert obs.shape == (n, 29) and act.shape == (n, 8) and adv.shape == (n,)
    offs = np.concatenate([[0], np.cumsum(lens)])[:-1]
    has = lens > 0
    np.testing.assert_array_equal(obs[torch.from_numpy(offs[has]).cuda()].cpu().numpy(), start[has])
    # early termination takes the FIRST surviving rows in index order: among the branches the budget ended at the last
    # two steps, lower ids are never longer than higher ids that survived the same step
    assert abs(float(adv.double().mean())) < 1e-4 and abs(float(adv.double().std(unbiased=False)) - 1.0) < 1e-4
    assert abs(float(cadv.double().mean())) < 1e-5
    for t_ in (ret, cret, logp, val, cval, mu, obs, act):
        assert bool(torch.isfinite(t_).all())
    assert set(np.unique(cost.cpu().numpy())) <= {0.0, 1.0}
    assert pool.n_alive == B and pool.size == 0


@pytest.mark.parametrize("task,B,T", [("AntSafe-v2", 70001, 6), ("HumanoidSafe-v2", 66000, 9), ("HopperSafe-v2", 70000, 4),
                                      ("AntSafe-v2", 70001, 12)])
def test_get_after_a_short_ragged_rollout_matches_a_gather(hip_lib, task, B, T):
    """ModelBuffer.get() after 'uncertainty' rollouts that end within a few steps, at sizes where the flatten takes its
    64-branch tiles (<= 8 steps) and its 16-branch ones (11 steps): every one of the twelve outputs against a plain gather
    from the [step][branch] buffers in branch-major order (buffers/modelbuffer.py:184-226)."""
    _need_gpu()
    import bench
    from cmbpo_amd import synthetic
    w = bench.build_world(0, task)
    sampler, pool, env, policy = bench.build_hip(w, task, B, torch.device("cuda:0"), maxroll=T, mode="uncertainty")
    start = synthetic.start_states(np.random.default_rng(11), B, task)
    start_d = torch.from_numpy(start).cuda()
    # a limit a little under the rollouts' own uncertainty: branches die over the first steps, at different ones
    sampler.set_rollout_dkl(float(sampler.compute_dynamics_dkl(start_d[:5000], depth=3)) * 0.8)
    sampler.reset(start_d)
    while pool.n_alive > 0:
        sampler.sample()
    sampler.finish_all_paths()
    lens = pool.t["len"].clone()
    steps = int(lens.max())
    assert 1 <= steps <= T - 1 and int(lens.min()) < steps            # ragged
    raw = {k: pool.t[k].clone() for k in ("obs_buf", "act_buf", "mu_buf", "ls_buf", "adv_buf", "cadv_buf", "ret_buf",
                                          "cret_buf", "logp_buf", "val_buf", "cval_buf", "cost_buf")}
    res, diag = pool.get(as_tensors=True)
    # sample p of the output = step t of branch b, branches in order, steps in order
    b_idx = torch.repeat_interleave(torch.arange(B, device="cuda"), lens.long())
    first = torch.cumsum(lens.long(), 0) - lens.long()
    t_idx = torch.arange(b_idx.numel(), device="cuda") - first[b_idx]
    n = int(lens.sum())
    assert diag["poolm_batch_size"] == n == b_idx.numel()
    g = lambda name: raw[name][t_idx, b_idx]
    obs, act, adv, cadv, ret, cret, logp, val, cval, cost, ls, mu = res
    for got, name in ((obs, "obs_buf"), (act, "act_buf"), (ret, "ret_buf"), (cret, "cret_buf"), (logp, "logp_buf"),
                      (val, "val_buf"), (cval, "cval_buf"), (cost, "cost_buf"), (ls, "ls_buf"), (mu, "mu_buf")):
        assert torch.equal(got, g(name)), name
    a = g("adv_buf").double()
    torch.testing.assert_close(adv.double(), (a - a.mean()) / (a.std(unbiased=False) + 1e-8), rtol=1e-4, atol=1e-5)
    c = g("cadv_buf").double()
    torch.testing.assert_close(cadv.double(), c - c.mean(), rtol=1e-4, atol=1e-5)
