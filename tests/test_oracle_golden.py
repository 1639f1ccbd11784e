"""CPU: pin the oracle (oracle/refcpu.py) against golden vectors produced by the reference's own code
(tests/golden/make_golden.py, run in the build container against /root/reference).

Everything here is NumPy-vs-NumPy, so the bar is bit-exact unless a comment says otherwise.
"""
import os

import numpy as np
import pytest

from oracle import refcpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False)


def test_g1_statics_bit_exact():
    g = _load("g1_statics")
    with np.errstate(all="ignore"):
        for key, task in (("AntSafe", "AntSafe-v2"), ("HalfCheetahSafe", "HalfCheetahSafe-v2")):
            obs, act, nxt = g[key + "_obs"], g[key + "_act"], g[key + "_next"]
            term = refcpu.TERMS_BY_TASK[task](obs, act, nxt)
            cost = refcpu.COST_BY_TASK[task](obs, act, nxt)
            np.testing.assert_array_equal(term, g[key + "_term"])
            np.testing.assert_array_equal(np.asarray(cost, np.float64), np.asarray(g[key + "_cost"], np.float64))
            assert term.dtype == bool and term.shape == (obs.shape[0], 1)
    # the precedence quirk: in-range z, z_rot < -0.7 -> done; out-of-range z -> never done
    nxt, term = g["AntSafe_next"], g["AntSafe_term"][:, 0]
    assert term[5] and not term[2] and not term[3]
    assert term[7]           # gate 0 * z_rot(-inf) = nan -> `nan >= -0.7` is False -> done
    assert not term[6]       # non-finite elsewhere: gate 0, finite z_rot -> not done
    assert g["no_done"].dtype == bool and not g["no_done"].any()


def test_g2_average_dkl_bit_exact():
    g = _load("g2_dkl")
    with np.errstate(all="ignore"):
        d = refcpu.average_dkl(g["mu"], g["std"])
        k = refcpu.gaussian_kl_np(g["mu"][0], np.log(g["std"][0] + 1e-3), g["mu"][1], np.log(g["std"][1] + 1e-3))
    np.testing.assert_array_equal(d, g["average_dkl"])
    np.testing.assert_array_equal(k, g["pair_kl"])
    assert d.dtype == g["average_dkl"].dtype


def test_g3_discount_cumsum_stats_cg():
    g = _load("g3_gae_stats_cg")
    y = refcpu.discount_cumsum(g["x32"], 0.99, 0.95)
    assert y.dtype == np.float64 == g["gae_r"].dtype     # lfilter promotes float32 input to float64
    np.testing.assert_array_equal(y, g["gae_r"])
    np.testing.assert_array_equal(refcpu.discount_cumsum(g["x32"], 0.97, 0.5), g["gae_c"])
    np.testing.assert_array_equal(refcpu.discount_cumsum(g["x1"], 0.99, 0.95), g["gae_1d"])
    m, s = refcpu.mpi_statistics_scalar(g["stat_in"])
    assert m == g["stat_mean"] and s == g["stat_std"]
    A = g["cg_A"]
    x = refcpu.cg(lambda p: (A @ p).astype(np.float32), g["cg_b"].copy())
    np.testing.assert_array_equal(x, g["cg_x"])
    # known answers: constant rewards closed form, CG exact on a 2x2 SPD system
    c = 0.99 * 0.95
    closed = np.array([(1 - c ** (5 - t)) / (1 - c) for t in range(5)])
    np.testing.assert_allclose(refcpu.discount_cumsum(np.ones(5), 0.99, 0.95), closed, rtol=1e-14)
    M = np.array([[4.0, 1.0], [1.0, 3.0]])
    np.testing.assert_allclose(refcpu.cg(lambda p: M @ p, np.array([1.0, 2.0])), np.linalg.solve(M, [1.0, 2.0]),
                               rtol=1e-6)


def oracle_from_world(w, task, T, mode, dkl_lim):
    model = lambda x: refcpu.ens_forward(x, w["ws"], w["bs"], w["sc_in"], w["sc_out"])
    policy = lambda obs, eps: refcpu.policy_forward(obs, w["pol"], eps)
    v = lambda obs: refcpu.ens_predict_mean(obs, *w["v"])[:, 0]
    vc = lambda obs: refcpu.ens_predict_mean(obs, *w["vc"])[:, 0]
    return refcpu.RolloutOracle(model, policy, v, vc, task, w["obs_dim"], w["act_dim"], T, mode, dkl_lim)


def replay_trace(g, make_world):
    task, B, T = str(g["task"]), int(g["B"]), int(g["T"])
    w = make_world(int(g["seed"]), task, int(g["hidden"]), out_scale=float(g["out_scale"]),
                   q_boost=float(g["q_boost"]))
    orc = oracle_from_world(w, task, T, str(g["mode"]), float(g["dkl_lim"]))
    orc.reset(g["start"])
    budget = int(g["budget"]) or None
    alive, totals, ratios = [], [], []
    with np.errstate(all="ignore"):
        for s in range(len(g["n_rows"])):
            n = int(g["n_rows"][s])
            ratio = orc.sample(g["eps"][s, :n], g["inds"][s, :n], max_samples=budget)
            alive.append(orc.alive.copy())
            totals.append(orc.tot["samples"])
            ratios.append(ratio)
        orc.finish_all()
        res, diag = orc.get()
    return orc, np.array(alive), np.array(totals), np.array(ratios), res, diag


TRACES = ["g5_trace_ant_unc", "g5_trace_ant_term", "g5_trace_hcs_sched", "g5_trace_hopper_budget",
          "g5_trace_humanoid_512"]


@pytest.mark.parametrize("name", TRACES)
def test_g5_rollout_oracle_matches_reference_trace(name):
    import sys
    sys.path.insert(0, GOLD)
    from worlds import build_world
    g = _load(name)
    orc, alive, totals, ratios, res, diag = replay_trace(g, build_world)
    np.testing.assert_array_equal(alive, g["alive"])                    # masks per step: bit-exact
    np.testing.assert_array_equal(totals, g["total_samples"])
    np.testing.assert_allclose(ratios, g["alive_ratio"], rtol=0, atol=0)
    assert diag["poolm_batch_size"] == int(g["poolm_batch_size"])
    for k, arr in zip(NAMES, res):
        np.testing.assert_array_equal(arr, g["get_" + k], err_msg=k)     # same NumPy ops: bit-exact
        assert arr.dtype == g["get_" + k].dtype, k
    assert diag["poolm_ret_mean"] == g["poolm_ret_mean"] and diag["poolm_cret_mean"] == g["poolm_cret_mean"]
    d = orc.diagnostics()
    for k, v in d.items():
        # float32 running sums in the reference (python int 0 + np.float32): order-of-accumulation level
        np.testing.assert_allclose(v, float(g["diag_" + k.replace("/", "__")]), rtol=1e-6, err_msg=k)
    np.testing.assert_allclose(orc.dkl_acc, g["dkl_acc"], rtol=1e-12)


def test_g7_update_pi_logic_matches_reference():
    """oracle.refupdate.update_pi vs the reference's CPOAgent.update_pi (recorded with a fake session that
    evaluates the same restated graph): case selection, dual variables, step, backtracking, margin."""
    from oracle import refupdate
    g = _load("g7_update_pi")
    D, A, H, T = int(g["obs_dim"]), int(g["act_dim"]), int(g["hidden"]), int(g["T"])
    cases = set()
    for si, name in enumerate(g["names"]):
        pre = f"s{si}_"
        batch = {k: g[pre + "b_" + k] for k in ("obs", "act", "adv", "cadv", "logp_old", "cost", "mu_old",
                                                 "log_std_old")}
        graph = refupdate.PolicyGraph(D, A, batch, max_path_length=T, hidden=H)
        params = g[pre + "params"]
        calls = dict(hvp=0, evals=0)

        def Hx(v):
            calls["hvp"] += 1
            return graph.hvp(params, v, 0.1)

        def set_and_eval(p):
            calls["evals"] += 1
            return graph.evals(np.asarray(p, np.float32))

        def grads():
            gg, bb, lo, sc = graph.grads(params)
            return gg, bb, lo, sc, float(graph.cur_cret_avg())

        agent = refupdate.AgentState(T, constrained=bool(g[pre + "constrained"]))
        agent.margin = float(g[pre + "margin_in"])
        with np.errstate(all="ignore"):
            new_params, info = refupdate.update_pi(agent, dict(grads=grads, Hx=Hx, set_and_eval=set_and_eval), params,
                                                   0.01, float(g[pre + "cost_lim"]), [float(g[pre + "real_cost"])] * 300)
        assert info["OptimCase"] == int(g[pre + "OptimCase"]), name
        assert info["BacktrackIters"] == int(g[pre + "BacktrackIters"]), name
        assert calls["hvp"] == int(g[pre + "hvp_calls"]) and calls["evals"] == int(g[pre + "eval_calls"]), name
        for k in ("Optim_A", "Optim_B", "Optim_c", "Optim_q", "Optim_r", "Optim_s", "Optim_Lam", "Optim_Nu", "Margin"):
            np.testing.assert_allclose(np.float64(info[k]), g[pre + k], rtol=1e-12, atol=0, err_msg=f"{name}:{k}")
        np.testing.assert_array_equal(np.asarray(new_params, np.float32), g[pre + "new_params"], err_msg=name)
        cases.add(info["OptimCase"])
    assert cases == {0, 1, 2, 3, 4}


def test_oracle_graph_analytic_cross_checks():
    """The TF half has no reference outputs to pin against ('parity unpinned'): check the restated graph
    against central finite differences (float64) and the explicit Fisher form instead."""
    import sys
    import torch
    sys.path.insert(0, GOLD)
    from worlds import make_update_batch
    from oracle import refupdate
    rng = np.random.default_rng(3)
    D, A, H, n, T = 5, 2, 16, 64, 10
    params, batch = make_update_batch(rng, n, D, A, H, 0.3, 1.0, T)
    graph = refupdate.PolicyGraph(D, A, batch, max_path_length=T, hidden=H, dtype=torch.float64, ent_reg=0.01)
    p64 = params.astype(np.float64)
    gg, bb, lo, sc = graph.grads(p64)
    v = rng.standard_normal(p64.shape)
    h = 1e-6
    fd = lambda f: (f(p64 + h * v) - f(p64 - h * v)) / (2 * h)
    ev = lambda q: graph.evals(q)
    np.testing.assert_allclose(np.dot(gg, v), fd(lambda q: ev(q)[1]), rtol=1e-6)
    np.testing.assert_allclose(np.dot(bb, v), fd(lambda q: ev(q)[2]), rtol=1e-6, atol=1e-10)
    # Hessian of the KL by finite differences of its gradient direction: v^T H v
    kl = lambda q: ev(q)[0]
    h2 = 1e-4
    vHv_fd = (kl(p64 + h2 * v) - 2 * kl(p64) + kl(p64 - h2 * v)) / h2 ** 2
    hv = graph.hvp(p64, v, damping=0.0)
    np.testing.assert_allclose(np.dot(v, hv), vHv_fd, rtol=1e-5)
    # at theta = theta_old (mu_old produced by the same parameters) the exact Hessian is the Fisher form
    # (mu_old was rounded to float32, so a (mu_old - mu) * d2mu term of order 1e-7 remains)
    np.testing.assert_allclose(graph.fisher_vp(p64, v, 0.1), graph.hvp(p64, v, 0.1), rtol=1e-4, atol=1e-6)
    assert abs(kl(p64)) < 1e-6      # KL(p || p) = 0 (up to the 1e-8 in the denominator)


def cpobuffer_oracle(g):
    """buffers/cpobuffer.py:179-207,249-290 restated with the oracle's GAE / statistics."""
    lengths = g["lengths"]
    adv, ret, cadv, cret = [np.zeros(int(lengths.sum()), np.float32) for _ in range(4)]
    lo = 0
    for p, L in enumerate(lengths):
        hi = lo + int(L)
        lv = np.zeros((1,)) if g["zero_val"][p] else g["last_val"][p:p + 1]
        a, r = refcpu.gae_rows(g["rew"][None, lo:hi], g["val"][None, lo:hi], lv[None, 0], 0.99, 0.95)
        ca, cr = refcpu.gae_rows(g["cost"][None, lo:hi], g["cval"][None, lo:hi], g["last_cval"][None, p], 0.97, 0.5)
        adv[lo:hi], ret[lo:hi], cadv[lo:hi], cret[lo:hi] = a[0], r[0], ca[0], cr[0]
        lo = hi
    m, s = refcpu.mpi_statistics_scalar(adv)
    cm, _ = refcpu.mpi_statistics_scalar(cadv)
    return (adv - m) / (s + 1e-8), cadv - cm, ret, cret


def test_g8_cpobuffer_oracle_matches_reference():
    g = _load("g8_cpobuffer")
    adv, cadv, ret, cret = cpobuffer_oracle(g)
    np.testing.assert_array_equal(ret, g["get_ret"])
    np.testing.assert_array_equal(cret, g["get_cret"])
    np.testing.assert_array_equal(adv.astype(np.float32), g["get_adv"])
    np.testing.assert_array_equal(cadv.astype(np.float32), g["get_cadv"])
