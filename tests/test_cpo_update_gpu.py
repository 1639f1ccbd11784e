"""GPU parity of the CPO update kernels (loss/grad, Fisher-vector product, line-search eval) and of the
whole update_policy against the oracle (torch-CPU autograd restatement of the TF graph + the update_pi
logic pinned by golden G7).

Tolerances (fp32, sums over N samples in a different order than the oracle):
  * gradients / FVP: |d| <= 1e-3*|ref| + 3e-5*max|ref|;  scalar losses / KL: rtol 2e-4, atol 1e-6;
  * full update: same OptimCase and BacktrackIters, duals (lam, nu) rtol 2e-2 (ten CG iterations amplify
    rounding), accepted parameters within 2e-2 of the oracle's step norm.
"""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)

from oracle import refupdate  # noqa: E402


@pytest.fixture(autouse=True, params=[1, 0], ids=["splitf16", "fp32mfma"])
def pi_path(request, hip_lib):
    """Every test of this file on both arithmetic paths of the 128 x 128 products (cmbpo_set_pi_matrix_path)."""
    before = hip_lib.cmbpo_get_pi_matrix_path()
    hip_lib.cmbpo_set_pi_matrix_path(request.param)
    yield request.param
    hip_lib.cmbpo_set_pi_matrix_path(before)


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _close(got, ref, rtol=1e-3, arel=3e-5, msg=""):
    np.testing.assert_allclose(got, ref, rtol=rtol, atol=arel * float(np.max(np.abs(ref))) + 1e-12, err_msg=msg)


def _setup(D, A, n, seed, cost_p=0.3, cadv_scale=1.0, T=35, hidden=128):
    from worlds import make_update_batch
    from cmbpo_amd.cpo_update import PolicyOps
    rng = np.random.default_rng(seed)
    params, batch = make_update_batch(rng, n, D, A, hidden, cost_p, cadv_scale, T)
    # move off theta_old a little so ratio != 1 and KL != 0 are exercised too
    graph = refupdate.PolicyGraph(D, A, batch, max_path_length=T, hidden=hidden)
    ops = PolicyOps(D, A, hidden, device="cuda:0")
    ops.set_params(params)
    ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"],
             batch["mu_old"], batch["log_std_old"])
    return rng, params, batch, graph, ops


@pytest.mark.parametrize("D,A,n,hidden", [(29, 8, 1037, 128), (20, 6, 64, 128), (47, 17, 500, 128), (21, 3, 31, 128),
                                          (64, 32, 257, 128), (3, 1, 100, 128),
                                          # configs/baseconfig/base.py:7: the default policy width (fp32 MFMAs on either path)
                                          (29, 8, 1037, 256), (47, 17, 300, 256), (3, 1, 33, 256)])
def test_loss_grad_fvp_eval_match_oracle(hip_lib, D, A, n, hidden):
    _need_gpu()
    rng, params, batch, graph, ops = _setup(D, A, n, seed=D * 100 + A, hidden=hidden)
    for shift in (0.0, 0.03):
        p = (params + shift * rng.standard_normal(params.shape)).astype(np.float32)
        ops.set_params(p)
        g_ref, b_ref, lo_ref, sc_ref = graph.grads(p)
        g, sg = ops.loss_grad(0)
        b, sb = ops.loss_grad(1)
        assert sg[0] == n and sb[0] == n
        _close(g, g_ref, msg="flat_g")
        _close(b, b_ref, msg="flat_b")
        np.testing.assert_allclose(-sg[1] / n, lo_ref, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(sg[2] / n, sc_ref, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(sg[4] / n, float(batch["cost"].mean()), rtol=1e-6)
        kl_ref, lo2, sc2 = graph.evals(p)
        s = ops.evals()
        np.testing.assert_allclose(s[3] / n, kl_ref, rtol=2e-4, atol=1e-7)
        np.testing.assert_allclose(-s[1] / n, lo2, rtol=2e-4, atol=1e-6)
        np.testing.assert_allclose(s[2] / n, sc2, rtol=2e-4, atol=1e-6)
    # Fisher-vector product at theta_old (where it equals TF's double back-prop of d_kl)
    ops.set_params(params)
    for k in range(3):
        v = rng.standard_normal(params.shape).astype(np.float32)
        if k == 2:
            v[:] = 0
            v[-A:] = 1.0          # log_std block only
        hv = ops.fvp(v) + np.float32(0.1) * v
        _close(hv, graph.hvp(params, v, 0.1), msg=f"hvp {k}")
        _close(hv, graph.fisher_vp(params, v, 0.1), msg=f"fisher {k}")


def test_fvp_is_linear_and_symmetric(hip_lib):
    """Size-independent properties at the bench size: H(a u + b v) = a Hu + b Hv, u.Hv = v.Hu, v.Hv >= 0."""
    _need_gpu()
    rng, params, batch, graph, ops = _setup(29, 8, 50000, seed=9)
    u = rng.standard_normal(params.shape).astype(np.float32)
    v = rng.standard_normal(params.shape).astype(np.float32)
    hu, hv = ops.fvp(u), ops.fvp(v)
    huv = ops.fvp((2 * u - 3 * v).astype(np.float32))
    _close(huv, 2 * hu - 3 * hv, rtol=2e-3, arel=1e-4)
    np.testing.assert_allclose(np.dot(u, hv), np.dot(v, hu), rtol=2e-3)
    assert np.dot(v, hv) >= 0 and np.dot(u, hu) >= 0


def test_fvp_power_of_two_scaling_is_exact_and_zero_stays_zero(hip_lib):
    """Every lift of the f16 path is a power of two derived from the operand it lifts: a direction scaled by 2^k gives the
    same pieces under a lift scaled by 2^-k, so the product scales bit for bit (as it does for fp32 MFMAs); a zero
    direction takes the lift-of-zero branches and returns exact zeros."""
    _need_gpu()
    rng, params, batch, graph, ops = _setup(29, 8, 777, seed=5)
    ops.loss_grad(0)      # saves the activations: the products below run on the saved-activation kernel
    v = rng.standard_normal(params.shape).astype(np.float32)
    hv = ops.fvp(v)
    for k in (-30, 20):
        a = np.float32(2.0 ** k)
        np.testing.assert_array_equal(ops.fvp(v * a), hv * a, err_msg=f"2^{k}")
    np.testing.assert_array_equal(ops.fvp(np.zeros_like(v)), np.zeros_like(v))


def test_fvp_wide_range_direction_agrees_between_the_matrix_paths(hip_lib, pi_path):
    """One lift per matrix: rows of a direction 1e7 apart in magnitude still come out at fp32 accuracy where it matters
    (errors measured against the largest component, as for every gradient-like vector here)."""
    _need_gpu()
    rng, params, batch, graph, ops = _setup(29, 8, 500, seed=6)
    v = rng.standard_normal(params.shape).astype(np.float32)
    scale = (10.0 ** rng.uniform(-4, 3, size=params.shape)).astype(np.float32)
    v = v * scale
    got = ops.fvp(v) + np.float32(0.1) * v
    _close(got, graph.fisher_vp(params, v, 0.1), msg="wide-range direction")


def test_fvp_and_gradient_are_additive_over_the_batch_at_a_million_samples(hip_lib):
    """Size-independent property at a batch the oracle cannot take: raw sums over a batch equal the sums over its two
    (unequal, tile-misaligned) parts -- every tile, the ragged last one and the per-workgroup partial vectors included."""
    _need_gpu()
    from worlds import make_update_batch
    from cmbpo_amd.cpo_update import PolicyOps
    D, A, n, cut = 29, 8, 1_000_003, 345_679
    rng = np.random.default_rng(11)
    params, batch = make_update_batch(rng, n, D, A, 128, 0.3, 1.0, 35)
    v = rng.standard_normal(params.shape).astype(np.float32)
    keys = ("obs", "act", "adv", "cadv", "logp_old", "cost", "mu_old", "log_std_old")

    def run(lo, hi):
        ops = PolicyOps(D, A, 128, device="cuda:0")
        ops.set_params(params)
        ops.bind(*[batch[k][lo:hi] for k in keys])
        g, s = ops.loss_grad(0)
        hv = ops.fvp(v)
        return g.astype(np.float64) * (hi - lo), hv.astype(np.float64) * (hi - lo), s

    g_all, h_all, s_all = run(0, n)
    g_a, h_a, s_a = run(0, cut)
    g_b, h_b, s_b = run(cut, n)
    for name, whole, parts in (("gradient", g_all, g_a + g_b), ("fvp", h_all, h_a + h_b)):
        np.testing.assert_allclose(whole, parts, rtol=2e-4, atol=2e-5 * float(np.abs(whole).max()), err_msg=name)
    np.testing.assert_allclose(s_all[:5], (s_a + s_b)[:5], rtol=1e-9)


SCENARIOS = [   # name, cost_p, cadv_scale, cost_lim, constrained, real_cost, seed
    ("feasible", 0.05, 1.0, 10.0, True, 3.0, 11),
    ("violating", 0.9, 1.0, 10.0, True, 25.0, 12),
    ("zero_costgrad", 0.0, 0.0, 10.0, True, 1.0, 13),
    ("unconstrained", 0.5, 1.0, 10.0, False, 10.0, 14),
    ("feasible_tight", 0.3, 3.0, 10.0, True, 9.0, 15),
]


@pytest.mark.parametrize("name,cost_p,cadv_scale,cost_lim,constrained,real_cost,seed,hidden",
                         [sc + (128,) for sc in SCENARIOS] + [sc + (256,) for sc in SCENARIOS[:2]])
def test_update_policy_matches_oracle(hip_lib, name, cost_p, cadv_scale, cost_lim, constrained, real_cost, seed, hidden):
    _need_gpu()
    from worlds import make_update_batch
    from cmbpo_amd.cpo_policy import CPOPolicy
    D, A, n, T = 29, 8, 4000, 35
    rng = np.random.default_rng(seed)
    params, batch = make_update_batch(rng, n, D, A, hidden, cost_p, cadv_scale, T)

    class _Space:
        def __init__(self, d):
            self.shape = (d,)

    pol = CPOPolicy(_Space(D), _Space(A), a_hidden_layer_sizes=(hidden, hidden), vf_hidden_layer_sizes=(128, 128),
                    vf_ensemble_size=3, vf_elites=2, vf_activation="swish", vf_loss="MSE", device="cuda:0",
                    constrain_cost=constrained, cost_lim=cost_lim, target_kl=0.01, max_path_length=T)
    pol.set_params(params)
    pol.real_c_buffer = [real_cost] * 300
    z = np.zeros(n, np.float32)
    buf = [batch["obs"], batch["act"], batch["adv"], batch["cadv"], z, z, batch["logp_old"], z, z, batch["cost"],
           batch["log_std_old"], batch["mu_old"]]
    info = pol.update_policy(buf)
    new_params = pol.actor.get_flat_params()

    graph = refupdate.PolicyGraph(D, A, batch, max_path_length=T, hidden=hidden)
    agent = refupdate.AgentState(T, constrained=constrained)

    def grads():
        g, b, lo, sc = graph.grads(params)
        return g, b, lo, sc, float(graph.cur_cret_avg())

    ref_params, ref = refupdate.update_pi(
        agent, dict(grads=grads, Hx=lambda v: graph.hvp(params, v, 0.1),
                    set_and_eval=lambda p: graph.evals(np.asarray(p, np.float32))),
        params, 0.01, cost_lim, [real_cost] * 300)
    assert info["OptimCase"] == ref["OptimCase"], name
    assert info["BacktrackIters"] == ref["BacktrackIters"] and info["accepted"] == ref["accepted"], name
    for k in ("Optim_c", "Optim_q", "Optim_r", "Optim_s", "Optim_Lam", "Optim_Nu", "Margin"):
        np.testing.assert_allclose(float(info[k]), float(ref[k]), rtol=2e-2, atol=1e-7, err_msg=f"{name}:{k}")
    step_norm = float(np.linalg.norm(ref["step"])) + 1e-12
    assert float(np.linalg.norm(info["step"] - ref["step"])) <= 2e-2 * step_norm, name
    assert float(np.linalg.norm(new_params - ref_params)) <= 2e-2 * step_norm + 1e-6, name
    st = pol.logger.stored
    for k in ("LossPi", "SurrCost", "SurrAdv", "Entropy", "KL", "LossPiDelta", "SurrCostDelta", "Optim_A", "OptimCase",
              "BacktrackIters"):
        assert k in st, k
    if ref["accepted"]:
        assert float(st["KL"]) <= 0.01 * 1.05
    # the rollout actor was re-packed ON THE DEVICE from the parameters the update left there (cmbpo_mlp_load_policy_flat):
    # the same forward, bit for bit, as an actor that loads the same parameters through the host
    from cmbpo_amd.cpo_policy import GaussianActor
    fresh = GaussianActor(D, A, (hidden, hidden), device="cuda:0")
    fresh.set_params(new_params)
    g = torch.Generator(device="cuda").manual_seed(seed)
    obs_t = torch.randn(300, D, device="cuda", generator=g)
    eps_t = torch.randn(300, A, device="cuda", generator=g)

    def fwd(actor):
        out = {"pi": torch.empty(300, A, device="cuda"), "logp_pi": torch.empty(300, device="cuda"),
               "mu": torch.empty(300, A, device="cuda"), "log_std": torch.empty(300, A, device="cuda")}
        actor.forward_device(obs_t, eps_t, out)
        return out
    o_upd, o_host = fwd(pol.actor), fwd(fresh)
    for k in o_upd:
        assert torch.equal(o_upd[k], o_host[k]), k
    np.testing.assert_array_equal(np.asarray(pol.ops.get_params()), new_params)


def test_cg_solve_graph_matches_eager_loop(hip_lib):
    """cmbpo_pi_cg_solve with the hipGraph replay == the same solve with the iterations launched one by one."""
    _need_gpu()
    from cmbpo_amd import _lib
    rng, params, batch, graph, ops = _setup(29, 8, 5000, seed=3)
    b = torch.from_numpy(rng.standard_normal(params.shape).astype(np.float32)).cuda()
    outs = []
    for use_graph in (True, False, True):
        ops.use_graph = use_graph
        x = torch.zeros_like(b)
        ops.cg_dev(b, x, 0.1)
        outs.append(x.cpu().numpy().copy())
    np.testing.assert_array_equal(outs[0], outs[1])        # same kernels, same order: bitwise equal
    np.testing.assert_array_equal(outs[0], outs[2])
    assert _lib.lib().cmbpo_pi_cg_graph_launches() >= 2    # capture is available on this stack and was replayed
    # repeated solves with the same buffers replay ONE captured graph (the cache key compares equal: no re-capture)
    ops.use_graph = True
    x = torch.zeros_like(b)
    ops.cg_dev(b, x, 0.1)
    caps = _lib.lib().cmbpo_pi_cg_graph_captures()
    for _ in range(3):
        x.zero_()
        ops.cg_dev(b, x, 0.1)
    assert _lib.lib().cmbpo_pi_cg_graph_captures() == caps
    np.testing.assert_array_equal(x.cpu().numpy(), outs[0])
    # and it is the CG of the oracle's operator
    ref = refupdate.cg(lambda v: graph.hvp(params, v, 0.1), b.cpu().numpy()) if hasattr(refupdate, "cg") else None
    if ref is not None:
        _close(outs[0], ref, rtol=2e-2, arel=2e-3)


@pytest.mark.parametrize("D,A,n,hidden", [(29, 8, 5003, 128), (47, 17, 500, 128), (21, 3, 31, 128), (29, 8, 2000, 256)])
def test_fvp_from_saved_activations_is_bit_identical(hip_lib, D, A, n, hidden):
    """cmbpo_pi_keep_activations: the Fisher-vector products that read the hidden activations cmbpo_pi_loss_grad saved
    == the ones that recompute the forward chain, bit for bit (eager products and the whole CG solve); the saved
    images are dropped with the parameters they belong to."""
    _need_gpu()
    from cmbpo_amd import _lib
    rng, params, batch, graph, ops = _setup(D, A, n, seed=7 * D + A, hidden=hidden)
    uses = lambda: _lib.lib().cmbpo_pi_saved_activation_uses(ops._h)
    v = rng.standard_normal(params.shape).astype(np.float32)
    b = torch.from_numpy(rng.standard_normal(params.shape).astype(np.float32)).cuda()
    hv_re = ops.fvp(v)                                   # nothing saved since set_params: recomputed
    x_re = torch.zeros_like(b)
    ops.cg_dev(b, x_re, 0.1)
    assert uses() == 0
    g0, _ = ops.loss_grad(0)                             # saves h1 / h2 of the batch
    hv_sv = ops.fvp(v)
    assert uses() == 1
    np.testing.assert_array_equal(hv_sv, hv_re)
    x_sv = torch.zeros_like(b)
    ops.cg_dev(b, x_sv, 0.1)
    assert uses() > 1
    np.testing.assert_array_equal(x_sv.cpu().numpy(), x_re.cpu().numpy())
    g1, _ = ops.loss_grad(0)                             # the gradient itself is unchanged by saving / not saving
    np.testing.assert_array_equal(g0, g1)
    # new parameters: the saved images are stale and must not be used
    p2 = (params + 0.05 * rng.standard_normal(params.shape)).astype(np.float32)
    ops.set_params(p2)
    before = uses()
    hv2 = ops.fvp(v)
    assert uses() == before
    ops.loss_grad(1)                                     # either gradient call saves
    hv2_sv = ops.fvp(v)
    assert uses() == before + 1
    np.testing.assert_array_equal(hv2_sv, hv2)
    assert np.max(np.abs(hv2 - hv_re)) > 0
    # switched off: never read
    ops.keep_activations = False
    ops.bind(batch["obs"], batch["act"], batch["adv"], batch["cadv"], batch["logp_old"], batch["cost"],
             batch["mu_old"], batch["log_std_old"])
    ops.loss_grad(0)
    before = uses()
    np.testing.assert_array_equal(ops.fvp(v), hv2)
    assert uses() == before
