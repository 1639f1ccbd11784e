"""Seeded synthetic worlds shared by the golden-vector generator and the tests.

Pure functions of a seed: nothing here touches the reference tree, so tests can import it on the GPU box.
"""
import numpy as np


def build_world(seed, task, hidden, E=7, out_scale=1.0, q_boost=0.0):
    """Seeded weights shared by the generator and the tests (tests rebuild them from `seed`)."""
    from cmbpo_amd import synthetic
    rng = np.random.default_rng(seed)
    obs_dim, act_dim = synthetic.ENV_DIMS[task]
    ws, bs = synthetic.ensemble_weights(rng, E, obs_dim + act_dim, hidden, 2 * (obs_dim + 1), bias_scale=0.05,
                                        out_scale=out_scale)
    if q_boost:
        bs[2][:, 0, 2] += q_boost      # pushes a quaternion dim so AntSafe's z_rot < -0.7 branch fires
    sc_in = synthetic.scaler(rng, obs_dim + act_dim, hit_clamp=False)
    sc_out = synthetic.scaler(rng, obs_dim + 1, hit_clamp=False)
    sc_out = (sc_out[0], (sc_out[1] * 0.01).astype(np.float32))
    pol = synthetic.policy_params(rng, obs_dim, act_dim)
    crit = []
    for _ in range(2):
        cw, cb = synthetic.ensemble_weights(rng, 3, obs_dim, 128, 1, bias_scale=0.05)
        crit.append((cw, cb, synthetic.scaler(rng, obs_dim, hit_clamp=False),
                     synthetic.scaler(rng, 1, hit_clamp=False)))
    elites = [0, 2, 3, 5, 6][: max(1, E - 2)]
    return dict(obs_dim=obs_dim, act_dim=act_dim, ws=ws, bs=bs, sc_in=sc_in, sc_out=sc_out, pol=pol,
                v=crit[0], vc=crit[1], elites=elites)



def make_update_batch(rng, n, obs_dim, act_dim, hidden, cost_p, cadv_scale, T):
    from oracle import refupdate
    from cmbpo_amd import synthetic
    params = np.concatenate([p.reshape(-1) for p in synthetic.policy_params(rng, obs_dim, act_dim, hidden)])
    params = (params + rng.standard_normal(params.shape) * 0.02).astype(np.float32)
    obs = rng.standard_normal((n, obs_dim)).astype(np.float32)
    g0 = refupdate.PolicyGraph(obs_dim, act_dim, dict(obs=obs, act=np.zeros((n, act_dim)), adv=np.zeros(n),
                                                      cadv=np.zeros(n), logp_old=np.zeros(n), cost=np.zeros(n),
                                                      mu_old=np.zeros((n, act_dim)),
                                                      log_std_old=np.zeros((n, act_dim))), hidden=hidden)
    import torch
    with torch.no_grad():
        mu, ls = g0._mu(torch.as_tensor(params))
    mu, ls = mu.numpy(), ls.numpy()
    act = (mu + rng.standard_normal(mu.shape).astype(np.float32) * np.exp(ls)).astype(np.float32)
    batch = dict(obs=obs, act=act, adv=rng.standard_normal(n).astype(np.float32),
                 cadv=(rng.standard_normal(n) * cadv_scale).astype(np.float32),
                 cost=(rng.random(n) < cost_p).astype(np.float32), mu_old=mu.astype(np.float32),
                 log_std_old=np.tile(ls[None], (n, 1)).astype(np.float32))
    g1 = refupdate.PolicyGraph(obs_dim, act_dim, {**batch, "logp_old": np.zeros(n)}, hidden=hidden)
    with torch.no_grad():
        batch["logp_old"] = g1.logp(torch.as_tensor(params)).numpy().astype(np.float32)
    return params, batch
