"""Seeded synthetic weights / inputs of the reference's shapes (no MuJoCo, no checkpoints).

Initialisers follow the reference's variable constructors so magnitudes are realistic:
  * ensembles: W ~ TruncNormal(0, 1/(2 sqrt(in))), b = 0  (models/pens/fc.py:135-166); tests may
    ask for non-zero biases to exercise the bias path;
  * policy: Glorot-uniform kernels, zero biases (tf.layers.dense default), log_std = -0.5
    (network/ac_network.py:104).
Env dims are the derived ones of SURVEY §8: HalfCheetahSafe 20/6, HopperSafe 21/3, AntSafe 29/8,
HumanoidSafe 47/17.
"""
import numpy as np

ENV_DIMS = {
    "HalfCheetahSafe-v2": (20, 6),
    "HopperSafe-v2": (21, 3),
    "AntSafe-v2": (29, 8),
    "HumanoidSafe-v2": (47, 17),
}


def _trunc_normal(rng, shape, std):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def ensemble_weights(rng, ensemble, in_dim, hidden, out_width, bias_scale=0.0, out_scale=1.0):
    """[W0,W1,W2], [b0,b1,b2] in the reference layout W[E,in,out], b[E,1,out]."""
    dims = [(in_dim, hidden), (hidden, hidden), (hidden, out_width)]
    ws, bs = [], []
    for li, (i, o) in enumerate(dims):
        w = _trunc_normal(rng, (ensemble, i, o), 1.0 / (2.0 * np.sqrt(i)))
        if li == 2:
            w *= np.float32(out_scale)
        b = (rng.standard_normal((ensemble, 1, o)) * bias_scale).astype(np.float32)
        ws.append(w)
        bs.append(b)
    return ws, bs


def scaler(rng, dim, hit_clamp=True):
    """(mu[1,dim], var[1,dim]); some variances below 1e-4 so the sigma clamp 1e-2 is exercised."""
    mu = (rng.standard_normal((1, dim)) * 0.1).astype(np.float32)
    var = rng.uniform(0.25, 4.0, size=(1, dim)).astype(np.float32)
    if hit_clamp and dim >= 4:
        var[0, :: max(dim // 3, 1)] = np.float32(1e-6)
    return mu, var


def policy_params(rng, obs_dim, act_dim, hidden=128):
    """[W0,b0,W1,b1,W2,b2,log_std] in TF creation order (network/ac_network.py:26-36,104)."""
    out = []
    for i, o in [(obs_dim, hidden), (hidden, hidden), (hidden, act_dim)]:
        lim = np.sqrt(6.0 / (i + o))
        out.append(rng.uniform(-lim, lim, size=(i, o)).astype(np.float32))
        out.append(np.zeros(o, dtype=np.float32))
    out.append(np.full(act_dim, -0.5, dtype=np.float32))
    return out


def start_states(rng, n, task):
    """Start observations that keep AntSafe branches alive (z in [0.3, 0.9], small quaternion)."""
    obs_dim, _ = ENV_DIMS[task]
    obs = (rng.standard_normal((n, obs_dim)) * 0.1).astype(np.float32)
    if task == "AntSafe-v2":
        obs[:, 0] = rng.uniform(0.3, 0.9, size=n).astype(np.float32)
    return obs
