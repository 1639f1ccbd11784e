"""Ensemble checkpoint interchange: the two files ``PE.save`` writes
(``models/pens/pe.py:736-764``) and ``PE._load_structure`` / ``finalize`` read back (:355-361, :766-783).

``<name>_<timestep>.nns``  one line per layer, the ``repr`` of the reference's FC layer (``models/pens/fc.py:46-50``):
    FC(output_dim=512, input_dim=37, activation='swish', weight_decay=2.5e-07, ensemble_size=7)
  the last line carries the end activation (None) and, for a probabilistic ensemble, HALF the variable's width.
``<name>_<timestep>.mat``  scipy ``savemat`` of {"0": ..., "1": ...}: nonoptvars first (scaler_in mu, var [1, in];
  scaler_out mu, var [1, out] -- each pair only when that scaler is enabled), then per layer weights [E, in, out] and
  biases [E, 1, out].

Pure file format, no device code: weights trained by the reference load here and vice versa.
"""
import os

import numpy as np
from scipy.io import loadmat, savemat


def layer_line(output_dim, input_dim, activation, weight_decay, ensemble_size):
    return "FC(output_dim={!r}, input_dim={!r}, activation={!r}, weight_decay={!r}, ensemble_size={!r})".format(
        int(output_dim), int(input_dim), activation, weight_decay, int(ensemble_size))


def parse_layer_line(line):
    """The parser of pe.py:770-781 (``line[3:-2]`` strips ``FC(`` and ``)\\n``)."""
    kwargs = dict(argval.split("=") for argval in line.rstrip("\n")[3:-1].split(", "))
    return dict(
        input_dim=int(kwargs["input_dim"]), output_dim=int(kwargs["output_dim"]),
        weight_decay=None if kwargs["weight_decay"] == "None" else float(kwargs["weight_decay"]),
        activation=None if kwargs["activation"] == "None" else kwargs["activation"][1:-1],
        ensemble_size=int(kwargs["ensemble_size"]))


def _paths(model_dir, name, timestep):
    stem = name if timestep is None else "{}_{}".format(name, timestep)
    return os.path.join(model_dir, stem + ".nns"), os.path.join(model_dir, stem + ".mat")


def save_ensemble(model_dir, name, timestep, weights, biases, activation, decays, probabilistic, scaler_in=None,
                  scaler_out=None, logvar_bounds=None):
    """weights[l]: [E, in, out]; biases[l]: [E, 1, out]; scaler_*: (mu[1, d], var[1, d]) or None; logvar_bounds:
    (max_logvar[1, d], min_logvar[1, d]) of an 'NLL' model -- the last two optvars (pe.py:208-209) -- or None."""
    nns, mat = _paths(model_dir, name, timestep)
    n = len(weights)
    with open(nns, "w+") as f:
        for l, w in enumerate(weights):
            E, i, o = w.shape
            last = l == n - 1
            f.write("%s\n" % layer_line(o // 2 if (last and probabilistic) else o, i, None if last else activation,
                                        None if decays is None else float(decays[l]), E))
    var_vals, k = {}, 0
    for sc in (scaler_in, scaler_out):
        if sc is not None:
            for a in sc:
                var_vals[str(k)] = np.asarray(a, np.float32).reshape(1, -1)
                k += 1
    for w, b in zip(weights, biases):
        var_vals[str(k)] = np.asarray(w, np.float32)
        var_vals[str(k + 1)] = np.asarray(b, np.float32).reshape(w.shape[0], 1, w.shape[2])
        k += 2
    if logvar_bounds is not None:
        for a in logvar_bounds:
            var_vals[str(k)] = np.asarray(a, np.float32).reshape(1, -1)
            k += 1
    savemat(mat, var_vals)
    return nns, mat


def load_ensemble(model_dir, name, timestep=None, use_scaler_in=False, use_scaler_out=False):
    """-> dict(layers=[...], weights, biases, scaler_in, scaler_out).  Which scalers the file holds is not recorded
    in it (the reference knows from its constructor arguments), so the caller says."""
    nns, mat = _paths(model_dir, name, timestep)
    with open(nns, "r") as f:
        layers = [parse_layer_line(line) for line in f if line.strip()]
    d = loadmat(mat)
    k = 0

    def take():
        nonlocal k
        a = np.asarray(d[str(k)], np.float32)
        k += 1
        return a

    sc_in = (take(), take()) if use_scaler_in else None
    sc_out = (take(), take()) if use_scaler_out else None
    weights, biases = [], []
    for l in layers:
        w, b = take(), take()
        E, i = l["ensemble_size"], l["input_dim"]
        # savemat / loadmat keep 3-D shapes; a width-1 trailing axis survives as well
        w = w.reshape(E, i, -1)
        weights.append(w)
        biases.append(b.reshape(E, 1, w.shape[2]))
    bounds = None
    if str(k) in d and str(k + 1) in d and str(k + 2) not in d:
        # an 'NLL' model: max_logvar, min_logvar [1, out_dim] close the optvars (pe.py:208-209)
        bounds = (take(), take())
        width = weights[-1].shape[2]
        if any(a.size not in (width, width // 2) for a in bounds):
            raise ValueError("%s: trailing variables are not max / min_logvar of this structure" % mat)
    if str(k) in d:
        raise ValueError("%s holds more variables than %s describes (scaler flags wrong?)" % (mat, nns))
    return dict(layers=layers, weights=weights, biases=biases, scaler_in=sc_in, scaler_out=sc_out,
                logvar_bounds=bounds)
