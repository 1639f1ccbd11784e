"""CPU: the .nns / .mat ensemble checkpoint format (models/pens/pe.py:736-783, fc.py:46-50).
No reference checkpoint ships with the reference tree, so the format is pinned by its writer's own conventions:
the FC repr line, the key order of the .mat file and the loader's parser, restated in cmbpo_amd/checkpoint.py."""
import os
import sys

import numpy as np
import pytest
from scipy.io import loadmat

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmbpo_amd import checkpoint  # noqa: E402


def test_layer_line_is_the_reference_repr_and_round_trips():
    line = checkpoint.layer_line(512, 37, "swish", 2.5e-07, 7)
    assert line == "FC(output_dim=512, input_dim=37, activation='swish', weight_decay=2.5e-07, ensemble_size=7)"
    assert checkpoint.parse_layer_line(line + "\n") == dict(input_dim=37, output_dim=512, weight_decay=2.5e-07,
                                                            activation="swish", ensemble_size=7)
    last = checkpoint.layer_line(30, 512, None, 1e-06, 7)
    assert last == "FC(output_dim=30, input_dim=512, activation=None, weight_decay=1e-06, ensemble_size=7)"
    assert checkpoint.parse_layer_line(last)["activation"] is None


@pytest.mark.parametrize("prob,scalers", [(True, (True, True)), (False, (True, False)), (False, (False, False))])
def test_save_load_round_trip(tmp_path, prob, scalers):
    rng = np.random.default_rng(0)
    E, I, H, D = 3, 5, 8, 2
    O = 2 * D if prob else D
    ws = [rng.standard_normal(s).astype(np.float32) for s in ((E, I, H), (E, H, H), (E, H, O))]
    bs = [rng.standard_normal((E, 1, s)).astype(np.float32) for s in (H, H, O)]
    sc_in = (rng.standard_normal((1, I)).astype(np.float32), rng.random((1, I)).astype(np.float32)) if scalers[0] else None
    sc_out = (rng.standard_normal((1, D)).astype(np.float32), rng.random((1, D)).astype(np.float32)) if scalers[1] else None
    nns, mat = checkpoint.save_ensemble(str(tmp_path), "DynEns", 40, ws, bs, "swish", (2.5e-7, 5e-7, 1e-6), prob,
                                        sc_in, sc_out)
    assert os.path.basename(nns) == "DynEns_40.nns" and os.path.basename(mat) == "DynEns_40.mat"
    lines = open(nns).read().splitlines()
    assert len(lines) == 3 and lines[2].startswith("FC(output_dim=%d, input_dim=8, activation=None" % D)
    raw = loadmat(mat)
    n_sc = 2 * sum(scalers)
    assert sorted(k for k in raw if not k.startswith("__")) == sorted(str(i) for i in range(n_sc + 6))
    if scalers[0]:
        np.testing.assert_array_equal(raw["0"], sc_in[0])       # nonoptvars first: scaler_in mu, var
        np.testing.assert_array_equal(raw["1"], sc_in[1])
    np.testing.assert_array_equal(raw[str(n_sc)], ws[0])        # then W0, b0, W1, ...
    ck = checkpoint.load_ensemble(str(tmp_path), "DynEns", 40, *scalers)
    for a, b in zip(ck["weights"] + ck["biases"], ws + bs):
        np.testing.assert_array_equal(a, b)
    assert (ck["scaler_in"] is None) == (sc_in is None) and (ck["scaler_out"] is None) == (sc_out is None)
    if sc_out is not None:
        np.testing.assert_array_equal(ck["scaler_out"][1], sc_out[1])
    # the reference loads <name>.nns / <name>.mat (no timestep): same reader
    os.rename(nns, os.path.join(tmp_path, "DynEns.nns"))
    os.rename(mat, os.path.join(tmp_path, "DynEns.mat"))
    ck2 = checkpoint.load_ensemble(str(tmp_path), "DynEns", None, *scalers)
    np.testing.assert_array_equal(ck2["weights"][2], ws[2])
    if any(scalers):     # scaler flags that do not match the file are caught, not silently mis-assigned
        with pytest.raises(ValueError):
            checkpoint.load_ensemble(str(tmp_path), "DynEns", None, False, False)


def test_single_output_critic_shapes_survive(tmp_path):
    rng = np.random.default_rng(1)
    E, I, H = 3, 4, 8
    ws = [rng.standard_normal(s).astype(np.float32) for s in ((E, I, H), (E, H, H), (E, H, 1))]
    bs = [rng.standard_normal((E, 1, s)).astype(np.float32) for s in (H, H, 1)]
    checkpoint.save_ensemble(str(tmp_path), "VEnsemble", 3, ws, bs, "swish", None, False)
    ck = checkpoint.load_ensemble(str(tmp_path), "VEnsemble", 3)
    assert ck["weights"][2].shape == (E, H, 1) and ck["biases"][2].shape == (E, 1, 1)
    np.testing.assert_array_equal(ck["weights"][2], ws[2])
    assert ck["layers"][0]["weight_decay"] is None
