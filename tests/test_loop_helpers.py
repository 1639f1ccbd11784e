"""CPU: the trainer loop's small pure functions against golden G10 (recorded from the reference's own
format_samples_for_dyn, update_dict and CMBPO._set_rollout_length; tests/golden/make_golden.py gen_loop_helpers)."""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
G = np.load(os.path.join(HERE, "golden", "g10_loop_helpers.npz"))


def test_format_samples_for_dyn():
    from cmbpo_amd.cmbpo import format_samples_for_dyn
    samples = {k[2:]: G[k] for k in G.files if k.startswith("s_")}
    for tag, kw in (("r", dict(append_r=True, append_c=False)), ("rc", dict(append_r=True, append_c=True)),
                    ("none", dict(append_r=False, append_c=False))):
        x, y = format_samples_for_dyn(samples, **kw)
        np.testing.assert_array_equal(x, G["dyn_in_" + tag])
        np.testing.assert_array_equal(y, G["dyn_out_" + tag])
        assert x.dtype == G["dyn_in_" + tag].dtype and y.dtype == G["dyn_out_" + tag].dtype


def test_update_dict():
    from cmbpo_amd.cmbpo import update_dict
    a = {"k1": 1.0, "k2": 4.0, "only_a": 7.0}
    b = {"k1": 3.0, "k2": -2.0, "only_b": 5.0}
    for i in range(3):
        wa, wb = G[f"ud{i}_w"]
        d = update_dict(a, b, weight_a=float(wa), weight_b=float(wb))
        assert sorted(d) == list(G[f"ud{i}_keys"])
        np.testing.assert_array_equal(np.array([d[k] for k in sorted(d)]), G[f"ud{i}_vals"])


def test_rollout_length_schedule():
    from cmbpo_amd.cmbpo import CMBPO
    for row in G["schedule"]:
        schedule, epoch, want = [int(v) for v in row[:4]], int(row[4]), int(row[5])
        seen = []
        fake = types.SimpleNamespace(_rollout_schedule=schedule, _epoch=epoch,
                                     model_sampler=types.SimpleNamespace(set_max_path_length=seen.append))
        CMBPO._set_rollout_length(fake)
        assert fake._rollout_length == want and seen == [want], (schedule, epoch)
