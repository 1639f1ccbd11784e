"""The real-environment sampler mirror (cpo_sampler.CpoSampler) against golden G12: the REFERENCE's
samplers/cpo_sampler.py:125-235 + buffers/cpobuffer.py driven by the scripted toy environment and stub policy of
tests/toyworld.py (recorded by tests/golden/make_golden.py --cpo-sampler-only).

CPU: with a recording pool the mirror must make the reference's exact sequence of store() / finish_path() calls
(arguments, bootstrap values AND their dtypes -- float64 zeros promote the reward deltas downstream), return the same
values, keep the same path statistics and log the same series.  GPU: through this repo's CPOBuffer (HIP GAE) the 12-array
get() list equals the reference buffer's."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import toyworld  # noqa: E402

GOLD = os.path.join(HERE, "golden", "g12_cpo_sampler.npz")
NAMES = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]


def test_cpo_sampler_makes_the_reference_call_sequence():
    import cmbpo_amd  # noqa: F401
    from cmbpo_amd.cpo_sampler import CpoSampler
    g = np.load(GOLD, allow_pickle=False)
    pool = toyworld.RecordingPool()
    res = toyworld.drive(CpoSampler, pool)
    np.testing.assert_array_equal(np.array(pool.stores), g["stores"])        # every argument of every store()
    np.testing.assert_array_equal(np.array(pool.finishes), g["finishes"])    # where paths end, bootstraps, their dtypes
    assert g["finishes"][:, 3].any() and not g["finishes"][:, 3].all()      # both kinds of value bootstrap occur
    for k, v in res.items():
        np.testing.assert_array_equal(np.asarray(v), g["rec_" + k], err_msg=k)
    assert res["n_episodes"] == 11 and res["policy_resets"] == 11


@pytest.mark.gpu
def test_cpo_sampler_through_the_hip_buffer_matches_the_reference_buffer(hip_lib):
    torch = pytest.importorskip("torch")
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmbpo_amd.cpo_sampler import CpoSampler
    from cmbpo_amd.cpobuffer import CPOBuffer
    g = np.load(GOLD, allow_pickle=False)
    D, A = toyworld.ToyEnv.D, toyworld.ToyEnv.A
    buf = CPOBuffer(size=128, archive_size=512, observation_space=toyworld.Space(D), action_space=toyworld.Space(A),
                    device="cuda:0")
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    res = toyworld.drive(CpoSampler, buf)
    np.testing.assert_array_equal(res["rets"], g["rec_rets"])
    got, diag = buf.get()
    for k, arr in zip(NAMES, got):
        ref = g["get_" + k]
        assert arr.shape == ref.shape and arr.dtype == ref.dtype, k
        if k in ("adv", "cadv"):
            np.testing.assert_allclose(arr, ref, rtol=1e-5, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(arr, ref, err_msg=k)     # ret / cret bit-exact (float64 recurrence)
    np.testing.assert_allclose(diag["poolr_ret_mean"], float(g["poolr_ret_mean"]), rtol=1e-6)
    np.testing.assert_allclose(diag["poolr_cret_mean"], float(g["poolr_cret_mean"]), rtol=1e-6)
