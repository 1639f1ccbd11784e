"""GPU: CPOBuffer (HIP GAE + normalisation behind the reference's buffer API) vs golden G8 recorded from the
reference's own CPOBuffer.  ret / cret are bit-exact (float64 recurrence, float32 casts, incl. the float64-zero
bootstrap promotion); normalised advantages within 1e-5 (statistics summed in a different order)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NAMES = ["obs", "act", "adv", "cadv", "ret", "cret", "logp", "val", "cval", "cost", "log_std", "mu"]


class _Space:
    def __init__(self, d):
        self.shape = (d,)


def test_cpobuffer_matches_reference(hip_lib):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmbpo_amd.cpobuffer import CPOBuffer
    g = np.load(os.path.join(GOLD, "g8_cpobuffer.npz"), allow_pickle=False)
    D, A = int(g["D"]), int(g["A"])
    buf = CPOBuffer(size=64, archive_size=256, observation_space=_Space(D), action_space=_Space(A), device="cuda:0")
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    i = 0
    for p, L in enumerate(g["lengths"]):
        for _ in range(int(L)):
            buf.store(g["obs"][i], g["act"][i], g["obs"][i], g["rew"][i], g["val"][i], g["cost"][i], g["cval"][i],
                      g["logp"][i], {"mu": g["mu"][i], "log_std": g["log_std"][i]}, False, 3)
            i += 1
        lv = np.zeros((1,)) if g["zero_val"][p] else g["last_val"][p:p + 1]
        buf.finish_path(lv, g["last_cval"][p:p + 1])
    assert buf.size == i
    res, diag = buf.get()
    for k, arr in zip(NAMES, res):
        ref = g["get_" + k]
        assert arr.shape == ref.shape and arr.dtype == ref.dtype, k
        if k in ("adv", "cadv"):
            np.testing.assert_allclose(arr, ref, rtol=1e-5, atol=1e-6, err_msg=k)
        else:
            np.testing.assert_array_equal(arr, ref, err_msg=k)
    np.testing.assert_allclose(diag["poolr_ret_mean"], float(g["poolr_ret_mean"]), rtol=1e-6)
    assert buf.arch_size == int(g["arch_size"]) and buf.size == 0
    assert list(buf.epochs_list) == [3]
    arch = buf.get_archive(["observations", "returns", "pi_infos"])
    np.testing.assert_array_equal(arch["returns"], g["get_ret"])
    np.testing.assert_array_equal(arch["mu"], g["get_mu"])
    # Boltzmann start-state sampling plumbing (algorithms/cmbpo.py:241-245)
    dist = buf.boltz_dist(np.array([0.02]), alpha=2)
    np.testing.assert_allclose(dist.sum(), 1.0, rtol=1e-6)
    batch = buf.distributed_batch_from_archive(16, dist, fields=["observations", "pi_infos"])
    assert batch["observations"].shape == (16, D) and batch["log_std"].shape == (16, A)
    ep = buf.epoch_batch(8, buf.epochs_list, fields=["observations", "pi_infos"])
    assert ep["observations"].shape == (1, 8, D)


def test_archive_accessors_match_reference_over_epochs(hip_lib):
    """Golden G11: three epochs moved to the archive, then the start-state sampling accessors of the trainer loop
    (buffers/cpobuffer.py:292-530) under the same np.random seed -- identical draws, identical batches."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from cmbpo_amd.cpobuffer import CPOBuffer
    g = np.load(os.path.join(GOLD, "g11_cpobuffer_archive.npz"))
    D, A = int(g["D"]), int(g["A"])

    class _S:
        def __init__(self, d):
            self.shape = (d,)

    buf = CPOBuffer(40, 100, _S(D), _S(A))
    buf.initialize({"mu": [A], "log_std": [A]}, gamma=0.99, lam=0.95, cost_gamma=0.97, cost_lam=0.5)
    i = p = 0
    lengths = iter(g["path_lengths"])
    for epoch, n_paths in zip(g["plan_epochs"], g["plan_lengths"]):
        for _ in range(int(n_paths)):
            for _ in range(int(next(lengths))):
                buf.store(g["obs"][i], g["act"][i], g["obs"][i] + 1, g["rew"][i], g["val"][i], g["cost"][i], g["cval"][i],
                          g["logp"][i], {"mu": g["mu"][i], "log_std": g["log_std"][i]}, False, int(epoch))
                i += 1
            buf.finish_path(g["last"][p, 0:1], g["last"][p, 1:2])
            p += 1
        buf.get()
    assert buf.arch_size == int(g["arch_size"]) and list(buf.epochs_list) == list(g["epochs_list"])
    assert buf.max_ep == int(g["max_ep"]) and buf.min_ep == int(g["min_ep"])
    dist = buf.boltz_dist(g["kls"], alpha=2)
    np.testing.assert_array_equal(dist, g["boltz"])
    np.random.seed(5)
    b = buf.distributed_batch_from_archive(23, dist, fields=["observations", "pi_infos"])
    np.testing.assert_array_equal(b["observations"], g["dist_obs"])
    np.testing.assert_array_equal(b["mu"], g["dist_mu"])
    e = buf.epoch_batch(7, buf.epochs_list, fields=["observations", "pi_infos"])
    np.testing.assert_array_equal(e["observations"], g["ep_obs"])
    np.testing.assert_array_equal(e["log_std"], g["ep_ls"])
    r = buf.rand_batch_from_archive(11, fields=["observations", "rewards"])
    np.testing.assert_array_equal(r["observations"], g["rand_obs"])
    np.testing.assert_array_equal(r["rewards"], g["rand_rew"])
    arch = buf.get_archive(["observations", "actions", "next_observations", "rewards", "costs", "terminals", "epochs"])
    for k, v in arch.items():
        np.testing.assert_array_equal(v, g["arch_" + k], err_msg=k)
