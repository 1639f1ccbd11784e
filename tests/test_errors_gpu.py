"""GPU: error behaviour of the C-ABI -- every entry point validates its arguments on the host (shapes against what the
kernels and their grids assume) and returns a negative code with a message; nothing is launched on bad input.  The
reference raises Python / TF shape errors at the same call sites (e.g. models/pens/fc.py:92 'Invalid input dimension')."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def test_handles_reject_bad_shapes_and_unloaded_use(hip_lib):
    _need_gpu()
    from cmbpo_amd import _lib
    from cmbpo_amd._lib import CmbpoHipError
    from cmbpo_amd.pens import PE, EnsembleMLP
    lib = _lib.lib()
    h = C.c_void_p()
    for bad in (dict(ensemble=0), dict(hidden=200), dict(in_dim=0), dict(out_width=500)):
        kw = dict(ensemble=3, in_dim=8, hidden=128, out_width=4)
        kw.update(bad)
        rc = lib.cmbpo_mlp_create(C.byref(h), kw["ensemble"], kw["in_dim"], kw["hidden"], kw["out_width"], 0, 0)
        assert rc == -1 and len(lib.cmbpo_last_error()) > 10, bad
    # HEAD_PROB needs an even output width
    assert lib.cmbpo_mlp_create(C.byref(h), 3, 8, 128, 5, 0, _lib.HEAD_PROB) == -1
    pe = PE(8, 2, hidden_dims=(128, 128), num_networks=3, num_elites=2, loss="MSPE", device="cuda:0")
    x = torch.zeros(4, 8, device="cuda")
    with pytest.raises(CmbpoHipError, match="not loaded"):        # CMBPO_ESTATE
        pe.predict_ensemble(x)
    pe.init_weights(np.random.RandomState(0))
    with pytest.raises(CmbpoHipError, match="in_dim"):
        pe.predict_ensemble(torch.zeros(4, 9, device="cuda"))
    with pytest.raises(ValueError):                               # fc.py:92
        pe.predict_ensemble(torch.zeros(3, 4, 8, device="cuda"))
    m2d, v2d = pe.predict(x)                                      # probabilistic: member mean + disagreement (pe.py:326-333)
    me, ve = pe.predict_ensemble(x)
    torch.testing.assert_close(m2d, me.mean(0))
    torch.testing.assert_close(v2d, ve.mean(0) + me.var(0, unbiased=False), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        PE(8, 2, hidden_dims=(200, 200, 200), num_networks=3, num_elites=2, device="cuda:0")
    # trainer: wrong dims / oversized batch / unsupported loss
    tr = pe._ensure_trainer(64)
    t = torch.zeros(10, 2, device="cuda")
    with pytest.raises(CmbpoHipError, match="batch"):
        tr.step(x, t, None, 0, 65)
    with pytest.raises(CmbpoHipError, match="in_dim|target_dim"):
        tr.step(torch.zeros(4, 7, device="cuda"), t, None, 0, 4)
    ce = PE(8, 2, hidden_dims=(128, 128), num_networks=3, num_elites=2, loss="CE", device="cuda:0")
    with pytest.raises(NotImplementedError):
        ce.train(np.zeros((10, 8), np.float32), np.zeros((10, 2), np.float32), max_epochs=1)
    with pytest.raises(CmbpoHipError, match="NLL"):               # 'NLL' needs a probabilistic head
        _lib.check(lib.cmbpo_trainer_set_loss(ce._ensure_trainer(32)._h, _lib.LOSS_NLL), "cmbpo_trainer_set_loss")
    del EnsembleMLP


def test_rollout_and_update_argument_checks(hip_lib):
    _need_gpu()
    from cmbpo_amd import _lib
    lib = _lib.lib()
    rs = _lib.RolloutStruct()
    rs.B, rs.T, rs.obs_dim, rs.act_dim = 8, 4, 3, 2           # every pointer NULL
    for fn in ("cmbpo_rollout_reset", "cmbpo_rollout_decide", "cmbpo_rollout_store", "cmbpo_rollout_compact"):
        assert getattr(lib, fn)(C.byref(rs), None) < 0, fn
        assert len(lib.cmbpo_last_error()) > 10
    h = C.c_void_p()
    assert lib.cmbpo_pi_create(C.byref(h), 29, 64, 8) == -1      # hidden must be 128 or 256
    assert lib.cmbpo_pi_create(C.byref(h), 100, 128, 8) == -1    # obs_dim > 64
    assert lib.cmbpo_pi_create(C.byref(h), 29, 128, 8) == 0
    b = _lib.PiBatchStruct()
    b.n, b.obs_dim, b.act_dim = 16, 29, 8
    assert lib.cmbpo_pi_eval(h, C.byref(b), None, None) == -4    # parameters not set: CMBPO_ESTATE
    lib.cmbpo_pi_destroy(h)
    assert lib.cmbpo_gae_segments(0, *([None] * 8), 0.99, 0.95, 0.99, 0.95, *([None] * 5)) <= 0
    assert lib.cmbpo_version() > 0
