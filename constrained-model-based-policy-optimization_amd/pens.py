"""Probabilistic ensemble (PE) -- host-side mirror of the reference model plug-in.

Mirrors the prediction interface of ``models/pens/pe.py:28`` (class ``PE``) and
``models/pens/pe_factory.py:9-72`` (``build_PE``) behind the contract of
``models/base_model.py:3-43`` (``EnsembleModel``): ``predict``,
``predict_ensemble``, ``is_probabilistic``, ``is_ensemble``, ``in_dim``,
``out_dim``, ``elite_inds`` and the ``scaler_out.cached_mu/cached_var`` attributes
the trainer reads (``algorithms/cmbpo.py:289-290``).

All arithmetic runs in the hand-written HIP kernels behind the C-ABI
(``csrc/ens_mlp.hip``); torch is used only for device memory and streams.
NumPy in -> NumPy out (drop-in for the reference call sites); torch CUDA tensors
in -> torch CUDA tensors out (device-resident rollout).

Training (``PE.train``, ``models/pens/pe.py:457-646``) is SURVEY §8(f) row N1 and
not part of this path: weights come in through :meth:`PE.set_weights`.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_ACTS = {"swish": _lib.ACT_SWISH, "tanh": _lib.ACT_TANH}


class _CachedScaler:
    """The attributes of TensorStandardScaler the callers read (models/pens/utils.py:90-115)."""

    def __init__(self, dim):
        self.cached_mu = np.zeros([1, dim], dtype=np.float32)
        self.cached_var = np.ones([1, dim], dtype=np.float32)
        self.fitted = False


class EnsembleMLP:
    """Owner of one ``cmbpo_mlp_t`` handle (packed weights on the device)."""

    def __init__(self, ensemble, in_dim, hidden, out_width, activation, head, device=None):
        self.device = torch.device(device if device is not None else "cuda")
        self.ensemble, self.in_dim, self.hidden = int(ensemble), int(in_dim), int(hidden)
        self.out_width, self.head = int(out_width), head
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_mlp_create(C.byref(self._h), self.ensemble, self.in_dim,
                                                   self.hidden, self.out_width,
                                                   _ACTS[activation], head), "cmbpo_mlp_create")
        self._keep = None

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().cmbpo_mlp_destroy(h)
            except Exception:
                pass

    @property
    def handle(self):
        return self._h

    def load(self, weights, biases, in_scaler=None, out_scaler=None, log_std=None):
        """weights: [W0[E,in,H], W1[E,H,H], W2[E,H,O]]; biases: [b0[E,H]|[E,1,H], ...]."""
        E, I, H, O = self.ensemble, self.in_dim, self.hidden, self.out_width
        shapes = [(E, I, H), (E, H, H), (E, H, O)]
        ws, bs = [], []
        for w, b, shp in zip(weights, biases, shapes):
            w = np.ascontiguousarray(np.asarray(w, dtype=np.float32).reshape(shp))
            b = np.ascontiguousarray(np.asarray(b, dtype=np.float32).reshape(shp[0], shp[2]))
            ws.append(w)
            bs.append(b)

        def vec(v, n):
            return None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(n))

        out_dim = O // 2 if self.head == _lib.HEAD_PROB else O
        in_mu, in_var = (vec(in_scaler[0], I), vec(in_scaler[1], I)) if in_scaler else (None, None)
        out_mu, out_var = (vec(out_scaler[0], out_dim), vec(out_scaler[1], out_dim)) if out_scaler else (None, None)
        ls = vec(log_std, out_dim)
        self._keep = (ws, bs, in_mu, in_var, out_mu, out_var, ls)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_mlp_load(
                self._h, _lib.ptr(ws[0]), _lib.ptr(bs[0]), _lib.ptr(ws[1]), _lib.ptr(bs[1]),
                _lib.ptr(ws[2]), _lib.ptr(bs[2]), _lib.ptr(in_mu), _lib.ptr(in_var),
                _lib.ptr(out_mu), _lib.ptr(out_var), _lib.ptr(ls), _lib.current_stream()),
                "cmbpo_mlp_load")


def _to_dev(x, device):
    """(tensor on device, was_numpy)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous(), False
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device), True


class PE:
    """Ensemble of E three-layer MLPs with optional in/out standard scalers.

    ``loss`` decides the head exactly as the reference does
    (``models/pens/pe.py:413-415``): 'NLL'/'MSPE' -> probabilistic (mean, var),
    anything else -> deterministic single head.
    """

    def __init__(self, in_dim, out_dim, name="BNN", hidden_dims=(512, 512), num_networks=7,
                 num_elites=5, loss="MSPE", activation="swish", use_scaler_in=False,
                 use_scaler_out=False, device=None, **_unused):
        hidden_dims = tuple(int(h) for h in hidden_dims)
        if len(hidden_dims) != 2 or hidden_dims[0] != hidden_dims[1]:
            raise ValueError("the HIP path supports two equal hidden layers (all shipped configs: "
                             "(512,512) dynamics, (128,128) critics); got %r" % (hidden_dims,))
        self.name = name
        self.loss_type = loss
        self.num_nets, self.num_elites = int(num_networks), int(num_elites)
        self._in_dim, self._out_dim = int(in_dim), int(out_dim)
        self.hidden = hidden_dims[0]
        self.use_scaler_in, self.use_scaler_out = bool(use_scaler_in), bool(use_scaler_out)
        self.scaler_in = _CachedScaler(self._in_dim) if use_scaler_in else None
        self.scaler_out = _CachedScaler(self._out_dim) if use_scaler_out else None
        # pe.py:103-108: elites are random until a training run ranks the members
        self._model_inds = np.random.randint(self.num_nets, size=self.num_elites)
        head = _lib.HEAD_PROB if self.is_probabilistic else _lib.HEAD_DETMEAN
        width = 2 * self._out_dim if self.is_probabilistic else self._out_dim
        self.mlp = EnsembleMLP(self.num_nets, self._in_dim, self.hidden, width, activation, head, device)
        self.device = self.mlp.device
        self.finalized = False

    # -- reference properties (pe.py:405-435) ---------------------------------
    @property
    def is_probabilistic(self):
        return "NLL" in self.loss_type or "MSPE" in self.loss_type

    @property
    def is_ensemble(self):
        return self.num_nets > 1

    @property
    def in_dim(self):
        return self._in_dim

    @property
    def out_dim(self):
        # the reference returns None for probabilistic models (pe.py:426-430, a missing
        # `return`); callers only use it for bookkeeping, so the real width is returned.
        return self._out_dim

    @property
    def elite_inds(self):
        return self._model_inds

    def set_elites(self, inds):
        self._model_inds = list(int(i) for i in inds)

    # -- weights ---------------------------------------------------------------
    def set_weights(self, weights, biases, scaler_in=None, scaler_out=None):
        """Load weights in the reference variable layout (fc.py:135-166, utils.py:100-115).

        weights/biases: per layer W[E,in,out], b[E,1,out]; scaler_*: (mu[1,dim], var[1,dim]).
        """
        if self.use_scaler_in and scaler_in is None:
            scaler_in = (self.scaler_in.cached_mu, self.scaler_in.cached_var)
        if self.use_scaler_out and scaler_out is None:
            scaler_out = (self.scaler_out.cached_mu, self.scaler_out.cached_var)
        if scaler_in is not None and self.scaler_in is not None:
            self.scaler_in.cached_mu = np.asarray(scaler_in[0], np.float32).reshape(1, -1)
            self.scaler_in.cached_var = np.asarray(scaler_in[1], np.float32).reshape(1, -1)
        if scaler_out is not None and self.scaler_out is not None:
            self.scaler_out.cached_mu = np.asarray(scaler_out[0], np.float32).reshape(1, -1)
            self.scaler_out.cached_var = np.asarray(scaler_out[1], np.float32).reshape(1, -1)
        self.mlp.load(weights, biases,
                      in_scaler=scaler_in if self.use_scaler_in else None,
                      out_scaler=scaler_out if self.use_scaler_out else None)
        self.finalized = True

    # -- prediction --------------------------------------------------------------
    def predict_ensemble(self, inputs, act=None, row_idx=None, out=None):
        """(mean, var)[E, B, out] for 2-D inputs (pe.py:688-697).

        ``inputs`` is x[B, in] (or obs[B, obs] with ``act`` given separately, which saves
        the concat of fake_env.py:81).  The 3-D input path (pe.py:705-713) is unused by the
        trainer and not provided.
        """
        if not self.is_probabilistic:
            raise NotImplementedError("factored prediction of a deterministic ensemble is not on the hot path")
        if len(inputs.shape) != 2:
            raise ValueError("Invalid input dimension.")  # fc.py:92
        x, was_np = _to_dev(inputs, self.device)
        a = None
        if act is not None:
            a, _ = _to_dev(act, self.device)
        obs_dim = x.shape[1]
        act_dim = 0 if a is None else a.shape[1]
        n = x.shape[0]
        ld = n if out is None else out[0].shape[1]
        if out is None:
            mean = torch.empty((self.num_nets, n, self._out_dim), dtype=torch.float32, device=self.device)
            var = torch.empty_like(mean)
        else:
            mean, var = out
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_ens_forward(
                self.mlp.handle, _lib.ptr(x), obs_dim, _lib.ptr(a), act_dim, _lib.ptr(row_idx), None,
                n if row_idx is None else row_idx.shape[0], ld, _lib.ptr(mean), _lib.ptr(var),
                _lib.current_stream()), "cmbpo_ens_forward")
        if was_np:
            return mean.cpu().numpy(), var.cpu().numpy()
        return mean, var

    def predict(self, inputs, row_idx=None, out=None):
        """Mean over ALL members (pe.py:338-343, 648-669); deterministic ensembles only."""
        if self.is_probabilistic:
            raise NotImplementedError("PE.predict of a probabilistic ensemble is not on the hot path "
                                      "(FakeEnv uses predict_ensemble, fake_env.py:88-91)")
        assert len(inputs.shape) == 2
        x, was_np = _to_dev(inputs, self.device)
        n = x.shape[0]
        res = out if out is not None else torch.empty((n, self._out_dim), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_ens_predict_mean(
                self.mlp.handle, _lib.ptr(x), x.shape[1], _lib.ptr(row_idx), None,
                n if row_idx is None else row_idx.shape[0], _lib.ptr(res), _lib.current_stream()),
                "cmbpo_ens_predict_mean")
        return res.cpu().numpy() if was_np else res

    def train(self, *args, **kwargs):
        raise NotImplementedError("PE.train (models/pens/pe.py:457-646) is SURVEY §8(f) row N1, outside this path; "
                                  "load trained weights with set_weights().")


def build_PE(in_dim, out_dim, name="BNN", hidden_dims=(200, 200, 200), num_networks=7, num_elites=5,
             loss="MSPE", activation="swish", output_activation=None, decay=1e-4, lr=1e-3,
             lr_decay=None, decay_steps=None, use_scaler_in=False, use_scaler_out=False,
             clip_loss=False, kl_cliprange=0.1, max_logvar=.5, min_logvar=-6, session=None, device=None):
    """Same signature as models/pens/pe_factory.py:9-29; optimizer arguments are accepted and ignored."""
    if output_activation is not None:
        raise NotImplementedError("output activations are unused by every shipped config")
    return PE(in_dim, out_dim, name=name, hidden_dims=hidden_dims, num_networks=num_networks,
              num_elites=num_elites, loss=loss, activation=activation, use_scaler_in=use_scaler_in,
              use_scaler_out=use_scaler_out, device=device)
