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

Training (``PE.train``, ``models/pens/pe.py:457-646``) keeps the
reference's host control flow (holdout split, bootstrap indices, epoch loop, early stopping,
elite ranking) and runs every ``sess.run(train_op)`` / ``sess.run(self.loss)`` as HIP kernels
over device-resident data (``csrc/ens_train.hip``).  Losses: 'MSPE' (dynamics) and 'MSE'
(critics), the two every shipped config uses.
"""
import ctypes as C
import itertools
import time
from collections import OrderedDict

import numpy as np
import torch

from . import _lib, checkpoint

_ACTS = {"swish": _lib.ACT_SWISH, "tanh": _lib.ACT_TANH}


class _CachedScaler:
    """The attributes of TensorStandardScaler the callers read (models/pens/utils.py:90-115) and its running
    fit (:119-138, :220-231).  The TF variables are float32, so every fit rounds the running moments to float32
    before they are cached."""

    def __init__(self, dim):
        self.cached_mu = np.zeros([1, dim], dtype=np.float32)
        self.cached_var = np.ones([1, dim], dtype=np.float32)
        self.cached_count = 0
        self.fitted = False

    def fit(self, data):
        if isinstance(data, torch.Tensor):
            n = data.shape[0]
            d64 = data.double()
            b_mu = d64.mean(dim=0, keepdim=True).cpu().numpy()
            b_var = d64.var(dim=0, unbiased=False, keepdim=True).cpu().numpy()
        else:
            n = data.shape[0]
            b_mu = np.mean(data, axis=0, keepdims=True)
            b_var = np.var(data, axis=0, keepdims=True)
        delta = b_mu - self.cached_mu
        tot = self.cached_count + n
        new_mu = self.cached_mu + delta * n / tot
        m2 = self.cached_var * self.cached_count + b_var * n + np.square(delta) * self.cached_count * n / tot
        self.cached_mu = np.asarray(new_mu, np.float32)
        self.cached_var = np.asarray(m2 / tot, np.float32)
        self.cached_count = float(np.float32(tot))
        self.fitted = True


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

    def set_scalers(self, in_scaler=None, out_scaler=None):
        """New scaler moments (mu, var) without touching the weights (after TensorStandardScaler.fit)."""
        out_dim = self.out_width // 2 if self.head == _lib.HEAD_PROB else self.out_width

        def vec(v, n):
            return None if v is None else np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(n))

        in_mu, in_var = (vec(in_scaler[0], self.in_dim), vec(in_scaler[1], self.in_dim)) if in_scaler else (None, None)
        out_mu, out_var = (vec(out_scaler[0], out_dim), vec(out_scaler[1], out_dim)) if out_scaler else (None, None)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().cmbpo_mlp_set_scalers(self._h, _lib.ptr(in_mu), _lib.ptr(in_var), _lib.ptr(out_mu),
                                                        _lib.ptr(out_var), _lib.current_stream()),
                       "cmbpo_mlp_set_scalers")


class EnsembleTrainer:
    """Owner of one ``cmbpo_trainer_t``: master weights, Adam moments and the activations of one step."""

    def __init__(self, mlp, max_batch, lr, decays, loss=None):
        self.mlp, self.max_batch = mlp, int(max_batch)
        self._h = C.c_void_p()
        dec = (C.c_double * 3)(*[float(d) for d in decays])
        with torch.cuda.device(mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_create(C.byref(self._h), mlp.handle, self.max_batch, float(lr), dec),
                       "cmbpo_trainer_create")
            if loss == "NLL":     # 'MSPE' / 'MSE' are the handle's defaults for its head
                _lib.check(_lib.lib().cmbpo_trainer_set_loss(self._h, _lib.LOSS_NLL), "cmbpo_trainer_set_loss")

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:
                _lib.lib().cmbpo_trainer_destroy(h)
            except Exception:
                pass

    def _shapes(self):
        m = self.mlp
        return [(m.ensemble, m.in_dim, m.hidden), (m.ensemble, m.hidden, m.hidden), (m.ensemble, m.hidden, m.out_width)]

    def set_weights(self, weights, biases):
        ws = [np.ascontiguousarray(np.asarray(w, np.float32).reshape(s)) for w, s in zip(weights, self._shapes())]
        bs = [np.ascontiguousarray(np.asarray(b, np.float32).reshape(s[0], s[2])) for b, s in zip(biases, self._shapes())]
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_set_weights(
                self._h, _lib.ptr(ws[0]), _lib.ptr(bs[0]), _lib.ptr(ws[1]), _lib.ptr(bs[1]), _lib.ptr(ws[2]),
                _lib.ptr(bs[2]), _lib.current_stream()), "cmbpo_trainer_set_weights")

    def get_weights(self):
        """([W0, W1, W2], [b0, b1, b2]) in the reference variable layout W[E,in,out], b[E,1,out]."""
        ws = [np.empty(s, np.float32) for s in self._shapes()]
        bs = [np.empty((s[0], 1, s[2]), np.float32) for s in self._shapes()]
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_get_weights(
                self._h, _lib.ptr(ws[0]), _lib.ptr(bs[0]), _lib.ptr(ws[1]), _lib.ptr(bs[1]), _lib.ptr(ws[2]),
                _lib.ptr(bs[2]), _lib.current_stream()), "cmbpo_trainer_get_weights")
        return ws, bs

    def get_moments(self, which):
        """Adam first (which=0) / second (which=1) moments, same layout as get_weights."""
        ws = [np.empty(s, np.float32) for s in self._shapes()]
        bs = [np.empty((s[0], 1, s[2]), np.float32) for s in self._shapes()]
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_get_moments(
                self._h, int(which), _lib.ptr(ws[0]), _lib.ptr(bs[0]), _lib.ptr(ws[1]), _lib.ptr(bs[1]),
                _lib.ptr(ws[2]), _lib.ptr(bs[2]), _lib.current_stream()), "cmbpo_trainer_get_moments")
        return ws, bs

    def set_moments(self, which, weights, biases, steps_done):
        ws = [np.ascontiguousarray(np.asarray(w, np.float32).reshape(s)) for w, s in zip(weights, self._shapes())]
        bs = [np.ascontiguousarray(np.asarray(b, np.float32).reshape(s[0], s[2])) for b, s in zip(biases, self._shapes())]
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_set_moments(
                self._h, int(which), _lib.ptr(ws[0]), _lib.ptr(bs[0]), _lib.ptr(ws[1]), _lib.ptr(bs[1]),
                _lib.ptr(ws[2]), _lib.ptr(bs[2]), int(steps_done), _lib.current_stream()), "cmbpo_trainer_set_moments")

    def reset_optimizer(self):
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_reset_optimizer(self._h, _lib.current_stream()),
                       "cmbpo_trainer_reset_optimizer")

    def step(self, inputs, targets, idx_ptr, idx_stride, batch):
        """One train_op on device tensors inputs[N,in] / targets[N,D]; member e uses rows idx[e*idx_stride + b]."""
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_step(
                self._h, _lib.ptr(inputs), inputs.shape[1], _lib.ptr(targets), targets.shape[1], idx_ptr,
                int(idx_stride), int(batch), _lib.current_stream()), "cmbpo_trainer_step")

    def epoch(self, inputs, targets, idx, batch):
        """All minibatches of one epoch over idx[E, n] (int32 device tensor), enqueued in one call."""
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_epoch(
                self._h, _lib.ptr(inputs), inputs.shape[1], _lib.ptr(targets), targets.shape[1], _lib.ptr(idx),
                int(idx.shape[1]), int(idx.shape[1]), int(batch), _lib.current_stream()), "cmbpo_trainer_epoch")

    def losses(self, inputs, targets, idx, idx_stride, n_rows, out=None):
        """`self.loss` per member (device tensor [E]); idx: int32 device tensor of rows (idx_stride 0: shared)."""
        if out is None:
            out = torch.empty(self.mlp.ensemble, dtype=torch.float32, device=self.mlp.device)
        with torch.cuda.device(self.mlp.device):
            _lib.check(_lib.lib().cmbpo_trainer_losses(
                self._h, _lib.ptr(inputs), inputs.shape[1], _lib.ptr(targets), targets.shape[1], _lib.ptr(idx),
                int(idx_stride), int(n_rows), _lib.ptr(out), _lib.current_stream()), "cmbpo_trainer_losses")
        return out

    @property
    def steps_done(self):
        return int(_lib.lib().cmbpo_trainer_steps_done(self._h))


def _to_dev(x, device):
    """(tensor on device, was_numpy)."""
    if isinstance(x, torch.Tensor):
        return x.to(device=device, dtype=torch.float32).contiguous(), False
    return torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).to(device), True


class TrainControl:
    """Host control flow of ``PE.train`` (models/pens/pe.py:364-403, 457-646): holdout split, bootstrap indices,
    epoch / minibatch loop, per-epoch shuffle, early stopping on the holdout losses, elite ranking.  The numerics
    sit behind hooks (``_begin_train``, ``_begin_epoch``, ``_train_batch`` / ``_train_epoch``, ``_holdout_losses``,
    ``_shuffle_on_device``, ``_finish_train``) that :class:`PE` implements with the HIP trainer; the class needs
    ``num_nets``, ``num_elites`` and ``name``."""

    def _start_train(self):
        self._snapshots = {i: (None, 1e10) for i in range(self.num_nets)}
        self._epochs_since_update = 0

    def _save_best(self, epoch, holdout_losses):
        """pe.py:367-389 (the per-member state snapshot taken there is never read back and is not kept)."""
        updated = False
        for i in range(len(holdout_losses)):
            current = holdout_losses[i]
            _, best = self._snapshots[i]
            improvement = (best - current) / best
            if improvement > 0.01:
                self._snapshots[i] = (epoch, current)
                updated = True
        if updated:
            self._epochs_since_update = 0
        else:
            self._epochs_since_update += 1
        return self._epochs_since_update > self._max_epochs_since_update

    def _end_train(self, holdout_losses):
        sorted_inds = np.argsort(holdout_losses)
        self._model_inds = sorted_inds[:self.num_elites].tolist()

    def _check_train_args(self, kwargs):
        pass

    def _train_epoch(self, n, batch_size):
        """The minibatch loop (pe.py:541-563); returns the number of gradient updates."""
        n_batches = int(np.ceil(n / batch_size))
        for batch_num in range(n_batches):
            self._train_batch(batch_num, min(batch_size, n - batch_num * batch_size))
        return n_batches

    def train(self, inputs, targets, batch_size=32, max_epochs=None, max_epochs_since_update=5,
              min_epoch_before_break=0, hide_progress=False, holdout_ratio=0.0, max_logging=5000,
              max_grad_updates=None, timer=None, max_t=None, rng=None, shuffle_on_device=False, **kwargs):
        """Trains / continues training the ensemble.

        ``rng`` supplies permutation / randint / uniform (default: the ``numpy.random`` module, the generator the
        reference draws from, in the same order).  ``shuffle_on_device`` replaces the per-epoch host argsort of
        shuffle_rows (pe.py:483-485) by the same shuffle drawn on the GPU (other random numbers, same distribution).
        The progress-bar evaluation of the training loss (pe.py:566-581, display only) is not run."""
        self._check_train_args(kwargs)
        rng = np.random if rng is None else rng
        self._max_epochs_since_update = max_epochs_since_update
        self._start_train()
        break_train = False

        # Split into training and holdout sets (pe.py:487-493)
        num_holdout = min(int(inputs.shape[0] * holdout_ratio), max_logging)
        permutation = rng.permutation(inputs.shape[0])
        train_rows, holdout_rows = permutation[num_holdout:], permutation[:num_holdout]
        n = train_rows.shape[0]
        self._begin_train(inputs, targets, train_rows, holdout_rows, batch_size)

        idxs = rng.randint(n, size=[self.num_nets, n])
        epoch_iter = range(max_epochs) if max_epochs else itertools.count()
        t0 = time.time()
        grad_updates = 0
        epoch = -1
        for epoch in epoch_iter:
            self._begin_epoch(idxs)
            grad_updates += self._train_epoch(n, batch_size)
            # shuffle_rows (pe.py:483-485, :564)
            if shuffle_on_device:
                idxs = self._shuffle_on_device(idxs)
            else:
                order = np.argsort(rng.uniform(size=idxs.shape), axis=-1)
                idxs = idxs[np.arange(idxs.shape[0])[:, None], order]
            if not hide_progress and holdout_ratio >= 1e-12:
                break_train = self._save_best(epoch, self._holdout_losses())
            t = time.time() - t0
            if (break_train and epoch > min_epoch_before_break) or (max_grad_updates and grad_updates > max_grad_updates):
                break
            if max_t and t > max_t:
                break
        if timer:
            timer.stamp('bnn_train')
        holdout_losses = self._holdout_losses()
        if timer:
            timer.stamp('bnn_holdout')
        self._end_train(holdout_losses)
        if timer:
            timer.stamp('bnn_end')
        self._finish_train()
        self.train_epochs, self.train_grad_updates = epoch + 1, grad_updates
        val_loss = (np.sort(holdout_losses)[:self.num_elites]).mean()
        return OrderedDict({f'{self.name}/val_loss': val_loss})


class PE(TrainControl):
    """Ensemble of E three-layer MLPs with optional in/out standard scalers.

    ``loss`` decides the head exactly as the reference does
    (``models/pens/pe.py:413-415``): 'NLL'/'MSPE' -> probabilistic (mean, var),
    anything else -> deterministic single head.
    """

    def __init__(self, in_dim, out_dim, name="BNN", hidden_dims=(512, 512), num_networks=7,
                 num_elites=5, loss="MSPE", activation="swish", use_scaler_in=False,
                 use_scaler_out=False, device=None, lr=1e-3, decay=1e-4, max_logvar=.5, min_logvar=-6, **_unused):
        hidden_dims = tuple(int(h) for h in hidden_dims)
        if len(hidden_dims) != 2 or hidden_dims[0] != hidden_dims[1]:
            raise ValueError("the HIP path supports two equal hidden layers (all shipped configs: "
                             "(512,512) dynamics, (128,128) critics); got %r" % (hidden_dims,))
        self.name = name
        self.loss_type = loss
        self.activation = activation
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
        # optimizer arguments of pe_factory.py:50-60 (weight decay per layer: decay/4, decay/2, decay)
        self.lr = float(lr)
        self.decays = (decay / 4.0, decay / 2.0, decay)
        self._trainer = None
        self._weights_on_device = False     # True once a train step has moved the masters past mlp._keep
        # 'NLL' models own two more optimised variables (pe.py:198-209).  They never enter the network
        # (_compile_outputs, pe.py:789-838); their only gradient is the constant +-0.01 of the regulariser (pe.py:263),
        # so their value is a function of the number of optimiser steps -- kept here for checkpoint interchange.
        self._lv_base = None
        if self.loss_type == "NLL":
            one = np.ones((1, self._out_dim), np.float32)
            self._lv_base, self._lv_steps0 = (one * np.float32(max_logvar), one * np.float32(min_logvar)), 0

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
        if self._trainer is not None:
            self._trainer.set_weights(*self.mlp._keep[:2])
        self._weights_on_device = False
        self.finalized = True

    def init_weights(self, rng=None):
        """Fresh variables as FC.construct_vars draws them (models/pens/fc.py:123-166): truncated normal weights
        (two-sigma rejection, as tf.truncated_normal_initializer) with stddev 1 / (2 sqrt(in)), zero biases."""
        rng = np.random if rng is None else rng
        E, I, H = self.num_nets, self._in_dim, self.hidden
        O = 2 * self._out_dim if self.is_probabilistic else self._out_dim
        ws, bs = [], []
        for k, n in ((I, H), (H, H), (H, O)):
            std = 1.0 / (2.0 * np.sqrt(k))
            w = rng.standard_normal((E, k, n))
            bad = np.abs(w) > 2.0
            while bad.any():
                w[bad] = rng.standard_normal(int(bad.sum()))
                bad = np.abs(w) > 2.0
            ws.append((w * std).astype(np.float32))
            bs.append(np.zeros((E, 1, n), np.float32))
        self.set_weights(ws, bs)
        return ws, bs

    def get_weights(self):
        """([W0, W1, W2], [b0, b1, b2]) -- the optvars the reference checkpoints (models/pens/pe.py:736-764)."""
        if self._trainer is not None and self._weights_on_device:
            return self._trainer.get_weights()
        ws, bs = self.mlp._keep[:2]
        return [w.copy() for w in ws], [b.reshape(b.shape[0], 1, b.shape[1]).copy() for b in bs]

    def sync_weights(self, comm, root=0):
        """Multi-GPU jobs train every ensemble as independent replicas (the reference never shards training either);
        this makes `root`'s weights, scalers and elites the ones every rank uses afterwards."""
        if comm is None or comm.world == 1:
            return
        ws, bs = self.get_weights()
        parts = [np.asarray(a, np.float32).reshape(-1) for a in ws + bs]
        if self.use_scaler_in:
            parts += [self.scaler_in.cached_mu.reshape(-1), self.scaler_in.cached_var.reshape(-1)]
        if self.use_scaler_out:
            parts += [self.scaler_out.cached_mu.reshape(-1), self.scaler_out.cached_var.reshape(-1)]
        parts.append(np.asarray(self._model_inds, np.float32).reshape(-1))
        flat = torch.from_numpy(np.concatenate(parts).astype(np.float32)).to(self.device)
        comm.broadcast(flat, root)
        flat = flat.cpu().numpy()
        out, off = [], 0
        for a in parts:
            out.append(flat[off:off + a.size])
            off += a.size
        k = len(ws)
        new_ws = [o.reshape(w.shape) for o, w in zip(out[:k], ws)]
        new_bs = [o.reshape(b.shape) for o, b in zip(out[k:2 * k], bs)]
        i = 2 * k
        sc_in = sc_out = None
        if self.use_scaler_in:
            sc_in, i = (out[i].reshape(1, -1), out[i + 1].reshape(1, -1)), i + 2
        if self.use_scaler_out:
            sc_out, i = (out[i].reshape(1, -1), out[i + 1].reshape(1, -1)), i + 2
        self.set_weights(new_ws, new_bs, sc_in, sc_out)
        self._model_inds = [int(round(v)) for v in out[i]]

    def reset(self):
        """pe.py:405-411: re-draw the layer variables (the optimizer state is NOT reset there either)."""
        self.init_weights()

    # -- max_logvar / min_logvar of 'NLL' models ---------------------------------------------------------------------
    @staticmethod
    def _const_grad_adam_drift(c, n, lr, b1=0.9, b2=0.999, eps=1e-8):
        """Total displacement of a variable after n tf.train.AdamOptimizer steps under the constant gradient c:
        m_t = c (1 - b1^t), v_t = c^2 (1 - b2^t), lr_t = lr sqrt(1 - b2^t) / (1 - b1^t)."""
        if n <= 0:
            return 0.0
        t = np.arange(1, int(n) + 1, dtype=np.float64)
        root = np.sqrt(1.0 - b2 ** t)
        return float(np.sum(lr * c * root / (abs(c) * root + eps)))

    def _logvar_bounds(self):
        if self._lv_base is None:
            return None
        n = (self._trainer.steps_done if self._trainer is not None else 0) - self._lv_steps0
        hi = self._lv_base[0] - np.float32(self._const_grad_adam_drift(0.01, n, self.lr))
        lo = self._lv_base[1] - np.float32(self._const_grad_adam_drift(-0.01, n, self.lr))
        return hi.astype(np.float32), lo.astype(np.float32)

    @property
    def max_logvar(self):
        b = self._logvar_bounds()
        return None if b is None else b[0]

    @property
    def min_logvar(self):
        b = self._logvar_bounds()
        return None if b is None else b[1]

    # -- checkpoints (pe.py:736-783; file format in checkpoint.py) ----------------
    def save(self, savedir, timestep):
        """Writes <name>_<timestep>.nns / .mat exactly as the reference does (structure + nonoptvars + optvars)."""
        if not self.finalized:
            raise RuntimeError()
        ws, bs = self.get_weights()
        sc = lambda s: (s.cached_mu, s.cached_var)
        return checkpoint.save_ensemble(
            savedir, self.name, timestep, ws, bs, self.activation, self.decays, self.is_probabilistic,
            sc(self.scaler_in) if self.use_scaler_in else None, sc(self.scaler_out) if self.use_scaler_out else None,
            logvar_bounds=self._logvar_bounds())

    def load(self, model_dir, timestep=None):
        """Loads <name>[_<timestep>].nns / .mat (the reference reads <name>.nns / <name>.mat, pe.py:355-361,766-783)."""
        ck = checkpoint.load_ensemble(model_dir, self.name, timestep, self.use_scaler_in, self.use_scaler_out)
        O = 2 * self._out_dim if self.is_probabilistic else self._out_dim
        want = [(self._in_dim, self.hidden), (self.hidden, self.hidden), (self.hidden, O)]
        got = [tuple(w.shape[1:]) for w in ck["weights"]]
        if got != want or any(w.shape[0] != self.num_nets for w in ck["weights"]):
            raise ValueError("checkpoint structure %r x %d does not match this ensemble %r x %d"
                             % (got, ck["weights"][0].shape[0], want, self.num_nets))
        if ck["layers"][0]["activation"] != self.activation:
            raise ValueError("checkpoint activation %r != %r" % (ck["layers"][0]["activation"], self.activation))
        if (ck["logvar_bounds"] is None) != (self._lv_base is None):
            raise ValueError("checkpoint %s max_logvar / min_logvar, this ensemble's loss is %r"
                             % ("holds" if ck["logvar_bounds"] is not None else "lacks", self.loss_type))
        self.set_weights(ck["weights"], ck["biases"], ck["scaler_in"], ck["scaler_out"])
        if self._lv_base is not None:
            hi, lo = ck["logvar_bounds"]
            self._lv_base = (hi.reshape(1, -1).astype(np.float32), lo.reshape(1, -1).astype(np.float32))
            self._lv_steps0 = self._trainer.steps_done if self._trainer is not None else 0
        return ck

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
        """Mean over ALL members (pe.py:338-343, 648-669); probabilistic ensembles return (mean, var) with the
        disagreement of the member means added to the variance (pe.py:326-333; the member forward is the HIP kernel, the
        reduction over the 7 members is two torch calls -- no caller of the training loop uses it)."""
        if self.is_probabilistic:
            # pe.py:326-333: mean over members; var = mean member variance + variance of the member means
            assert len(inputs.shape) == 2
            x, was_np = _to_dev(inputs, self.device)
            mean, var = self.predict_ensemble(x)
            m2d = mean.mean(dim=0)
            v2d = var.mean(dim=0) + ((mean - m2d) ** 2).mean(dim=0)
            return (m2d.cpu().numpy(), v2d.cpu().numpy()) if was_np else (m2d, v2d)
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

    # -- training hooks (device side of TrainControl.train) -------------------------
    def _ensure_trainer(self, batch_size):
        if self._trainer is not None and self._trainer.max_batch >= batch_size:
            return self._trainer
        if not self.finalized:
            self.init_weights()
        ws, bs = self.get_weights()
        # a larger batch than any seen before re-creates the step buffers; weights and Adam state move over
        old = self._trainer
        state = None if old is None else (old.get_moments(0), old.get_moments(1), old.steps_done)
        self._trainer = EnsembleTrainer(self.mlp, max(int(batch_size), 32), self.lr, self.decays, self.loss_type)
        self._trainer.set_weights(ws, bs)
        if state is not None:
            (mw, mb), (vw, vb), steps = state
            self._trainer.set_moments(0, mw, mb, steps)
            self._trainer.set_moments(1, vw, vb, steps)
        return self._trainer

    def validate(self, inputs, targets):
        """pe.py:440-451: mean `self.loss` of the num_elites best members on (inputs, targets)."""
        x, _ = _to_dev(inputs, self.device)
        t, _ = _to_dev(targets, self.device)
        tr = self._trainer if self._trainer is not None else self._ensure_trainer(min(x.shape[0], 4096))
        idx = torch.arange(x.shape[0], dtype=torch.int32, device=self.device)
        losses = tr.losses(x, t, idx, 0, x.shape[0]).cpu().numpy()
        return np.sort(losses)[:self.num_elites].mean()

    def _check_train_args(self, kwargs):
        if self.loss_type not in ("MSPE", "MSE", "NLL"):
            raise NotImplementedError("HIP training covers 'MSPE', 'MSE' (the shipped configs) and 'NLL'; got %r"
                                      % (self.loss_type,))
        if kwargs.get("weights") is not None or kwargs.get("old_pred") is not None:
            raise NotImplementedError("weighted / clipped losses (vf_clipping) are off in every shipped config")

    def _begin_train(self, inputs, targets, train_rows_h, holdout_rows_h, batch_size):
        """Data to the device once, scaler fit on the training rows (pe.py:518-523), trainer handle."""
        c = self._ctx = {}
        c["x"], _ = _to_dev(inputs, self.device)
        t_dev, _ = _to_dev(targets, self.device)
        c["t"] = t_dev[:, None] if t_dev.dim() == 1 else t_dev
        to_dev = lambda a: torch.from_numpy(np.ascontiguousarray(a.astype(np.int32))).to(self.device)
        c["train_rows"], c["holdout_rows"] = to_dev(train_rows_h), to_dev(holdout_rows_h)

        def rows_of(a, rows_h, rows_d):   # numpy data is fitted in numpy, exactly like the reference
            if isinstance(a, torch.Tensor):
                a = a[:, None] if a.dim() == 1 else a
                return a[rows_d.long()]
            a = a[:, None] if a.ndim == 1 else a
            return a[rows_h]

        if self.use_scaler_in:
            self.scaler_in.fit(rows_of(inputs, train_rows_h, c["train_rows"]))
        if self.use_scaler_out:
            self.scaler_out.fit(rows_of(targets, train_rows_h, c["train_rows"]))
        c["tr"] = self._ensure_trainer(batch_size)
        self.mlp.set_scalers(
            (self.scaler_in.cached_mu, self.scaler_in.cached_var) if self.use_scaler_in else None,
            (self.scaler_out.cached_mu, self.scaler_out.cached_var) if self.use_scaler_out else None)
        c["loss_buf"] = torch.empty(self.num_nets, dtype=torch.float32, device=self.device)
        c["batch_size"] = int(batch_size)

    def _begin_epoch(self, idxs_h):
        """Rows of the full data set each member visits this epoch: train_rows[idxs] (inputs[batch_idxs], pe.py:543)."""
        c = self._ctx
        if isinstance(idxs_h, torch.Tensor):
            idxs = idxs_h
        else:
            idxs = torch.from_numpy(np.ascontiguousarray(idxs_h.astype(np.int32))).to(self.device)
        c["gidx"] = c["train_rows"][idxs.long()].contiguous()

    def _train_batch(self, batch_num, rows):
        c = self._ctx
        n = c["gidx"].shape[1]
        c["tr"].step(c["x"], c["t"], c["gidx"].data_ptr() + 4 * batch_num * c["batch_size"], n, rows)
        self._weights_on_device = True

    def _train_epoch(self, n, batch_size):
        c = self._ctx
        c["tr"].epoch(c["x"], c["t"], c["gidx"], batch_size)
        self._weights_on_device = True
        return int(np.ceil(n / batch_size))

    def _shuffle_on_device(self, idxs):
        if not isinstance(idxs, torch.Tensor):
            idxs = torch.from_numpy(np.ascontiguousarray(idxs.astype(np.int32))).to(self.device)
        order = torch.argsort(torch.rand(idxs.shape, device=self.device), dim=-1)
        return torch.gather(idxs, 1, order)

    def _holdout_losses(self):
        c = self._ctx
        if c["holdout_rows"].shape[0] == 0:
            # the reference feeds empty arrays here and ranks NaN losses (argsort keeps the member order)
            return np.full(self.num_nets, np.nan, np.float32)
        return c["tr"].losses(c["x"], c["t"], c["holdout_rows"], 0, c["holdout_rows"].shape[0],
                              out=c["loss_buf"]).cpu().numpy()

    def _finish_train(self):
        self._ctx = None


def build_PE(in_dim, out_dim, name="BNN", hidden_dims=(200, 200, 200), num_networks=7, num_elites=5,
             loss="MSPE", activation="swish", output_activation=None, decay=1e-4, lr=1e-3,
             lr_decay=None, decay_steps=None, use_scaler_in=False, use_scaler_out=False,
             clip_loss=False, kl_cliprange=0.1, max_logvar=.5, min_logvar=-6, session=None, device=None):
    """Same signature as models/pens/pe_factory.py:9-29 (Adam with a constant learning rate; lr_decay is unused by
    every shipped config)."""
    if output_activation is not None:
        raise NotImplementedError("output activations are unused by every shipped config")
    if lr_decay is not None:
        raise NotImplementedError("learning-rate decay is unused by every shipped config")
    if clip_loss:
        raise NotImplementedError("clipped value losses (vf_clipping) are off in every shipped config")
    return PE(in_dim, out_dim, name=name, hidden_dims=hidden_dims, num_networks=num_networks,
              num_elites=num_elites, loss=loss, activation=activation, use_scaler_in=use_scaler_in,
              use_scaler_out=use_scaler_out, device=device, lr=lr, decay=decay, max_logvar=max_logvar,
              min_logvar=min_logvar)
