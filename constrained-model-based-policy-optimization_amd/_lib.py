"""ctypes binding of the C-ABI in ``include/cmbpo_hip.h`` (libcmbpo_hip.so).

This is the stub a maintainer of the reference would add:
the reference has no FFI of its own, its hot path is TF ``sess.run`` + NumPy.
The library is built in-tree by ``__graft_entry__.build()``; if it is missing
the import of any product module fails loudly -- there is no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcmbpo_hip.so")

ACT_SWISH, ACT_TANH = 0, 1
HEAD_PROB, HEAD_DETMEAN, HEAD_GAUSS_PI = 0, 1, 2
LOSS_DEFAULT, LOSS_NLL = 0, 1      # cmbpo_trainer_set_loss
ENS_FP32, ENS_SPLIT_BF16, ENS_SPLIT_F16 = 0, 1, 2    # cmbpo_set_ens_matrix_path
TASK_DEFAULT, TASK_HCS, TASK_ANTSAFE = 0, 1, 2

# models/statics.py:56-69 -- task name -> rule id
TASK_IDS = {
    "default": TASK_DEFAULT,
    "HalfCheetah-v2": TASK_DEFAULT,
    "HalfCheetahSafe-v2": TASK_HCS,
    "AntSafe-v2": TASK_ANTSAFE,
}

_p = C.c_void_p
_i = C.c_int
_sz = C.c_size_t

# name -> (restype, argtypes); every symbol include/cmbpo_hip.h declares.
SIGNATURES = {
    "cmbpo_last_error": (C.c_char_p, []),
    "cmbpo_version": (_i, []),
    "cmbpo_set_block_rows": (_i, [_i]),
    "cmbpo_set_stagger": (_i, [_i]),
    "cmbpo_set_dispatch_mode": (_i, [_i]),
    "cmbpo_set_ens_matrix_path": (_i, [_i]),
    "cmbpo_get_ens_matrix_path": (_i, []),
    "cmbpo_set_ens_f16_min_rows": (_i, [_i]),
    "cmbpo_set_ens_f16_row_tiles": (_i, [_i]),
    "cmbpo_mlp_create": (_i, [C.POINTER(_p), _i, _i, _i, _i, _i, _i]),
    "cmbpo_mlp_destroy": (None, [_p]),
    "cmbpo_mlp_load": (_i, [_p] * 13),
    "cmbpo_mlp_load_policy_flat": (_i, [_p, _p, _p]),
    "cmbpo_ens_forward": (_i, [_p, _p, _i, _p, _i, _p, _p, _i, _i, _p, _p, _p]),
    "cmbpo_ens_predict_mean": (_i, [_p, _p, _i, _p, _p, _i, _p, _p]),
    "cmbpo_critic_pair_supported": (_i, [_p, _p]),
    "cmbpo_critic_pair_predict": (_i, [_p, _p, _p, _i, _p, _p, _i, _p, _p, _p]),
    "cmbpo_policy_forward": (_i, [_p, _p, _i, _p, _p, _p, _i, _p, _p, _p, _p, _p]),
    "cmbpo_fakeenv_post": (_i, [_i, _i, _i, _i, _p, _p, _i, _p, _p, _p, _p, _p, _i,
                                _p, _p, _p, _p, _p, _p, _p, _p]),
}



class RolloutStruct(C.Structure):
    """ctypes image of ``cmbpo_rollout_t`` (include/cmbpo_hip.h), field for field."""
    _fields_ = (
        [(n, C.c_int32) for n in ("B", "T", "obs_dim", "act_dim", "max_path_length", "ptr",
                                  "uncertainty_mode", "rank", "world")]
        + [("max_samples", C.c_int64), ("dkl_lim", C.c_double), ("gamma", C.c_double), ("lam", C.c_double),
           ("cost_gamma", C.c_double), ("cost_lam", C.c_double)]
        + [(n, C.c_void_p) for n in ("alive_idx", "alive_idx_out", "iscal", "dscal")]
        + [("use_host_budget", C.c_int32), ("host_rank_off", C.c_int32), ("host_excess", C.c_int64)]
        + [(n, C.c_void_p) for n in (
            "alive", "fin_code", "len",
            "cur_obs", "next_obs", "act_t", "logp_t", "mu_t", "ls_t", "v_t", "vc_t", "v_n", "vc_n",
            "rew_t", "cost_t", "dkl_t", "epv_t", "term_t",
            "dkl_acc", "path_ret", "path_cost", "path_dyn_var", "store_part",
            "obs_buf", "act_buf", "mu_buf", "ls_buf",
            "rew_buf", "val_buf", "cost_buf", "cval_buf", "logp_buf",
            "adv_buf", "ret_buf", "cadv_buf", "cret_buf")]
    )


# iscal / dscal slots (CMBPO_I_* / CMBPO_D_* in the header)
I_N_ALIVE, I_N_UNC, I_N_FIN_PRE, I_N_STORED, I_N_ALIVE_OUT, I_SIZE, I_N_FIN_POST = 0, 1, 2, 3, 4, 5, 6
(D_TOTAL_SAMPLES, D_TOTAL_COST, D_TOTAL_REW, D_TOTAL_VS, D_TOTAL_CVS, D_TOTAL_DKL, D_TOTAL_DYN_EP_VAR,
 D_MAX_DKL, D_MAX_PATH_RETURN, D_DKL_SUM_T, D_STEP_MAX_DKL, D_SUM_PATH_RET, D_SUM_PATH_COST) = range(13)

_rp = C.POINTER(RolloutStruct)
SIGNATURES.update({
    "cmbpo_rollout_reset": (_i, [_rp, _p]),
    "cmbpo_rollout_decide": (_i, [_rp, _p]),
    "cmbpo_rollout_count": (_i, [_rp, _p]),
    "cmbpo_rollout_finish": (_i, [_rp, _i, _p]),
    "cmbpo_rollout_store": (_i, [_rp, _p]),
    "cmbpo_rollout_book_pre_max_rows": (_i, []),
    "cmbpo_rollout_book_pre": (_i, [_rp, _i, _p]),
    "cmbpo_rollout_book_post": (_i, [_rp, _i, _p]),
    "cmbpo_rollout_read_scalars": (_i, [_rp, _p, _p]),
    "cmbpo_rollout_compact": (_i, [_rp, _p]),
    "cmbpo_rollout_step": (_i, [_rp, _i, _p, _p, _p, _p, _i, _i, _p, _p, _p, _p, _p]),
    "cmbpo_rollout_run": (_i, [_rp, _i, _p, _p, _p, _p, _i, _i, _p, _p, C.c_long, C.c_long, _p, _p, _i, C.c_double, _i, _p,
                               C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), _p]),
    "cmbpo_buffer_offsets": (_i, [_rp, _p, _p]),
    "cmbpo_buffer_moments": (_i, [_rp, _i, _p, _p]),
    "cmbpo_buffer_prepare": (_i, [_rp, _p, _p, _p]),
    "cmbpo_buffer_flatten": (_i, [_rp, _p, _p, C.POINTER(C.c_void_p), _p]),
})



class PiBatchStruct(C.Structure):
    """ctypes image of ``cmbpo_pi_batch_t``."""
    _fields_ = [("n", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32)] + \
               [(k, C.c_void_p) for k in ("obs", "act", "adv", "cadv", "logp_old", "cost", "mu_old", "logstd_old")]


_bp = C.POINTER(PiBatchStruct)
SIGNATURES.update({
    "cmbpo_pi_create": (_i, [C.POINTER(_p), _i, _i, _i]),
    "cmbpo_pi_destroy": (None, [_p]),
    "cmbpo_pi_num_params": (_i, [_p]),
    "cmbpo_pi_set_params": (_i, [_p, _p, _p]),
    "cmbpo_pi_loss_grad": (_i, [_p, _bp, _i, _p, _p, _p]),
    "cmbpo_pi_fvp": (_i, [_p, _bp, _p, _p, _p]),
    "cmbpo_pi_keep_activations": (_i, [_p, _i]),
    "cmbpo_pi_saved_activation_uses": (C.c_long, [_p]),
    "cmbpo_set_pi_matrix_path": (None, [_i]),
    "cmbpo_get_pi_matrix_path": (_i, []),
    "cmbpo_pi_eval": (_i, [_p, _bp, _p, _p]),
    "cmbpo_cg_init": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "cmbpo_cg_step": (_i, [_i, _p, C.c_double, C.c_float, _p, _p, _p, _p, _p]),
    "cmbpo_pi_cg_solve": (_i, [_p, _bp, _p, C.c_double, C.c_float, _i, _p, _p, _p, _p, _p, _i, _p]),
    "cmbpo_pi_cg_release": (None, [_p]),
    "cmbpo_pi_cg_iter": (_i, [_p, _bp, C.c_double, C.c_float, _p, _p, _p, _p, _p]),
    "cmbpo_pi_cg_commit": (_i, [_p, _p]),
    "cmbpo_pi_cg_graph_launches": (C.c_long, []),
    "cmbpo_pi_cg_graph_captures": (C.c_long, []),
    "cmbpo_vec_lincomb": (_i, [_i, C.c_float, _p, C.c_float, _p, _p, _p]),
    "cmbpo_vec_dots": (_i, [_i, _i, C.POINTER(_p), C.POINTER(_p), _p, _p]),
    "cmbpo_gae_segments": (_i, [_i] + [_p] * 8 + [C.c_double] * 4 + [_p] * 5),
    "cmbpo_adv_normalize": (_i, [_i, _p, _p, _p, _p]),
})

# ensemble training (csrc/ens_train.hip)
SIGNATURES.update({
    "cmbpo_trainer_create": (_i, [C.POINTER(_p), _p, _i, C.c_float, _p]),
    "cmbpo_trainer_destroy": (None, [_p]),
    "cmbpo_trainer_set_loss": (_i, [_p, _i]),
    "cmbpo_trainer_set_weights": (_i, [_p] * 8),
    "cmbpo_trainer_get_weights": (_i, [_p] * 8),
    "cmbpo_trainer_reset_optimizer": (_i, [_p, _p]),
    "cmbpo_trainer_get_moments": (_i, [_p, _i] + [_p] * 7),
    "cmbpo_trainer_set_moments": (_i, [_p, _i] + [_p] * 6 + [C.c_long, _p]),
    "cmbpo_mlp_set_scalers": (_i, [_p] * 6),
    "cmbpo_trainer_step": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _p]),
    "cmbpo_trainer_epoch": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _i, _p]),
    "cmbpo_trainer_losses": (_i, [_p, _p, _i, _p, _i, _p, _i, _i, _p, _p]),
    "cmbpo_trainer_steps_done": (C.c_long, [_p]),
})

_lib = None


class CmbpoHipError(RuntimeError):
    pass


def lib():
    """Load (once) and return the ctypes handle; raise if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CmbpoHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
        # torch first: its wheel bundles its own libamdhip64 / libhsa-runtime64, and the library must bind to THAT
        # runtime (the one that owns the tensors it is handed).  Loaded the other way round the process ends up with
        # two HIP runtimes and the second one finds no device.
        import torch  # noqa: F401
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().cmbpo_last_error().decode("utf-8", "replace")
        raise CmbpoHipError(f"{what} failed (rc={rc}): {msg}")


def ptr(t):
    """Device (or host) pointer of a torch tensor / numpy array, None -> NULL."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    """Raw hipStream_t of torch's current stream on the current device (the C call behind
    ``torch.cuda.current_stream().cuda_stream``, which costs ~8 us of Python per call -- it is asked for before every
    kernel launch, seven times per rollout step)."""
    import torch
    try:
        return torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice())
    except AttributeError:      # other torch builds: the public, slower route
        return torch.cuda.current_stream().cuda_stream
