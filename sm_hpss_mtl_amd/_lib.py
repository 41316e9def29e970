"""ctypes binding of libsmh.so (C ABI: include/smh.h).  No CPU fallback: a missing library or a missing
GPU is an error, never a silent detour."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# (SMH_LIBSMH_PATH: another build of the same library, for A/B timing of two kernel versions on one box -- tools only)
LIB_PATH = os.environ.get("SMH_LIBSMH_PATH") or os.path.join(HERE, "libsmh.so")

SMH_OK, SMH_E_INVALID, SMH_E_HIP, SMH_E_WORKSPACE, SMH_E_DEVICE = 0, -1, -2, -3, -4


class FrontendCfg(C.Structure):
    _fields_ = [("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop", C.c_int32), ("n_mels", C.c_int32),
                ("l_harm", C.c_int32), ("l_perc", C.c_int32), ("log_db", C.c_int32), ("mel_sr", C.c_float)]


class ModelCfg(C.Structure):
    _fields_ = [("n_feat", C.c_int32), ("patch_size", C.c_int32), ("n_classes", C.c_int32),
                ("nb_filters", C.c_int32), ("kernel_size", C.c_int32), ("nb_stacks", C.c_int32),
                ("n_dilations", C.c_int32), ("block_variant", C.c_int32)]


class CnnCfg(C.Structure):
    _fields_ = [("kind", C.c_int32), ("in_h", C.c_int32), ("in_w", C.c_int32), ("n_classes", C.c_int32),
                ("n_mels", C.c_int32), ("n_fft", C.c_int32), ("fc_width", C.c_int32), ("fs", C.c_float)]


_vp, _i, _sz, _fp = C.c_void_p, C.c_int, C.c_size_t, C.c_void_p  # device pointers travel as void*

# name -> (restype, argtypes): exactly the declarations of include/smh.h
SIGNATURES = {
    "smh_last_error": (C.c_char_p, []),
    "smh_version": (_i, []),
    "smh_device_count": (_i, []),
    "smh_ctx_create": (_i, [C.POINTER(FrontendCfg), C.POINTER(_vp)]),
    "smh_ctx_destroy": (None, [_vp]),
    "smh_ctx_feat_rows": (_i, [_vp]),
    "smh_ctx_mel_basis": (_i, [_vp, _vp]),
    "smh_num_frames": (_i, [_i, _i, _i]),
    "smh_tiled_frames": (_i, [_i, _i]),
    "smh_num_patches": (_i, [_i, _i, _i]),
    "smh_patch_start": (_i, [_i, _i, _i, _i]),
    "smh_stft_mag_f32": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "smh_hpss_median_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _vp]),
    "smh_hpss_median_ex_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _i, _vp]),
    "smh_median_time_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _vp]),
    "smh_median_time_ex_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _i, _vp]),
    "smh_median_freq_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _vp]),
    "smh_softmask_f32": (_i, [_vp, _fp, _fp, _fp, _sz, _fp, _fp, _vp]),
    "smh_mel_f32": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "smh_power_to_db_sq_f32": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "smh_standardize_rows_f32": (_i, [_vp, _fp, _i, _i, _fp, _vp]),
    "smh_extract_patches_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _i, _i, _fp, _vp]),
    "smh_harm_buffer_floats": (_sz, [_i, _i]),
    "smh_features_blocked_ok": (_i, [_vp, _i, _i]),
    "smh_features_f32": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _fp, _fp, _vp, _vp]),
    "smh_features_ex_f32": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _vp, _vp]),
    "smh_frontend_workspace_bytes": (_sz, [_vp, _i, _i]),
    "smh_frontend_f32": (_i, [_vp, _fp, _i, _i, _i, _i, _fp, _fp, _vp, _sz, _fp, _fp, _fp, _vp]),
    "smh_frontend_ragged_sizes": (_i, [_vp, C.POINTER(C.c_longlong), C.POINTER(C.c_int), _i, _i, _i, C.POINTER(C.c_longlong),
                                       C.POINTER(C.c_longlong), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]),
    "smh_frontend_ragged_f32": (_i, [_vp, _fp, C.POINTER(C.c_longlong), C.POINTER(C.c_int), _i, _i, _i, _fp, _fp, _vp, _sz, _vp]),
    "smh_normalize_workspace_bytes": (_sz, [_i, _i]),
    "smh_silence_workspace_bytes": (_sz, [_i, _i, _i]),
    "smh_normalize_f32": (_i, [_fp, _i, _i, _fp, _vp, _sz, _vp]),
    "smh_rms_f32": (_i, [_fp, _i, _i, _i, _i, _fp, _vp]),
    "smh_remove_silence_f32": (_i, [_fp, _i, _i, _fp, _i, _i, _i, _i, C.c_double, C.c_double, _fp, _vp, _vp, _vp, _vp,
                                    _sz, _vp]),
    "smh_preprocess_signal_f32": (_i, [_fp, _i, _i, _i, _i, _i, _fp, _vp, _vp, _sz, _vp]),
    "smh_mix_signals_f32": (_i, [_fp, _fp, _i, _i, _i, _fp, _fp, _vp, _sz, _vp]),
    "smh_medfilt1d_f32": (_i, [_fp, _i, _i, _i, _fp, _vp]),
    "smh_noise_augment_f32": (_i, [_fp, _fp, _sz, C.c_float, C.c_ulonglong, C.c_ulonglong, _vp]),
    "smh_dropout_masks_f32": (_i, [_fp, _sz, C.c_float, _sz, C.c_float, C.c_ulonglong, C.c_ulonglong, _vp]),
    "smh_cnn_trainer_create": (_i, [_vp, _i, C.POINTER(C.c_void_p)]),
    "smh_cnn_trainer_destroy": (None, [_vp]),
    "smh_cnn_trainer_grad_ptr": (_vp, [_vp]),
    "smh_cnn_trainer_bucket_floats": (_sz, [_vp]),
    "smh_cnn_trainer_copy_state": (_i, [_vp, _vp, _vp]),
    "smh_cnn_trainer_num_dropouts": (_i, [_vp]),
    "smh_cnn_trainer_dropout_info": (_i, [_vp, _i, C.POINTER(C.c_size_t), C.POINTER(C.c_float)]),
    "smh_cnn_train_step_f32": (_i, [_vp, _vp, _vp, _i, _vp, _vp, _vp, _vp, _vp]),
    "smh_cnn_trainer_apply_f32": (_i, [_vp, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "smh_scale_data_f64": (_i, [_vp, _i, _i, _vp, _vp, _vp, _vp]),
    "smh_data_statistics_f64": (_i, [_vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "smh_model_create": (_i, [C.POINTER(ModelCfg), C.POINTER(_vp)]),
    "smh_model_destroy": (None, [_vp]),
    "smh_model_num_params": (_sz, [_vp]),
    "smh_model_set_weights": (_i, [_vp, _vp, _sz, _vp]),
    "smh_model_out_dim": (_i, [_vp]),
    "smh_model_forward_f32": (_i, [_vp, _fp, _i, _fp, _fp, _vp]),
    "smh_model_status": (_i, [_vp, _vp]),
    "smh_model_w0_ptr": (_vp, [_vp]),
    "smh_features_l0_f32": (_i, [_vp, _fp, _fp, _fp, _i, _i, _i, _i, _i, _fp, _fp, _fp, _fp, _vp, _vp]),
    "smh_model_forward_x0_f32": (_i, [_vp, _fp, _i, _fp, _fp, _vp]),
    "smh_model_eval_losses_f32": (_i, [_vp, _fp, _fp, _i, C.c_double, C.POINTER(C.c_double), C.c_double, _vp, _vp]),
    "smh_model_dense_workspace_bytes": (_sz, [_vp, _i]),
    "smh_model_forward_dense_f32": (_i, [_vp, _fp, _i, _i, _vp, _sz, _fp, _vp]),
    "smh_model_forward_bf16": (_i, [_vp, _fp, _i, _fp, _vp]),
    "smh_model_forward_bf16_ex": (_i, [_vp, _fp, _i, _fp, _i, _vp]),
    "smh_model_forward_x0_bf16": (_i, [_vp, _fp, _i, _fp, _i, _vp]),
    "smh_model_get_weights": (_i, [_vp, _vp, _sz, _vp]),
    "smh_cnn_create": (_i, [C.POINTER(CnnCfg), C.POINTER(_vp)]),
    "smh_cnn_destroy": (None, [_vp]),
    "smh_cnn_num_params": (_sz, [_vp]),
    "smh_cnn_out_dim": (_i, [_vp]),
    "smh_cnn_feat_dim": (_i, [_vp]),
    "smh_cnn_num_tensors": (_i, [_vp]),
    "smh_cnn_tensor_info": (_i, [_vp, _i, C.c_char_p, _i, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(_sz)]),
    "smh_cnn_set_weights": (_i, [_vp, _vp, _sz, _vp]),
    "smh_cnn_get_weights": (_i, [_vp, _vp, _sz, _vp]),
    "smh_cnn_workspace_bytes": (_sz, [_vp, _i]),
    "smh_cnn_forward_f32": (_i, [_vp, _fp, _i, _fp, _fp, _vp, _sz, _vp]),
    "smh_cnn_forward_bf16": (_i, [_vp, _fp, _i, _fp, _fp, _vp, _sz, _vp]),
    "smh_trainer_create": (_i, [_vp, _i, C.POINTER(_vp)]),
    "smh_trainer_destroy": (None, [_vp]),
    "smh_trainer_grad_ptr": (_vp, [_vp]),
    "smh_train_step_f32": (_i, [_vp, _fp, _fp, _i, _fp, _fp, _vp, _fp, _vp]),
    "smh_trainer_apply_sgd_f32": (_i, [_vp, C.c_float, C.c_float, C.c_float, C.c_float, _vp]),
    "smh_trainer_bucket_floats": (_sz, [_vp]),
    "smh_trainer_copy_state": (_i, [_vp, _vp, _vp]),
    "smh_trainer_reset_state": (_i, [_vp, _vp]),
    "smh_trainer_set_deterministic": (_i, [_vp, _i, _vp]),
    "smh_trainer_set_dtype": (_i, [_vp, _i]),
    "smh_trainer_apply_f32": (_i, [_vp, _i, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint, _vp]),
}

_lib = None


def load():
    """Load libsmh.so and type every entry point.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libsmh.so not found at %s -- build it with `python -m sm_hpss_mtl_amd.build` "
            "(there is no CPU fallback for the HIP path)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    return load().smh_last_error().decode("utf-8", "replace")


def current_stream():
    """torch's current HIP stream as a ctypes pointer for the C ABI.  torch.cuda.current_stream() builds a Stream object through
    several device-index lookups (8 us per call: a quarter of the generator's host time per file); the raw-stream query is one C call."""
    import torch
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
    except AttributeError:  # a torch without the private query
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc: int, what: str = "libsmh") -> int:
    """Map the C status to the Python exceptions the reference's callers would see
    (numpy/librosa shape errors -> ValueError; runtime failures -> RuntimeError)."""
    if rc >= 0:
        return rc
    msg = "%s: %s" % (what, last_error())
    if rc == SMH_E_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def require_gpu():
    lib = load()
    if lib.smh_device_count() <= 0:
        raise RuntimeError("no HIP device visible: the sm_hpss_mtl_amd compute path is GPU-only")
    return lib
