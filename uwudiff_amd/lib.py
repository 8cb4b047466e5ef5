"""ctypes binding of libuwu_hip.so (the C ABI declared in include/uwu_hip.h).

The product path has NO fallback: if the library is missing or a call fails, this module raises.
PyTorch is only the owner of device memory and streams -- every entry takes raw device pointers.
"""
import ctypes
import os
from ctypes import c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libuwu_hip.so")

F32, BF16 = 0, 1
PT = {"epsilon": 0, "v_prediction": 1, "sample": 2, "rectified_flow": 3}
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_DGELU, EPI_BIAS_SILU, EPI_ACCUM = 0, 1, 2, 3, 4, 5


class UwuError(RuntimeError):
    pass


class DitDesc(ctypes.Structure):
    _fields_ = (
        [(n, c_int32) for n in ("B", "T", "D", "H", "L", "mlp_ratio", "in_ch", "out_ch", "patch", "img", "dtype",
                                "cond_dim", "freq_dim")]
        + [("ln_eps", c_float), ("mod_total", c_int32)]
        + [("w", c_void_p), ("w32", c_void_p), ("g32", c_void_p)]
        + [(n, c_int64) for n in ("off_patch_w", "off_patch_b", "off_t_w1", "off_t_b1", "off_t_w2", "off_t_b2",
                                  "off_y_w", "off_y_b", "off_mod_w", "off_mod_b", "off_final_w", "off_final_b",
                                  "off_layer0", "layer_stride")]
        + [("pos", c_void_p), ("ws", c_void_p), ("ws_bytes", c_size_t), ("layer_done", c_void_p)]
        + [("side_stream", c_void_p)]
        + [("rope", c_int32), ("off_rope_h", c_int64), ("off_rope_w", c_int64), ("pos_xy", c_void_p)]
        + [("fp8", c_int32), ("f8_scale", c_void_p), ("f8_amax", c_void_p), ("f8_fmt", c_void_p)]
        + [("checkpoint", c_int32)]
    )


P = c_void_p
_SIGS = {
    "uwu_last_error": (c_char_p, []),
    "uwu_version": (c_int, []),
    "uwu_env_refresh": (c_int, []),
    "uwu_memset_zero": (c_int, [P, ctypes.c_uint64, P]),
    "uwu_schedule_gather": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, c_int, P, P]),
    "uwu_rf_time_to_sigma": (c_int, [P, c_float, P, c_int, c_int, P, P, P]),
    "uwu_qsample": (c_int, [P, P, P, c_int, c_int64, P, P, P]),
    "uwu_philox_raw": (c_int, [P, c_int64, ctypes.c_uint64, ctypes.c_uint64, P]),
    "uwu_philox_normal": (c_int, [P, c_int64, ctypes.c_uint64, ctypes.c_uint64, P]),
    "uwu_draw_timesteps": (c_int, [P, c_int, c_int, ctypes.c_uint64, ctypes.c_uint64, P]),
    "uwu_draw_u01": (c_int, [P, c_int, ctypes.c_uint64, ctypes.c_uint64, P]),
    "uwu_qsample_draw": (c_int, [P, P, c_int, c_int64, c_int, c_float, c_float, P, P, P, ctypes.c_uint64, ctypes.c_uint64, P]),
    "uwu_qsample_norm": (c_int, [P, P, P, c_int, c_int64, c_float, c_float, P, P, P]),
    "uwu_ctx_place": (c_int, [P, c_int, P, P] + [c_int] * 7 + [P]),
    "uwu_loss_fwd_bwd": (c_int, [P, P, P, P, c_int, P, c_int, c_int, c_int, c_int, c_int64, P, P, P, P, P, P]),
    "uwu_scale_inplace": (c_int, [P, c_int, c_int64, P, P]),
    "uwu_scale_into": (c_int, [P, P, c_int, c_int64, P, P]),
    "uwu_cast_i64_to_f32": (c_int, [P, P, c_int64, P]),
    "uwu_sampler_step": (c_int, [P, P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, P]),
    "uwu_aggregate_concat": (c_int, [P, P, P, c_int, c_int, c_int64, c_int, ctypes.c_uint64, P]),
    "uwu_aggregate_split": (c_int, [P, P, P, c_int, c_int, c_int64, P]),
    "uwu_aggregate_first": (c_int, [P, P, P, c_int, c_int64, P]),
    "uwu_sampler_combine": (c_int, [P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, P]),
    "uwu_scale_copy": (c_int, [P, P, c_int64, c_float, P]),
    "uwu_grad_sqnorm_clip": (c_int, [P, c_int64, c_float, c_float, P, P, P]),
    "uwu_adamw_step": (c_int, [P, P, P, P, P, c_int64, c_float, c_float, c_float, c_float, c_float, c_int, c_float,
                               P, c_int, P]),
    "uwu_cast_f32_to_bf16": (c_int, [P, P, c_int64, P]),
    "uwu_cast_bf16_to_f32": (c_int, [P, P, c_int64, P]),
    "uwu_comm_unique_id": (c_int, [P]),
    "uwu_comm_init": (c_int, [P, c_int, c_int, ctypes.POINTER(c_void_p)]),
    "uwu_allreduce_flat": (c_int, [P, P, c_int64, P]),
    "uwu_comm_destroy": (c_int, [P]),
    "uwu_gemm": (c_int, [P, P, P, P, P, P] + [c_int] * 13 + [P]),
    "uwu_gemm_fp8_scratch_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uwu_gemm_fp8": (c_int, [P, P, P, P, P, P] + [c_int] * 9 + [P, P, P, ctypes.c_size_t, P]),
    "uwu_fp8_amax": (c_int, [P, c_int, c_int64, P, P]),
    "uwu_fp8_update_scales": (c_int, [P, P, P, c_int, c_float, P]),
    "uwu_fp8_quantize": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, P, P, P]),
    "uwu_gemm_fp8_emit": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, c_int, P, c_int, P,
                                  P, P]),
    "uwu_gemm_wgrad_scratch_bytes": (ctypes.c_size_t, [c_int, c_int, c_int]),
    "uwu_gemm_wgrad": (c_int, [P, P, P, P] + [c_int] * 8 + [P, ctypes.c_size_t, P]),
    "uwu_gemm_prof_enable": (c_int, [c_int]),
    "uwu_gemm_prof_collect": (c_int, [c_int, P, P, P]),
    "uwu_prof_enable": (c_int, [c_int]),
    "uwu_prof_collect": (c_int, [c_int, c_int, P, P, P, P]),
    "uwu_colsum": (c_int, [P, c_int, c_int, c_int, c_int, P, c_int, P]),
    "uwu_colsum_batched": (c_int, [P, c_int, c_int, c_int, c_int, c_int, P, c_int, P]),
    "uwu_skinny_linear_ok": (c_int, [c_int, c_int, c_int]),
    "uwu_skinny_linear_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_skinny_linear_dgrad": (c_int, [P, P, P, c_int, c_int, c_int, P]),
    "uwu_skinny_linear_wgrad": (c_int, [P, P, P, P, c_int, c_int, c_int, P]),
    "uwu_add_ln_modulate_fwd": (c_int, [P, P, P, P, P, c_int, P, P, P, P, c_int, c_int, c_int, c_float, c_int, c_int, P]),
    "uwu_add_ln_modulate_fwd_q8": (c_int, [P, P, P, P, P, c_int, P, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int, c_float, P]),
    "uwu_add_ln_modulate_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P]),
    "uwu_attention_fwd": (c_int, [P, P, P, P, P] + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_attention_bwd": (c_int, [P] * 10 + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_attention_bias_fwd": (c_int, [P] * 6 + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_attention_bias_bwd": (c_int, [P] * 11 + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_axial_rope_fwd": (c_int, [P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "uwu_axial_rope_bwd": (c_int, [P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "uwu_axial_rope_bwd_shared": (c_int, [P, P, P, c_int, P, P, P, P, P, c_int64, c_int, c_int, c_int, c_int, P]),
    "uwu_axial_rope_table": (c_int, [P, P, P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_attention_rope_fwd": (c_int, [P] * 6 + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_attention_rope_bwd": (c_int, [P] * 10 + [c_int] * 9 + [c_float, c_int, P]),
    "uwu_timestep_embedding": (c_int, [P, c_int, c_int, c_float, P, c_int, P]),
    "uwu_silu_fwd": (c_int, [P, P, c_int64, c_int, P]),
    "uwu_silu_bwd": (c_int, [P, P, P, c_int64, c_int, P]),
    "uwu_add": (c_int, [P, P, P, c_int64, c_int, P]),
    "uwu_transpose_bf16": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_patchify": (c_int, [P, P, c_int, c_int, c_int, c_int, c_int, c_int, P]),
    "uwu_unpatchify": (c_int, [P, c_int, P, c_int, c_int, c_int, c_int, c_int, P]),
    "uwu_add_pos": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_groupnorm_fwd": (c_int, [P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_int, c_int, P]),
    "uwu_groupnorm_bwd": (c_int, [P] * 10 + [c_int] * 6 + [P]),
    "uwu_im2col3x3": (c_int, [P, P] + [c_int] * 6 + [P]),
    "uwu_col2im3x3": (c_int, [P, P] + [c_int] * 6 + [P]),
    "uwu_conv3x3_implicit_ok": (c_int, [c_int] * 7),
    "uwu_conv3x3_fwd": (c_int, [P, P, P, P] + [c_int] * 7 + [P]),
    "uwu_conv3x3_dgrad": (c_int, [P, P, P] + [c_int] * 7 + [P]),
    "uwu_conv3x3_wgrad_scratch_bytes": (ctypes.c_size_t, [c_int, c_int, c_int64]),
    "uwu_conv3x3_wgrad": (c_int, [P, P, P, P] + [c_int] * 7 + [P, ctypes.c_size_t, P]),
    "uwu_geglu_fwd": (c_int, [P, P, c_int64, c_int, c_int, P]),
    "uwu_geglu_bwd": (c_int, [P, P, P, c_int64, c_int, c_int, P]),
    "uwu_upsample2x": (c_int, [P, P] + [c_int] * 6 + [P]),
    "uwu_add_rowvec": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_nchw_to_cl": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_cl_to_nchw": (c_int, [P, P, c_int, c_int, c_int, c_int, P]),
    "uwu_dit_workspace_bytes": (c_size_t, [ctypes.POINTER(DitDesc)]),
    "uwu_dit_forward": (c_int, [ctypes.POINTER(DitDesc), P, P, P, P, P]),
    "uwu_dit_backward": (c_int, [ctypes.POINTER(DitDesc), P, P]),
    "uwu_dit_backward_cond": (c_int, [ctypes.POINTER(DitDesc), P, P]),
    "uwu_dit_layer_param_stride": (c_int64, [c_int, c_int]),
}

_lib = None


def exported_symbols():
    return sorted(_SIGS)


def load():
    """Load libuwu_hip.so and declare every signature.  Raises if absent (no CPU fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UwuError(
            f"{LIB_PATH} not found: build it with `python -m uwudiff_amd.build` "
            "(the HIP extension is required; there is no fallback path)"
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().uwu_last_error().decode(errors="replace")
        raise UwuError(f"{what} failed (code {rc}): {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise UwuError(f"unsupported dtype {t.dtype}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL).  Tensors must be contiguous CUDA(HIP) tensors."""
    if t is None:
        return None
    if not t.is_cuda:
        raise UwuError("libuwu_hip kernels need device tensors (got a CPU tensor); there is no CPU path")
    if not t.is_contiguous():
        raise UwuError("libuwu_hip kernels need contiguous tensors")
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def call(name, *args):
    lib = load()
    rc = getattr(lib, name)(*args)
    check(rc, name)
