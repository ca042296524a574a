"""ctypes binding of libsurfd_hip.so (include/surfd_hip.h).

The product path has no CPU fallback: if the library is missing or a call fails this module
raises — it never routes around the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# SURFD_LIB: a differently built copy of the same library (A/B timing of kernel variants, tools/build_variants.py)
LIB_PATH = os.environ.get("SURFD_LIB") or os.path.join(_HERE, "lib", "libsurfd_hip.so")
GRID_MAX_LEVELS = 8

c_i64p = C.POINTER(C.c_int64)
c_f32p = C.POINTER(C.c_float)


class UNetCfg(C.Structure):
    _fields_ = [("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
                ("num_res_blocks", C.c_int), ("n_mult", C.c_int), ("channel_mult", C.c_int * 8),
                ("n_attn", C.c_int), ("attention_resolutions", C.c_int * 8), ("num_heads", C.c_int),
                ("context_dim", C.c_int), ("num_classes", C.c_int)]


class SamplerCfg(C.Structure):
    _fields_ = [("sampler", C.c_int), ("num_steps", C.c_int), ("clip_denoised", C.c_int), ("eta", C.c_float),
                ("timestep_map", c_i64p),
                ("coef1", c_f32p), ("coef2", c_f32p), ("log_variance", c_f32p),
                ("sqrt_recip_ab", c_f32p), ("sqrt_recipm1_ab", c_f32p), ("ab", c_f32p), ("ab_prev", c_f32p)]


class GridStats(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("levels", C.c_int * GRID_MAX_LEVELS),
                ("fwd_points", C.c_int64 * GRID_MAX_LEVELS), ("grad_points", C.c_int64)]


_P = C.c_void_p
_SIGS = {
    "surfd_last_error": (C.c_char_p, []),
    "surfd_abi_version": (C.c_int, []),
    "surfd_device_count": (C.c_int, []),
    "surfd_build_config": (C.c_char_p, []),
    "surfd_profile_enable": (C.c_int, [C.c_int]),
    "surfd_profile_read": (C.c_int, [C.c_int, c_i64p, C.POINTER(C.c_double)]),
    "surfd_unet_debug_read": (C.c_int, [_P, C.POINTER(C.c_longlong), C.c_int]),
    "surfd_unet_create": (C.c_int, [C.POINTER(UNetCfg), C.POINTER(_P)]),
    "surfd_unet_destroy": (None, [_P]),
    "surfd_unet_num_params": (C.c_int, [_P]),
    "surfd_unet_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_int)]),
    "surfd_unet_set_param": (C.c_int, [_P, C.c_char_p, _P, c_i64p, C.c_int, _P]),
    "surfd_unet_finalize": (C.c_int, [_P, _P]),
    "surfd_unet_forward": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "surfd_unet_set_precision": (C.c_int, [_P, C.c_int]),
    "surfd_unet_set_cu_budget": (C.c_int, [_P, C.c_int]),
    "surfd_unet_set_wide": (C.c_int, [_P, C.c_int]),
    "surfd_unet_saturation_count": (C.c_int, [_P, C.c_int, c_i64p, _P]),
    "surfd_unet_loop_progress": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "surfd_unet_debug_only_op": (C.c_int, [_P, C.c_int]),
    "surfd_unet_debug_run_module": (C.c_int, [_P, C.c_char_p, _P, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    "surfd_sample_loop": (C.c_int, [_P, C.POINTER(SamplerCfg), _P, _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "surfd_sample_loop_begin": (C.c_int, [_P, C.POINTER(SamplerCfg), _P, _P, _P, _P, C.c_int, C.c_int, _P]),
    "surfd_sample_loop_run": (C.c_int, [_P, C.c_int, C.POINTER(C.c_int), _P]),
    "surfd_sample_loop_end": (C.c_int, [_P, _P, _P]),
    "surfd_ddpm_step": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "surfd_ddim_step": (C.c_int, [_P, _P, _P, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                  C.c_int, _P, C.c_int64, _P]),
    "surfd_decoder_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "surfd_decoder_destroy": (None, [_P]),
    "surfd_decoder_num_params": (C.c_int, [_P]),
    "surfd_decoder_param_info": (C.c_int, [_P, C.c_int, C.POINTER(C.c_char_p), c_i64p, C.POINTER(C.c_int)]),
    "surfd_decoder_set_param": (C.c_int, [_P, C.c_char_p, _P, c_i64p, C.c_int, _P]),
    "surfd_decoder_finalize": (C.c_int, [_P, _P]),
    "surfd_decoder_set_precision": (C.c_int, [_P, C.c_int]),
    "surfd_decoder_saturation_count": (C.c_int, [_P, C.c_int, c_i64p, _P]),
    "surfd_decoder_sustained_clock": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double), _P]),
    "surfd_decoder_set_grid_blocks": (C.c_int, [_P, C.c_int]),
    "surfd_decoder_bind_latents": (C.c_int, [_P, _P, C.c_int, _P]),
    "surfd_decoder_logits_emb": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, _P]),
    "surfd_decoder_udf": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, _P, _P]),
    "surfd_decoder_udf_grad": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P, _P, _P, _P]),
    "surfd_grid_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "surfd_grid_destroy": (None, [_P]),
    "surfd_grid_set_thresholds": (C.c_int, [_P, c_f32p, C.c_int, C.c_float, C.c_float, C.c_float]),
    "surfd_grid_fill": (C.c_int, [_P, _P, C.c_int, _P, _P, _P]),
    "surfd_grid_fill_batch": (C.c_int, [_P, C.c_int, _P, _P, _P, _P, _P]),
    "surfd_grid_fill_dense": (C.c_int, [_P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "surfd_grid_get_stats": (C.c_int, [_P, C.POINTER(GridStats), _P]),
    "surfd_grid_get_totals": (C.c_int, [_P, C.POINTER(GridStats), c_i64p, C.c_int, _P]),
    "surfd_grid_begin": (C.c_int, [_P, _P, _P, _P]),
    "surfd_grid_level_points": (C.c_int, [_P, C.c_int, _P, C.c_int64, c_i64p, _P]),
    "surfd_grid_level_commit": (C.c_int, [_P, C.c_int, _P, C.c_int64, _P]),
    "surfd_grid_grad_points": (C.c_int, [_P, _P, C.c_int64, c_i64p, _P]),
    "surfd_grid_grad_commit": (C.c_int, [_P, _P, C.c_int64, _P]),
    "surfd_grid_shard_begin": (C.c_int, [_P, _P, _P, _P]),
    "surfd_grid_shard_level_eval": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "surfd_grid_shard_pack": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P, _P]),
    "surfd_grid_shard_level_commit": (C.c_int, [_P, C.c_int, _P, C.c_int64, C.c_int, _P]),
    "surfd_grid_shard_grad_eval": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_int64, _P]),
    "surfd_grid_shard_grad_commit": (C.c_int, [_P, _P, C.c_int64, C.c_int, _P]),
    "surfd_grid_shard_overflows": (C.c_int, [_P, c_i64p, C.c_int, _P]),
    "surfd_mc_udf": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "surfd_mc_iso": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.POINTER(_P)]),
    "surfd_band_create": (C.c_int, [C.c_int, C.c_int64, C.POINTER(_P)]),
    "surfd_band_destroy": (None, [_P]),
    "surfd_band_compact": (C.c_int, [_P, _P, _P, C.c_float, _P]),
    "surfd_band_fetch": (C.c_int, [_P, _P, c_i64p, C.POINTER(_P), C.POINTER(_P)]),
    "surfd_mc_band_threshold": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "surfd_mc_scratch_create": (C.c_int, [C.c_int, C.POINTER(_P)]),
    "surfd_mc_scratch_destroy": (None, [_P]),
    "surfd_mc_udf_band": (C.c_int, [_P, _P, _P, C.c_int64, C.c_int, C.POINTER(_P)]),
    "surfd_mc_num_vertices": (C.c_int64, [_P]),
    "surfd_mc_num_faces": (C.c_int64, [_P]),
    "surfd_mc_copy": (C.c_int, [_P, _P, _P, _P, _P]),
    "surfd_mc_destroy": (None, [_P]),
    "surfd_write_obj": (C.c_int, [C.c_char_p, _P, C.c_int64, _P, C.c_int64]),
    "surfd_xattn_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(_P)]),
    "surfd_xattn_destroy": (None, [_P]),
    "surfd_xattn_set_param": (C.c_int, [_P, C.c_char_p, _P, c_i64p, C.c_int, _P]),
    "surfd_xattn_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    "surfd_mc_lut_count": (C.c_int, []),
    "surfd_mc_lut": (C.c_int, [C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.POINTER(C.c_byte)), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
}
EXPORTED_SYMBOLS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension must be built first "
                "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(L, name)          # AttributeError here = stale/partial library: rebuild
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().surfd_last_error()
        raise RuntimeError(f"libsurfd_hip error {rc}: {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """device (or host) address of a contiguous tensor, None -> NULL"""
    if t is None:
        return None
    assert t.is_contiguous(), "libsurfd_hip needs contiguous buffers"
    return t.data_ptr()


def stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def shape_arr(shape):
    return (C.c_int64 * max(len(shape), 1))(*shape)
