"""ctypes binding of ``libmotioned.so`` (``include/motioned.h``).

The library is the ONLY compute path of this package: there is no PyTorch / CPU fallback.  Loading
fails loudly when the shared object is missing (build it with ``python -m motioneditor_amd.build``
or ``__graft_entry__.build()``); compute calls fail loudly on a non-zero return code.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

LIB_PATH = Path(__import__("os").environ.get("ME_LIB") or Path(__file__).resolve().parent / "libmotioned.so")   # ME_LIB: an A/B build of the same ABI (tools/)

ME_OK, ME_EINVAL, ME_EHIP = 0, -1, -2
ABI_VERSION = 9
GATHER_DENSE, GATHER_CONV3, GATHER_TCONV = 0, 1, 2
SEG_PLAIN, SEG_DUAL_CUR, SEG_DUAL_PREV, SEG_DUAL_BIN = 0, 1, 2, 3

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmArgs(C.Structure):
    _fields_ = [
        ("X", _vp), ("W", _vp), ("C", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32),
        ("ldx", _i32), ("ldc", _i32), ("gather", _i32),
        ("Hin", _i32), ("Win", _i32), ("Hout", _i32), ("Wout", _i32), ("stride", _i32), ("ups", _i32), ("pad0", _i32),
        ("frames", _i32), ("npix", _i32), ("chunk", _i32),
        ("frame0", _i32), ("frames_total", _i32), ("halo_prev", _i32), ("halo_next", _i32),
        ("bias", _vp), ("rowvec", _vp), ("ldrv", _i32), ("rows_per_vec", _i32),
        ("res", _vp), ("ldr", _i32), ("res2", _vp), ("ldr2", _i32), ("geglu", _i32), ("act", _i32), ("alpha", _f32), ("res_rows", _i32), ("res2_rows", _i32),
        ("work", _vp), ("work_bytes", _i64), ("splits_", _i32), ("sel_rows", _i32),
        ("C2", _vp), ("c2_col0", _i32), ("c2_dh", _i32), ("c2_hs", _i64), ("m_off", _i32),
        ("ln_stats", _vp), ("ln_colsum", _vp), ("ln_cvec", _vp), ("ln_stride", _i64), ("ln_parts", _i32), ("ln_eps", _f32),
        ("ln_out", _vp), ("ln_out_stride", _i64),
    ]


class ConvSmallArgs(C.Structure):
    _fields_ = [
        ("inp", _vp), ("W", _vp), ("bias", _vp), ("out", _vp),
        ("n_img", _i32), ("Cin", _i32), ("Cout", _i32), ("H", _i32), ("Wd", _i32),
        ("img_stride", _i64), ("ch_stride", _i64),
        ("in_is_f16", _i32), ("silu", _i32), ("frames", _i32), ("frame_stride", _i64),
    ]


class AttnArgs(C.Structure):
    _fields_ = [
        ("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp),
        ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32),
        ("heads", _i32), ("dh", _i32),
        ("n_items", _i32), ("nq", _i32), ("nk", _i32), ("nseg", _i32),
        ("seg_item", _vp), ("seg_mode", _vp), ("mask", _vp), ("scale", _f32), ("general_dual", _i32),
        ("vsum", _vp), ("n_kv_items", _i32), ("q_items", _i32), ("lse", _vp),
        ("hsk", _i64), ("hsv", _i64), ("hsq", _i64), ("item_order", _vp),
    ]


class AttnBwdArgs(C.Structure):
    _fields_ = [
        ("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp), ("dO", _vp), ("lse", _vp), ("dQ", _vp), ("dK", _vp), ("dV", _vp), ("delta", _vp),
        ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32), ("lddo", _i32), ("lddq", _i32), ("lddk", _i32), ("lddv", _i32),
        ("heads", _i32), ("dh", _i32),
        ("n_items", _i32), ("nq", _i32), ("nk", _i32), ("nseg", _i32), ("n_kv_items", _i32),
        ("seg_item", _vp), ("inv_ptr", _vp), ("inv_item", _vp), ("scale", _f32),
    ]


class GemmDwArgs(C.Structure):
    _fields_ = [
        ("dY", _vp), ("X", _vp), ("dW", _vp), ("work", _vp),
        ("M", _i32), ("N", _i32), ("K", _i32), ("lddy", _i32), ("ldx", _i32), ("dy_is_f16", _i32),
        ("taps", _i32), ("tap", _i32), ("gather", _i32), ("frames", _i32), ("npix", _i32), ("chunk", _i32), ("alpha", _f32),
    ]


class TAttnArgs(C.Structure):
    _fields_ = [
        ("Q", _vp), ("K", _vp), ("V", _vp), ("O", _vp),
        ("ldq", _i32), ("ldk", _i32), ("ldv", _i32), ("ldo", _i32),
        ("heads", _i32), ("dh", _i32),
        ("batch", _i32), ("frames", _i32), ("npix", _i32),
        ("kv_map", _i32 * 8), ("scale", _f32),
        ("q_frames", _i32), ("q_frame0", _i32), ("kv_parts", _i32), ("q_parts", _i32),
    ]


class GroupNormArgs(C.Structure):
    _fields_ = [
        ("X", _vp), ("Y", _vp), ("gamma", _vp), ("beta", _vp), ("stats", _vp),
        ("rows", _i32), ("rows_per_group", _i32),
        ("C", _i32), ("ldx", _i32), ("ldy", _i32), ("groups", _i32), ("eps", _f32), ("silu", _i32),
    ]


class LayerNormArgs(C.Structure):
    _fields_ = [
        ("X", _vp), ("Y", _vp), ("gamma", _vp), ("beta", _vp),
        ("rows", _i32), ("C", _i32), ("ldx", _i32), ("ldy", _i32), ("eps", _f32),
    ]


class PlanStats(C.Structure):
    _fields_ = [("launches", _i64), ("event_records", _i64), ("event_waits", _i64), ("arg_bytes", _i64), ("replays", _i64), ("streams", _i32)]


class PlanNodeInfo(C.Structure):
    _fields_ = [("kind", _i32), ("stream", _i32), ("event", _i32), ("grid", C.c_uint32 * 3), ("block", C.c_uint32 * 3), ("lds_bytes", C.c_uint32),
                ("n_args", _i32), ("arg_bytes", _i64)]


PLAN_LAUNCH, PLAN_RECORD, PLAN_WAIT = 0, 1, 2

# every symbol include/motioned.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "me_abi_version": (C.c_int, []),
    "me_gemm_work_bytes": (_i64, [C.POINTER(GemmArgs)]),
    "me_last_error": (C.c_char_p, []),
    "me_last_kernel": (C.c_char_p, []),
    "me_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "me_gemm": (C.c_int, [C.POINTER(GemmArgs), _vp]),
    "me_conv_small": (C.c_int, [C.POINTER(ConvSmallArgs), _vp]),
    "me_attn": (C.c_int, [C.POINTER(AttnArgs), _vp]),
    "me_tattn": (C.c_int, [C.POINTER(TAttnArgs), _vp]),
    "me_groupnorm": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "me_groupnorm_scratch_bytes": (C.c_int64, [_i32, _i32, _i32]),
    "me_groupnorm_stats": (C.c_int, [C.POINTER(GroupNormArgs), _vp]),
    "me_groupnorm_apply": (C.c_int, [C.POINTER(GroupNormArgs), _i64, _vp]),
    "me_layernorm": (C.c_int, [C.POINTER(LayerNormArgs), _vp]),
    "me_ln_stats": (C.c_int, [_vp, _i32, _i64, _i32, _vp, _i64, _vp]),
    "me_softmax_rows": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "me_axpy_rows": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _f32, _vp]),
    "me_attn_vsum_bytes": (_i64, [_i32, _i32]),
    "me_attn_fallback_blocks": (_i64, [_i32]),
    "me_geglu_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "me_layernorm_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _i32, _i64, _i32, C.c_float, _vp]),
    "me_groupnorm_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, C.c_float, _i32, _vp, _vp]),
    "me_groupnorm_bwd_scratch_bytes": (_i64, [_i32, _i32, _i32]),
    "me_tattn_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _i32, _i32, _i32, _i32, _i32, _i32, C.c_float, _vp]),
    "me_softmax_bwd_rows": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, C.c_float, _vp]),
    "me_relu_bwd": (C.c_int, [_vp, _i32, _vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "me_copy_rows": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _vp]),
    "me_copy_blocks": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i32, _i64, _i32, _i64, _i64, _i64, _i64, _vp]),
    "me_silu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "me_relu": (C.c_int, [_vp, _vp, _i64, _vp]),
    "me_timestep_embed": (C.c_int, [_vp, _i32, _i32, _f32, _vp]),
    "me_cfg_ddim": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _vp]),
    "me_timestep_embed_dev": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "me_cfg_ddim_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp, _vp]),
    "me_gaussian_sample": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _i32, _f32, _vp]),
    "me_attn_bwd": (C.c_int, [C.POINTER(AttnBwdArgs), _vp]),
    "me_gemm_dw": (C.c_int, [C.POINTER(GemmDwArgs), _vp]),
    "me_gemm_dw_work_bytes": (_i64, [_i32, _i32, _i32]),
    "me_colsum": (C.c_int, [_vp, _vp, _i32, _i32, _i64, _i32, _f32, _vp, _vp]),
    "me_colsum_work_bytes": (_i64, [_i32]),
    "me_layernorm_bwd_params": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _i32, _i64, _i32, _f32, _f32, _vp, _vp]),
    "me_layernorm_bwd_params_work_bytes": (_i64, [_i64, _i32]),
    "me_grad_acc": (C.c_int, [_vp, _i32, _vp, _i32, _i32, _i64, _i32, _f32, _i32, _i32, _vp]),
    "me_sumsq_absmax": (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    "me_sumsq_work_bytes": (_i64, []),
    "me_adamw": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _f32, _f32, _f32, _f32, _f32, _f32, _f32, _vp, _f32, _f32, _vp]),
    "me_cast_f16": (C.c_int, [_vp, _vp, _i64, _vp]),
    "me_cast_rows_f16": (C.c_int, [_vp, _i32, _vp, _i32, _i64, _i32, _i32, _vp]),
    "me_mse_seed": (C.c_int, [_vp, _vp, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _i32, _i32, _f32, _f32, _f32, _f32, _vp]),
    "me_nchw_to_rows": (C.c_int, [_vp, _i32, _vp, _i64, _i64, _i32, _i32, _i32, _vp]),
    "me_rows_to_nchw": (C.c_int, [_vp, _i64, _i64, _vp, _i32, _i32, _i32, _i32, _vp]),
    # the step as one call (csrc/plan.hip); me_plan* is an opaque pointer
    "me_plan_begin": (C.c_int, [C.POINTER(_vp), _vp]),
    "me_plan_event_record": (C.c_int, [_vp, C.POINTER(_i32)]),
    "me_plan_event_wait": (C.c_int, [_vp, _i32]),
    "me_plan_end": (C.c_int, [_vp]),
    "me_plan_recording": (C.c_int, []),
    "me_plan_bind": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "me_denoise_step": (C.c_int, [_vp, _vp, _vp, _vp, _f32, _f32, _f32, _f32, _vp]),
    "me_plan_info": (C.c_int, [_vp, C.POINTER(PlanStats)]),
    "me_plan_node": (C.c_int, [_vp, _i64, C.POINTER(PlanNodeInfo), _vp, _i64]),
    "me_plan_destroy": (None, [_vp]),
}

_lib = None


class MotionedError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libmotioned.so once; raise (never fall back) when it is absent."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise MotionedError(
                f"{LIB_PATH} is missing: the HIP library is the only compute path of motioneditor_amd "
                "(no CPU / PyTorch fallback). Build it with `python -m motioneditor_amd.build`.")
        # PyTorch-ROCm bundles its own libamdhip64; libmotioned must bind to THAT runtime instance (it launches on
        # torch's streams and torch's allocations).  Loading the .so before torch would pull /opt/rocm's copy in first
        # and every launch would then fail with hipErrorNoDevice -- so make sure torch is loaded first.
        import torch  # noqa: F401
        L = C.CDLL(str(LIB_PATH))
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(L, name)  # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        if L.me_abi_version() != ABI_VERSION:
            raise MotionedError(f"libmotioned ABI {L.me_abi_version()} != {ABI_VERSION}")
        _lib = L
    return _lib


_SYNC = bool(__import__("os").environ.get("ME_SYNC"))   # debugging aid: synchronise after every call so that a device fault names its launch


def check(rc: int, what: str = "") -> None:
    if _SYNC and rc == ME_OK:
        import sys
        import torch
        print(f"[ME_SYNC] {what} {lib().me_last_kernel().decode()}", file=sys.stderr, flush=True)
        torch.cuda.synchronize()
    if rc != ME_OK:
        msg = lib().me_last_error().decode(errors="replace")
        if rc == ME_EINVAL:
            raise ValueError(f"libmotioned {what}: {msg}")
        raise MotionedError(f"libmotioned {what}: {msg} (rc={rc})")
