"""ctypes binding of libdm_hip.so (include/dm_hip.h).  There is NO fallback: if the library is
missing or a call fails, this raises."""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported BEFORE libdm_hip.so is dlopen'ed: the PyTorch-ROCm wheel bundles its own
# libamdhip64; loading ours first would bring a second HIP runtime (the system one) into the process,
# and kernels launched through it see "no ROCm-capable device".  With torch first, our DT_NEEDED
# libamdhip64 resolves to the already-loaded runtime, so streams and device pointers are shared.
import torch  # noqa: F401  (side effect: loads the HIP runtime the whole process will use)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdm_hip.so")

c_void_p, c_long, c_int, c_float, c_ulonglong = C.c_void_p, C.c_long, C.c_int, C.c_float, C.c_ulonglong


class ConvArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_long), ("w", c_void_p), ("bias", c_void_p),
                ("chan_add", c_void_p), ("res", c_void_p), ("ldr", c_long), ("y", c_void_p),
                ("ldy", c_long), ("yf", c_void_p), ("ldyf", c_long),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("mode", c_int), ("n_valid", c_int), ("pool2", c_int), ("act_silu_in", c_int),
                ("drop_p", c_float), ("drop_seed", c_ulonglong),
                ("norm_scale", c_void_p), ("norm_shift", c_void_p)]


class GemmArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_long), ("w", c_void_p), ("bias", c_void_p),
                ("chan_add", c_void_p), ("res", c_void_p), ("ldr", c_long), ("y", c_void_p),
                ("ldy", c_long), ("yf", c_void_p), ("ldyf", c_long),
                ("M", c_long), ("K", c_int), ("N", c_int), ("rows_per_batch", c_int),
                ("n_valid", c_int), ("act_silu_in", c_int)]


class WgradArgs(C.Structure):
    _fields_ = [("x", c_void_p), ("ldx", c_long), ("dy", c_void_p), ("lddy", c_long),
                ("dw", c_void_p), ("dbias", c_void_p),
                ("B", c_int), ("H", c_int), ("W", c_int), ("Cin", c_int), ("Cout", c_int),
                ("taps", c_int), ("mode", c_int), ("act_silu_in", c_int), ("n_valid", c_int),
                ("k_valid", c_int), ("workspace", c_void_p), ("ws_bytes", c_long),
                ("norm_scale", c_void_p), ("norm_shift", c_void_p)]


class AttnArgs(C.Structure):
    _fields_ = [("q", c_void_p), ("ldq", c_long), ("k", c_void_p), ("ldk", c_long),
                ("v", c_void_p), ("ldv", c_long), ("o", c_void_p), ("ldo", c_long),
                ("lse", c_void_p), ("d_o", c_void_p), ("lddo", c_long),
                ("dq", c_void_p), ("lddq", c_long), ("dk", c_void_p), ("lddk", c_long),
                ("dv", c_void_p), ("lddv", c_long), ("delta", c_void_p),
                ("B", c_int), ("heads", c_int), ("Nq", c_int), ("Nk", c_int), ("hd", c_int),
                ("scale", c_float), ("drop_p", c_float), ("drop_seed", c_ulonglong),
                ("workspace", c_void_p), ("ws_bytes", c_long)]


class Cross1Args(C.Structure):
    _fields_ = [("v", c_void_p), ("ldv", c_long), ("w_out", c_void_p), ("res", c_void_p), ("ldr", c_long),
                ("y", c_void_p), ("ldy", c_long), ("dy", c_void_p), ("lddy", c_long), ("u", c_void_p),
                ("dw_out", c_void_p), ("dv", c_void_p), ("lddv", c_long), ("workspace", c_void_p), ("ws_bytes", c_long),
                ("B", c_int), ("N", c_int), ("heads", c_int), ("C", c_int),
                ("drop_p", c_float), ("drop_seed", c_ulonglong)]


_SIGS = {
    "dm_abi_version": (c_int, []),
    "dm_target_arch": (C.c_char_p, []),
    "dm_last_hip_error": (C.c_char_p, []),
    "dm_note_hip_error": (None, [c_int]),
    "dm_last_variant": (C.c_char_p, []),
    "dm_set_dropout_counter": (c_int, [c_void_p]),
    "dm_bump_dropout_counter": (c_int, [c_void_p, c_void_p]),
    "dm_packed_weight_bytes": (c_long, [c_int, c_int, c_int]),
    "dm_pack_weights_batch": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dm_pack_weight": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dm_conv3x3": (c_int, [C.POINTER(ConvArgs), c_void_p]),
    "dm_gemm": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "dm_wgrad_workspace_bytes": (c_long, [c_int] * 7),
    "dm_wgrad": (c_int, [C.POINTER(WgradArgs), c_void_p]),
    "dm_gn_workspace_bytes": (c_long, [c_int] * 4),
    "dm_gn_silu_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p,
                               c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_void_p]),
    "dm_gn_prepare": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_int, c_int, c_int, c_int, c_float, c_void_p, c_long, c_void_p]),
    "dm_gn_silu_bwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p,
                               c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int,
                               c_int, c_int, c_float, c_ulonglong, c_void_p, c_void_p, c_long, c_void_p]),
    "dm_attn_fwd": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "dm_attn_bwd": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "dm_attn_bwd_workspace_bytes": (c_long, [c_int] * 5),
    "dm_cross1_fwd": (c_int, [C.POINTER(Cross1Args), c_void_p]),
    "dm_cross1_bwd": (c_int, [C.POINTER(Cross1Args), c_void_p]),
    "dm_cross1_bwd_workspace_bytes": (c_long, [c_int] * 3),
    "dm_nchw_f32_to_nhwc_bf16": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p]),
    "dm_nhwc_to_nchw_f32": (c_int, [c_void_p, c_int, c_long, c_void_p, c_int, c_int, c_int, c_void_p]),
    "dm_cast_f32_to_bf16": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "dm_cast_bf16_to_f32": (c_int, [c_void_p, c_void_p, c_long, c_void_p]),
    "dm_add_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "dm_silu_bwd_bf16": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "dm_film_fwd": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_float, c_ulonglong,
                            c_void_p]),
    "dm_film_bwd_chunks": (c_int, [c_int]),
    "dm_film_bwd": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int,
                            c_int, c_void_p]),
    "dm_colsum_bf16": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dm_batchsum_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dm_batchsum2_cast_f32": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "dm_qsample_vtarget": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long, c_void_p]),
    "dm_mse_workspace_bytes": (c_long, [c_int]),
    "dm_mse_loss_fwd_bwd": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_long,
                                    c_void_p, c_long, c_void_p]),
    "dm_init_device": (c_int, []),
    "dm_mse_noalloc_max_batch": (c_int, []),
    "dm_mse_loss_fwd_bwd_noalloc": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_int, c_long,
                                            c_void_p]),
    "dm_cfg_ddpm_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long,
                                 c_void_p]),
    "dm_cfg_dpmpp2m_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_long,
                                    c_void_p]),
    "dm_sqnorm_workspace_bytes": (c_long, []),
    "dm_grad_norm_clipcoef": (c_int, [c_void_p, c_long, c_float, c_float, c_void_p, c_void_p, c_long, c_void_p]),
    "dm_adamw_ema_step": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_float,
                                  c_float, c_float, c_float, c_float, c_int, C.c_double, c_float, c_int, c_void_p]),
    "dm_opt_state_bytes": (c_int, []),
    "dm_opt_state_init": (c_int, [c_void_p, c_int, c_int, c_void_p]),
    "dm_opt_state_advance": (c_int, [c_void_p, C.c_double, c_int, c_int, C.c_double, C.c_double, c_void_p]),
    "dm_adamw_ema_step_dev": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p,
                                      c_float, c_float, c_float, c_float, C.c_double, c_float, c_int, c_void_p]),
    "dm_ring_fetch": (c_int, [c_void_p, c_void_p, c_long, c_int, c_void_p, c_void_p]),
    # ---- fp32 compute mode (include/dm_hip.h, last section)
    "dm_packed_weight_f32_bytes": (c_long, [c_int, c_int, c_int]),
    "dm_pack_weight_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dm_conv3x3_f32": (c_int, [C.POINTER(ConvArgs), c_void_p]),
    "dm_gemm_f32": (c_int, [C.POINTER(GemmArgs), c_void_p]),
    "dm_pool2_f32": (c_int, [c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_int, c_void_p]),
    "dm_wgrad_f32_workspace_bytes": (c_long, [c_int] * 7),
    "dm_wgrad_f32": (c_int, [C.POINTER(WgradArgs), c_void_p]),
    "dm_gn_silu_fwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_void_p, c_void_p, c_long, c_void_p,
                                   c_int, c_int, c_int, c_int, c_float, c_int, c_void_p, c_long, c_void_p]),
    "dm_gn_f32_workspace_bytes": (c_long, [c_int, c_int]),
    "dm_gn_silu_bwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_void_p, c_void_p,
                                   c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_int, c_float, c_ulonglong, c_void_p, c_void_p, c_long, c_void_p]),
    "dm_attn_fwd_f32": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "dm_attn_bwd_f32": (c_int, [C.POINTER(AttnArgs), c_void_p]),
    "dm_nchw_to_nhwc_f32": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_long, c_int, c_void_p]),
    "dm_add_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "dm_silu_bwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_long, c_long, c_int, c_void_p]),
    "dm_colsum_f32": (c_int, [c_void_p, c_long, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "dm_film_fwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_void_p, c_long, c_int, c_int, c_int, c_float, c_ulonglong,
                                c_void_p]),
    "dm_film_bwd_f32": (c_int, [c_void_p, c_long, c_void_p, c_long, c_void_p, c_void_p, c_long, c_void_p, c_int, c_int,
                                c_int, c_void_p]),
}

EXPORTS = tuple(_SIGS)


class DmError(RuntimeError):
    pass


_lib = None


def lib():
    """Load (once) and return the ctypes handle.  Raises if the HIP library is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DmError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
                          f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU/torch fallback.")
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(h, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = h
    return _lib


def check(code: int, what: str) -> None:
    if code != 0:
        detail = {-1: "bad argument", -3: "size beyond the entry point's fixed maximum"}.get(code, "unknown")
        if code == -2:
            detail = "launch error: " + (lib().dm_last_hip_error() or b"?").decode()
        raise DmError(f"{what} failed with code {code} ({detail})")
