"""ctypes binding of libpromptir_hip.so (the C ABI declared in include/promptir_hip.h).

There is deliberately NO fallback: if the shared library is missing, was built for
another architecture, or lacks a symbol, importing this module raises.  The product
path never routes through PyTorch eager ops or the CPU oracle.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PIR_LIB", os.path.join(_HERE, "libpromptir_hip.so"))  # PIR_LIB: A/B builds in tools/
ABI_VERSION = 8

c_float_p = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())
c_long = C.c_long
c_int = C.c_int
c_size_t = C.c_size_t
c_float = C.c_float
stream_t = C.c_void_p


class GemmNN(C.Structure):
    _fields_ = [
        ("A", c_float_p), ("a_s1", c_long), ("a_s2", c_long), ("a_sm", c_long), ("a_sk", c_long),
        ("X", c_float_p), ("x_s1", c_long), ("x_s2", c_long), ("ldx", c_long),
        ("Y", c_float_p), ("y_s1", c_long), ("y_s2", c_long), ("ldy", c_long),
        ("R", c_float_p), ("r_s1", c_long), ("r_s2", c_long), ("ldr", c_long),
        ("rowscale", c_float_p), ("rs_s1", c_long), ("rs_s2", c_long),
        ("M", c_int), ("K", c_int), ("N", c_int), ("O1", c_int), ("O2", c_int),
        ("A3", c_float_p), ("a3_kp", c_int),
    ]


class GemmNT(C.Structure):
    _fields_ = [
        ("X", c_float_p), ("x_s1", c_long), ("x_s2", c_long), ("x_sr", c_long), ("ldx", c_long),
        ("Y", c_float_p), ("y_s1", c_long), ("y_s2", c_long), ("y_sr", c_long), ("ldy", c_long),
        ("G", c_float_p), ("g_so", c_long), ("g_si", c_long), ("g_sj", c_long),
        ("M1", c_int), ("M2", c_int), ("N", c_int), ("O1", c_int), ("O2", c_int), ("BR", c_int),
        ("shift_dh", c_int), ("shift_dw", c_int), ("H", c_int), ("W", c_int),
        ("ws", c_float_p), ("ws_floats", c_size_t),
        ("alpha", c_float), ("accumulate", c_int),
    ]


class SplitDesc(C.Structure):
    _fields_ = [("W", c_float_p), ("out", c_float_p), ("st", c_long), ("sm", c_long), ("sk", c_long),
                ("M", c_int), ("K", c_int), ("taps", c_int), ("flip", c_int)]


P, L, I, Z, F, S = c_float_p, c_long, c_int, c_size_t, c_float, stream_t

# name -> (restype, argtypes); must list EVERY symbol of include/promptir_hip.h
SIGNATURES = {
    "pir_abi_version": (I, []),
    "pir_arch": (C.c_char_p, []),
    "pir_tune_set": (I, [I, I]),
    "pir_build_flags": (I, []),
    "pir_gemm_nn": (I, [C.POINTER(GemmNN), S]),
    "pir_gemm_nn_ws": (I, [C.POINTER(GemmNN), P, Z, S]),
    "pir_gemm_nn_ws_floats": (Z, [C.POINTER(GemmNN)]),
    "pir_gemm_nn_plan": (I, [C.POINTER(GemmNN)]),
    "pir_split_bf16x3_bytes": (Z, [I, I]),
    "pir_split_bf16x3": (I, [P, I, I, L, L, P, S]),
    "pir_conv3x3": (I, [P, L, L, L, I, P, L, P, L, P, L, I, I, I, I, I, S]),
    "pir_gemm_nt_ws_floats": (Z, [I, I, I, I, I]),
    "pir_gemm_nt": (I, [C.POINTER(GemmNT), S]),
    "pir_gemm_nt_partials": (I, [C.POINTER(GemmNT), C.POINTER(C.c_int), S]),
    "pir_gemm_nt_group": (I, [C.POINTER(GemmNT), I, S]),
    "pir_gemm_nt_ws_needed": (Z, [C.POINTER(GemmNT)]),
    "pir_gemm_nt_group_ws_needed": (Z, [C.POINTER(GemmNT), I, I]),
    "pir_split_bf16x3_batch": (I, [P, P, I, S]),
    "pir_split_bf16x3_taps": (I, [P, I, I, L, L, L, I, P, S]),
    "pir_conv3x3_x3": (I, [P, I, P, L, P, L, P, L, I, I, I, I, I, S]),
    "pir_conv3x3_x3_ws": (I, [P, I, P, L, P, L, P, L, I, I, I, I, I, P, Z, S]),
    "pir_conv3x3_x3_ws_floats": (Z, [I, I, I, I, I]),
    "pir_conv3x3_wgrad_ws_floats": (Z, [I, I, I, I, I]),
    "pir_conv3x3_wgrad": (I, [P, L, P, L, P, I, I, I, I, I, P, Z, I, S]),
    "pir_layernorm_fwd": (I, [P, L, P, P, P, L, P, P, I, I, I, S]),
    "pir_layernorm_bwd_ws_floats": (Z, [I, I, I]),
    "pir_layernorm_bwd": (I, [P, L, P, L, P, I, P, P, P, L, P, L, P, P, P, Z, I, I, I, S]),
    "pir_dwconv3x3": (I, [P, L, P, I, P, L, I, I, I, I, S]),
    "pir_dwconv3x3_gate": (I, [P, L, P, P, L, I, I, I, I, S]),
    "pir_dwconv3x3_gate_bwd": (I, [P, L, P, P, L, P, L, I, I, I, I, S]),
    "pir_dwconv3x3_wgrad_ws_floats": (Z, [I, I, I, I]),
    "pir_dwconv3x3_wgrad": (I, [P, L, P, L, P, P, Z, I, I, I, I, S]),
    "pir_dwconv3x3_bwd_ws_floats": (Z, [I, I, I, I]),
    "pir_dwconv3x3_bwd": (I, [P, L, P, L, P, P, L, P, P, Z, I, I, I, I, S]),
    "pir_gdfn_dwconv_bwd_ws_floats": (Z, [I, I, I, I]),
    "pir_gdfn_dwconv_bwd": (I, [P, L, P, P, L, P, L, P, P, Z, I, I, I, I, S]),
    "pir_gdfn_fused_ws_bytes": (Z, [I, I, I, I]),
    "pir_gdfn_fused_fwd": (I, [P, L, P, P, P, I, P, P, L, P, Z, P, P, I, I, I, I, I, S]),
    "pir_row_sumsq": (I, [P, L, P, I, I, I, S]),
    "pir_mdta_softmax_fwd": (I, [P, P, I, P, P, I, I, I, S]),
    "pir_mdta_softmax_bwd": (I, [P, P, P, P, I, P, P, P, P, P, I, I, I, S]),
    "pir_mdta_softmax_fwd_parts": (I, [P, I, P, I, P, P, P, I, I, I, S]),
    "pir_mdta_softmax_bwd_parts": (I, [P, I, P, P, P, I, P, P, P, P, P, I, I, I, S]),
    "pir_dwconv3x3_sumsq_floats": (Z, [I, I, I]),
    "pir_dwconv3x3_sumsq": (I, [P, L, P, P, L, P, Z, I, C.POINTER(C.c_int), I, I, I, I, S]),
    "pir_pixel_unshuffle2": (I, [P, L, P, L, I, I, I, I, S]),
    "pir_pixel_shuffle2": (I, [P, L, P, L, I, I, I, I, S]),
    "pir_spatial_mean": (I, [P, L, P, I, I, I, S]),
    "pir_prompt_mix_fwd": (I, [P, P, P, P, I, I, I, S]),
    "pir_prompt_resize_fwd": (I, [P, P, P, L, I, I, I, I, I, I, S]),
    "pir_prompt_resize_bwd_ws_floats": (Z, [I, I, I, I, I, I]),
    "pir_prompt_resize_bwd": (I, [P, L, P, P, P, P, P, Z, I, I, I, I, I, I, S]),
    "pir_prompt_mix_bwd": (I, [P, P, P, P, P, P, P, L, I, I, I, I, I, S]),
    "pir_tiles_gather": (I, [P, L, P, I, I, I, I, I, I, I, I, I, I, I, I, I, S]),
    "pir_tiles_blend": (I, [P, P, L, I, I, I, I, I, I, I, I, I, I, I, I, I, S]),
    "pir_l1_loss": (I, [P, P, P, P, F, F, P, L, S]),
    "pir_l1_loss_grad": (I, [P, P, P, F, P, L, S]),
    "pir_copy_strided4": (I, [P, L, L, L, L, P, I, I, I, I, S]),
    "pir_degrade_gaussian": (I, [P, P, P, P, L, I, S]),
    "pir_crop_augment_u8": (I, [P, P, P, P, P, P, I, I, S]),
    "pir_copy_planes": (I, [P, L, P, L, I, I, L, S]),
    "pir_add": (I, [P, P, P, L, S]),
    "pir_bias_add": (I, [P, L, P, I, I, I, S]),
    "pir_bias_grad": (I, [P, L, P, I, I, I, S]),
    "pir_gelu_gate": (I, [P, L, P, L, I, I, I, S]),
    "pir_gelu_gate_bwd": (I, [P, L, P, L, P, L, I, I, I, S]),
    "pir_reduce_partials": (I, [P, L, I, F, I, P, L, S]),
    "pir_reduce_defer": (I, [S, I]),
    "pir_reduce_flush": (I, [S]),
    "pir_reduce_pending": (I, [S]),
    "pir_reduce_defer_limit": (L, [L]),
    "pir_ln_conv1x1_fwd": (I, [P, L, P, P, P, I, P, L, P, P, I, I, I, I, S]),
    "pir_conv1x1_wgrad_ln": (I, [P, L, P, L, P, P, P, P, P, P, Z, I, I, I, I, S]),
    "pir_conv1x1_dgrad_ln_bwd": (I, [P, L, P, I, I, P, L, P, P, P, P, L, P, L, P, P, P, Z, I, I, I, S]),
    "pir_mdta_dqk": (I, [P, P, L, L, P, P, P, L, L, I, I, I, I, S]),
    "pir_adamw_step": (I, [P, P, P, P, L, F, F, F, F, F, L, F, S]),
}


class HipLibraryError(RuntimeError):
    pass


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise HipLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C promptir_amd/csrc`). promptir_amd has no CPU / eager fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise HipLibraryError(f"{LIB_PATH} does not export {name}; rebuild the library") from e
        fn.restype = restype
        fn.argtypes = argtypes
    got = lib.pir_abi_version()
    if got != ABI_VERSION:
        raise HipLibraryError(f"ABI mismatch: library reports {got}, binding expects {ABI_VERSION}; rebuild")
    raw_tune = lib.pir_tune_set

    def tune_set(knob, value):      # every knob change is counted: host-side caches of "which shapes a kernel serves" key on it
        KNOB_EPOCH[0] += 1
        return raw_tune(knob, value)

    lib.pir_tune_set = tune_set
    return lib


KNOB_EPOCH = [0]


lib = _load()


def check(status: int, what: str) -> None:
    if status != 0:
        kind = {-22: "invalid argument", -12: "workspace too small"}.get(status, f"hipError {status}")
        raise RuntimeError(f"promptir_hip: {what} failed ({kind})")
