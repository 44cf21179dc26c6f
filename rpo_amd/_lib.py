"""ctypes binding of librpo_hip.so (include/rpo_amd.h).

There is NO fallback: if the shared object is missing or a symbol cannot be
resolved this module raises, and every op in ``rpo_amd.ops`` raises with it.
The oracle under ``oracle/`` is never imported from here.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librpo_hip.so")

RPO_F32, RPO_BF16, RPO_F16 = 0, 1, 2
EPI_NONE, EPI_BIAS, EPI_BIAS_QGELU, EPI_BIAS_RESID, EPI_QGELU_BWD, EPI_PATCH, EPI_LN_BIAS, EPI_LN_BIAS_QGELU = range(8)
E_BADARG, E_SHAPE, E_DTYPE, E_ALIGN, E_WORKSPACE = -1, -2, -3, -4, -5      # include/rpo_amd.h RPO_E_*

c_i64, c_i32, c_f32, c_vp = C.c_int64, C.c_int32, C.c_float, C.c_void_p


class ImageDesc(C.Structure):
    """struct rpo_image_desc (include/rpo_amd.h)."""
    _fields_ = [("src_offset", c_i64), ("width", c_i32), ("height", c_i32),
                ("crop_x", c_i32), ("crop_y", c_i32), ("crop_w", c_i32), ("crop_h", c_i32),
                ("resize_w", c_i32), ("resize_h", c_i32), ("win_x", c_i32), ("win_y", c_i32),
                ("flip", c_i32), ("reserved", c_i32)]


class GemmArgs(C.Structure):
    """struct rpo_gemm_args (include/rpo_amd.h)."""
    _fields_ = [
        ("A", c_vp), ("lda", c_i64),
        ("W", c_vp), ("ldw", c_i64),
        ("C", c_vp), ("ldc", c_i64),
        ("M", c_i32), ("N", c_i32), ("K", c_i32),
        ("in_dtype", c_i32), ("out_dtype", c_i32), ("epilogue", c_i32),
        ("bias", c_vp),
        ("resid", c_vp), ("ldr", c_i64),
        ("aux", c_vp), ("ldaux", c_i64),
        ("aux_row0", c_i32),
        ("skip_row0", c_i32), ("skip_col0", c_i32),
        ("group", c_i32),
        ("split_k", c_i32),
        ("split_stride", c_i64),
        ("tile_config", c_i32),
        ("out2", c_vp), ("ldout2", c_i64),
        ("ln_stats", c_vp),
        ("ln_colsum", c_vp),
        ("ln_eps", c_f32),
        ("seg_rows0", c_i32), ("seg_rows1", c_i32), ("seg1_row0", c_i32),
        ("ln_group", c_i32),
        ("aux_dtype", c_i32),
        ("prefetch", c_vp), ("prefetch_bytes", c_i64),
        ("resid_hi", c_vp), ("resid_lo", c_vp), ("ldr16", c_i64),
        ("out_lo", c_vp),
        ("c_row0", c_i32),
    ]


class LnBwdArgs(C.Structure):
    """struct rpo_ln_bwd_args (include/rpo_amd_experimental.h: the measured-slower experiments, -DRPO_EXPERIMENTAL build)."""
    _fields_ = [("dy", c_vp), ("lddy", c_i64), ("x", c_vp), ("ldx", c_i64), ("gamma", c_vp),
                ("dres", c_vp), ("lddres", c_i64), ("dx", c_vp), ("lddx", c_i64),
                ("dx_cast", c_vp), ("cast_dtype", c_i32), ("ldcast", c_i64),
                ("rows", c_i32), ("d", c_i32), ("eps", c_f32), ("dy_splits", c_i32), ("dy_split_stride", c_i64)]


class AttnBwdArgs(C.Structure):
    """struct rpo_attn_bwd_args (include/rpo_amd_experimental.h: the measured-slower experiments, -DRPO_EXPERIMENTAL build)."""
    _fields_ = [("q_rows", c_vp), ("ldq", c_i64), ("k", c_vp), ("v", c_vp), ("ldkv", c_i64),
                ("dx", c_vp), ("lddx", c_i64), ("w_out_t", c_vp), ("ldw", c_i64), ("dq", c_vp), ("lddq", c_i64),
                ("groups", c_i32), ("H", c_i32), ("keys", c_i32), ("Kp", c_i32),
                ("key_len", c_vp), ("key_stride", c_i32), ("scale", c_f32)]


class ChainLayer(C.Structure):
    """struct rpo_chain_layer (include/rpo_amd_experimental.h: the measured-slower experiments, -DRPO_EXPERIMENTAL build)."""
    _fields_ = [("w_proj_t", c_vp), ("w_fc_t", c_vp), ("w_out_t", c_vp), ("w_q_t", c_vp), ("aux", c_vp),
                ("x_ln2", c_vp), ("x_ln1", c_vp), ("ln2_w", c_vp), ("ln1_w", c_vp),
                ("q_rows", c_vp), ("k", c_vp), ("v", c_vp)]


class ChainBwdArgs(C.Structure):
    """struct rpo_chain_bwd_args (include/rpo_amd_experimental.h: the measured-slower experiments, -DRPO_EXPERIMENTAL build)."""
    _fields_ = [("layer", C.POINTER(ChainLayer)),
                ("layers", c_i32), ("units", c_i32), ("Kp", c_i32), ("d", c_i32), ("H", c_i32), ("keys", c_i32),
                ("dtype", c_i32),
                ("key_len", c_vp), ("key_stride", c_i32),
                ("ldx", c_i64), ("ldq", c_i64), ("ldkv", c_i64),
                ("dxa", c_vp), ("dxb", c_vp), ("dxc", c_vp), ("du", c_vp), ("dq", c_vp),
                ("dy", c_vp), ("dy_stride", c_i64),
                ("scale", c_f32), ("eps", c_f32),
                ("wgs_per_group", c_i32),
                ("state", c_vp),
                ("timeline", c_vp)]


# name -> (restype, argtypes); must list every symbol include/rpo_amd.h declares
SIGNATURES = {
    "rpo_version": (c_i32, []),
    "rpo_gemm_stats_group": (c_i32, [C.POINTER(GemmArgs)]),
    "rpo_gemm_hilo_ok": (c_i32, [C.POINTER(GemmArgs)]),
    "rpo_error_string": (C.c_char_p, [c_i32]),
    "rpo_gemm_nt": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "rpo_gemm_ws": (c_i32, [C.POINTER(GemmArgs), c_vp]),
    "rpo_gemm_ws_pack": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "rpo_gemm_ws_ok": (c_i32, [C.POINTER(GemmArgs)]),
    "rpo_layernorm_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_layernorm_bwd": (c_i32, [c_vp, c_i32, c_i64, c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp,
                                  c_i32, c_i64, c_i32, c_i32, c_f32, c_i32, c_i64, c_vp]),
    "rpo_im2col_patches": (c_i32, [c_vp, c_vp, c_i32, c_i64, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "rpo_img_embed_norm": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32,
                                   c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_img_embed_norm_rows": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32,
                                        c_i32, c_i32, c_i32, c_i32, c_f32, c_i32, c_i32, c_vp]),
    "rpo_img_assemble": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp]),
    "rpo_broadcast_rows": (c_i32, [c_vp, c_vp, c_i64, c_i32, c_i32, c_i32, c_vp]),
    "rpo_reduce_groups": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i32, c_i32, c_vp]),
    "rpo_attn_readonly_fwd": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32,
                                      c_f32, c_vp]),
    "rpo_attn_readonly_fwd_rows": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32, c_i32, c_i32, c_i32,
                                      c_f32, c_i32, c_vp]),
    "rpo_attn_readonly_bwd_proj": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32,
                                           c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_attn_readonly_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_i32,
                                      c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_text_attn_fwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp, c_i32, c_i32,
                                  c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_text_attn_bwd": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_i64, c_i32, c_vp,
                                  c_i32, c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_text_attn_bwd_dense": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_vp, c_i64, c_vp, c_vp, c_vp, c_i64, c_i32, c_vp,
                                        c_i32, c_i32, c_i32, c_f32, c_vp]),
    "rpo_metanet_fwd": (c_i32, [c_vp] * 8 + [c_i32] * 4 + [c_vp]),
    "rpo_metanet_bwd": (c_i32, [c_vp] * 8 + [c_i32] * 4 + [c_vp]),
    "rpo_head_workspace_floats": (c_i64, [c_i32, c_i32, c_i32, c_i32]),
    "rpo_head_fwd_bwd": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32,
                                 c_vp, c_vp]),
    "rpo_head_fwd_bwd_act": (c_i32, [c_vp, c_vp, c_vp, c_f32, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_i32, c_i32, c_i32,
                                     c_i32, c_i32, c_vp, c_vp]),
    "rpo_sgd_step": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp]),
    "rpo_sgd_step_guarded": (c_i32, [c_vp, c_vp, c_vp, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_vp, c_vp]),
    "rpo_convert": (c_i32, [c_vp, c_i64, c_vp, c_i32, c_i64, c_i32, c_i32, c_vp]),
    "rpo_probe_mfma": (c_i32, [c_i32, c_vp, c_vp, c_vp, c_vp]),
    "rpo_probe_peak_mfma": (c_i32, [c_i32, c_i32, c_i32, c_vp, c_vp, c_vp]),
    "rpo_probe_peak_copy": (c_i32, [c_vp, c_vp, c_i64, c_vp]),
    "rpo_preprocess_ksize": (c_i32, [c_i32, c_i32]),
    "rpo_preprocess_workspace_bytes": (C.c_size_t, [c_i32, c_i32, c_i32, c_i32]),
    "rpo_preprocess_batch": (c_i32, [c_vp, c_i64, c_vp, c_vp, c_i32, c_i32, c_i32, c_i32, c_vp, c_vp, c_vp, c_vp,
                                     C.c_size_t, c_vp]),
}

# include/rpo_amd_experimental.h: bound only from the -DRPO_EXPERIMENTAL build of the library (RPO_EXPERIMENTAL=1)
EXPERIMENTAL_SIGNATURES = {
    "rpo_gemm_nt_pair": (c_i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), c_vp]),
    "rpo_mlp_fused": (c_i32, [C.POINTER(GemmArgs), C.POINTER(GemmArgs), c_vp, c_i32, c_vp]),
    "rpo_layernorm_bwd_pair": (c_i32, [C.POINTER(LnBwdArgs), C.POINTER(LnBwdArgs), c_vp]),
    "rpo_attn_bwd_proj_pair": (c_i32, [C.POINTER(AttnBwdArgs), C.POINTER(AttnBwdArgs), c_i32, c_vp]),
    "rpo_chain_state_bytes": (C.c_size_t, []),
    "rpo_chain_bwd": (c_i32, [C.POINTER(ChainBwdArgs), c_vp]),
    "rpo_chain_bwd_ok": (c_i32, [C.POINTER(ChainBwdArgs)]),
}
EXPERIMENTAL = os.environ.get("RPO_EXPERIMENTAL") == "1"


def xenv(name: str, default: str = "0") -> str:
    """Switch of a measured-slower EXPERIMENT: read only when RPO_EXPERIMENTAL=1 -- which also
    selects the -DRPO_EXPERIMENTAL build of the library -- otherwise the default.  The default train step, the eval path
    and the sibling trainers never take these branches."""
    return os.environ.get(name, default) if EXPERIMENTAL else default

_lib = None


class RPOLibraryError(RuntimeError):
    pass


def load(path: str | None = None):
    """dlopen the HIP library and bind every declared symbol (fails loudly)."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.environ.get("RPO_HIP_LIB")                  # RPO_HIP_LIB: A/B a variant build (tools/)
    if p is None and EXPERIMENTAL:                             # the experiments' library, built on first use
        from .build import build_library
        p = build_library(experimental=True)
    p = p or LIB_PATH
    # PyTorch-ROCm bundles its own libamdhip64.so.7; librpo_hip.so must bind to THAT runtime
    # (same SONAME as /opt/rocm's) or the two would hold separate device contexts and torch's
    # pointers/streams would be foreign to our kernels.  Importing torch first makes the dynamic
    # linker resolve our DT_NEEDED entry to the already-resident copy.
    import torch  # noqa: F401
    if not os.path.exists(p):
        raise RPOLibraryError(
            f"{p} not found: build it with `python -m rpo_amd.build` (hipcc, gfx950). "
            "rpo_amd has no CPU fallback.")
    try:
        lib = C.CDLL(p)
    except OSError as e:  # e.g. libamdhip64 missing
        raise RPOLibraryError(f"cannot load {p}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RPOLibraryError(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    for name, (res, args) in EXPERIMENTAL_SIGNATURES.items():   # present only in the -DRPO_EXPERIMENTAL build
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.restype = res
            fn.argtypes = args
    if path is None:
        _lib = lib
    return lib


def experimental():
    """The library with the experiments' entry points (include/rpo_amd_experimental.h), or an error that says how to get it."""
    lib = load()
    if not hasattr(lib, "rpo_chain_bwd"):
        raise RPOLibraryError("this entry point belongs to the measured-slower experiments of rounds 3 / 4: set "
                              "RPO_EXPERIMENTAL=1 (loads rpo_amd/build/librpo_hip_exp.so, built with -DRPO_EXPERIMENTAL)")
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().rpo_error_string(int(rc))
        raise RPOLibraryError(f"{what or 'rpo call'} failed: {msg.decode() if msg else rc} (code {rc})")
