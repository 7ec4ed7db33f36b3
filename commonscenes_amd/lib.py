"""ctypes binding of libcommonscenes_hip.so (the C ABI declared in include/commonscenes_hip.h).

The HIP library IS the compute path: loading fails loudly when the .so is missing, and no function
in this package falls back to PyTorch/CPU arithmetic.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

# torch must be imported BEFORE the library is dlopen'ed: both need libamdhip64.so.7 and whichever is
# mapped first serves the whole process.  torch ships its own HIP + HSA runtime pair; letting
# /opt/rocm's HIP runtime in first leaves two HSA runtimes loaded and every launch fails with
# hipErrorNoDevice.  Device buffers and streams are torch's, so its runtime is the one to share.
import torch  # noqa: F401

from .build import LIB_PATH

ABI_VERSION = 18     # cs_abi_version() of the library this module's SIGNATURES table describes
CS_OK = 0
CS_EINVAL = -22
CS_ENOMEM = -12
ACT_NONE, ACT_RELU, ACT_SILU, ACT_GELU, ACT_GEGLU = 0, 1, 2, 3, 4
MATH_FP32, MATH_F16X3 = 0, 1
MATH_F16 = 2        # attention only: plain fp16 operands (reduced precision, opt-in)
# The numerics mode model classes start in: F16X3 (fp32-grade, the benchmarked mode) unless CS_MATH=fp32 asks for the
# fp32-input MFMA kernels.  `set_math()` switches per model.
DEFAULT_MATH = MATH_FP32 if os.environ.get("CS_MATH", "f16x3").lower() == "fp32" else MATH_F16X3

_f = C.c_void_p   # device float*
_i = C.c_int
_l = C.c_int64
_s = C.c_void_p   # hipStream_t
_fl = C.c_float


class CsConvGemm(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("w", C.c_void_p), ("out", C.c_void_p), ("bias", C.c_void_p),
        ("scale", C.c_void_p), ("shift", C.c_void_p), ("rowvec", C.c_void_p), ("res", C.c_void_p),
        ("nb", C.c_int32), ("din", C.c_int32), ("hin", C.c_int32), ("win", C.c_int32),
        ("dout", C.c_int32), ("hout", C.c_int32), ("wout", C.c_int32),
        ("cin", C.c_int32), ("cout", C.c_int32),
        ("lda", C.c_int32), ("ldw", C.c_int32), ("ldo", C.c_int32), ("ldr", C.c_int32),
        ("ldrv", C.c_int32),
        ("kd", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
        ("sd", C.c_int32), ("sh", C.c_int32), ("sw", C.c_int32),
        ("pd", C.c_int32), ("ph", C.c_int32), ("pw", C.c_int32),
        ("ud", C.c_int32), ("uh", C.c_int32), ("uw", C.c_int32),
        ("act", C.c_int32), ("rv_rows", C.c_int32), ("math", C.c_int32), ("tile", C.c_int32),
        ("w_lo", C.c_void_p), ("acc_scale", C.c_float), ("a_scale", C.c_float),
        ("x_lo", C.c_void_p), ("a_format", C.c_int32), ("splitk", C.c_int32),
        ("splitk_ws", C.c_void_p), ("status", C.c_void_p),
        ("gn_part", C.c_void_p), ("gn_ld", C.c_int32), ("gn_rows", C.c_int32), ("out_format", C.c_int32),
        ("out_scale", C.c_float), ("a_bound", C.c_void_p),
        ("splitk_sync", C.c_void_p), ("splitk_sync_words", C.c_int32),
    ]


class CsTransformerStats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("rq", "rk", "rv", "ro", "bo", "rx", "bx", "rg", "bg", "r2", "b2", "rpi", "bpi", "g1",
                                        "be1", "g3", "be3")]


class CsGnSeg(C.Structure):
    _fields_ = [("part", C.c_void_p), ("ld", C.c_int32), ("col0", C.c_int32), ("ch0", C.c_int32), ("nch", C.c_int32),
                ("tiles_per_sample", C.c_int32), ("ncls", C.c_int32), ("nb_src", C.c_int32), ("reserved", C.c_int32)]


class CsDebug(C.Structure):
    """include/commonscenes_hip.h: the debug / A-B switches, parsed ONCE from the CS_* environment by the library"""
    _fields_ = [(n, C.c_int32) for n in (
        "no_split16", "no_pair16", "no_upfold", "no_splitk", "no_fused_geglu", "no_tapcol", "tapcol_tile", "no_cfg_split",
        "concat_copy", "tile512", "no_pw", "no_slab4", "no_attn_img", "attn_nw8", "no_up2_direct", "no_up2_batch",
        "plan_pow2", "slice_tile2", "no_gn_parts", "no_pair_epilogue", "no_dyn_scale", "no_tok_rules", "no_fused_reduce",
        "no_gn_fold", "no_kwave", "no_static_scales", "no_wino", "wino_min_rows", "no_wino43", "wino43_min_rows",
        "no_wino_tail")] + [
        ("split16_min_rows", C.c_int64), ("cfg_split_min_rows", C.c_int64), ("gn_small_group", C.c_int64)]


class CsUnetConfig(C.Structure):
    _fields_ = [
        ("in_channels", C.c_int32), ("out_channels", C.c_int32), ("model_channels", C.c_int32),
        ("num_res_blocks", C.c_int32), ("n_mult", C.c_int32), ("channel_mult", C.c_int32 * 8),
        ("n_attn_res", C.c_int32), ("attention_resolutions", C.c_int32 * 8),
        ("num_heads", C.c_int32), ("context_dim", C.c_int32),
        ("d", C.c_int32), ("h", C.c_int32), ("w", C.c_int32), ("math", C.c_int32),
        ("use_spatial_transformer", C.c_int32), ("dims", C.c_int32),
    ]


class CsVqvaeConfig(C.Structure):
    _fields_ = [
        ("ch", C.c_int32), ("out_ch", C.c_int32), ("n_mult", C.c_int32), ("ch_mult", C.c_int32 * 8),
        ("num_res_blocks", C.c_int32), ("z_channels", C.c_int32), ("resolution", C.c_int32),
        ("n_embed", C.c_int32), ("embed_dim", C.c_int32), ("math", C.c_int32),
    ]


_pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every symbol declared in include/commonscenes_hip.h
SIGNATURES = {
    "cs_conv_gemm": (_i, [C.POINTER(CsConvGemm), _s]),
    "cs_conv_gemm_plan": (_i, [C.POINTER(CsConvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "cs_conv_gemm_epilogue_caps": (_i, [C.POINTER(CsConvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cs_conv_wino_ok": (_i, [C.POINTER(CsConvGemm)]),
    "cs_conv_wino_plan": (_i, [C.POINTER(CsConvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]),
    "cs_conv_wino_plan_info": (_i, [C.POINTER(CsConvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cs_conv_wino_positions": (_i, [C.POINTER(CsConvGemm), _s]),
    "cs_conv_wino_output": (_i, [C.POINTER(CsConvGemm), _s]),
    "cs_conv_gemm_launch_info": (_i, [C.POINTER(CsConvGemm), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "cs_debug": (C.POINTER(CsDebug), []),
    "cs_debug_set": (None, [C.POINTER(CsDebug)]),
    "cs_norm_a_scale": (_fl, [_fl, _fl, _l]),
    "cs_bound_a_scale": (_fl, [_fl]),
    "cs_weight_rowstats": (_i, [_f, _i, _i, _f, _s]),
    "cs_transformer_static_scales": (_i, [C.POINTER(CsTransformerStats), _i, _l, _i, _fl, _fl, _fl, _f]),
    "cs_conv_wants_split16": (_i, [_l, _i, _i, _i, _i, _i]),
    "cs_tapcol_ok": (_i, [_i, _i, _i, _i]),
    "cs_tapcol_tile": (_i, [_l, _i]),
    "cs_groupnorm_finalize_parts": (_i, [C.POINTER(CsGnSeg), _i, _i, _i, _i, _i, _fl, _f, _f, _s]),
    "cs_groupnorm_parts": (_i, [_f, C.POINTER(CsGnSeg), _i, _f, _f, _f, _i, _i, _i, _i, _i, _i, _fl, _i, _f, _f, _s]),
    "cs_conv_up2_info": (_i, [_i, _i, _i, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                              C.POINTER(C.c_int32)]),
    "cs_fold_upsample_weight": (_i, [_f, _f, _i, _i, _i, _i, _i, _s]),
    "cs_conv_gemm_up2_ws_bytes": (_l, [C.POINTER(CsConvGemm)]),
    "cs_conv_gemm_up2": (_i, [C.POINTER(CsConvGemm), _pp, _pp, C.POINTER(C.c_float), C.c_void_p, _s]),
    "cs_conv3d_3x3x3_s111": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _s]),
    "cs_conv3d_3x3x3_s122": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _s]),
    "cs_gemm_tokens": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _s]),
    "cs_relayout_weight": (_i, [_f, _f, _i, _i, _i, _i, _i, _s]),
    "cs_pack_weight_f16x3": (_i, [_f, _f, _f, _i, _i, _i, _fl, _s]),
    "cs_pack_weight_f16x3_tapcol": (_i, [_f, _f, _f, _i, _i, _i, _fl, _s]),
    "cs_pack_weight_f16x3_wino": (_i, [_f, _f, _f, _i, _i, _fl, _s]),
    "cs_pack_weight_f16x3_wino_v": (_i, [_f, _f, _f, _i, _i, _fl, _i, _i, _i, _s]),
    "cs_tapsum27": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "cs_groupnorm_ws_bytes": (_l, [_i, _i]),
    "cs_groupnorm_stats": (_i, [_f, _i, _i, _i, _i, _i, _fl, _f, _f, _s]),
    "cs_groupnorm_stats_bound": (_i, [_f, _i, _i, _i, _i, _i, _fl, _f, _f, _f, _s]),
    "cs_groupnorm_apply": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _s]),
    "cs_groupnorm_apply_split16": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_groupnorm_apply_wino16": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_groupnorm_apply_wino16_range": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_groupnorm_apply_wino_range": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _i, _f, _s]),
    "cs_layernorm_pair16": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _fl, _fl, _f, _s]),
    "cs_groupnorm_apply_range": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _s]),
    "cs_groupnorm_apply_split16_range": (_i, [_f, _f, _f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_groupnorm": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _fl, _i, _f, _f, _s]),
    "cs_groupnorm_silu_ndhwc": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _fl, _f, _f, _s]),
    "cs_layernorm": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _fl, _s]),
    "cs_attn_selfattn": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _s]),
    "cs_attn_selfattn_f16x3": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_attn_selfattn_f16x3_scaled": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _fl, _fl, _fl, _f, _s]),
    "cs_attn_f16x3_ws_bytes": (_l, [_i, _i, _i, _i, _i]),
    "cs_attn_selfattn_f16x3_ws": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _f, _s]),
    "cs_attn_selfattn_f16x3_ws_scaled": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _fl, _fl, _fl, _f, _f, _s]),
    "cs_attnblock_static_scales": (_i, [_fl, _fl, _l, _i, _fl, _fl, _fl, _f]),
    "cs_attn_selfattn_f16": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _i, _i, _i, _i, _i, _fl, _f, _s]),
    "cs_geglu": (_i, [_f, _f, _i, _i, _i, _i, _s]),
    "cs_copy_rows": (_i, [_f, _f, _l, _i, _i, _i, _s]),
    "cs_add_rowvec": (_i, [_f, _f, _l, _i, _i, _i, _i, _s]),
    "cs_nchw_to_ndhwc": (_i, [_f, _f, _i, _i, _i, _i, _s]),
    "cs_absmax": (_i, [_f, _l, _f, _s]),
    "cs_ndhwc_to_nchw": (_i, [_f, _f, _i, _i, _i, _i, _s]),
    "cs_timestep_embedding": (_i, [_f, _f, _i, _i, _fl, _s]),
    "cs_ddim_cfg_update": (_i, [_f, _f, _f, _f, _f, _l, _l, _fl, _fl, _fl, _fl, _fl, _i, _s]),
    "cs_ddim_coefficients": (_i, [_fl, _fl, _fl, _fl, _f]),
    "cs_ddim_cfg_update_dev": (_i, [_f, _f, _f, _f, _f, _l, _l, _f, _fl, _i, _s]),
    "cs_plms_update": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _l, _l, _i, _fl, _fl, _fl, _fl, _i, _s]),
    "cs_chamfer_backward": (_i, [_f, _f, _f, _f, _f, _f, _f, _f, _i, _i, _i, _s]),
    "cs_emd_approxmatch": (_i, [_f, _f, _f, _f, _i, _i, _i, _s]),
    "cs_emd_matchcost": (_i, [_f, _f, _f, _f, _i, _i, _i, _s]),
    "cs_emd_matchcost_grad": (_i, [_f, _f, _f, _f, _f, _i, _i, _i, _s]),
    "cs_mc_blocks_per_object": (_i, [_i]),
    "cs_mc_count": (_i, [_f, _i, _i, _fl, _i, _f, _s]),
    "cs_mc_emit": (_i, [_f, _i, _i, _fl, _i, _f, _f, _f, _f, _f, _f, _fl, _fl, _s]),
    "cs_chamfer_nm_distance": (_i, [_f, _f, _f, _f, _i, _i, _i, _s]),
    "cs_unet_create": (_i, [C.POINTER(CsUnetConfig), _pp]),
    "cs_unet_destroy": (None, [C.c_void_p]),
    "cs_unet_param_count": (_i, [C.c_void_p]),
    "cs_unet_param_info": (_i, [C.c_void_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int64 * 5), C.POINTER(C.c_int),
                                C.POINTER(C.c_int64)]),
    "cs_unet_raw_bytes": (_l, [C.c_void_p]),
    "cs_unet_arena_bytes": (_l, [C.c_void_p]),
    "cs_unet_context_floats": (_l, [C.c_void_p]),
    "cs_unet_set_context_bounds": (_i, [C.c_void_p, C.POINTER(C.c_float), _i]),
    "cs_unet_pack": (_i, [C.c_void_p, _f, _f, _s]),
    "cs_unet_workspace_bytes": (_l, [C.c_void_p, _i, _i]),
    "cs_unet_context": (_i, [C.c_void_p, _f, _f, _i, _f, _f, _f, _l, _s]),
    "cs_unet_step": (_i, [C.c_void_p, _f, _f, _f, _f, _f, _i, _i, _f, _f, _l, _s]),
    "cs_vq_argmin_lookup": (_i, [_f, _f, _f, _f, _l, _i, _i, _i, _i, _s]),
    "cs_gcn_gather_cat": (_i, [_f, _f, _f, _f, _i, _i, _i, _i, _f, _s]),
    "cs_gcn_segment_mean": (_i, [_f, _f, _f, _i, _i, _i, _i, _i, _f, _s]),
    "cs_gcn_csr_ints": (_l, [_i, _i]),
    "cs_gcn_csr_build": (_i, [_f, _f, _i, _i, _f, _s]),
    "cs_gcn_segment_mean_csr": (_i, [_f, _f, _f, _i, _i, _i, _i, _s]),
    "cs_embedding": (_i, [_f, _f, _f, _i, _i, _i, _i, _f, _s]),
    "cs_log_softmax": (_i, [_f, _f, _i, _i, _i, _i, _s]),
    "cs_synth_fill": (_i, [_f, _l, C.c_uint64, C.c_double, C.c_double, _s]),
    "cs_vqvae_create": (_i, [C.POINTER(CsVqvaeConfig), _pp]),
    "cs_vqvae_destroy": (None, [C.c_void_p]),
    "cs_vqvae_param_count": (_i, [C.c_void_p]),
    "cs_vqvae_param_info": (_i, [C.c_void_p, _i, C.POINTER(C.c_char_p), C.POINTER(C.c_int64 * 5), C.POINTER(C.c_int),
                                 C.POINTER(C.c_int64)]),
    "cs_vqvae_raw_bytes": (_l, [C.c_void_p]),
    "cs_vqvae_arena_bytes": (_l, [C.c_void_p]),
    "cs_vqvae_pack": (_i, [C.c_void_p, _f, _f, _s]),
    "cs_vqvae_workspace_bytes": (_l, [C.c_void_p, _i]),
    "cs_vqvae_decode": (_i, [C.c_void_p, _f, _f, _f, _f, _i, _i, _f, _f, _l, _s]),
    "cs_abi_version": (_i, []),
}

_LIB = None


class NativeLibraryMissing(RuntimeError):
    pass


def load(path: Path | None = None) -> C.CDLL:
    """Load the HIP library; raise (never fall back) if it is absent or lacks a symbol."""
    global _LIB
    if _LIB is not None and path is None:
        return _LIB
    # CS_LIB_PATH (A/B timing of compile-time what-if builds only: tools/build_alt.sh writes them next to the product
    # library): another build of the SAME sources and ABI; never set in the product
    alt = os.environ.get("CS_LIB_PATH")
    p = Path(path) if path else (Path(alt) if alt else LIB_PATH)
    if not p.exists():
        raise NativeLibraryMissing(
            f"{p} not found. Build it with `python -m commonscenes_amd.build` "
            "(hipcc --offload-arch=gfx950); there is no CPU fallback.")
    lib = C.CDLL(str(p))
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise NativeLibraryMissing(f"{p} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    got = lib.cs_abi_version()
    if got != ABI_VERSION:
        raise NativeLibraryMissing(f"{p} has ABI version {got}, this package expects {ABI_VERSION}: rebuild it "
                                   "(`python -m commonscenes_amd.build --force`)")
    if path is None:
        _LIB = lib
    return lib


class CsError(RuntimeError):
    pass


class CsOverflowError(CsError):
    """A CS_MATH_F16X3 kernel met an activation beyond the fp16 range (CS_STATUS_F16X3_OVERFLOW)."""


class CsSplitKTimeout(CsError):
    """A reducer of a fused split-K launch gave up waiting for a slice that never reached a CU (CS_STATUS_SPLITK_TIMEOUT):
    the launch was not resident.  The host classes re-run with the two-kernel form."""


STATUS_F16X3_OVERFLOW = 1
STATUS_INTERNAL = 2
STATUS_SPLITK_TIMEOUT = 4


def check(rc: int, what: str) -> None:
    if rc != CS_OK:
        kind = {CS_EINVAL: "invalid argument", CS_ENOMEM: "workspace too small"}.get(rc, f"hipError_t {rc}")
        raise CsError(f"{what} failed: {kind}")


def debug() -> CsDebug:
    """The library's CsDebug struct (live view): the ONE parse of the CS_* A/B switches, shared by every host."""
    return load().cs_debug().contents


def debug_set(**fields) -> None:
    """Flip CsDebug switches for the REST of the process (no restore): what a host does when it must leave a route for good,
    e.g. `debug_set(no_fused_reduce=1)` after CS_STATUS_SPLITK_TIMEOUT."""
    new = CsDebug.from_buffer_copy(debug())
    for k, v in fields.items():
        if not hasattr(new, k):
            raise AttributeError(f"CsDebug has no field {k!r}")
        setattr(new, k, int(v))
    load().cs_debug_set(C.byref(new))


class debug_override:
    """`with lib.debug_override(no_gn_parts=1): ...` -- flip switches inside one process (tests, A/B tools); restores the
    previous state on exit.  cs_debug_set copies the struct, so the library and every host see the change at once."""

    def __init__(self, **fields):
        self.fields = fields

    def __enter__(self):
        cur = debug()
        self.prev = CsDebug.from_buffer_copy(cur)
        new = CsDebug.from_buffer_copy(cur)
        for k, v in self.fields.items():
            if not hasattr(new, k):
                raise AttributeError(f"CsDebug has no field {k!r}")
            setattr(new, k, int(v))
        load().cs_debug_set(C.byref(new))
        return self

    def __exit__(self, *exc):
        load().cs_debug_set(C.byref(self.prev))
        return False
