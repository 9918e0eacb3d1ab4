"""ctypes binding of libomgsr_hip.so (include/omgsr_hip.h).

`import torch` happens BEFORE the library is mapped so that the process holds a single HIP runtime
(torch bundles libamdhip64.so.7; the library's DT_NEEDED resolves to the already-loaded SONAME).
There is no CPU fallback: if the library is missing or the device is not gfx950 the product path
raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

from .build import lib_path

ABI_VERSION = 17


class OmgsrError(RuntimeError):
    pass


class IgemmArgs(C.Structure):
    _fields_ = [
        ("in_", C.c_void_p), ("weight", C.c_void_p), ("bias", C.c_void_p), ("gate", C.c_void_p),
        ("residual", C.c_void_p), ("out", C.c_void_p),
        ("N", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("Cin", C.c_int32),
        ("Cout", C.c_int32), ("Cout_pad", C.c_int32), ("K_pad", C.c_int32),
        ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("pad_top", C.c_int32),
        ("pad_left", C.c_int32), ("upsample", C.c_int32),
        ("Ho", C.c_int32), ("Wo", C.c_int32),
        ("act", C.c_int32), ("out_dtype", C.c_int32), ("out_layout", C.c_int32),
        ("t_rows", C.c_int32), ("t_ld", C.c_int32), ("out_ld", C.c_int32),
        ("batch", C.c_int32),
        ("in_bstride", C.c_int64), ("w_bstride", C.c_int64), ("out_bstride", C.c_int64),
        ("alpha", C.c_float), ("weight_cm", C.c_void_p), ("workspace", C.c_void_p),
        ("gn_partial", C.c_void_p), ("gn_groups", C.c_int32), ("gn_entries", C.c_int32),
        ("res_el", C.c_int32), ("in_split", C.c_int32), ("sample_rows", C.c_int64), ("out_lo_off", C.c_int32),
        ("in_ld", C.c_int32), ("w_split", C.c_int32), ("weight_ph", C.c_void_p), ("mx_chunks16", C.c_int32), ("mx_scale_w1", C.c_int32), ("mx_scale_a1", C.c_int32),
        ("mx_scale_w2", C.c_int32), ("mx_scale_a2", C.c_int32), ("out_mx", C.c_int32), ("group_tiles", C.c_int32), ("overflow_flag", C.c_void_p),
        ("gn_scale_shift", C.c_void_p), ("gn_nimg", C.c_int32), ("gn_act", C.c_int32), ("in_el", C.c_int32), ("mx_fmt", C.c_int32),
    ]


GN_MAX_GROUPS = 8


class GnMergeArgs(C.Structure):
    _fields_ = [("partial", C.c_void_p * GN_MAX_GROUPS), ("count", C.c_double * GN_MAX_GROUPS),
                ("weight", C.c_float * GN_MAX_GROUPS), ("tiles", C.c_int32 * GN_MAX_GROUPS),
                ("nslot", C.c_int32 * GN_MAX_GROUPS), ("entries", C.c_int32 * GN_MAX_GROUPS), ("ngroups", C.c_int32)]


class AttnArgs(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("k", C.c_void_p), ("vt", C.c_void_p), ("o", C.c_void_p),
        ("B", C.c_int32), ("H", C.c_int32), ("D", C.c_int32), ("Lq", C.c_int32), ("Lk", C.c_int32),
        ("q_ld", C.c_int64), ("k_ld", C.c_int64), ("vt_ld", C.c_int64), ("o_ld", C.c_int64),
        ("q_bstride", C.c_int64), ("k_bstride", C.c_int64), ("vt_bstride", C.c_int64), ("o_bstride", C.c_int64),
        ("scale", C.c_float), ("o_lo_off", C.c_int32), ("o_mx", C.c_int32), ("q_lo_off", C.c_int32), ("k_lo_off", C.c_int32), ("p_split", C.c_int32),
        ("reserved1", C.c_int32), ("vt_lo_off", C.c_int64),
    ]


class TimingEntry(C.Structure):
    _fields_ = [("kind", C.c_int32), ("ms", C.c_float), ("flops", C.c_double), ("bytes", C.c_double),
                ("m", C.c_int64), ("n", C.c_int64), ("k", C.c_int64), ("variant", C.c_int32), ("stage", C.c_int32)]


_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> (restype, argtypes): every symbol include/omgsr_hip.h declares
SIGNATURES = {
    "omgsr_abi_version": (C.c_int, []),
    "omgsr_check_device": (C.c_int, []),
    "omgsr_set_compute_dtype": (C.c_int, [C.c_int]),
    "omgsr_get_compute_dtype": (C.c_int, []),
    "omgsr_error_string": (C.c_char_p, [C.c_int]),
    "omgsr_igemm": (C.c_int, [C.POINTER(IgemmArgs), _P]),
    "omgsr_igemm_multi_plan": (C.c_int, [C.POINTER(IgemmArgs), _I]),
    "omgsr_igemm_multi": (C.c_int, [C.POINTER(IgemmArgs), _I, _P]),
    "omgsr_set_batch_invariant": (C.c_int, [C.c_int]),
    "omgsr_set_attention_defer_max": (C.c_int, [C.c_float]),
    "omgsr_igemm_workspace_bytes": (C.c_int64, [C.POINTER(IgemmArgs)]),
    "omgsr_igemm_gn_slots": (C.c_int32, [C.POINTER(IgemmArgs)]),
    "omgsr_igemm_gn_entries": (C.c_int32, [C.POINTER(IgemmArgs)]),
    "omgsr_igemm_gn_fusable": (C.c_int32, [C.POINTER(IgemmArgs)]),
    "omgsr_igemm_out_mx6_ok": (C.c_int32, [C.POINTER(IgemmArgs)]),
    "omgsr_groupnorm_scale_shift": (C.c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "omgsr_groupnorm_finalize": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, C.c_double, _F, _P]),
    "omgsr_groupnorm_partial": (C.c_int, [_P, _P, _I, _L, _I, _I, _I, _P]),
    "omgsr_groupnorm_finalize_merged": (C.c_int, [C.POINTER(GnMergeArgs), _P, _P, _P, _I, _I, _F, _P]),
    "omgsr_groupnorm_apply_shared": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "omgsr_image_to_model_input": (C.c_int, [_P, _P, _I, _I, _I, _I, _P]),
    "omgsr_colorfix_workspace_bytes": (C.c_int64, [_I, _I, _I, _I]),
    "omgsr_colorfix": (C.c_int, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "omgsr_groupnorm_nchunk": (C.c_int, [_L]),
    "omgsr_groupnorm_stats": (C.c_int, [_P, _P, _P, _P, _P, _I, _L, _I, _I, _F, _I, _P]),
    "omgsr_groupnorm_apply": (C.c_int, [_P, _P, _P, _P, _P, _P, _I, _L, _I, _I, _I, _I, _I, _P, _I, _P, _P]),
    "omgsr_layernorm": (C.c_int, [_P, _P, _P, _P, _L, _I, _F, _I, _I, _P]),
    "omgsr_resize_nearest_exact_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "omgsr_transpose_split": (C.c_int, [_P, _P, _I, _I, _I, _L, _P]),
    "omgsr_to_operand": (C.c_int, [_P, _P, _L, _I, _I, _P, _P]),
    "omgsr_attention": (C.c_int, [C.POINTER(AttnArgs), _P]),
    "omgsr_softmax_rows": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "omgsr_softmax_rows_split": (C.c_int, [_P, _P, _L, _I, _I, _P]),
    "omgsr_rmsnorm_rope": (C.c_int, [_P, _P, _P, _I, _P, _P, _I, _I, _I, _I, _L, _I, _I, _F, _P]),
    "omgsr_nchw_to_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "omgsr_nhwc_to_nchw": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _F, _I, _P]),
    "omgsr_copy_channels": (C.c_int, [_P, _P, _L, _I, _I, _I, _I, _I, _P]),
    "omgsr_vae_sample": (C.c_int, [_P, _P, _P, _L, _I, _I, _F, _F, _I, _P]),
    "omgsr_axpby": (C.c_int, [_P, _P, _P, _L, _F, _F, _F, _F, _I, _I, _P]),
    "omgsr_linear_f32": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "omgsr_tile_accumulate": (C.c_int, [_P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "omgsr_tile_normalise": (C.c_int, [_P, _P, _P, _I, _L, _I, _I, _I, _P]),
    "omgsr_crop_nhwc": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "omgsr_paste_nhwc": (C.c_int, [_P, _P] + [_I] * 13 + [_P]),
    "omgsr_flux_pack": (C.c_int, [_P, _P, _I, _I, _I, _I, _I, _I, _I, _P]),
    "omgsr_resample_u8": (C.c_int, [_P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _P]),
    "omgsr_timing_enable": (C.c_int, [C.c_int]),
    "omgsr_timing_reset": (C.c_int, []),
    "omgsr_timing_stage": (C.c_int, [C.c_int]),
    "omgsr_timing_collect": (C.c_int, [C.POINTER(TimingEntry), C.c_int]),
    "omgsr_mfma_peak": (C.c_int, [_I, C.POINTER(C.c_float), _P]),
}

_lib = None


def load(path: str | None = None):
    """Map the library (once) and bind every entry point. Raises OmgsrError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = path or os.environ.get("OMGSR_HIP_LIB") or lib_path()
    if not os.path.isfile(path):
        raise OmgsrError(
            f"{path} not found: build it with `python -m omgsr_amd.build` (hipcc --offload-arch=gfx950). "
            "The OMGSR MI355X path has no CPU fallback.")
    lib = C.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise OmgsrError(f"{path} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.omgsr_abi_version() != ABI_VERSION:
        raise OmgsrError(f"ABI mismatch: library {lib.omgsr_abi_version()} vs binding {ABI_VERSION}; rebuild")
    _lib = lib
    return lib


def check(code: int, what: str = "") -> None:
    if code != 0:
        msg = load().omgsr_error_string(code)
        raise OmgsrError(f"{what}: {msg.decode() if msg else code} (code {code})")
