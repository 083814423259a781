"""ctypes binding of libflow2gan_hip.so (the C ABI in include/flow2gan_hip.h).

The product path has no CPU fallback: importing this module without the built library, or calling
an op with tensors that are not on an MI355X, raises.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (F2G_LIB_PATH: lab builds of the same library, tools/micro/ -- measurement aid only)
LIB_PATH = os.environ.get("F2G_LIB_PATH") or os.path.join(_HERE, "libflow2gan_hip.so")

c_float_p = C.c_void_p  # raw device pointers travel as integers


class Operand(C.Structure):
    _fields_ = [
        ("base", C.c_void_p), ("rows", C.c_int32), ("cols", C.c_int32),
        ("P1", C.c_int32), ("P0", C.c_int32), ("seglen", C.c_int32),
        ("step1", C.c_int32), ("pad1", C.c_int32), ("L1", C.c_int32),
        ("step0", C.c_int32), ("pad0", C.c_int32), ("unit", C.c_int32), ("L0u", C.c_int32),
        ("seq_stride", C.c_int64), ("line_stride", C.c_int64),
        ("reflect", C.c_int32), ("split", C.c_int32),
        ("alpha", C.c_void_p), ("lrelu_src", C.c_void_p),
        ("lrelu_slope", C.c_float), ("unbounded", C.c_int32),
    ]


class Epilogue(C.Structure):
    _fields_ = [
        ("C", C.c_void_p), ("ldc", C.c_int64), ("P0o", C.c_int32), ("c_bf16", C.c_int32),
        ("seq_stride_o", C.c_int64), ("row_stride_o", C.c_int64), ("off_o", C.c_int64),
        ("bias", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int64), ("gamma", C.c_void_p),
        ("aux", C.c_void_p), ("ldaux", C.c_int64), ("alpha_n", C.c_void_p),
        ("colsum_alpha", C.c_void_p), ("colsum", C.c_void_p),
        ("lrelu_slope", C.c_float), ("scale", C.c_float),
        ("accumulate", C.c_int32), ("atomic", C.c_int32),
        ("prelu_slope", C.c_void_p), ("prelu_out", C.c_void_p), ("ld_prelu_out", C.c_int64),
        ("mask_src", C.c_void_p), ("fm_ref", C.c_void_p), ("fm_wdev", C.c_void_p),
        ("mask_slope", C.c_float), ("fm_w", C.c_float),
        ("x3_out", C.c_void_p), ("colsum_part_ld", C.c_int64),
    ]


class MultiEntry(C.Structure):
    _fields_ = [("out", C.c_void_p), ("inp", C.c_void_p), ("kind", C.c_int32), ("blocks", C.c_int32),
                ("n", C.c_int32 * 4), ("s", C.c_int64 * 4)]


MULTI_MAX = 48


class MultiDesc(C.Structure):
    _fields_ = [("n", C.c_int32), ("_pad", C.c_int32), ("e", MultiEntry * MULTI_MAX)]


class GemmDesc(C.Structure):
    _fields_ = [("A", Operand), ("B", Operand), ("E", Epilogue), ("form", C.c_int32),
                ("split_k", C.c_int32), ("precision", C.c_int32), ("_pad3", C.c_int32)]


class DwnormFwd(C.Structure):
    _fields_ = [
        ("x", C.c_void_p), ("ldx", C.c_int64), ("z", C.c_void_p), ("ldz", C.c_int64),
        ("B", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("K", C.c_int32),
        ("lens", C.c_void_p), ("w_dw", C.c_void_p), ("b_dw", C.c_void_p), ("beta", C.c_void_p),
        ("log_scale", C.c_void_p), ("cproj", C.c_void_p), ("ldcp", C.c_int64),
        ("Fc", C.c_int32), ("up", C.c_int32), ("te", C.c_void_p), ("ldte", C.c_int64),
        ("rstd", C.c_void_p), ("z_format", C.c_int32), ("_pad", C.c_int32),
    ]


class DwnormBwd(C.Structure):
    _fields_ = [
        ("f", DwnormFwd), ("gz", C.c_void_p), ("ldgz", C.c_int64), ("du", C.c_void_p),
        ("lddu", C.c_int64), ("g_cproj", C.c_void_p), ("g_te", C.c_void_p),
        ("g_beta", C.c_void_p), ("g_log_scale", C.c_void_p), ("partials", C.c_void_p),
        ("g_cproj_store", C.c_int32), ("_pad2", C.c_int32),
    ]


class DwconvBwd(C.Structure):
    _fields_ = [
        ("du", C.c_void_p), ("lddu", C.c_int64), ("x", C.c_void_p), ("ldx", C.c_int64),
        ("gx", C.c_void_p), ("ldgx", C.c_int64),
        ("B", C.c_int32), ("F", C.c_int32), ("C", C.c_int32), ("K", C.c_int32),
        ("lens", C.c_void_p), ("w_dw", C.c_void_p), ("gres", C.c_void_p), ("ldgres", C.c_int64),
        ("gamma", C.c_void_p), ("g_w", C.c_void_p), ("g_b", C.c_void_p), ("g_gamma", C.c_void_p),
        ("partials", C.c_void_p),
    ]


class Conv32Desc(C.Structure):  # == f2g_conv32_desc
    _fields_ = [("x", C.c_void_p), ("x_seq", C.c_int64), ("x_line", C.c_int64),
                ("S", C.c_int32), ("H", C.c_int32), ("Win", C.c_int32), ("Wout", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("lrelu_slope", C.c_float),
                ("precision", C.c_int32), ("y", C.c_void_p), ("y_seq", C.c_int64), ("y_line", C.c_int64),
                ("mask_src", C.c_void_p), ("fm_ref", C.c_void_p), ("fm_wdev", C.c_void_p),
                ("mask_slope", C.c_float), ("fm_w", C.c_float), ("colsum", C.c_void_p)]


class Conv2chDesc(C.Structure):  # == f2g_conv2ch_desc
    _fields_ = [("x", C.c_void_p), ("x_seq", C.c_int64), ("x_line", C.c_int64),
                ("S", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("_pad", C.c_int32),
                ("w", C.c_void_p), ("bias", C.c_void_p), ("lrelu_slope", C.c_float),
                ("_pad2", C.c_int32), ("y", C.c_void_p), ("gw", C.c_void_p), ("wt", C.c_void_p),
                ("gx", C.c_void_p), ("gx_seq", C.c_int64), ("gx_line", C.c_int64)]


class FftDesc(C.Structure):  # == f2g_fft_desc
    _fields_ = [("x", C.c_void_p), ("x_stride", C.c_int64), ("hop", C.c_int32), ("n_fft", C.c_int32),
                ("F", C.c_int32), ("rows", C.c_int32), ("window", C.c_void_p), ("twiddle", C.c_void_p),
                ("spec", C.c_void_p), ("ld_spec", C.c_int64), ("interleaved", C.c_int32),
                ("spec_cols", C.c_int32), ("frames", C.c_void_p), ("ld_frames", C.c_int64),
                ("reflect_T", C.c_int32), ("_pad", C.c_int32)]


class OlaMultiDesc(C.Structure):  # == f2g_ola_multi_desc
    _fields_ = [("frames", C.c_void_p * 4), ("ldf", C.c_int64 * 4), ("F", C.c_int32 * 4),
                ("n_fft", C.c_int32 * 4), ("hop", C.c_int32 * 4), ("window", C.c_void_p * 4),
                ("wbranch", C.c_void_p * 4), ("n", C.c_int32), ("_pad", C.c_int32)]


class Mpd0Desc(C.Structure):  # == f2g_mpd0_desc
    _fields_ = [("x", C.c_void_p), ("S", C.c_int32), ("H", C.c_int32), ("Hout", C.c_int32),
                ("halo", C.c_int32), ("w", C.c_void_p), ("bias", C.c_void_p), ("slope", C.c_float),
                ("_pad", C.c_int32), ("y", C.c_void_p)]


class MpdPostDesc(C.Structure):  # == f2g_mpdpost_desc
    _fields_ = [("y", C.c_void_p), ("S", C.c_int32), ("H", C.c_int32), ("halo", C.c_int32),
                ("_pad", C.c_int32), ("w", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p),
                ("g", C.c_void_p), ("mask_src", C.c_void_p), ("fm_ref", C.c_void_p), ("fm_wdev", C.c_void_p),
                ("mask_slope", C.c_float), ("fm_w", C.c_float), ("colsum", C.c_void_p), ("x3_out", C.c_void_p)]


class FusedMlpDesc(C.Structure):  # == f2g_fused_mlp_desc
    _fields_ = [("z", C.c_void_p), ("ldz", C.c_int64), ("wp", C.c_void_p), ("b1", C.c_void_p),
                ("alpha", C.c_void_p), ("b2", C.c_void_p), ("res", C.c_void_p), ("ldres", C.c_int64),
                ("gamma", C.c_void_p), ("out", C.c_void_p), ("ldo", C.c_int64),
                ("rows", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("parts", C.c_int32)]


class SadamGroup(C.Structure):  # == f2g_sadam_group
    _fields_ = [("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float),
                ("scalar_lr_scale", C.c_float), ("eps", C.c_float), ("param_min_rms", C.c_float),
                ("param_max_rms", C.c_float), ("scalar_max", C.c_float),
                ("clipping_scale", C.c_float), ("size_update_period", C.c_int32),
                ("clipping_update_period", C.c_int32), ("step", C.c_int32),
                ("first", C.c_int32), ("count", C.c_int32)]


# numpy layouts of the device tables (== f2g_sadam_tensor / f2g_sadam_chunk)
SADAM_TENSOR_DTYPE = [("p", "<u8"), ("g", "<u8"), ("v", "<u8"), ("m", "<u8"), ("numel", "<i8"),
                      ("group", "<i4"), ("is_scalar", "<i4")]
SADAM_CHUNK_DTYPE = [("tensor", "<i4"), ("count", "<i4"), ("offset", "<i8")]
SADAM_NCOEF, SADAM_TSTATE, SADAM_GSTATE = 12, 10, 1028

_P, _I, _L, _F = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (stream appended automatically)
_SIGS = {
    "f2g_gemm": [C.POINTER(GemmDesc)],
    "f2g_split_bf16": [_P, _P, _L],
    "f2g_to_bf16": [_P, _P, _L],
    "f2g_dwnorm_fwd": [C.POINTER(DwnormFwd)],
    "f2g_dwnorm_bwd": [C.POINTER(DwnormBwd)],
    "f2g_dwconv_bwd": [C.POINTER(DwconvBwd)],
    "f2g_biasnorm_fwd": [_P, _L, _P, _L, _I, _I, _P, _P],
    "f2g_biasnorm_bwd": [_P, _L, _P, _L, _P, _L, _I, _I, _P, _P, _P, _P],
    "f2g_istft_ola": [_P, _L, _P, _I, _I, _I, _I, _I, _P, _P, _F, _I],
    "f2g_istft_ola_bwd": [_P, _P, _L, _I, _I, _I, _I, _I, _P, _P, _F],
    "f2g_frames_fold": [_P, _L, _P, _I, _I, _I, _I, _I, _I],
    "f2g_axpby_rows": [_P, _P, _P, _P, _P, _F, _F, _I, _I],
    "f2g_clamp": [_P, _P, _F, _F, _L],
    "f2g_silu": [_P, _P, _L],
    "f2g_silu_bwd": [_P, _P, _P, _L],
    "f2g_time_embedding": [_P, _P, _I, _I, _F],
    "f2g_mask_rows": [_P, _L, _I, _I, _I, _P],
    "f2g_colsum": [_P, _P, _L, _P, _L, _I, _I],
    "f2g_rows_fold_up": [_P, _L, _P, _L, _I, _I, _I, _I, _I],
    "f2g_bct_to_rows": [_P, _L, _P, _I, _I, _I],
    "f2g_rows_to_bct": [_P, _P, _L, _I, _I, _I],
    "f2g_permute4": [_P, _P, _I, _I, _I, _I, _L, _L, _L, _L],
    "f2g_limit_grad": [_P, _P, _F, _F, _L],
    "f2g_copy3": [_P, _L, _L, _P, _L, _L, _I, _I, _I, _I],
    "f2g_spec_power": [_P, _L, _P, _L, _I, _I, _I],
    "f2g_spec_power_bwd": [_P, _L, _P, _L, _P, _I, _I, _I],
    "f2g_fm_spec_loss": [_P, _P, _P, _P, _I, _I, _I, _P, _F, _F, _F, _F, _F],
    "f2g_masked_mse": [_P, _P, _P, _P, _I, _I, _P, _F],
    "f2g_l1_loss": [_P, _P, _P, _P, _I, _I, _L, _F, _F, _P],
    "f2g_hinge_loss": [_P, _P, _P, _L, _F, _F, _P],
    "f2g_peaknorm_fwd": [_P, _P, _P, _I, _I],
    "f2g_peaknorm_bwd": [_P, _P, _P, _P, _I, _I],
    "f2g_lrelu_bwd": [_P, _P, _P, _F, _P, _F, _I, _I, _L],
    "f2g_reflect_pad": [_P, _P, _I, _I, _I, _I],
    "f2g_fft_frames": [C.POINTER(FftDesc), _I],
    "f2g_mpd0_fwd": [C.POINTER(Mpd0Desc)],
    "f2g_mpd0_wgrad": [C.POINTER(Mpd0Desc), _P],
    "f2g_mpd0_dgrad": [C.POINTER(Mpd0Desc), _P],
    "f2g_mpdpost_fwd": [C.POINTER(MpdPostDesc)],
    "f2g_mpdpost_dgrad": [C.POINTER(MpdPostDesc)],
    "f2g_mpdpost_wgrad": [C.POINTER(MpdPostDesc), _P],
    "f2g_period_fold": [_P, _P, _I, _I, _I, _I],
    "f2g_period_fold_bwd": [_P, _P, _I, _I, _I, _I, _I],
    "f2g_fill": [_P, _F, _L],
    "f2g_bucket_arm": [_P, _L, C.c_int32],
    "f2g_scale": [_P, _F, _L],
    "f2g_log_clip": [_P, _L, _F],
    "f2g_conv32_s2_fwd": [C.POINTER(Conv32Desc)],
    "f2g_conv32_s2_dgrad": [C.POINTER(Conv32Desc)],
    "f2g_conv33_fwd": [C.POINTER(Conv32Desc)],
    "f2g_conv33_wgrad": [C.POINTER(Conv32Desc), _P],
    "f2g_conv32_s2_wgrad": [C.POINTER(Conv32Desc), _P],
    "f2g_conv2ch_fwd": [C.POINTER(Conv2chDesc)],
    "f2g_conv2ch_wgrad": [C.POINTER(Conv2chDesc)],
    "f2g_conv2ch_dgrad": [C.POINTER(Conv2chDesc)],
    "f2g_convpost_fwd": [C.POINTER(Conv2chDesc)],
    "f2g_convpost_wgrad": [C.POINTER(Conv2chDesc)],
    "f2g_convpost_dgrad": [C.POINTER(Conv2chDesc)],
    "f2g_wave_stats": [_P, _L, _L, _I, _I, _P, _P],
    "f2g_wave_gain": [_P, _L, _P, _L, _L, _I, _I, _I, _P, _P, _P],
    "f2g_sadam_stats": [_P, _P, _I, _P, _I],
    "f2g_sadam_prepare": [_P, C.POINTER(SadamGroup), _P, _P, _P, _P],
    "f2g_lrelu_bwd_colsum": [_P, _P, _P, _F, _P, _F, _I, _I, _L, _P],
    "f2g_zero_halo": [_P, _I, _I, _I, _I, _I],
    "f2g_sadam_update": [_P, _P, _I, _P],
    "f2g_mlp_pack": [_P, _P, _L, _P, _L, _I, _I],
    "f2g_fused_mlp": [C.POINTER(FusedMlpDesc)],
    "f2g_fused_block": [C.POINTER(DwnormFwd), C.POINTER(FusedMlpDesc)],
    "f2g_fused_block_multi": [C.POINTER(DwnormFwd), C.POINTER(FusedMlpDesc), C.c_int32],
    "f2g_istft_ola_multi": [C.POINTER(OlaMultiDesc), _P, _I, _I, _F, _I],
    "f2g_split_bf16x3": [_P, _P, _L, _I, _I],
    "f2g_multi": [C.POINTER(MultiDesc)],
}
EXPORTS = sorted(list(_SIGS) + ["f2g_version", "f2g_last_error", "f2g_gemm_last_path",
                                 "f2g_gemm_lean_ok", "f2g_gemm_wgrad_lean", "f2g_fused_mlp_ok",
                                 "f2g_dwnorm_bwd_workspace", "f2g_split_bf16x3_bytes", "f2g_gemm_x6_ok",
                                 "f2g_gemm_colsum_part_rows", "f2g_set_option", "f2g_get_option",
                                 "f2g_dwconv_bwd_workspace", "f2g_sadam_chunk_elems"])


class F2GError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C flow2gan_amd/csrc`). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, args in _SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = list(args) + [C.c_void_p]
        fn.restype = C.c_int
    for name in ("f2g_dwnorm_bwd_workspace", "f2g_dwconv_bwd_workspace"):
        fn = getattr(lib, name)
        fn.argtypes = [C.c_int32] * 4
        fn.restype = C.c_int64
    lib.f2g_gemm_lean_ok.argtypes = [C.POINTER(GemmDesc)]
    lib.f2g_gemm_lean_ok.restype = C.c_int
    lib.f2g_gemm_wgrad_lean.argtypes = [C.POINTER(GemmDesc)]
    lib.f2g_gemm_wgrad_lean.restype = C.c_int
    lib.f2g_split_bf16x3_bytes.argtypes = [C.c_int32, C.c_int32]
    lib.f2g_split_bf16x3_bytes.restype = C.c_int64
    lib.f2g_gemm_x6_ok.argtypes = [C.POINTER(GemmDesc)]
    lib.f2g_gemm_x6_ok.restype = C.c_int
    lib.f2g_gemm_colsum_part_rows.argtypes = [C.POINTER(GemmDesc)]
    lib.f2g_gemm_colsum_part_rows.restype = C.c_int32
    lib.f2g_fused_mlp_ok.argtypes = [C.c_int32, C.c_int32]
    lib.f2g_fused_mlp_ok.restype = C.c_int
    lib.f2g_sadam_chunk_elems.argtypes = []
    lib.f2g_sadam_chunk_elems.restype = C.c_int32
    lib.f2g_set_option.argtypes = [C.c_char_p, C.c_int32]
    lib.f2g_set_option.restype = C.c_int
    lib.f2g_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int32)]
    lib.f2g_get_option.restype = C.c_int
    lib.f2g_version.restype = C.c_char_p
    lib.f2g_last_error.restype = C.c_char_p
    return lib


lib = _load()


def version() -> str:
    return lib.f2g_version().decode()


def set_option(name: str, value: int) -> int:
    """Set a dispatch option of the library (include/flow2gan_hip.h: f2g_set_option); returns the previous
    value so that a test can restore it."""
    old = C.c_int32()
    if lib.f2g_get_option(name.encode(), C.byref(old)) or lib.f2g_set_option(name.encode(), int(value)):
        raise F2GError(f"unknown library option {name!r}")
    return old.value


def get_option(name: str) -> int:
    v = C.c_int32()
    if lib.f2g_get_option(name.encode(), C.byref(v)):
        raise F2GError(f"unknown library option {name!r}")
    return v.value


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def ptr(t) -> int | None:
    """Device pointer of a tensor (None stays NULL).  Refuses host tensors: no CPU path."""
    if t is None:
        return None
    if not t.is_cuda:
        raise F2GError("flow2gan_amd ops need tensors on an MI355X (got a CPU tensor)")
    if t.dtype not in (torch.float32, torch.int32, torch.bfloat16):   # bf16: GEMM operand images
        raise F2GError(f"unsupported dtype {t.dtype}")
    return t.data_ptr()


# F2G_DRYRUN=1 (measurement aid only): build every descriptor but launch nothing, to time the
# pure host cost of issuing a step.  Results are garbage by construction.
_DRYRUN = os.environ.get("F2G_DRYRUN", "0") == "1"


# Set by ops.WeightBatch while re-layout operations are being collected for one f2g_multi launch: any OTHER
# kernel call first flushes what is pending (program order between the batch and everything else is kept).
PRE_CALL = None


def call(name: str, *args):
    if PRE_CALL is not None and name != "f2g_multi":
        PRE_CALL()
    if _DRYRUN:
        stream_ptr()
        return
    rc = getattr(lib, name)(*args, stream_ptr())
    if rc != 0:
        raise F2GError(f"{name} failed with code {rc}: {lib.f2g_last_error().decode()}")
