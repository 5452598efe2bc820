"""ctypes binding of libemap_hip.so (include/emap_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` (``emap_amd/csrc/build.sh``).
There is no CPU fallback: if the library is missing, or a call is made without a GPU, the error is
raised to the caller.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMAP_HIP_LIB") or os.path.join(_HERE, "lib", "libemap_hip.so")   # env override: A/B builds

PREC_BF16 = 0
PREC_BF16X3 = 1
PREC_F16 = 2
PREC_F16X3 = 3
PREC_F16X3M = 4    # f16x3 with MX-fp6 cross terms in the forward sweep of the value+gradient pass too (include/emap_hip.h)
PREC_F16X3E = 5    # f16x3 with f16 cross terms in both sweeps of the value+gradient pass (no MX fp6): the wider-margin mode
PRECISIONS = {"bf16": PREC_BF16, "bf16x3": PREC_BF16X3, "f16": PREC_F16, "f16x3": PREC_F16X3, "f16x3m": PREC_F16X3M, "f16x3e": PREC_F16X3E}
UDF_TYPES = {"abs": 0, "square": 1, "sdf": 2}
MAX_LIN = 12
ABI_VERSION = 10

F_NAN_SAMPLES = 1
F_NAN_GRADERR = 2
F_MLP_NONFINITE = 4


class NetConfig(C.Structure):
    _fields_ = [("d_hidden", C.c_int32), ("n_lin", C.c_int32), ("skip_l", C.c_int32), ("multires", C.c_int32),
                ("d_out", C.c_int32), ("udf_type", C.c_int32), ("scale", C.c_float)]


class CompositeOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in (
        "weights", "alpha", "mid_z", "dists", "inside_sphere", "gradient_mag", "gradients_flip", "edge", "depth",
        "weight_sum", "normals", "scalars")]


class RenderParams(C.Structure):
    _fields_ = [("n_rays", C.c_int32), ("n_samples", C.c_int32), ("n_importance", C.c_int32),
                ("up_sample_steps", C.c_int32), ("inv_s", C.c_float), ("beta", C.c_float), ("gamma", C.c_float),
                ("cos_anneal_ratio", C.c_float), ("has_cos_anneal", C.c_int32), ("flip_saturation", C.c_float),
                ("near_surface", C.c_float), ("sparse_scale", C.c_float), ("background", C.c_float),
                ("has_background", C.c_int32), ("variance_dev", C.c_void_p), ("beta_dev", C.c_void_p),
                ("gamma_dev", C.c_void_p), ("beta_min", C.c_float), ("reserved", C.c_int32)]


class CompositeGrads(C.Structure):
    _fields_ = [("d_edge", C.c_void_p), ("d_depth", C.c_void_p), ("d_gradient_error", C.c_void_p),
                ("d_gradient_error_near_surface", C.c_void_p), ("scalars", C.c_void_p), ("d_variance", C.c_void_p),
                ("d_beta", C.c_void_p), ("d_gamma", C.c_void_p), ("grad_scale", C.c_float), ("accumulate", C.c_int32),
                ("zero_tail", C.c_void_p), ("n_zero_tail", C.c_int64)]


class ParamGrads(C.Structure):
    _fields_ = [("g_host", C.POINTER(C.c_void_p)), ("v_host", C.POINTER(C.c_void_p)), ("dg_host", C.POINTER(C.c_void_p)),
                ("dv_host", C.POINTER(C.c_void_p)), ("db_host", C.POINTER(C.c_void_p)), ("weight_norm", C.c_int32),
                ("accumulate", C.c_int32), ("grad_scale", C.c_float), ("reserved", C.c_int32)]


class RayDataset(C.Structure):
    _fields_ = [("edges", C.c_void_p), ("pixel_order", C.c_void_p), ("n_edge", C.c_void_p), ("density", C.c_void_p),
                ("kinv", C.c_void_p), ("pose", C.c_void_p), ("image_perm", C.c_void_p), ("n_images", C.c_int32), ("H", C.c_int32),
                ("W", C.c_int32), ("reserved", C.c_int32)]


class RayBatch(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("rays_o", "rays_v", "edge", "depth_scale", "ndc_uv", "p_cam", "pixels", "img_idx", "t_rand")]


# every symbol include/emap_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "emap_abi_version": (C.c_int, []),
    "emap_last_error": (C.c_char_p, []),
    "emap_set_grad_mode": (C.c_int, [C.c_int]),
    "emap_set_fused_sampling": (C.c_int, [C.c_int]),
    "emap_set_fused_composite": (C.c_int, [C.c_int]),
    "emap_set_value_tile_mode": (C.c_int, [C.c_int]),
    "emap_packed_bytes": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.POINTER(C.c_size_t)]),
    "emap_pack_weights": (C.c_int, [C.POINTER(NetConfig), C.POINTER(_P), C.POINTER(_P), C.POINTER(_P), _P, C.c_int, _P]),
    "emap_udf_fwd": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, _P, C.c_int64, _P, _P]),
    "emap_udf_scratch_bytes": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.c_int64, C.POINTER(C.c_size_t)]),
    "emap_udf_fwd_grad": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, _P, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "emap_embed": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "emap_null_direction": (C.c_int, [_P, C.c_int64, C.c_int, _P, _P]),
    "emap_sample_pdf": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "emap_sample_pdf_u": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "emap_upsample_step": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, C.c_float, C.c_float, C.c_float,
                                     _P, _P, _P, _P]),
    "emap_merge_sorted": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "emap_composite_fwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                                     C.POINTER(CompositeOut), _P, _P, _P]),
    "emap_composite_fwd_p": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.POINTER(RenderParams),
                                       C.POINTER(CompositeOut), _P, _P, _P]),
    "emap_render_workspace_bytes": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.POINTER(RenderParams), C.POINTER(C.c_size_t)]),
    "emap_render_fwd": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, C.POINTER(RenderParams), _P, _P, _P, _P, _P, _P,
                                  _P, _P, _P, C.POINTER(CompositeOut), _P, C.c_size_t, _P, _P]),
    "emap_composite_bwd": (C.c_int, [_P, _P, _P, _P, _P, _P, C.c_int, C.c_int, _P, C.POINTER(RenderParams),
                                     C.POINTER(CompositeGrads), _P, _P, _P, _P]),
    "emap_udf_vjp_workspace_bytes": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.c_int64, C.POINTER(C.c_size_t)]),
    "emap_udf_vjp": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, _P, C.c_int64, _P, _P, C.POINTER(ParamGrads), _P, C.c_size_t,
                               _P, _P]),
    "emap_render_bwd_workspace_bytes": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.POINTER(RenderParams), C.POINTER(C.c_size_t)]),
    "emap_render_bwd": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, C.POINTER(RenderParams), _P, _P, _P, _P, _P, _P, _P,
                                  C.POINTER(CompositeGrads), C.POINTER(ParamGrads), _P, C.c_size_t, _P, _P]),
    "emap_render_bwd_absmax_offset": (C.c_int, [C.POINTER(NetConfig), C.c_int, C.POINTER(RenderParams), C.POINTER(C.c_size_t)]),
    "emap_render_bwd_staged": (C.c_int, [C.POINTER(NetConfig), _P, C.c_int, C.POINTER(RenderParams), _P, _P, _P, _P, _P, _P, _P,
                                         C.POINTER(CompositeGrads), C.POINTER(ParamGrads), _P, C.c_size_t, _P, _P, C.c_int]),
    "emap_sample_rays": (C.c_int, [C.POINTER(RayDataset), C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_uint64, _P, _P, C.POINTER(RayBatch), _P]),
    "emap_train_stats": (C.c_int, [_P, _P, _P, C.c_int, C.c_float, _P, _P, _P]),
    "emap_train_loss": (C.c_int, [_P, C.c_float, C.c_float, C.c_float, _P, _P]),
    "emap_adam_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_double, C.c_double, C.c_float, _P]),
    "emap_adam_step_masked": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_int64, C.c_float, C.c_float, C.c_double, C.c_double, C.c_float, _P, _P, _P]),
    "emap_ar_local_bytes": (C.c_int, [C.c_int64, C.POINTER(C.c_size_t)]),
    "emap_ar_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p), _P]),
    "emap_ar_open": (C.c_int, [_P, C.POINTER(C.c_void_p)]),
    "emap_ar_close": (C.c_int, [_P]),
    "emap_ar_free": (C.c_int, [_P]),
    "emap_ar_allreduce_sum": (C.c_int, [_P, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.c_size_t, _P]),
    "emap_ar_error": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "emap_ar_set_timeout_ms": (C.c_int, [C.c_int64]),
    "emap_profile_enable": (C.c_int, [C.c_int]),
    "emap_profile_read": (C.c_int, [C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "emap_profile_read_kernel": (C.c_int, [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "emap_profile_read_clock": (C.c_int, [C.c_int, C.POINTER(C.c_float)]),
    "emap_linspace_host": (None, [C.c_float, C.c_float, C.c_int, C.POINTER(C.c_float)]),
}

_lib = None


class EmapLibraryError(RuntimeError):
    pass


def lib():
    """Load (once) and return the bound library.  Raises EmapLibraryError if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmapLibraryError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or emap_amd/csrc/build.sh). "
            "emap_amd has no CPU fallback.")
    try:
        l = C.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise EmapLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(l, name)
        except AttributeError as e:
            raise EmapLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if l.emap_abi_version() != ABI_VERSION:
        raise EmapLibraryError("libemap_hip.so ABI version mismatch")
    _lib = l
    return l


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = lib().emap_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libemap_hip {what} failed (rc={rc}): {msg}")


def require_cuda(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(
            f"emap_amd: `{name}` is on {t.device}; the HIP path needs tensors on an MI355X (cuda) device. "
            "There is no CPU fallback.")


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def stream_ptr(device=None):
    """Current stream of `device` (default: the current device).  Callers that take tensors from an explicit device wrap the
    C call in ``torch.cuda.device(t.device)`` and pass ``t.device`` here, so kernels land on the tensor's device and stream."""
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def on_device(t: torch.Tensor):
    """Context manager: make t's device current for the C call (kernel launches and the per-function LDS attribute are per device)."""
    return torch.cuda.device(t.device)


def f32c(t: torch.Tensor) -> torch.Tensor:
    """fp32 contiguous view/copy (the kernels' only layout)."""
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()
