"""ctypes binding of the C ABI in include/clair_hip.h.

There is no CPU fallback: if the HIP library is missing or a call fails, this module raises.  The
library is built in-tree by ``clair_torch_amd.build`` (hipcc, gfx950) and loaded from
``clair_torch_amd/lib/libclair_hip.so``.
"""
import ctypes
import os

from . import build as _build

# include/clair_hip.h constants
DTYPE_U8, DTYPE_U16, DTYPE_F32 = 0, 1, 2
INTERP_LOOKUP, INTERP_LINEAR, INTERP_CATMULL, INTERP_NONE = 0, 1, 2, 3
STD_NONE, STD_CONSTANT, STD_MULTIPLIER, STD_EXPLICIT = 0, 1, 2, 3
WEIGHT_NONE, WEIGHT_GAUSS = 0, 1
LAYOUT_NCHW, LAYOUT_NHWC, LAYOUT_NHWC_BGR = 0, 1, 2
MERGE_FIRST_BATCH, MERGE_FINALIZE, MERGE_MEAN_OUT_F32, MERGE_F64_MOMENTS = 1, 2, 4, 8
MERGE_REFERENCE_ORDER, MERGE_CLOSED_FORM, MERGE_STD_HINT, MERGE_REQUIRE_ONE_LAUNCH, MERGE_OUT_AS_INPUT = 16, 32, 64, 128, 256
ERR_NO_GRADIENT_PATH = -4

ABI_VERSION = 3
EXPORTS = ("ct_abi_version", "ct_error_string", "ct_hdr_merge_batch", "ct_hdr_merge_batches", "ct_linearize_std", "ct_linearize_fwd",
           "ct_linearize_bwd", "ct_pair_residual_fwd", "ct_pair_residual_bwd", "ct_pair_residual_bwd_workspace", "ct_flatfield_sums",
           "ct_flatfield_apply", "ct_video_stats_batch", "ct_dark_field_blur", "ct_hdr_merge_kernel_name",
           "ct_merge_set_retry_counter", "ct_norm_constants", "ct_index_constants", "ct_pivot_index_constants",
           "ct_pivot_floor_constants", "ct_pivot_interval_constants", "ct_band_stats", "ct_band_stats_workspace")


class Geometry(ctypes.Structure):
    _fields_ = [("channels", ctypes.c_int32), ("h_tile", ctypes.c_int64), ("width", ctypes.c_int64),
                ("h_global", ctypes.c_int64), ("row_offset", ctypes.c_int64), ("image_stride", ctypes.c_int64),
                ("layout", ctypes.c_int32)]


class Icrf(ctypes.Structure):
    _fields_ = [("lut_dev", ctypes.c_void_p), ("n_points", ctypes.c_int32), ("interp", ctypes.c_int32)]


class PairParams(ctypes.Structure):
    _fields_ = [("lower", ctypes.c_float), ("upper", ctypes.c_float), ("weight_scale", ctypes.c_float),
                ("use_relative", ctypes.c_int32), ("use_uncertainty_weighting", ctypes.c_int32),
                ("std_mode", ctypes.c_int32), ("std_value", ctypes.c_float), ("pair_band", ctypes.c_int32)]


class NativeLibraryError(RuntimeError):
    pass


_lib = None


def library_path():
    return _build.LIB_PATH


def load():
    """Load (once) the HIP shared library; raises NativeLibraryError if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise NativeLibraryError(
            f"{path} is missing: build it with `python -m clair_torch_amd.build` (hipcc, gfx950). "
            "clair_torch_amd has no CPU fallback for its kernels.")
    lib = ctypes.CDLL(path)
    vp, i32, i64, f32, u32 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_float, ctypes.c_uint32
    gp, ip = ctypes.POINTER(Geometry), ctypes.POINTER(Icrf)
    lib.ct_abi_version.restype = i32
    lib.ct_error_string.restype = ctypes.c_char_p
    lib.ct_error_string.argtypes = [i32]
    lib.ct_hdr_merge_batch.restype = i32
    lib.ct_hdr_merge_batch.argtypes = [vp, i32, f32, i32, gp, vp, i32, f32, vp, ip, i32, vp, vp, vp, vp, vp, u32, vp]
    lib.ct_hdr_merge_batches.restype = i32
    lib.ct_hdr_merge_batches.argtypes = [vp, vp, vp, i32, i32, f32, gp, i32, f32, vp, ip, i32, vp, vp, vp, vp, vp, u32, vp]
    lib.ct_pivot_interval_constants.restype = i32
    lib.ct_pivot_interval_constants.argtypes = [f32, i32, i32, i32, ctypes.POINTER(f32)]
    lib.ct_linearize_std.restype = i32
    lib.ct_linearize_std.argtypes = [vp, i32, f32, i64, gp, vp, i32, f32, ip, vp, vp, vp]
    lib.ct_linearize_fwd.restype = i32
    lib.ct_linearize_fwd.argtypes = [vp, i64, gp, ip, vp, vp]
    lib.ct_linearize_bwd.restype = i32
    lib.ct_linearize_bwd.argtypes = [vp, vp, i64, gp, ip, vp, vp, vp]
    pp = ctypes.POINTER(PairParams)
    lib.ct_pair_residual_fwd.restype = i32
    lib.ct_pair_residual_fwd.argtypes = [vp, i32, f32, i32, gp, vp, ip, vp, vp, vp, i32, pp, i32, vp, vp, vp]
    lib.ct_pair_residual_bwd.restype = i32
    lib.ct_pair_residual_bwd.argtypes = [vp, i32, f32, i32, gp, vp, ip, vp, i32, vp, vp, vp, pp, vp, vp, vp, vp, i64, vp]
    lib.ct_pair_residual_bwd_workspace.restype = i64
    lib.ct_pair_residual_bwd_workspace.argtypes = [i32, i32, i32]
    lib.ct_hdr_merge_kernel_name.restype = ctypes.c_char_p
    lib.ct_hdr_merge_kernel_name.argtypes = [i32, f32, i32, i32, u32]
    lib.ct_merge_set_retry_counter.restype = None
    lib.ct_merge_set_retry_counter.argtypes = [vp]
    lib.ct_flatfield_sums.restype = i32
    lib.ct_flatfield_sums.argtypes = [vp, i32, vp, i32, i64, vp, vp]
    lib.ct_flatfield_apply.restype = i32
    lib.ct_flatfield_apply.argtypes = [vp, i32, i64, vp, i32, vp, vp, vp, vp, i32, i64, vp]
    lib.ct_dark_field_blur.restype = i32
    lib.ct_dark_field_blur.argtypes = [vp, i32, f32, i32, gp, vp, vp, i32, f32, vp, vp, i32, f32, f32, vp, vp, vp]
    lib.ct_band_stats_workspace.restype = i64
    lib.ct_band_stats_workspace.argtypes = [i32]
    lib.ct_band_stats.restype = i32
    lib.ct_band_stats.argtypes = [vp, vp, i32, i64, vp, i64, vp, vp]
    lib.ct_video_stats_batch.restype = i32
    lib.ct_video_stats_batch.argtypes = [vp, i32, f32, i32, gp, ip, f32, vp, vp, vp]
    if lib.ct_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"{path}: ABI version {lib.ct_abi_version()} != {ABI_VERSION}; rebuild the library")
    _lib = lib
    return lib


def check(rc, what):
    """Translate a C status code into the exception type the reference raises at that point."""
    if rc == 0:
        return
    msg = load().ct_error_string(rc).decode()
    if rc == ERR_NO_GRADIENT_PATH:
        # torch.autograd.grad's error text in the reference (hdr_merge.py:108, linearization.py:100)
        raise RuntimeError("element 0 of tensors does not require grad and does not have a grad_fn")
    if rc in (-1, -5):
        raise ValueError(f"{what}: {msg}")
    raise NativeLibraryError(f"{what}: {msg} (code {rc})")
