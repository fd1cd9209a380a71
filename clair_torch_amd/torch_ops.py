"""The C-ABI entry points as ``torch.library`` custom ops (namespace ``clair_hip``), SURVEY 8(b).

``clair_torch_amd.ops`` (ctypes) stays the implementation; this module registers the same kernels with PyTorch's
dispatcher so that they are visible as ``torch.ops.clair_hip.*``, carry schemas and fake (meta) implementations -- i.e.
they trace under FakeTensorMode / torch.compile and export -- and, for the ICRF sampler, an autograd formula that calls
the backward kernel.  Tensor-only signatures (optional tensors, ints, floats, strings): the streaming state, geometry
and pair lists are passed as their component tensors / scalars.

    torch.ops.clair_hip.icrf_forward(image, lut, interp, h_global, row_offset) -> Tensor      (differentiable)
    torch.ops.clair_hip.icrf_backward(image, grad_out, lut, interp, need_image, need_lut, h_global, row_offset)
    torch.ops.clair_hip.hdr_merge(stack, exposures, lut?, interp, gaussian, std?, std_mode, std_value, max_code,
                                  h_global, row_offset, layout) -> (mean float64, std float32)
    torch.ops.clair_hip.linearize_std(frames, lut, interp, std?, std_mode, std_value, max_code, layout) -> (lin, std)
    torch.ops.clair_hip.pair_residual_sums(stack, i_idx, j_idx, ratio, lut?, interp, lower, upper, relative,
                                           unc_weight, std_mode, std_value, max_code, level) -> (P, C, 5) float64
    torch.ops.clair_hip.pair_residual_lut_grad(stack, i_idx, j_idx, ratio, coef, lut, interp, lower, upper, relative,
                                               max_code) -> (C, L) float64

CPU tensors are refused by the kernels' front-end exactly as through ``ops`` (there is no CPU path).
"""
from typing import Optional, Tuple

import torch

from . import ops

_LIB = "clair_hip"


def _tile(h_global: int, row_offset: int):
    return None if h_global <= 0 else ops.TileGeometry(h_global=h_global, row_offset=row_offset)


def _chw(t: torch.Tensor, layout: str):
    return (t.shape[1], t.shape[2], t.shape[3]) if layout == "nchw" else (t.shape[3], t.shape[1], t.shape[2])


@torch.library.custom_op(f"{_LIB}::icrf_forward", mutates_args=())
def icrf_forward(image: torch.Tensor, lut: torch.Tensor, interp: str, h_global: int = 0, row_offset: int = 0) -> torch.Tensor:
    return ops.icrf_forward(image, lut, interp, _tile(h_global, row_offset))


@icrf_forward.register_fake
def _(image, lut, interp, h_global=0, row_offset=0):
    return torch.empty_like(image, dtype=torch.float32)


@torch.library.custom_op(f"{_LIB}::icrf_backward", mutates_args=())
def icrf_backward(image: torch.Tensor, grad_out: torch.Tensor, lut: torch.Tensor, interp: str, need_image: bool,
                  need_lut: bool, h_global: int = 0, row_offset: int = 0) -> Tuple[torch.Tensor, torch.Tensor]:
    gx, gl = ops.icrf_backward(image, grad_out, lut, interp, need_image, need_lut, _tile(h_global, row_offset))
    return (gx if gx is not None else image.new_zeros(0), gl if gl is not None else lut.new_zeros(0))


@icrf_backward.register_fake
def _(image, grad_out, lut, interp, need_image, need_lut, h_global=0, row_offset=0):
    return (torch.empty_like(image) if need_image else image.new_empty(0),
            torch.empty_like(lut, dtype=torch.float32) if need_lut else lut.new_empty(0))


def _icrf_setup(ctx, inputs, output):
    image, lut, interp, h_global, row_offset = inputs
    ctx.save_for_backward(image, lut)
    ctx.meta = (interp, h_global, row_offset)


def _icrf_backward(ctx, grad_out):
    image, lut = ctx.saved_tensors
    interp, h_global, row_offset = ctx.meta
    need_image = ctx.needs_input_grad[0] and interp != "lookup"
    need_lut = ctx.needs_input_grad[1]
    gx, gl = icrf_backward(image, grad_out.contiguous(), lut, interp, need_image, need_lut, h_global, row_offset)
    return (gx if need_image else None), (gl if need_lut else None), None, None, None


icrf_forward.register_autograd(_icrf_backward, setup_context=_icrf_setup)


@torch.library.custom_op(f"{_LIB}::hdr_merge", mutates_args=())
def hdr_merge(stack: torch.Tensor, exposures: torch.Tensor, lut: Optional[torch.Tensor], interp: str, gaussian: bool,
              std: Optional[torch.Tensor], std_mode: str, std_value: float, max_code: float, h_global: int = 0,
              row_offset: int = 0, layout: str = "nchw") -> Tuple[torch.Tensor, torch.Tensor]:
    mean, sd = ops.hdr_merge_batch(stack, exposures, lut=lut, interp=interp if lut is not None else None,
                                   gaussian_weight=gaussian, std=std, std_mode=std_mode, std_value=std_value,
                                   max_code=max_code if max_code > 0 else None, tile=_tile(h_global, row_offset),
                                   layout=layout)
    return mean, (sd if sd is not None else mean.new_zeros(0, dtype=torch.float32))


@hdr_merge.register_fake
def _(stack, exposures, lut, interp, gaussian, std, std_mode, std_value, max_code, h_global=0, row_offset=0, layout="nchw"):
    chw = _chw(stack, layout)
    has_std = std is not None or std_mode != "none"
    return (stack.new_empty(chw, dtype=torch.float64),
            stack.new_empty(chw if has_std else (0,), dtype=torch.float32))


@torch.library.custom_op(f"{_LIB}::linearize_std", mutates_args=())
def linearize_std(frames: torch.Tensor, lut: torch.Tensor, interp: str, std: Optional[torch.Tensor], std_mode: str,
                  std_value: float, max_code: float, layout: str = "nchw") -> Tuple[torch.Tensor, torch.Tensor]:
    return ops.linearize_frames(frames, lut, interp, std=std, std_mode=std_mode, std_value=std_value,
                                max_code=max_code if max_code > 0 else None, want_std=True, layout=layout)


@linearize_std.register_fake
def _(frames, lut, interp, std, std_mode, std_value, max_code, layout="nchw"):
    shape = (frames.shape[0],) + tuple(_chw(frames, layout))
    return frames.new_empty(shape, dtype=torch.float32), frames.new_empty(shape, dtype=torch.float32)


@torch.library.custom_op(f"{_LIB}::pair_residual_sums", mutates_args=())
def pair_residual_sums(stack: torch.Tensor, i_idx: torch.Tensor, j_idx: torch.Tensor, ratio: torch.Tensor,
                       lut: Optional[torch.Tensor], interp: str, lower: float, upper: float, relative: bool,
                       unc_weight: bool, std_mode: str, std_value: float, max_code: float, level: int = 1) -> torch.Tensor:
    pairs = ops.PairList(i_idx, j_idx, ratio, stack.shape[0], stack.device)
    return ops.pair_residual_sums(stack, pairs, lut=lut, interp=interp if lut is not None else None, lower=lower,
                                  upper=upper, use_relative=relative, use_unc_weight=unc_weight, std_mode=std_mode,
                                  std_value=std_value, max_code=max_code if max_code > 0 else None, level=level)


@pair_residual_sums.register_fake
def _(stack, i_idx, j_idx, ratio, lut, interp, lower, upper, relative, unc_weight, std_mode, std_value, max_code, level=1):
    return stack.new_empty((i_idx.shape[0], stack.shape[1], 5), dtype=torch.float64)


@torch.library.custom_op(f"{_LIB}::pair_residual_lut_grad", mutates_args=())
def pair_residual_lut_grad(stack: torch.Tensor, i_idx: torch.Tensor, j_idx: torch.Tensor, ratio: torch.Tensor,
                           coef: torch.Tensor, lut: torch.Tensor, interp: str, lower: float, upper: float, relative: bool,
                           max_code: float) -> torch.Tensor:
    pairs = ops.PairList(i_idx, j_idx, ratio, stack.shape[0], stack.device)
    return ops.pair_residual_lut_grad(stack, pairs, coef, lut=lut, interp=interp, lower=lower, upper=upper,
                                      use_relative=relative, max_code=max_code if max_code > 0 else None)


@pair_residual_lut_grad.register_fake
def _(stack, i_idx, j_idx, ratio, coef, lut, interp, lower, upper, relative, max_code):
    return lut.new_empty(lut.shape, dtype=torch.float64)


@torch.library.custom_op(f"{_LIB}::band_stats", mutates_args=())
def band_stats(mean: torch.Tensor, std: Optional[torch.Tensor]) -> torch.Tensor:
    """ct_band_stats: (6, C) float64 per-channel min / max / sum of a merged band's mean and std (configuration C5)."""
    return ops.band_stats(mean, std)


@band_stats.register_fake
def _(mean, std):
    return mean.new_empty((6, mean.shape[0]), dtype=torch.float64)

