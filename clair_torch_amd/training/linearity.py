"""The per-pixel exposure-pair linearity statistic on the GPU: loss term for train_icrf and measure_linearity.

Host side of ct_pair_residual_fwd / ct_pair_residual_bwd (csrc/ct_pairs.hip).  The reference computes, per step,
mask -> pair weights -> model forward -> pixelwise residual -> weighted spatial mean -> sqrt(sum_p mean^2)
(clair_torch/training/icrf_training.py:105-136) with full (P,C,H,W) float64 temporaries and autograd; here the
forward is one kernel producing (P,C) sums, and the backward one kernel producing the (C,L) LUT gradient.

Multi-GPU: the image is sharded in row bands.  The loss is NOT a sum of per-tile losses (it is a ratio of sums
under a square root), but the five sums are additive, so the exact data-parallel form is
all_reduce(sums) -> identical loss on every rank -> local backward -> all_reduce(LUT gradient).
Both messages are a few KB: one flat float64 buffer each.
"""
from typing import Optional

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader

from .. import ops
from ..common.general_functions import get_valid_exposure_pairs
from ..common.typecheck import expect
from ..inference._staging import resolve_device, stage_images, std_arguments, normalise_transform_list
from ..models.base import ICRFModelBase


def _all_reduce_sum(t: torch.Tensor, group):
    """Sum over the ranks of ``group``: None = the default group when one with more than one rank is up, False = this
    rank only (a caller that shards nothing), a ProcessGroup = that group."""
    if group is False:
        return t
    if group is not None or (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def spatial_mean(sums: torch.Tensor):
    """(P,C,5) sums -> weighted spatial mean (general_functions.py:149-156: total weight clamped to 1e-8)."""
    return sums[..., 1] / sums[..., 0].clamp(min=1e-8)


def spatial_statistics(sums: torch.Tensor, centered_sums: torch.Tensor, have_err: bool):
    """-> (mean, std, error|None) with weighted_mean_and_std semantics (general_functions.py:149-170).
    ``centered_sums`` comes from a second kernel pass with center = mean, so the std is formed from
    sum (v - mean)^2 w m directly, like the reference, not from cancelling raw moments."""
    den = sums[..., 0].clamp(min=1e-8)
    mean = sums[..., 1] / den
    std = torch.sqrt(centered_sums[..., 2] / den)
    err = sums[..., 3] / sums[..., 4].clamp(min=1e-8) if have_err else None
    return mean, std, err


class _LinearityTerm(torch.autograd.Function):
    """lut (C,L) -> per-channel linearity loss sqrt(sum_p spatial_mean_pc^2) (icrf_training.py:133-136)."""

    @staticmethod
    def forward(ctx, lut, stack, pairs, kw, group):
        sums = ops.pair_residual_sums(stack, pairs, lut=lut, level=0, **kw)
        _all_reduce_sum(sums, group)
        den = sums[..., 0].clamp(min=1e-8)
        spatial = sums[..., 1] / den
        lin = torch.sqrt((spatial ** 2).sum(dim=0))
        ctx.stack, ctx.pairs, ctx.kw, ctx.group = stack, pairs, kw, group
        ctx.save_for_backward(lut, spatial, den, lin)
        ctx.mark_non_differentiable(spatial)
        return lin, spatial

    @staticmethod
    def backward(ctx, grad_lin, _grad_spatial):
        lut, spatial, den, lin = ctx.saved_tensors
        kw = ctx.kw
        coef = (grad_lin.to(torch.float64) / lin).unsqueeze(0) * spatial / den
        grad = ops.pair_residual_lut_grad(ctx.stack, ctx.pairs, coef, lut=lut, interp=kw["interp"], lower=kw["lower"],
                                          upper=kw["upper"], use_relative=kw["use_relative"], max_code=kw["max_code"],
                                          tile=kw["tile"], use_unc_weight=kw["use_unc_weight"], std=kw["std"],
                                          std_mode=kw["std_mode"], std_value=kw["std_value"], smean=spatial,
                                          layout=kw["layout"])
        _all_reduce_sum(grad, ctx.group)
        return grad.to(lut.dtype), None, None, None, None


def linearity_loss(lut: torch.Tensor, stack: torch.Tensor, pairs: ops.PairList, *, interp: str, lower: float,
                   upper: float, use_relative: bool, use_unc_weight: bool, std=None, std_mode="none", std_value=0.0,
                   max_code=None, tile=None, group=None, layout="nchw"):
    """Differentiable (w.r.t. ``lut``) per-channel linearity loss and the (P,C) spatial means.  ``layout`` "nhwc" /
    "nhwc_bgr": ``stack`` is (N,H,W,C) as decoded (cv_to_torch, general_functions.py:315-335, folded into the staging)."""
    if not (use_unc_weight and (std is not None or std_mode != "none")):
        std, std_mode, std_value = None, "none", 0.0  # uncertainties only enter through the weights (losses.py:93-100)
    kw = dict(interp=interp, lower=lower, upper=upper, use_relative=use_relative, use_unc_weight=use_unc_weight,
              std=std, std_mode=std_mode, std_value=std_value, max_code=max_code, tile=tile, layout=layout)
    return _LinearityTerm.apply(lut, stack, pairs, kw, group)


def measure_linearity(dataloader: DataLoader, device, use_uncertainty_weighting: bool = True,
                      use_relative_linearity_loss: bool = True, icrf_model: Optional[ICRFModelBase] = None,
                      gpu_transforms=None, tile=None, group=None):
    """clair_torch/inference/measure_linearity.py:17-74: spatial linearity statistics of the FIRST batch.

    Returns (exposure ratios (P,) float64, spatial mean (P,C), spatial std (P,C), spatial error (P,C) | None).
    Thresholds are the reference's hard-coded ones: ratio >= 0.2, valid pixels in [1/255, 254/255]."""
    expect(dataloader, DataLoader, "dataloader")
    expect(device, (str, torch.device), "device")
    expect(icrf_model, ICRFModelBase, "icrf_model", allow_none=True)
    dev = resolve_device(device)
    transforms = normalise_transform_list(gpu_transforms)
    for _, val_batch, std_batch, meta_batch in dataloader:
        images, max_code, layout = stage_images(val_batch, dev, transforms, want_layout=True)
        std, std_mode, std_value = std_arguments(std_batch, dataloader.dataset, dev)
        if std is not None and layout != "nchw":  # explicit uncertainty images are planar
            images, max_code, layout = stage_images(images, dev, transforms) + ("nchw",)
        exposures = meta_batch["exposure_time"].to(torch.float64)
        i_idx, j_idx, ratio = get_valid_exposure_pairs(exposures, 0.2)
        pairs = ops.PairList(i_idx, j_idx, ratio, images.shape[0], dev)
        lut = interp = None
        if icrf_model is not None:
            lut, interp = icrf_model.icrf.detach().to(dev), icrf_model.interp_name
        kw = dict(lut=lut, interp=interp, lower=1 / 255, upper=254 / 255, use_relative=use_relative_linearity_loss,
                  use_unc_weight=use_uncertainty_weighting, std=std, std_mode=std_mode, std_value=std_value,
                  max_code=max_code, level=1, tile=tile, layout=layout)
        sums = _all_reduce_sum(ops.pair_residual_sums(images, pairs, **kw), group)
        centered = _all_reduce_sum(ops.pair_residual_sums(images, pairs, center=spatial_mean(sums), **kw), group)
        mean, sd, err = spatial_statistics(sums, centered, std_mode != "none")
        return pairs.ratio, mean, sd, err
    raise ValueError("dataloader yielded no batches")
