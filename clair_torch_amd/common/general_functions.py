"""Small host-side functions on the boundary of the hot path (clair_torch/common/general_functions.py).

These are O(N^2) / O(C*L) bookkeeping on tiny tensors (pair selection) or API-compatible utilities; the per-pixel
work they feed is done by the HIP kernels.  weighted_mean_and_std, flat_field_mean and flatfield_correction are the
reference's public tensor helpers of the same names: plain torch expressions on whatever device their arguments live
on, kept so that code written against the reference's helpers keeps working.  The drop-in entry points do not call
them -- their arithmetic is fused into ct_pair_residual_fwd (the weighted spatial statistics) and ct_flatfield_sums /
ct_flatfield_apply (the flat-field epilogue).
"""
import math
from typing import Optional

import torch


def get_valid_exposure_pairs(increasing_exposure_values: torch.Tensor, exposure_ratio_threshold: Optional[float] = None):
    """All (i, j), i < j in triu order with ratio t_i / t_j >= threshold (reference general_functions.py:242-272)."""
    n = increasing_exposure_values.shape[0]
    dev = increasing_exposure_values.device
    ratios = increasing_exposure_values.view(n, 1) / increasing_exposure_values.view(1, n)
    i_idx, j_idx = torch.triu_indices(n, n, offset=1)
    i_idx, j_idx = i_idx.to(dev), j_idx.to(dev)
    ratio_pairs = ratios[i_idx, j_idx]
    if exposure_ratio_threshold is not None:
        keep = ratio_pairs >= exposure_ratio_threshold
        i_idx, j_idx, ratio_pairs = i_idx[keep], j_idx[keep], ratio_pairs[keep]
    return i_idx, j_idx, ratio_pairs


def get_pairwise_valid_pixel_mask(image_value_stack, i_idx, j_idx, image_std_stack=None, val_lower=0.0, val_upper=1.0,
                                  std_lower=None, std_upper=None):
    """Boolean (P,C,H,W) validity mask (reference general_functions.py:276-312).  The training / measurement
    kernels evaluate the same predicate per sample without materialising it; this function exists for API parity."""
    if val_lower > val_upper:
        raise ValueError("Lower threshold cannot be a larger value than upper threshold.")
    if std_lower is not None and std_upper is not None and std_lower > std_upper:
        raise ValueError("Lower threshold cannot be a larger value than upper threshold.")
    vi, vj = image_value_stack[i_idx], image_value_stack[j_idx]
    mask = (vi >= val_lower) & (vi <= val_upper) & (vj >= val_lower) & (vj <= val_upper)
    if image_std_stack is not None and (std_lower is not None or std_upper is not None):
        si, sj = image_std_stack[i_idx], image_std_stack[j_idx]
        mask = mask & (si >= std_lower) & (si <= std_upper) & (sj >= std_lower) & (sj <= std_upper)
    return mask


def weighted_mean_and_std(values: torch.Tensor, weights: Optional[torch.Tensor] = None, mask: Optional[torch.Tensor] = None,
                          dim=None, keepdim=False, eps=1e-8, compute_std: Optional[bool] = True):
    """Weighted mean and standard deviation over ``dim`` with an optional boolean mask (reference
    general_functions.py:118-178).  A mask multiplies values AND weights (so masked values count as zeros with zero
    weight); without weights the mask is the weight.  The total weight is clamped to ``eps``; where it is exactly zero
    mean and std are zero.  Returns (mean, std | None)."""
    std = None
    if mask is not None:
        m = mask.to(dtype=values.dtype)
        values = values * m
        weights = m if weights is None else weights * m
    if weights is None:
        mean = values.mean(dim=dim, keepdim=True)
        if compute_std:
            std = torch.sqrt(((values - mean) ** 2).mean(dim=dim, keepdim=True))
    else:
        total = weights.sum(dim=dim, keepdim=True).clamp(min=eps)
        empty = total == 0
        mean = (values * weights).sum(dim=dim, keepdim=True) / total
        if compute_std:
            std = torch.sqrt((((values - mean) ** 2) * weights).sum(dim=dim, keepdim=True) / total)
            std = torch.where(empty, torch.zeros_like(std), std)
        mean = torch.where(empty, torch.zeros_like(mean), mean)
    if not keepdim:
        mean = mean.squeeze(dim) if dim is not None else mean.squeeze()
        if compute_std:
            std = std.squeeze(dim) if dim is not None else std.squeeze()
    return mean, std


def flat_field_mean(flat_field: torch.Tensor, mid_area_side_fraction: float) -> torch.Tensor:
    """Mean over a centred ROI covering ``mid_area_side_fraction`` of each spatial side, per image and channel:
    (N, C, H, W) -> (N, C, 1, 1) (reference general_functions.py:182-210; both call sites pass 1.0 = whole image)."""
    if not 0.0 <= mid_area_side_fraction <= 1.0:
        raise ValueError("mid_area_side_fraction should be between 0.0 and 1.0")
    _, _, h, w = flat_field.shape
    dx, dy = math.floor(w * mid_area_side_fraction), math.floor(h * mid_area_side_fraction)
    first = (math.floor(1 / mid_area_side_fraction) - 1) / 2   # ROI position in units of its own size
    x0, x1 = math.floor(first * dx), math.floor((first + 1) * dx)
    y0, y1 = math.floor(first * dy), math.floor((first + 1) * dy)
    return flat_field[:, :, y0:y1, x0:x1].mean(dim=(-1, -2), keepdim=True)


def flatfield_correction(images: torch.Tensor, flatfield: torch.Tensor, flatfield_mean_val: torch.Tensor,
                         epsilon: float = 1e-6) -> torch.Tensor:
    """images / (flatfield + epsilon) * flatfield_mean_val with broadcasting (reference general_functions.py:214-238);
    raises ValueError when the shapes do not broadcast."""
    for other, name in ((flatfield, "flatfield"), (flatfield_mean_val, "flatfield_mean_val")):
        try:
            torch.broadcast_shapes(images.shape, other.shape)
        except RuntimeError as exc:
            raise ValueError(f"{name} shape {tuple(other.shape)} is not broadcastable to {tuple(images.shape)}") from exc
    return (images / (flatfield + epsilon)) * flatfield_mean_val
