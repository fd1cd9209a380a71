"""Loss pieces with the reference's names (clair_torch/training/losses.py).

The per-pixel work of pixelwise_linearity_loss / compute_spatial_linearity_loss / combined_gaussian_pair_weights
is fused into the pair-residual HIP kernels (training/linearity.py); what remains here are the O(C*L) curve
penalties (plain torch ops on the (C, L) LUT, differentiable) and the elementwise Gaussian weight kept for API use
(compute_hdr_image only tests ``weight_fn is not None``, hdr_merge.py:95).
"""
from typing import Optional

import torch

from ..common.general_functions import weighted_mean_and_std


def pixelwise_linearity_loss(image_value_stack: torch.Tensor, i_idx: torch.Tensor, j_idx: torch.Tensor,
                             ratio_pairs: torch.Tensor, image_std_stack: Optional[torch.Tensor] = None,
                             use_relative: bool = True):
    """Per-pixel residual |I_i - r I_j| (relative: divided by r I_j + 1e-6) of every exposure pair and, with
    uncertainties, its propagated standard uncertainty (reference losses.py:13-67).  Returns ((P,C,H,W), (P,C,H,W)|None).

    API-compatibility helper: it materialises the (P, C, H, W) tensors the reference does.  train_icrf and
    measure_linearity never call it -- ct_pair_residual_fwd forms the same residual per sample in registers and keeps
    only the (P, C) sums (training/linearity.py)."""
    r = ratio_pairs.view(-1, 1, 1, 1)
    v_i, v_j = image_value_stack[i_idx], image_value_stack[j_idx]
    expected = v_j * r
    residual = v_i - expected
    safe = expected + 1e-6
    if use_relative:
        residual = residual / safe
    err = None
    if image_std_stack is not None:
        s_i, s_j = image_std_stack[i_idx], image_std_stack[j_idx]
        if use_relative:
            err = torch.sqrt((s_i / safe) ** 2 + ((v_i * s_j) / (safe * v_j.clamp(min=1e-6))) ** 2 + 1e-6)
        else:
            err = torch.sqrt(s_i ** 2 + (r * s_j) ** 2)
    return residual.abs(), err


def compute_spatial_linearity_loss(pixelwise_losses: torch.Tensor, pixelwise_errors: Optional[torch.Tensor] = None,
                                   external_weights: Optional[torch.Tensor] = None,
                                   valid_mask: Optional[torch.Tensor] = None, use_uncertainty_weighting: bool = True):
    """Weighted spatial mean / std of the per-pixel residuals and the plain masked mean of their uncertainties
    (reference losses.py:70-108): weights = [1 / (err + 1e-6)] + external weights.  Returns (mean, std, error|None),
    each (P, C).  API-compatibility helper, see pixelwise_linearity_loss."""
    weights = None
    if pixelwise_errors is not None or external_weights is not None:
        weights = torch.zeros_like(pixelwise_losses)
        if pixelwise_errors is not None and use_uncertainty_weighting:
            weights = weights + 1 / (pixelwise_errors + 1e-6)
        if external_weights is not None:
            weights = weights + external_weights
    mean, std = weighted_mean_and_std(pixelwise_losses, weights=weights, mask=valid_mask, dim=(2, 3))
    error = None
    if pixelwise_errors is not None:
        error, _ = weighted_mean_and_std(pixelwise_errors, mask=valid_mask, dim=(2, 3))
    return mean, std, error


def gaussian_value_weights(image: torch.Tensor, scale: Optional[float] = 30.0) -> torch.Tensor:
    """exp(-scale * (image - 0.5)^2), reference losses.py:193-205."""
    return torch.exp(-scale * (image - 0.5) ** 2)


def combined_gaussian_pair_weights(image_stack, i_idx, j_idx, scale: Optional[float] = 10.0):
    """Sum of the two images' Gaussian weights per pair, reference losses.py:208-235."""
    if i_idx.ndim != 1 or j_idx.ndim != 1:
        raise ValueError("i_idx and j_idx must be one-dimensional")
    return gaussian_value_weights(image_stack[i_idx], scale) + gaussian_value_weights(image_stack[j_idx], scale)


def compute_monotonicity_penalty(curve: torch.Tensor, squared=True, per_channel: bool = False) -> torch.Tensor:
    """Penalty on non-increasing steps of a (C, L) curve, reference losses.py:111-133."""
    df = curve[:, 1:] - curve[:, :-1]
    down = (df <= 0).float()
    penalty = (down * df.pow(2) if squared else down * (-df)).sum(dim=1)
    return penalty if per_channel else torch.sum(penalty)


def compute_smoothness_penalty(curve: torch.Tensor, per_channel: bool = False) -> torch.Tensor:
    """Sum of squared second differences, reference losses.py:136-149."""
    penalty = (curve[:, :-2] - 2 * curve[:, 1:-1] + curve[:, 2:]).pow(2).sum(dim=1)
    return penalty if per_channel else torch.sum(penalty)


def compute_range_penalty(curve: torch.Tensor, epsilon: float = 1e-6, per_channel: bool = False) -> torch.Tensor:
    """Linear penalty for leaving [0, 1], reference losses.py:152-170."""
    penalty = (torch.relu(-curve) + torch.relu(curve - 1)).sum(dim=1)
    return penalty if per_channel else torch.sum(penalty)


def compute_endpoint_penalty(curve: torch.Tensor, per_channel: Optional[bool] = False) -> torch.Tensor:
    """(first - 0)^2 + (last - 1)^2 per channel, reference losses.py:173-190."""
    if curve.ndim == 1:
        curve = curve.unsqueeze(1)
    if curve.ndim not in (1, 2):
        raise ValueError(f"curve must have 1 or 2 dimensions, got {curve.ndim}")
    penalty = (curve[:, 0] - 0) ** 2 + (curve[:, -1] - 1) ** 2
    return penalty if per_channel else torch.sum(penalty)
