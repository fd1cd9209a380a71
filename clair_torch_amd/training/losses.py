"""Loss pieces with the reference's names (clair_torch/training/losses.py).

The per-pixel work of pixelwise_linearity_loss / compute_spatial_linearity_loss / combined_gaussian_pair_weights
is fused into the pair-residual HIP kernels (training/linearity.py); what remains here are the O(C*L) curve
penalties (plain torch ops on the (C, L) LUT, differentiable) and the elementwise Gaussian weight kept for API use
(compute_hdr_image only tests ``weight_fn is not None``, hdr_merge.py:95).
"""
from typing import Optional

import torch


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
