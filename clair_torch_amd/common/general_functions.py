"""Small host-side functions on the boundary of the hot path (clair_torch/common/general_functions.py).

These are O(N^2) / O(C*L) bookkeeping on tiny tensors (pair selection) or API-compatible utilities; the per-pixel
work they feed is done by the HIP kernels.
"""
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
