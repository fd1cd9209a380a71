"""Batched online weighted mean / mean+variance accumulators with the reference's interface
(clair_torch/common/statistics.py).  These are small device-agnostic tensor utilities kept for API compatibility;
the per-pixel hot loops do NOT go through them -- compute_hdr_image carries the WBOMean recurrence inside
ct_hdr_merge_batch and compute_video_mean_and_std carries the WBOMeanVar recurrence inside ct_video_stats_batch.
"""
from typing import Optional

import torch

from .enums import VarianceMode


class WBOMean:
    """Weighted batched online mean (reference statistics.py:14-109)."""

    def __init__(self, dim=0):
        if not isinstance(dim, int):
            raise TypeError(f"Expected dim as int or tuple of int, got {type(dim)}")  # statistics.py:27-30
        self._dim = (dim,)
        self._mean = 0.0
        self._sum_of_weights = 0.0

    mean = property(lambda self: self._mean)
    sum_of_weights = property(lambda self: self._sum_of_weights)
    dim = property(lambda self: self._dim)

    def internal_detach(self, *, in_place: bool = True):
        for name in self._state_names():
            value = getattr(self, name)
            if torch.is_tensor(value):
                setattr(self, name, value.detach_() if in_place else value.detach())

    def _state_names(self):
        return ("_mean", "_sum_of_weights")

    def _batch_moments(self, values, weights, want_m2):
        count = 1
        for d in self._dim:
            count *= values.shape[d]
        if weights is not None:
            w_sum = torch.sum(weights, dim=self._dim, keepdim=True)
            w2_sum = torch.sum(weights ** 2, dim=self._dim, keepdim=True)
            mean = torch.sum(weights * values, dim=self._dim, keepdim=True) / (w_sum + 1e-6)
            m2 = torch.sum(weights * (values - mean) ** 2, dim=self._dim, keepdim=True) if want_m2 else None
        else:
            mean = torch.mean(values, dim=self._dim, keepdim=True)
            w_sum = torch.full_like(mean, count, dtype=values.dtype)
            w2_sum = w_sum
            m2 = torch.sum((values - mean) ** 2, dim=self._dim, keepdim=True) if want_m2 else None
        return mean, w_sum, w2_sum, m2

    def update_values(self, batch_values: torch.Tensor, batch_weights: Optional[torch.Tensor] = None):
        mean_b, w_b, _, _ = self._batch_moments(batch_values, batch_weights, False)
        w = self._sum_of_weights + w_b
        self._mean = self._mean + (w_b / w) * (mean_b - self._mean)
        self._sum_of_weights = w
        return self._mean


class WBOMeanVar(WBOMean):
    """Weighted batched online mean and second moment (reference statistics.py:112-259)."""

    def __init__(self, dim=0, variance_mode: VarianceMode = VarianceMode.RELIABILITY_WEIGHTS):
        super().__init__(dim=dim)
        if variance_mode not in (VarianceMode.POPULATION, VarianceMode.RELIABILITY_WEIGHTS, VarianceMode.SAMPLE_FREQUENCY):
            raise ValueError(f"Unknown variance mode {variance_mode}")
        self._variance_mode = variance_mode
        self._m2 = 0.0
        self._sum_of_squared_weights = 0.0

    m2 = property(lambda self: self._m2)
    sum_of_squared_weights = property(lambda self: self._sum_of_squared_weights)

    def _state_names(self):
        return ("_mean", "_m2", "_sum_of_weights", "_sum_of_squared_weights")

    def variance(self):
        w, w2 = self._sum_of_weights, self._sum_of_squared_weights
        if self._variance_mode == VarianceMode.POPULATION:
            return self._m2 * (1 / w)
        if self._variance_mode == VarianceMode.SAMPLE_FREQUENCY:
            return self._m2 * (1 / (w - 1))
        return self._m2 * (1 / (w - w2 / w))

    def update_values(self, batch_values: torch.Tensor, batch_weights: Optional[torch.Tensor] = None):
        mean_b, w_b, w2_b, m2_b = self._batch_moments(batch_values, batch_weights, True)
        w_a, mean_a = self._sum_of_weights, self._mean
        w = w_a + w_b
        self._m2 = self._m2 + m2_b + (w_a * w_b / w) * (mean_b - mean_a) ** 2
        self._mean = mean_a + (w_b / w) * (mean_b - mean_a)
        self._sum_of_weights = w
        self._sum_of_squared_weights = self._sum_of_squared_weights + w2_b
        return self._mean, self._m2
