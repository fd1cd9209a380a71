"""ICRFModelDirect: one free parameter per LUT sample (clair_torch/models/icrf_model.py:89-127).

``state_dict`` keys match the reference (``_x_axis_datapoints``, ``_icrf``, ``direct_params.{c}``) so checkpoints
written by either implementation load in the other.  ICRFModelPCA is out of scope (broken at the reference commit,
SURVEY 0.7).
"""
from typing import Optional

import torch
from torch import nn

from ..common.enums import InterpMode
from .base import ICRFModelBase


class ICRFModelDirect(ICRFModelBase):
    def __init__(self, n_points: Optional[int] = 256, channels: Optional[int] = 3,
                 interpolation_mode: InterpMode = InterpMode.LINEAR, initial_power: float = 2.5,
                 icrf: Optional[torch.Tensor] = None):
        super().__init__(n_points, channels, interpolation_mode, initial_power, icrf)
        # icrf_model.py:108-110: the parameters always start from linspace^p, also when a curve was given;
        # they take over the curve at the first update_icrf()
        self.direct_params = nn.ParameterList([
            nn.Parameter(torch.linspace(0, 1, self.n_points) ** initial_power) for _ in range(self.channels)])

    def channel_params(self, c: int):
        return [self.direct_params[c]]

    def update_icrf(self):
        self._icrf = torch.stack([p for p in self.direct_params], dim=0)
