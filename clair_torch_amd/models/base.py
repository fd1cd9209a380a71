"""ICRF model base class with the reference's interface (clair_torch/models/base.py:17-259).

The curve is a (C, L) float32 LUT.  ``forward`` dispatches on the interpolation mode like the reference, but the
three samplers are HIP kernels (ct_linearize_fwd / ct_linearize_bwd, registered as the torch.library custom op
``clair_hip::icrf_forward`` with its autograd formula, clair_torch_amd/torch_ops.py) instead of chains of eager indexing ops; the backward returns the analytic image gradient and the LUT gradient, so optimisers
and ``torch.autograd.grad`` calls written against the reference keep working.

Reference behaviours kept on purpose: LINEAR and CATMULL pick the LUT row from the flat NCHW position
modulo C (base.py:173-176, 216-219); LOOKUP uses the true channel and carries no image gradient.
"""
from abc import ABC, abstractmethod
from typing import Optional

import torch
from torch import nn

from .. import torch_ops
from ..common.enums import INTERP_NAME, InterpMode
from ..common.typecheck import expect


class ICRFModelBase(nn.Module, ABC):
    def __init__(self, n_points: Optional[int] = 256, channels: Optional[int] = 3,
                 interpolation_mode: InterpMode = InterpMode.LINEAR, initial_power: float = 2.5,
                 icrf: Optional[torch.Tensor] = None):
        super().__init__()
        expect(interpolation_mode, InterpMode, "interpolation_mode")
        expect(icrf, torch.Tensor, "icrf", allow_none=True)
        if icrf is not None:  # a given curve overrides n_points / channels (base.py:61-62)
            channels, n_points = icrf.shape
        self._channels, self._n_points, self._initial_power = channels, n_points, initial_power
        self.register_buffer("_x_axis_datapoints", torch.linspace(0, 1, n_points))
        if icrf is None:
            icrf = self._initialize_default_icrf()
        self.register_buffer("_icrf", icrf)
        if interpolation_mode not in INTERP_NAME:
            raise ValueError(f"Unknown interpolation mode {interpolation_mode}")
        self.interpolation_mode = interpolation_mode

    icrf = property(lambda self: self._icrf)
    channels = property(lambda self: self._channels)
    n_points = property(lambda self: self._n_points)
    initial_power = property(lambda self: self._initial_power)
    x_axis_datapoints = property(lambda self: self._x_axis_datapoints)

    @abstractmethod
    def channel_params(self, c: int):
        """Optimisation parameters of channel ``c`` (one optimiser per channel in the reference's training script)."""

    @abstractmethod
    def update_icrf(self) -> None:
        """Rebuild ``_icrf`` from the parameters."""

    def _initialize_default_icrf(self) -> torch.Tensor:
        # base.py:128-133: linspace(0,1,L)^p for every channel, laid out (C, L)
        ramp = torch.linspace(0, 1, self.n_points).unsqueeze(1).repeat(1, self.channels) ** self.initial_power
        return torch.transpose(ramp, 0, 1)

    @property
    def interp_name(self) -> str:
        return INTERP_NAME[self.interpolation_mode]

    def forward(self, image: torch.Tensor) -> torch.Tensor:
        if self.interpolation_mode not in INTERP_NAME:
            raise ValueError(f"Unknown interpolation mode {self.interpolation_mode}")
        if not image.is_cuda:
            raise RuntimeError("ICRF model forward runs on MI355X (cuda/ROCm) tensors only; got a "
                               f"{image.device} tensor. clair_torch_amd has no CPU path.")
        if image.ndim != 4:
            raise ValueError(f"image must be (N, C, H, W), got {tuple(image.shape)}")
        # the dispatcher-registered op (clair_hip::icrf_forward): forward kernel + registered autograd formula
        return torch_ops.icrf_forward(image.to(torch.float32), self._icrf, self.interp_name)

    def plot_icrf(self) -> None:
        """Live plotting is UI (out of scope): headless no-op kept for call-site compatibility."""
        return None
